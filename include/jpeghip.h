/*
 * jpeghip.h -- C-ABI of the MI355X-native baseline-JPEG decode path (libjpeghip.so).
 *
 * Drop-in boundary behind maxasm/jpeg-decoder's Go marker parser (reference paths are into
 * /root/reference).  The reference has no FFI of its own; the boundary is the pair of call sites in
 * decodeStartOfScan:
 *
 *   cut 1  main.go:1022-1032  generateCodes + decodeHuffmanData(header, &BitReader{...})
 *   cut 2  main.go:1054-1057  dequantize; inverseDCT; spreadCoeffecients; convertColorSpace
 *
 * jd_decode() replaces both: the Go host marshals Header (main.go:1467-1491) into a jd_frame, hands
 * over the entropy-coded segment, and gets the cropped RGB8 image that writeBitMap (main.go:1171-1218)
 * would read out of Header.blocks.  The cgo stub is in INTEGRATION.md and jpeg-decoder_amd/go/.
 *
 * Plain C: fixed-width integers, pointers and sizes only.  No call retains a caller pointer after it
 * returns unless stated.  All functions return 0 on success or a negative JD_E_* code.
 */
#ifndef JPEGHIP_H
#define JPEGHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- frame description: what the Go Header holds at cut 1 ------------------------------------ */

typedef struct jd_component {
    uint8_t h, v;   /* sampling factors: ColorComponent.hSamplingFactor / vSamplingFactor (main.go:127-128) */
    uint8_t tq;     /* ColorComponent.qTableId (main.go:131)                                             */
    uint8_t td, ta; /* dcHuffmanTableId / acHuffmanTableId bound by SOS (main.go:966-973)                */
} jd_component;

typedef struct jd_huff {
    uint8_t bits[16];   /* HuffmanTable.codesOfLen (main.go:860-865)                                       */
    uint8_t vals[256];  /* HuffmanTable.symbols (main.go:866-870)                                          */
    uint16_t nvals;
    uint8_t present;
} jd_huff;

typedef struct jd_frame {
    uint16_t width, height;       /* Header.width / height (main.go:117-118); precision is fixed at 8     */
    uint8_t ncomp;                /* 1 or 3 (the reference rejects > 3, main.go:112-115)                  */
    jd_component comp[3];         /* frame order; component 0 is luma and fixes the MCU (main.go:165)     */
    uint16_t restart_interval;    /* Header.restartInterval (main.go:830-831); 0 = none                   */
    uint8_t qt[4][64];            /* QuantizationTable.table: NATURAL order, low byte only (main.go:61-75) */
    uint8_t qt_present;           /* bit i set = table id i defined                                       */
    jd_huff huff[2][4];           /* [0][id] = DC tables, [1][id] = AC tables (main.go:850-851)           */
} jd_frame;

/* ---- error codes -------------------------------------------------------------------------------
 * The reference prints "Error! ..." and os.Exit(1) (e.g. main.go:189-192); the library returns a code
 * and the host shim prints jd_strerror() and exits 1 to keep the CLI behaviour. */
enum {
    JD_OK = 0,
    JD_E_ARG = -1,          /* NULL pointer, size mismatch, bad device                                   */
    JD_E_UNSUPPORTED = -2,  /* sampling other than luma {1,2}x{1,2} / chroma 1x1, ncomp not 1 or 3        */
    JD_E_TABLE = -3,        /* missing or malformed quantisation / Huffman table                         */
    JD_E_HIP = -4,          /* HIP runtime error (jd_last_hip_error())                                   */
    JD_E_NOMEM = -5,
    JD_E_MARKER = -6,       /* marker other than RSTn inside the entropy-coded segment (main.go:1013-1016) */
    JD_E_RESTART = -7,      /* number / numbering of RSTn does not match DRI                             */
    JD_E_HUFFMAN = -8,      /* invalid Huffman code (scanSymbol returned 0xFF, main.go:189-192)           */
    JD_E_COEF = -9,         /* run/size walks past coefficient 63 (Go: map miss clobbers [0], Appendix C) */
    JD_E_TRUNCATED = -10,   /* entropy data ended before all MCUs were decoded (readBit -1, main.go:1273)  */
    JD_E_SYNC = -11,        /* internal: self-synchronising decode failed its consistency check          */
    JD_E_NOT_INIT = -12,
    JD_E_STATE = -13,       /* batch call out of order (e.g. wait before run)                            */
    JD_E_TOO_LARGE = -14    /* image does not fit a fixed-size workspace                                 */
};
const char* jd_strerror(int code);
int jd_last_hip_error(void);

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Binds the library to HIP devices (NULL/0 = every visible device).  Replaces nothing in the
 * reference (it has no device); call once before the first decode.  Thread-safe, idempotent. */
int jd_init(const int* device_ids, int n_devices);
void jd_shutdown(void);
int jd_device_count(void);

/* Validates a frame the way decodeStartOfFrame / decodeStartOfScan + getTable / getQuantizationTable
 * would need it (main.go:87-180, 432-452, 957-977) without touching the GPU. */
int jd_frame_check(const jd_frame* frame);

/* Geometry helpers (main.go:159-172): MCU grid and coefficient count for a frame. */
int jd_frame_geometry(const jd_frame* frame, uint32_t* mcus_x, uint32_t* mcus_y, uint32_t* blocks_per_mcu);

/* ---- one-shot decode: replaces main.go:1022-1032 + 1054-1057 ----------------------------------
 * ecs            entropy-coded segment, host memory.
 *                ecs_is_stuffed = 1: the raw bytes between the SOS header and EOI, FF00 stuffing and
 *                  RSTn markers included (what main.go:993-1021 consumes); the device unstuffs and
 *                  indexes the restart markers.
 *                ecs_is_stuffed = 0: the reference's `_bitstream` (main.go:1011,1018): unstuffed, RSTn
 *                  removed; rst_offsets[k] = byte offset in it where restart interval k+1 starts
 *                  (n_rst = number of RSTn markers; required when restart_interval > 0).
 * rgb_out        host memory, height rows of row_stride bytes (>= 3*width), top-down, R,G,B -- the
 *                crop writeBitMap performs (main.go:1198-1211) without the bottom-up/BGR file order.
 * Synchronous.  */
int jd_decode(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int ecs_is_stuffed,
              const uint32_t* rst_offsets, size_t n_rst, uint8_t* rgb_out, size_t row_stride, int device);

/* Parity hook for cut 1 alone (decodeHuffmanData, main.go:749-816): quantised coefficients, int16,
 * scan order [mcu][block-in-MCU][64] (luma blocks row-major inside the MCU, then Cb, Cr), ZIG-ZAG order,
 * DC already predicted (absolute), before dequantisation.  coef_cap counts int16 elements. */
int jd_decode_coefficients(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int ecs_is_stuffed,
                           const uint32_t* rst_offsets, size_t n_rst, int16_t* coef_out, size_t coef_cap,
                           int device);

/* ---- batch / device-resident form --------------------------------------------------------------
 * One jd_batch owns the device workspace for n images on one device and decodes all of them with a
 * fixed sequence of batched kernels (no per-image launches, no host round trips between kernels).
 * Typical use: create once per shape set, then {upload or bind, run, wait, download} per batch.
 * The reference decodes files one after another (main.go:1509-1512); this is its throughput form. */
typedef struct jd_batch jd_batch;

enum {
    JD_BATCH_KEEP_COEF = 1u,  /* keep coefficients readable after run (jd_batch_coefficients)           */
    JD_BATCH_PINNED_IO = 2u,  /* library-owned pinned staging + copy streams: jd_batch_stage_* / jd_batch_submit */
    JD_BATCH_EXTERNAL_IO = 4u /* every image will be bound (jd_batch_bind_ecs / _rgb): no ECS / RGB buffers of its own */
};

int jd_batch_create(int device, int n_images, const jd_frame* frames, const size_t* ecs_cap,
                    unsigned flags, jd_batch** out);
void jd_batch_destroy(jd_batch* b);

/* Input: either copy a stuffed ECS from host memory into the batch's own device buffer ... */
int jd_batch_upload(jd_batch* b, int image, const uint8_t* ecs, size_t ecs_len);
/* ... or the reference's unstuffed `_bitstream` + restart offsets (see jd_decode) ... */
int jd_batch_upload_unstuffed(jd_batch* b, int image, const uint8_t* bitstream, size_t len,
                              const uint32_t* rst_offsets, size_t n_rst);
/* ... or point at a stuffed ECS already resident in device memory (not copied; must stay valid until
 * jd_batch_wait returns).  d_ecs must be 16-byte aligned. */
int jd_batch_bind_ecs(jd_batch* b, int image, const void* d_ecs, size_t ecs_len);
/* Output: device memory for image i (height rows of row_stride bytes).  Without a bind the batch
 * allocates its own RGB buffer (row_stride = 3*width) that jd_batch_download copies out. */
int jd_batch_bind_rgb(jd_batch* b, int image, void* d_rgb, size_t row_stride);

/* Enqueue the whole decode on `hip_stream` (a hipStream_t; NULL = the batch's own stream).  Does not
 * allocate and does not wait for the device (descriptors go up by an asynchronous copy from pinned memory;
 * it only waits if an earlier run of the same batch is still in flight on a DIFFERENT stream).
 * Outputs are final once jd_batch_wait has returned: for images without restart markers the stream-ordered
 * result is complete unless the self-synchronising pass needed more cross-workgroup fix-ups than the fixed
 * number enqueued here (not observed; jd_batch_wait detects it, finishes the job and re-runs the consumers). */
int jd_batch_run(jd_batch* b, void* hip_stream);
/* Block until the last run (or submit) finished; returns the first per-image error (or JD_OK). */
int jd_batch_wait(jd_batch* b);
int jd_batch_status(const jd_batch* b, int image);
int jd_batch_download(jd_batch* b, int image, uint8_t* rgb_out, size_t row_stride);
/* Quantised coefficients of one image as int16 (layout of jd_decode_coefficients).  On the device they travel from the
 * Huffman kernels to the IDCT kernel in a byte form when the image's quantisers allow it; this call
 * expands them, on the device, into a temporary it allocates and frees.  Limits of the stored forms, beyond which the
 * image's status is JD_E_COEF: DC and AC 1..62 16 bits, AC 63 16 bits (int16 form) / 12 bits (byte form). */
int jd_batch_coefficients(jd_batch* b, int image, int16_t* coef_out, size_t coef_cap);

/* ---- pinned-memory pipeline (the reference's file loop main.go:1509-1512 + writer
 * main.go:1171-1218 in throughput form).  A batch created with JD_BATCH_PINNED_IO owns pinned host staging
 * for the stuffed ECS of all its images and for all its RGB output, plus copy streams.  Per use:
 *   jd_batch_stage_* for every image (the host writes / the library copies the ECS into the staging),
 *   jd_batch_submit (asynchronous: one H2D copy -> the kernels -> one D2H copy, chained by events),
 *   ... stage and submit OTHER batches meanwhile (two or three in flight keep both copy engines and the
 *   CUs busy) ...
 *   jd_batch_wait, then read the pixels in place through jd_batch_host_rgb.
 * C-allocated memory throughout, so a Go caller may keep the pointers across calls (cgo pointer rule). */
int jd_batch_stage_reserve(jd_batch* b, int image, size_t ecs_len, uint8_t** dst);   /* zero-copy: write ecs_len bytes at *dst */
int jd_batch_stage_ecs(jd_batch* b, int image, const uint8_t* ecs, size_t ecs_len);  /* memcpy form                          */
/* whole files: jd_parse + frame check against image first+k's frame + ECS copy, on `threads` host threads */
int jd_batch_stage_jpegs(jd_batch* b, int first, int count, const uint8_t* const* files, const size_t* lens, int threads);
int jd_batch_submit(jd_batch* b, void* hip_stream);
int jd_batch_host_rgb(jd_batch* b, int image, const uint8_t** rgb, size_t* row_stride);
/* Pinned host memory for callers that keep their own input/output buffers (hipHostMalloc / hipHostFree). */
int jd_host_alloc(size_t bytes, void** out);
void jd_host_free(void* p);

/* ---- allocation-free decode ------------------------------------------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device memory: serving loops and harnesses
 * that forbid hipMalloc (or run with the device's memory already handed out).  The same decode as
 * jd_batch_{create, bind_ecs, bind_rgb, run, wait} for one image, but the whole workspace (descriptors,
 * tables, unstuffed stream, restart index, sync records, coefficients) lives in a file-scope device array of
 * the library (12 MiB per device, held until the process ends) whose address jd_init looked up per device:
 * this call makes no hipMalloc / hipFree / hipHostMalloc and creates no stream or event.
 *   d_ecs    DEVICE memory, 16-byte aligned: the stuffed entropy-coded segment (as jd_batch_bind_ecs)
 *   d_rgb    DEVICE memory: height rows of row_stride bytes (as jd_batch_bind_rgb)
 *   limits   ecs_len <= JD_RESIDENT_MAX_ECS and MCUs x blocks-per-MCU x 64 <= JD_RESIDENT_MAX_COEF
 *            (1920 x 1080 4:2:0 fits; up to 2.7 MP 4:2:0 or 1.39 MP 4:4:4), else JD_E_TOO_LARGE
 * SYNCHRONOUS: enqueues on hip_stream (NULL = a stream of the library's), waits for that stream and reads the
 * per-image status back before it returns (pixels are then in d_rgb) -- so it must not be called while hip_stream
 * is being captured into a graph.  Allocation-free means the device / HIP side: the first call for a frame plans
 * on the host (table packing, a few small host allocations); repeated calls with the same jd_frame reuse that plan
 * and do no host allocation either.
 * Calls on one device share that device's array: the library serialises them on the host, so they never
 * overlap; two PROCESSES are independent (each has its own copy). */
#define JD_RESIDENT_MAX_ECS (2u << 20)
#define JD_RESIDENT_MAX_COEF (4u << 20)
int jd_decode_resident(const jd_frame* frame, const void* d_ecs, size_t ecs_len, void* d_rgb, size_t row_stride,
                       int device, void* hip_stream);

/* Instrumentation for bench.py / DESIGN.md: runs `iters` decodes of the batch on its own stream and
 * returns, per pipeline stage, the average device time in microseconds measured with hipEvents around
 * that stage's kernels.  names[i] points at static strings.  Returns the number of stages written. */
int jd_batch_profile(jd_batch* b, int iters, float* stage_us, const char** names, int cap);
/* The same measurement taken live: after jd_batch_timing(b, 1) every jd_batch_run (up to 64 between reads)
 * records hipEvents around its stages on the stream it runs on; jd_batch_stage_times synchronises, writes
 * the per-stage averages in microseconds and returns the number of runs averaged (then starts over). */
int jd_batch_timing(jd_batch* b, int enable);
int jd_batch_stage_times(jd_batch* b, float* stage_us, const char** names, int cap);
/* Bytes of device workspace the batch holds, and per-image facts measured by the last run. */
size_t jd_batch_workspace_bytes(const jd_batch* b);
int jd_batch_unstuffed_len(jd_batch* b, int image, size_t* len);

/* ---- progressive (SOF2) files: SURVEY row N4 ----------------------------------------------------
 * Replaces the SOF2 branch of decodeBandCoeffecients (main.go:241-428) and the scan chaining of
 * decodeStartOfScan (main.go:1035-1062).  The Go host keeps parsing SOS / DHT / DRI between the scans and hands
 * over one jd_scan per SOS: the scan header (main.go:954-987), the tables in force at that point (a DHT between
 * two scans may redefine an id, main.go:1049-1052) and where the scan's entropy-coded bytes lie in the file.
 * Coefficients accumulate over the scans exactly as Header.blocks does; dequantisation, IDCT, chroma spread and
 * colour conversion then run as for a baseline frame (main.go:1054-1057).  As in the reference, restart markers
 * inside a progressive scan are dropped and not acted upon. */
typedef struct jd_scan {
    uint8_t progressive;      /* 1: frame type SOF2 (the same in every scan of a file); 0: the single scan of an SOF0 file */
    uint8_t ncomp;            /* Header.componentsInScan (main.go:956)                                            */
    uint8_t comp[3];          /* frame index of each component of the scan, in the scan header's order               */
    uint8_t td[3], ta[3];     /* their DC / AC table ids (main.go:966-967)                                          */
    uint8_t ss, se, ah, al;   /* startOfSelection, endOfSelection, successiveApproximationHigh / Low (main.go:980-987) */
    uint64_t ecs_off, ecs_len; /* the scan's stuffed entropy-coded bytes: offset in the file and length               */
    jd_huff huff[2][4];       /* Header.huffmanTables as they stand when this scan starts                            */
} jd_scan;
/* Parser twin for SOF0 and SOF2 files (decodeJPEG ... decodeStartOfScan's header work, main.go:1076-1154, 942-1021,
 * 1035-1052): fills the frame and up to `cap` scans; *n_scans = scans in the file (JD_E_ARG if cap is too small). */
int jd_parse_scans(const uint8_t* file, size_t file_len, jd_frame* frame, jd_scan* scans, int cap, int* n_scans);
/* Decode a file given as frame + scans (host memory in, host memory out; synchronous). */
int jd_decode_scans(const jd_frame* frame, const jd_scan* scans, int n_scans, const uint8_t* file, size_t file_len,
                    uint8_t* rgb_out, size_t row_stride, int device);
/* The same for n_images progressive files at once (the reference's file loop, main.go:1509-1512): the scans of all of
 * them are dealt to wavefronts together, level by level -- a scan is a serial chain, so a batch is what fills the chip.
 * status[i] (may be NULL) receives image i's own result; the call returns the first error.  row_strides may be NULL
 * (3 * width).  Baseline files do not belong here (JD_E_UNSUPPORTED): they go through jd_batch_*. */
int jd_decode_scans_batch(int n_images, const jd_frame* frames, const jd_scan* const* scans, const int* n_scans,
                          const uint8_t* const* files, const size_t* file_lens, uint8_t* const* rgb_out,
                          const size_t* row_strides, int* status, int device);
/* Parity hook: the accumulated coefficients after the last scan, layout as jd_decode_coefficients. */
int jd_decode_scans_coefficients(const jd_frame* frame, const jd_scan* scans, int n_scans, const uint8_t* file,
                                 size_t file_len, int16_t* coef_out, size_t coef_cap, int device);

/* ---- host-side twin of the Go marker parser (test harness / CLI; Go keeps its own in production) --
 * Parses a baseline JPEG held in memory the way decodeJPEG and friends do (main.go:1076-1154, 42-180,
 * 818-888, 942-987) and returns the frame plus the extent of the stuffed entropy-coded segment
 * (start offset and length, EOI excluded). */
int jd_parse(const uint8_t* file, size_t file_len, jd_frame* frame, size_t* ecs_off, size_t* ecs_len);
/* Restart-interval-range sharding.  No counterpart in the reference (it has no restart handling, SURVEY
 * F3).  Cuts a DRI image into at most `parts` bands of whole MCU rows that decode independently -- each on its own GPU if
 * wanted: band k is an image of its own (bands[k].frame: the caller's frame with the band's height) whose stuffed
 * entropy-coded segment is ecs[ecs_off .. ecs_off + ecs_len) (the RSTn in front of it and the one behind it belong to no
 * band) and whose pixel rows go to rows y0 .. y0 + frame.height of the picture.  Every entry point above takes a band
 * as it takes an image.  Needs restart_interval > 0 dividing the MCUs per row (else JD_E_UNSUPPORTED); boundaries are
 * placed so that every band's first marker is RST0 again.  *n_bands <= parts (a picture has only so many MCU rows).
 * JD_E_RESTART if the segment holds fewer, or wrongly numbered, markers than the cuts need. */
typedef struct jd_band {
    jd_frame frame;
    size_t ecs_off, ecs_len;
    uint32_t y0;
} jd_band;
int jd_split_bands(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int parts, jd_band* bands, int* n_bands);
/* BMP exactly as writeBitMap lays it out (main.go:1171-1218): returns the file size; writes it if
 * out != NULL and cap is large enough. */
long jd_bmp_encode(const uint8_t* rgb, size_t row_stride, int width, int height, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
