// jd_internal.h -- shared between the host API (jd_api.cpp) and the kernels (jd_kernels.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

// ---- Huffman lookup tables -------------------------------------------------------------------------
// One table = JD_LUT_PRIMARY u16 entries indexed by the next JD_LUT_BITS bits of the stream, followed by
// 64-entry secondary tables for codes longer than JD_LUT_BITS (indexed by the following 6 bits).
//   bit 15 = 0 : [4:0] total bits consumed (code length + magnitude bits), [8:5] magnitude bit count,
//                [14:9] zig-zag advance - 1 (run for run/size symbols, 15 for ZRL, 63 for EOB, 0 for DC)
//   bit 15 = 1 : [14:0] offset (in entries, from the table start) of the secondary table
//   0          : no code starts with these bits (invalid)
#define JD_LUT_BITS 10
#define JD_LUT_PRIMARY (1 << JD_LUT_BITS)
#define JD_LUT_SECONDARY 64
// primary index bits of the sync pass's twin tables (multi-symbol AC entries: a wider window holds more symbols)
#ifndef JD_SLUT_BITS
#define JD_SLUT_BITS 10
#endif

// ---- writer tables (32-bit entries) ------------------------------------------------------------------
// The whole-block writer decodes through its own tables, laid out so that the symbol step needs no left shifts
// and almost no field extraction (on gfx950 right shifts, and/xor/add issue at twice the rate of left shifts,
// bit-field extracts and three-operand forms).  One table = 2^P primary entries indexed by the next P bits of the
// stream + 2^(16-P)-entry secondary tables indexed by the bits after them.
//   bit 31 = 0 : [4:0]   32 - (code length + magnitude bits): how far to shift the window right to bring the
//                        magnitude bits down -- a shift uses only these five bits, so the entry is the operand
//                [15:8]  byte advance inside the block: 2 * (run + 1); 32 for ZRL; 254 for EOB; 0 for DC
//                        (tables of images in the byte form: run + 1; 16; 127; 0)
//                [30:16] magnitude mask 2^size - 1
//   bit 31 = 1 : [30:0] offset (in entries, from the table start) of the secondary table
//   0          : no code starts with these bits (invalid)
// primary index bits: DC tables 9; AC tables 10 when the set then still fits two writer workgroups per CU (a long
// code costs the whole wave a second lookup), else 9.  Secondary tables are indexed by the 16 - primary following bits.
#define JD_W32_DC_BITS 9
#define JD_W32_AC_BITS_WIDE 10
#define JD_W32_AC_BITS_NARROW 9
#define JD_W32_NONE 0x80000000u     // no such code (a secondary-table pointer with offset 0)
#define JD_W32_EOB_ADV 254u
#define JD_W32_EOB_ADV8 127u        // byte form (write_segments8): the advance counts coefficients, not bytes

// ---- pair tables of the byte-form writer (AC; jd_pack_lut32p) ---------------------------------------------------------
// Same shape as the writer tables above (2^P primary entries + secondary tables), but an entry describes up to TWO
// symbols, and a single symbol is a pair whose second half is empty (size 0, advance 0): one code path in the decoder.
//   [4:0]   sh2: 32 - bits of both symbols (codes + magnitudes): the right shift that brings symbol 2's magnitude down,
//           and what the window advances by
//   [7:5]   size2: magnitude bits of symbol 2 (0 .. 7)
//   [15:8]  adv1: coefficients symbol 1 advances by (run + 1; 16 for ZRL; JD_WP_EOB_ADV for EOB)
//   [22:16] adv2: the same for symbol 2; 0 = there is none
//   [27:23] sh1: 32 - bits of symbol 1 (= sh2 in a single)
//   [31:28] size1: magnitude bits of symbol 1; >= 8 (bit 31) only in singles -- the decoder keeps such a value's high part
//   bit 31 set and adv1 = 0: [27:16] offset (in entries, from the table start) of a secondary table; offset 0 = no such code
// A block that ends with symbol 1 (coefficient 63 reached) ignores symbol 2 and advances by symbol 1 alone.
#define JD_WP_EOB_ADV 127u
#define JD_WP_SKIP (31u | (JD_WP_EOB_ADV << 8) | (31u << 23))   // stand-in for an invalid code: one bit, end of block

// self-synchronising decode: bits per sub-sequence (one lane each)
#ifndef JD_SUBSEQ_BITS
#define JD_SUBSEQ_BITS 4096u
#endif
#define JD_SYNC_WG 256
// The first JD_SYNC_WARM lanes of a sync workgroup re-decode the sub-sequences just before the ones the
// workgroup owns, so that the owned lanes start from a state that has had >= JD_SYNC_WARM sub-sequences to
// synchronise and the cross-workgroup fix-up launches normally find nothing to redo.
#ifndef JD_SYNC_WARM
#define JD_SYNC_WARM 3
#endif
#define JD_SYNC_OWN (JD_SYNC_WG - JD_SYNC_WARM)

// whole-block Huffman writer: waves per workgroup (they share the lookup tables) and LDS bytes per wave
// (per lane a 64-byte half-block slot + a 64-byte stream ring, plus one dump word)
#define JD_W_WAVES 8
#define JD_WAVE_LDS (64u * 128u + 64u)

// unstuff pass: bytes per workgroup
#ifndef JD_K0_THREADS
#define JD_K0_THREADS 1024
#endif
#define JD_K0_CHUNK (JD_K0_THREADS * 32)

// unstuffed stream: bytes of padding in front of / behind each image's data
#define JD_STREAM_FRONT 16
#define JD_STREAM_TAIL 48

// IDCT + colour kernel: MCUs per workgroup (one wave per block-in-MCU, one lane per MCU)
#define JD_TILE_MCUS 64

// jd_decode_resident: bytes of the file-scope device array it works in (coefficients JD_RESIDENT_MAX_COEF x 2 B,
// unstuffed stream JD_RESIDENT_MAX_ECS, and up to 2 MiB of descriptors, tables, restart index and sync records)
#define JD_RESIDENT_SCRATCH_BYTES (12u << 20)

// per-image status bits written by the kernels
enum {
    JD_ST_MARKER = 1u << 0,      // non-RST marker inside the ECS
    JD_ST_RST_COUNT = 1u << 1,   // number of RSTn != intervals - 1
    JD_ST_RST_SEQ = 1u << 2,     // RSTn numbering out of sequence
    JD_ST_HUFF = 1u << 3,        // invalid Huffman code
    JD_ST_COEF = 1u << 4,        // coefficient index ran past 63
    JD_ST_TRUNC = 1u << 5,       // data ended early / interval overrun
    JD_ST_SYNC = 1u << 6,        // sync result inconsistent
    JD_ST_COUNT = 1u << 7,       // decoded block count != expected
    JD_ST_SAMPLE = 1u << 8,      // a post-IDCT chroma sample left int16 (K2; reported last: a damaged stream that
                                 // already earned another bit usually decodes to such garbage as well)
    JD_ST_FOVF = 1u << 9         // single-decode path (k1f_*): the staging of a batch ran out (only speculative decodes of a
                                 // hostile stream can ask for that much); never reported: the host decodes the batch again
                                 // through the two-pass path
};

enum {
    JD_IMG_UNSTUFFED_INPUT = 1u,   // K0 is skipped, host filled stream + interval table
    JD_IMG_UNIFORM_TABLES = 2u,    // every component uses the same DC and the same AC table: the sync pass ignores the
                                   // block-in-MCU phase, the writer stores DC differences, k1_dc_scan predicts
    JD_IMG_COEF8 = 4u,             // coefficients travel from the Huffman writer to K2 as bytes: a 64-byte block in the low
                                   // plane, the rare wide block's high parts in the high plane (write_segments8)
    JD_IMG_FUSED = 8u              // no DRI, byte form: the single-decode path (k1f_*, below) instead of k1_sync + k1_write_subseq
};

// ---- single-decode path for streams without restart markers (k1f_*) --------------------------------------------------
// The two-pass path decodes every bit 2.6 times (blind pass, counting pass, corrections, writing pass) because a block's
// place in the coefficient array is a prefix sum over everything before it.  Here the WRITING decode runs first, on a
// guessed entry state, into STAGING (slots of 64 bytes), DC values relative to the sub-sequence's start.  Sub-sequence s
// owns the slots [s * f_ch0, (s + 1) * f_ch0) -- f_ch0 is a quarter more than the image's average blocks per sub-sequence,
// so consecutive sub-sequences lie next to each other and most need nothing else -- and takes overflow chunks of
// JD_F_CHUNK slots from a counter when it holds more (one atomic per wave and chunk).  Lanes whose guess turns out
// wrong (one in twelve) decode again into their own slots; a scan over the per-sub-sequence counts then gives every
// block its index, and k1f_index writes, per block, where it lives and what to add to its DC (JdKernelArgs::loc) -- K2
// reads coefficients through that.  Decodes per bit: 0.5 (blind) + 1 + 0.1.
// Sub-sequence state here: the first block START at or after a sub-sequence's end, as [15:0] its bit offset past that
// end, [23:16] its block-in-MCU index.
// Its sub-sequences are the two-pass path's or twice that (JdKernelArgs::fsub_bits, chosen per run by the batch's size):
// the blind decode costs the same 2048 bits per BOUNDARY, so half as many boundaries halve it, and a wrong guess wastes the
// same share of the work either way -- but every step of the critical path is one walk through a sub-sequence, so only a
// batch that fills the machine many times over takes the long ones.
#define JD_F_SUBSEQ_BITS_MAX 8192u
#define JD_F_CHUNK 32u
#define JD_F_ROUNDS 6              // re-decode rounds enqueued blindly; jd_batch_wait adds more while the last check finds work
#define JD_F_KMAX_LIMIT (JD_F_SUBSEQ_BITS_MAX / 2u / (JD_F_CHUNK - 2u) + 4u)   // chunk-table entries per sub-sequence: a block is at least two bits
#define JD_F_HIMAP_OFF(slot_cap) ((uint64_t)(slot_cap) * 64u)                                   // byte offsets in the image's region
#define JD_F_HIPOOL_OFF(slot_cap) (JD_F_HIMAP_OFF(slot_cap) + ((((uint64_t)(slot_cap) * 4u) + 63u) & ~(uint64_t)63u))
#ifndef JD_F_CUT_DIV
#define JD_F_CUT_DIV 16u           // first decode: a wave stops when no more than 1/16 of its lanes are still walking
#endif
#ifndef JD_F_REDO_LANES
#define JD_F_REDO_LANES 256u       // lanes per workgroup of the re-decode launches
#endif
#define JD_F_REDO_DIV 8u           // ... which have an eighth of the lanes a full pass has (each workgroup walks its share of the list)
#define JD_FSTATE(rel, c) (0x80000000u | ((rel) > 0xFFFFu ? 0xFFFFu : (rel)) | ((c) << 16))


struct JdImg {
    const uint8_t* ecs;          // stuffed ECS (device)
    uint8_t* rgb;                // output image (device)
    uint32_t ecs_len;
    uint32_t row_stride;
    uint32_t width, height;
    uint32_t mcux, mcuy, n_mcus;
    uint32_t hs, vs, ncomp, bpm; // bpm = blocks per MCU
    uint32_t restart_interval, n_intervals;
    uint32_t flags;
    uint32_t n_k0_chunks;
    uint32_t interval_base;      // index into interval_start[] (n_intervals + 1 entries)
    uint32_t subseq_base, n_subseq_cap, sync_wg_base, n_sync_wg;
    uint32_t lut_off, lut_len;   // u16 offset / length of this image's table set in the LUT buffer
    uint32_t dc_lut[3], ac_lut[3]; // per component: entry offset of its DC / AC table inside the set
                                   // the sync pass's twin set (multi-symbol AC tables, same layout) follows at
                                   // lut_off + lut_len
    uint32_t tile_base, n_tiles; // IDCT/colour tiles
    uint32_t s_len;              // u16 length of the sync twin set (at lut_off + lut_len)
    uint32_t sdc[3], sac[3];     // per component: entry offset of its DC / AC table inside the twin set
    uint32_t w_off, w_len;       // writer tables (32-bit entries): u16 offset in the LUT buffer (16-byte aligned) and
                                 // length in entries (multiple of 4)
    uint32_t wdc[3], wac[3];     // per component: entry offset of its DC / AC writer table inside that set
    uint32_t w_ac_bits;          // primary index bits of the AC writer tables
    uint64_t stream_off;         // byte offset of this image's region in the unstuffed buffer
    uint64_t coef_off;           // int16 offset in the coefficient buffer
    uint64_t loc_off;            // JD_IMG_FUSED: first entry of this image's blocks in JdKernelArgs::loc
    uint32_t slot_cap;           // JD_IMG_FUSED: staging slots of this image.  Its coefficient region: slot_cap x 64 B (low plane),
                                 // slot_cap x 4 B (per slot: where a WIDE block's high parts are), hi_cap x 64 B (those high parts,
                                 // handed out one by one: wide blocks are rare)
    uint32_t hi_cap;
    uint32_t f_ch0, f_ov_base;   // JD_IMG_FUSED: slots every sub-sequence owns; first slot of the overflow chunks (set per run
                                 // from the stream's length)
    uint32_t n_fwg;              // JD_IMG_FUSED: workgroups of 64 * JD_W_WAVES lanes that cover its sub-sequences (capacity)
    uint8_t q[3][64];            // quantisation table per component, natural order
};

struct JdIdctConst {             // main.go:1339-1354
    double M1, M2, M3, M4, M5, S0, S1, S2, S3, S4, S5, S6, S7;
};

// per-sub-sequence record of the self-synchronising pass
struct JdSub {
    uint32_t entry;              // packed entry state used for the last decode of this sub-sequence
    uint32_t exit;               // packed exit state (entry of the next sub-sequence)
    uint32_t nstart;             // blocks that START inside this sub-sequence
    uint32_t first;              // bit offset (from the sub-sequence start) of the first of them
    int32_t dc[3];               // sum of their DC differences, per component
    uint32_t pad;
};
// Checkpoints of a sub-sequence's FIRST full decode (k1_sync round 1): where that decode stood at the first block start past
// every fifth of the sub-sequence (state and running counts) plus its totals and exit state -- one self-consistent trajectory.  A later decode of the same
// sub-sequence from a corrected entry state stops at the first checkpoint whose state it reproduces (from there on it IS
// that trajectory) and takes the rest of the counts from the record.
#ifndef JD_CK_N
#define JD_CK_N 4
#endif
struct JdCkEntry {               // 32 bytes: written with two 16-byte non-temporal stores, never in parts
    uint32_t state;              // bits left << 12 | block-in-MCU << 8 | zig-zag index (0 = no checkpoint); totals: exit state
    uint32_t nstart;             // blocks started so far (totals: in the whole sub-sequence)
    int32_t dc[3];               // DC-difference sums so far (totals: of the whole sub-sequence)
    uint32_t pad[3];
};
struct JdSubCk { JdCkEntry e[JD_CK_N + 1]; };   // [JD_CK_N] = the decode's totals and exit state
static_assert(sizeof(JdSubCk) == 32 * (JD_CK_N + 1), "JdSubCk");
// packed state: [7:0] bit offset past the sub-sequence start, [15:8] block-in-MCU, [23:16] zig-zag index,
// bit 31 = valid
#define JD_STATE_VALID 0x80000000u
// exclusive prefix of a sub-sequence inside its sync workgroup (block starts, DC-difference sums)
struct JdSubPfx { uint32_t blk_off; int32_t dcp[3]; };

// dwords of one progressive-scan symbol table (jd_pack_lut_lane: 7 rows of 64, one row per vector register)
#define JD_PLUT_DWORDS 448
// one scan of a progressive file (kp_scan: one wave per scan)
struct JdPScan {
    uint32_t img;                // image whose coefficients the scan refines
    uint32_t stream_off, stream_len;   // where K0 puts its unstuffed bytes in the scan-stream buffer (16-byte aligned); its RAW length
    uint32_t ss, se, ah, al;     // spectral selection and successive approximation (main.go:980-987)
    uint32_t ncomp;              // components in the scan ...
    uint32_t comp[3];            // ... their frame indices, ascending (the reference walks the frame's components)
    uint32_t lut_off[3], lut_len[3];   // ... and the symbol table each decodes through (DC tables in a DC scan, the AC table
                                 // in an AC scan; none in a DC refinement): dword offset in the table buffer / JD_PLUT_DWORDS
};

struct JdKernelArgs {
    const JdImg* imgs;
    int n_images;
    const uint32_t* prefix;      // n_images + 1 exclusive prefix of workgroups per image for this kernel
    uint8_t* stream;             // unstuffed streams
    int16_t* coef;
    const uint16_t* lut;
    uint32_t* status;            // per image
    uint32_t* stream_len;        // per image: unstuffed byte count
    uint32_t* interval_start;
    unsigned long long* k0_state; // per K0 chunk: look-back status word (jd_kernels.hip)
    uint32_t* k0_ticket;         // per image: next chunk to hand out
    JdSub* sub;
    JdSubPfx* subpfx;
    JdSubCk* subck;              // per sub-sequence: checkpoints of its first full decode
    uint4* sync_tot;             // per sync workgroup: (block starts, dc0, dc1, dc2) totals -> exclusive prefix after spine
    uint32_t* changed;           // sync: set when a workgroup's exit state changed
    // single-decode path (JD_IMG_FUSED)
    uint2* loc;                  // per block: (staging slot, DC base of its component at its sub-sequence's start)
    uint32_t* chunk;             // per sub-sequence: fkmax overflow-chunk numbers (chunk k holds its blocks f_ch0 + JD_F_CHUNK * k ...)
    uint32_t* guess;             // per sub-sequence: blind guess of its exit state
    uint32_t* wl;                // per image (at subseq_base): sub-sequences to decode again
    uint32_t* falloc;            // per image: chunks handed out
    uint32_t* fhalloc;           // per image: high-part slots handed out
    uint32_t* fcount;            // [JD_F_ROUNDS + 1][n_images]: length of the re-decode list of each round
    uint32_t fkmax;
    uint32_t fsub_bits;          // bits per sub-sequence of the single-decode path in this run (4096 or 8192)
};

