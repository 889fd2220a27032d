/*
 * oracle.h -- CPU oracle for the baseline-JPEG hot path of maxasm/jpeg-decoder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (jpeg-decoder_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * there only as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement of the reference's Go algorithm (main.go +
 * zmap/map.go); every function in oracle.c cites the reference lines it
 * follows.  PARITY UNPINNED BY THE REFERENCE: the reference ships no tests,
 * golden vectors or known-answer outputs and no Go toolchain
 * exists in this image, so the Go binary cannot be run.  The oracle is pinned
 * instead against SURVEY.md Appendix B (SHA-256 of coefficients / samples /
 * RGB for the nine baseline files in the reference's test/ directory, made by
 * an independent restatement) -- see tests/test_oracle_golden.py.
 */
#ifndef JD_ORACLE_H
#define JD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_image orc_image;

/* error codes (the reference prints "Error! ..." and os.Exit(1); we return) */
enum {
    ORC_OK = 0,
    ORC_ERR_EOF = -1,           /* Buffer.advance hit end of file, main.go:22-25        */
    ORC_ERR_NOT_JPEG = -2,      /* main.go:1095-1098                                     */
    ORC_ERR_MARKER = -3,        /* unexpected / unsupported marker, main.go:1103-1149   */
    ORC_ERR_DQT = -4,           /* main.go:55-58, 77-82                                  */
    ORC_ERR_SOF = -5,           /* main.go:97-100, 112-115, 139-142                      */
    ORC_ERR_DRI = -6,           /* main.go:824-827                                       */
    ORC_ERR_HUFFMAN = -7,       /* invalid code / truncated data, main.go:189-192        */
    ORC_ERR_CORRUPT = -8,       /* stream states the reference handles by panic/clobber */
    ORC_ERR_UNSUPPORTED = -9,   /* progressive (SOF2) etc.: outside the hot path         */
    ORC_ERR_RESTART = -10,      /* restart-aware extension: marker bookkeeping mismatch  */
    ORC_ERR_IO = -11,
    ORC_ERR_NOMEM = -12
};

/* decode modes */
enum {
    ORC_MODE_REFERENCE = 0,     /* literal reference semantics: RSTn dropped, never aligned, DC never
                                   reset (main.go:999-1000, 750) -- fails on real DRI files */
    ORC_MODE_RESTART_AWARE = 1  /* EXTENSION (not in the reference): byte-align and reset DC predictors
                                   at every restart boundary, T.81 F.2.1.3.1                            */
};

/* cosine flavour for the IDCT constants */
enum {
    ORC_COS_GO = 0,             /* port of Go's pure-Go math.Cos (Cephes): what the Go binary computes */
    ORC_COS_LIBM = 1            /* glibc cos(): differs from the above by 1 ulp in S3 and S7           */
};
void orc_set_cos_flavour(int flavour);
/* returns the 13 constants M0,M1,M2,M3,M4,M5,S0..S7 as doubles in that order (14 values) */
void orc_get_idct_constants(double out[14]);

/* Full reference path on a JPEG held in memory: parse markers, entropy-decode,
 * dequantise, IDCT, spread, colour-convert.  Snapshots of every stage are kept. */
int orc_decode_mem(const uint8_t* data, size_t len, int mode, orc_image** out);
int orc_decode_file(const char* path, int mode, orc_image** out);
void orc_free(orc_image* img);
const char* orc_strerror(int code);
/* last error text of the calling thread ("Error! ..." as the reference would have printed) */
const char* orc_last_message(void);

/* geometry (main.go:117-118, 160-172) */
int orc_width(const orc_image*);
int orc_height(const orc_image*);
int orc_components(const orc_image*);
int orc_block_width(const orc_image*);
int orc_block_height(const orc_image*);
int orc_block_width_real(const orc_image*);
int orc_block_height_real(const orc_image*);
int orc_hfactor(const orc_image*, int comp);
int orc_vfactor(const orc_image*, int comp);
int orc_restart_interval(const orc_image*);
size_t orc_bitstream_len(const orc_image*);      /* len(_bitstream), main.go:1028 */

/* Stage snapshots, all laid out like Header.blocks: [blockHeightReal*blockWidthReal][3][64],
 * natural (de-zigzagged) order, int16 (values checked to fit).  Return ORC_OK or <0. */
int orc_get_coef(const orc_image*, int16_t* out);   /* after entropy decode, before dequantise      */
int orc_get_ycc(const orc_image*, int16_t* out);    /* after dequantise + IDCT, before spread        */
int orc_get_rgb(const orc_image*, uint8_t* out);    /* H*W*3, top-down, R,G,B (crop of the block grid) */

/* BMP exactly as writeBitMap lays it out (main.go:1171-1218); returns bytes written or <0.
 * out may be NULL to query the size. */
long orc_bmp_bytes(const orc_image*, uint8_t* out, size_t cap);

/* The pixel pipeline alone (main.go:1054-1057) on caller-supplied coefficients in Header.blocks
 * layout, int16 natural order; used for DRI twins and for timing. */
int orc_pixels_from_coef(const orc_image* geometry_and_tables, const int16_t* coef, uint8_t* rgb_out);

/* Timing hooks for bench.py's cpu_baseline: seconds spent in the in-memory hot path of the last
 * decode: [0] entropy decode (main.go:1032), [1] dequant+IDCT+spread+colour (main.go:1054-1057). */
void orc_last_timing(const orc_image*, double out[2]);

#ifdef __cplusplus
}
#endif
#endif
