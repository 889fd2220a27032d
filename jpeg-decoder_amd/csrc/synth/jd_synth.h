/* jd_synth.h -- synthetic image generator + baseline JPEG encoder (bench/test workload maker). */
#ifndef JD_SYNTH_H
#define JD_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct jds_image jds_image;
/* Noise image -> FDCT -> quantised coefficients.  hs,vs in {1,2} are the luma sampling factors (chroma 1x1),
 * ncomp in {1,3}, quality 1..100 (libjpeg scaling of Annex K.1/K.2), detail 0..3 picks the octave weights (q85 4:2:0: about 0.18 / 0.30 / 0.44 / 0.55 B/px).
 * Returns 0, -1 bad argument, -2 out of memory. */
int jds_create(int width, int height, int hs, int vs, int ncomp, int quality, uint64_t seed, int detail,
               jds_image** out);
void jds_destroy(jds_image*);
int jds_mcus(const jds_image*);
int jds_blocks_per_mcu(const jds_image*);
/* quantised coefficients, scan order [mcu][block-in-MCU][64], zig-zag order, absolute DC */
int jds_coefficients(const jds_image*, int16_t* out);
void jds_quant_tables(const jds_image*, uint8_t* out128);   /* [2][64] natural order */
/* Entropy-code into a complete baseline JPEG file (malloc'd; release with jds_free_buffer).
 * restart_interval 0 = no DRI/RSTn; >0 = DRI + RSTn every that many MCUs ("DRI twin": same coefficients). */
int jds_encode(const jds_image*, int restart_interval, uint8_t** out, size_t* out_len);
/* Variants of the file layout for the parser / decoder corner cases (same coefficients):
 *   JDS_SHARED_TABLES   every component codes through Huffman tables 0/0 (one DC + one AC table in the file)
 *   JDS_ZERO_BASED_IDS  component ids 0,1,2 instead of 1,2,3 (main.go:133-135, 153-157, 961-963)
 *   JDS_DQT_16BIT       16-bit DQT entries with a non-zero high byte, which the reference drops (main.go:60-68)
 *   JDS_RANDOM_TABLES   random valid Huffman tables instead of Annex K.3's (code lengths 1..16, symbols in random
 *                       order); flags >> 8 seeds them */
enum { JDS_SHARED_TABLES = 1u, JDS_ZERO_BASED_IDS = 2u, JDS_DQT_16BIT = 4u, JDS_RANDOM_TABLES = 8u };
int jds_encode_ex(const jds_image*, int restart_interval, unsigned flags, uint8_t** out, size_t* out_len);
void jds_free_buffer(uint8_t*);
#ifdef __cplusplus
}
#endif
#endif
