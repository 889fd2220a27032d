/*
 * jd_synth.c -- deterministic synthetic image generator + baseline (SOF0) JPEG encoder.
 *
 * Bench / test workload maker for SURVEY.md section 8(d) configs C3-C5 (the reference has no
 * encoder and no benchmark inputs).  Host-only C, no dependencies.
 *
 *   pixels : integer multi-octave value noise, lattice values from splitmix64(seed, ix, iy, comp, octave),
 *            bilinear in 8.8 fixed point -- bit-identical wherever this file is compiled.
 *   coding : forward DCT (float64, no contraction), Annex K.1/K.2 quantisation tables scaled with the
 *            libjpeg quality rule, Annex K.3 Huffman tables, optional DRI/RSTn.  The quantised
 *            coefficients are kept, so one image can be entropy-coded with and without restart markers
 *            ("DRI twins") and the coefficients themselves are ground truth for the
 *            Huffman stage.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "jd_synth.h"

static const uint8_t K1_LUMA[64] = {
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
static const uint8_t K2_CHROMA[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };

/* Annex K.3 typical Huffman tables: BITS then HUFFVAL */
static const uint8_t DC_LUM_BITS[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t DC_CHR_BITS[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t DC_VALS[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t AC_LUM_BITS[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t AC_LUM_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa };
static const uint8_t AC_CHR_BITS[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t AC_CHR_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa };

static int ZZ[64];
static double COSM[8][8];       /* COSM[u][x] = C(u)/2 * cos((2x+1)u*pi/16) */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_tables(void) {
    int r = 0, c = 0;
    for (int i = 0; i < 64; i++) {
        ZZ[i] = r * 8 + c;
        if (((r + c) & 1) == 0) { if (c == 7) r++; else if (r == 0) c++; else { r--; c++; } }
        else { if (r == 7) c++; else if (c == 0) r++; else { r++; c--; } }
    }
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            COSM[u][x] = (u == 0 ? sqrt(0.5) : 1.0) * 0.5 * cos((2 * x + 1) * u * M_PI / 16.0);
}

struct jds_image {
    int width, height, hs, vs, ncomp, quality;
    int mcux, mcuy, bpm;
    uint8_t qt[2][64];            /* natural order */
    int16_t* coef;                /* scan order [mcu][blk][64], zig-zag order, absolute DC */
};

/* ---- noise ---------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline int lattice(uint64_t seed, int comp, int oct, int ix, int iy) {
    uint64_t k = seed * 0x100000001B3ull + ((uint64_t)comp << 60) + ((uint64_t)oct << 56) +
                 ((uint64_t)(uint32_t)iy << 28) + (uint64_t)(uint32_t)ix;
    return (int)(splitmix64(k) & 0xFF);
}
/* value in [0,255]: octaves with cell sizes 64/16/4/1 (in luma pixels), weights wt[0..3] (sum 32) */
static int noise_px(uint64_t seed, int comp, int x, int y, int shift_x, int shift_y, const int* wt) {
    static const int cell_log2[4] = { 6, 4, 2, 0 };
    int acc = 0;
    /* chroma planes sample the luma-resolution noise field at the top-left of their footprint */
    int X = x << shift_x, Y = y << shift_y;
    for (int o = 0; o < 4; o++) {
        int s = cell_log2[o];
        if (wt[o] == 0) continue;
        int ix = X >> s, iy = Y >> s;
        int fx = ((X & ((1 << s) - 1)) << 8) >> s, fy = ((Y & ((1 << s) - 1)) << 8) >> s;   /* 0..255 */
        int v00 = lattice(seed, comp, o, ix, iy), v10 = lattice(seed, comp, o, ix + 1, iy);
        int v01 = lattice(seed, comp, o, ix, iy + 1), v11 = lattice(seed, comp, o, ix + 1, iy + 1);
        int top = v00 * (256 - fx) + v10 * fx, bot = v01 * (256 - fx) + v11 * fx;        /* 8.8 */
        int v = (top * (256 - fy) + bot * fy) >> 16;                                     /* 0..255 */
        acc += v * wt[o];
    }
    return acc >> 5;
}

/* ---- forward DCT + quantisation of one block ------------------------------------------------ */
static void fdct_quant(const int* px /*64, level-shifted*/, const uint8_t* q, int16_t* out_zz) {
    double tmp[64], f[64];
    for (int u = 0; u < 8; u++)              /* rows: tmp[y][u] */
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int x = 0; x < 8; x++) s += COSM[u][x] * (double)px[y * 8 + x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            double s = 0;
            for (int y = 0; y < 8; y++) s += COSM[v][y] * tmp[y * 8 + u];
            f[v * 8 + u] = s;
        }
    for (int i = 0; i < 64; i++) {
        int nat = ZZ[i];
        double r = f[nat] / (double)q[nat];
        int k = (int)(r < 0 ? r - 0.5 : r + 0.5);
        if (k > 1023) k = 1023;
        if (k < -1023) k = -1023;
        out_zz[i] = (int16_t)k;
    }
}

typedef struct { jds_image* im; uint64_t seed; int row0, row1; const int* wt; } work_t;

static void* worker(void* arg) {
    work_t* w = (work_t*)arg;
    jds_image* im = w->im;
    int px[64];
    for (int my = w->row0; my < w->row1; my++)
        for (int mx = 0; mx < im->mcux; mx++) {
            int16_t* dst = im->coef + ((size_t)my * im->mcux + mx) * im->bpm * 64;
            int b = 0;
            for (int v = 0; v < im->vs; v++)
                for (int u = 0; u < im->hs; u++, b++) {
                    int x0 = (mx * im->hs + u) * 8, y0 = (my * im->vs + v) * 8;
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++)
                            px[y * 8 + x] = noise_px(w->seed, 0, x0 + x, y0 + y, 0, 0, w->wt) - 128;
                    fdct_quant(px, im->qt[0], dst + b * 64);
                }
            int sx = im->hs == 2, sy = im->vs == 2;
            for (int c = 1; c < im->ncomp; c++, b++) {
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++)
                        px[y * 8 + x] = noise_px(w->seed, c, mx * 8 + x, my * 8 + y, sx, sy, w->wt) - 128;
                fdct_quant(px, im->qt[1], dst + b * 64);
            }
        }
    return NULL;
}

int jds_create(int width, int height, int hs, int vs, int ncomp, int quality, uint64_t seed,
               int detail, jds_image** out) {
    *out = NULL;
    if (width < 1 || height < 1 || width > 65535 || height > 65535) return -1;
    if (!(hs == 1 || hs == 2) || !(vs == 1 || vs == 2)) return -1;
    if (!(ncomp == 1 || ncomp == 3)) return -1;
    if (ncomp == 1 && (hs != 1 || vs != 1)) return -1;
    if (quality < 1 || quality > 100) return -1;
    pthread_once(&g_once, init_tables);
    jds_image* im = (jds_image*)calloc(1, sizeof *im);
    if (!im) return -2;
    im->width = width; im->height = height; im->hs = hs; im->vs = vs; im->ncomp = ncomp; im->quality = quality;
    im->mcux = (width + 8 * hs - 1) / (8 * hs);
    im->mcuy = (height + 8 * vs - 1) / (8 * vs);
    im->bpm = hs * vs + (ncomp - 1);
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; i++) {
        int a = (K1_LUMA[i] * scale + 50) / 100, b = (K2_CHROMA[i] * scale + 50) / 100;
        im->qt[0][i] = (uint8_t)(a < 1 ? 1 : a > 255 ? 255 : a);
        im->qt[1][i] = (uint8_t)(b < 1 ? 1 : b > 255 ? 255 : b);
    }
    size_t n = (size_t)im->mcux * im->mcuy * im->bpm * 64;
    im->coef = (int16_t*)malloc(n * sizeof(int16_t));
    if (!im->coef) { free(im); return -2; }
    /* octave weights (sum 32): detail 0 = smooth ... 3 = busy */
    static const int WT[4][4] = { { 20, 8, 4, 0 }, { 14, 9, 7, 2 }, { 12, 8, 8, 4 }, { 8, 8, 10, 6 } };
    const int* wt = WT[detail < 0 ? 0 : detail > 3 ? 3 : detail];
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = (int)(ncpu < 1 ? 1 : ncpu > 16 ? 16 : ncpu);
    if (nt > im->mcuy) nt = im->mcuy;
    pthread_t th[16]; work_t wk[16];
    for (int t = 0; t < nt; t++) {
        wk[t].im = im; wk[t].seed = seed; wk[t].wt = wt;
        wk[t].row0 = (int)((long)im->mcuy * t / nt);
        wk[t].row1 = (int)((long)im->mcuy * (t + 1) / nt);
        if (nt == 1 || pthread_create(&th[t], NULL, worker, &wk[t]) != 0) { worker(&wk[t]); th[t] = 0; wk[t].row0 = -1; }
    }
    for (int t = 0; t < nt; t++) if (wk[t].row0 >= 0) pthread_join(th[t], NULL);
    *out = im;
    return 0;
}

void jds_destroy(jds_image* im) { if (im) { free(im->coef); free(im); } }
int jds_mcus(const jds_image* im) { return im->mcux * im->mcuy; }
int jds_blocks_per_mcu(const jds_image* im) { return im->bpm; }
int jds_coefficients(const jds_image* im, int16_t* out) {
    memcpy(out, im->coef, (size_t)im->mcux * im->mcuy * im->bpm * 64 * sizeof(int16_t));
    return 0;
}
void jds_quant_tables(const jds_image* im, uint8_t* out128) { memcpy(out128, im->qt, 128); }

/* ---- entropy coder ---------------------------------------------------------------------------- */
typedef struct { uint16_t code[256]; uint8_t len[256]; } enc_table;

static void build_enc(const uint8_t* bits, const uint8_t* vals, enc_table* t) {
    memset(t, 0, sizeof *t);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) { t->code[vals[k]] = (uint16_t)code; t->len[vals[k]] = (uint8_t)l; code++; }
        code <<= 1;
    }
}

typedef struct { uint8_t* p; size_t n, cap; uint64_t acc; int nacc; int fail; } bitw;

static void bw_byte(bitw* w, uint8_t b) {
    if (w->n + 2 > w->cap) {
        size_t nc = w->cap * 2 + 4096;
        uint8_t* q = (uint8_t*)realloc(w->p, nc);
        if (!q) { w->fail = 1; return; }
        w->p = q; w->cap = nc;
    }
    w->p[w->n++] = b;
}
static void bw_ecs_byte(bitw* w, uint8_t b) { bw_byte(w, b); if (b == 0xFF) bw_byte(w, 0x00); }
static void bw_bits(bitw* w, unsigned v, int n) {
    if (n == 0) return;
    w->acc = (w->acc << n) | (v & ((1u << n) - 1));
    w->nacc += n;
    while (w->nacc >= 8) { bw_ecs_byte(w, (uint8_t)(w->acc >> (w->nacc - 8))); w->nacc -= 8; }
}
static void bw_flush_ones(bitw* w) {
    if (w->nacc > 0) { int pad = 8 - w->nacc; bw_bits(w, (1u << pad) - 1, pad); }
    w->acc = 0; w->nacc = 0;
}
static void bw_u16(bitw* w, unsigned v) { bw_byte(w, (uint8_t)(v >> 8)); bw_byte(w, (uint8_t)v); }

static inline int bit_size(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }

static void encode_block(bitw* w, const int16_t* zz, int* pred, const enc_table* dc, const enc_table* ac) {
    int diff = zz[0] - *pred;
    *pred = zz[0];
    int s = bit_size(diff);
    bw_bits(w, dc->code[s], dc->len[s]);
    if (s) bw_bits(w, (unsigned)(diff < 0 ? diff + (1 << s) - 1 : diff), s);
    int run = 0;
    for (int i = 1; i < 64; i++) {
        int v = zz[i];
        if (v == 0) { run++; continue; }
        while (run > 15) { bw_bits(w, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        s = bit_size(v);
        int sym = (run << 4) | s;
        bw_bits(w, ac->code[sym], ac->len[sym]);
        bw_bits(w, (unsigned)(v < 0 ? v + (1 << s) - 1 : v), s);
        run = 0;
    }
    if (run > 0) bw_bits(w, ac->code[0x00], ac->len[0x00]);
}

static void put_dht(bitw* w, int tc_th, const uint8_t* bits, const uint8_t* vals, int nvals) {
    bw_byte(w, (uint8_t)tc_th);
    for (int i = 0; i < 16; i++) bw_byte(w, bits[i]);
    for (int i = 0; i < nvals; i++) bw_byte(w, vals[i]);
}

/* JDS_RANDOM_TABLES: a random valid Huffman table over the symbol set `syms` -- a random full binary tree grown leaf by
 * leaf (half of the splits go to a deepest leaf, so codes of up to 16 bits are common), one deepest leaf left unused
 * (T.81 C.2: no code of all ones), the symbols dealt over the leaves in random order.  What the decoders' table
 * packers see of it: primary / secondary splits at every length, pair entries, shortest codes of 1 bit. */
static uint64_t tab_rnd(uint64_t* st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void random_table(uint64_t seed, const uint8_t* syms, int nsym, uint8_t bits[16], uint8_t* vals) {
    uint64_t st = seed;
    int depth[260], n = 1;
    depth[0] = 0;
    while (n < nsym + 1) {
        int pick = -1;
        if (tab_rnd(&st) & 1) {                                   /* a deepest leaf that may still be split */
            for (int i = 0; i < n; i++) if (depth[i] < 16 && (pick < 0 || depth[i] > depth[pick])) pick = i;
        } else {
            for (int tries = 0; tries < 64 && pick < 0; tries++) { int i = (int)(tab_rnd(&st) % (uint64_t)n); if (depth[i] < 16) pick = i; }
        }
        if (pick < 0) for (int i = 0; i < n; i++) if (depth[i] < 16 && (pick < 0 || depth[i] < depth[pick])) pick = i;
        depth[pick]++; depth[n++] = depth[pick];
    }
    int deepest = 0;
    for (int i = 1; i < n; i++) if (depth[i] > depth[deepest]) deepest = i;
    depth[deepest] = depth[n - 1]; n--;                           /* the unused leaf: canonical order puts all ones there */
    memset(bits, 0, 16);
    for (int i = 0; i < n; i++) bits[depth[i] - 1]++;
    for (int i = 0; i < nsym; i++) vals[i] = syms[i];
    for (int i = nsym - 1; i > 0; i--) { int j = (int)(tab_rnd(&st) % (uint64_t)(i + 1)); uint8_t t = vals[i]; vals[i] = vals[j]; vals[j] = t; }
}

int jds_encode(const jds_image* im, int restart_interval, uint8_t** out, size_t* out_len) {
    return jds_encode_ex(im, restart_interval, 0, out, out_len);
}

int jds_encode_ex(const jds_image* im, int restart_interval, unsigned flags, uint8_t** out, size_t* out_len) {
    const int shared = (flags & JDS_SHARED_TABLES) != 0, zero_ids = (flags & JDS_ZERO_BASED_IDS) != 0;
    const int wide_q = (flags & JDS_DQT_16BIT) != 0;
    *out = NULL; *out_len = 0;
    if (restart_interval < 0 || restart_interval > 65535) return -1;
    bitw w; memset(&w, 0, sizeof w);
    enc_table dcl, dcc, acl, acc_;
    uint8_t tb[4][16], tv[4][162];                              /* DC luma, AC luma, DC chroma, AC chroma as they go into the file */
    memcpy(tb[0], DC_LUM_BITS, 16); memcpy(tv[0], DC_VALS, 12); memcpy(tb[1], AC_LUM_BITS, 16); memcpy(tv[1], AC_LUM_VALS, 162);
    memcpy(tb[2], DC_CHR_BITS, 16); memcpy(tv[2], DC_VALS, 12); memcpy(tb[3], AC_CHR_BITS, 16); memcpy(tv[3], AC_CHR_VALS, 162);
    if (flags & JDS_RANDOM_TABLES) {
        const uint64_t ts = (uint64_t)(flags >> 8) * 4u + 1u;
        random_table(ts, DC_VALS, 12, tb[0], tv[0]); random_table(ts + 1, AC_LUM_VALS, 162, tb[1], tv[1]);
        random_table(ts + 2, DC_VALS, 12, tb[2], tv[2]); random_table(ts + 3, AC_LUM_VALS, 162, tb[3], tv[3]);
    }
    build_enc(tb[0], tv[0], &dcl); build_enc(tb[2], tv[2], &dcc);
    build_enc(tb[1], tv[1], &acl); build_enc(tb[3], tv[3], &acc_);
    /* SOI, APP0/JFIF */
    bw_u16(&w, 0xFFD8);
    bw_u16(&w, 0xFFE0); bw_u16(&w, 16);
    { const uint8_t j[14] = { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 }; for (int i = 0; i < 14; i++) bw_byte(&w, j[i]); }
    /* DQT (zig-zag order in the file) */
    int nq = im->ncomp == 1 ? 1 : 2;
    bw_u16(&w, 0xFFDB); bw_u16(&w, (unsigned)(2 + (wide_q ? 129 : 65) * nq));
    for (int t = 0; t < nq; t++) {
        bw_byte(&w, (uint8_t)(t | (wide_q ? 0x10 : 0)));
        /* 16-bit form: a non-zero high byte that the reference drops (main.go:60-68) */
        for (int i = 0; i < 64; i++) { if (wide_q) bw_byte(&w, (uint8_t)(1 + (i & 3))); bw_byte(&w, im->qt[t][ZZ[i]]); }
    }
    /* SOF0 */
    bw_u16(&w, 0xFFC0); bw_u16(&w, (unsigned)(8 + 3 * im->ncomp)); bw_byte(&w, 8);
    bw_u16(&w, (unsigned)im->height); bw_u16(&w, (unsigned)im->width); bw_byte(&w, (uint8_t)im->ncomp);
    for (int c = 0; c < im->ncomp; c++) {
        bw_byte(&w, (uint8_t)(c + (zero_ids ? 0 : 1)));
        bw_byte(&w, c == 0 ? (uint8_t)((im->hs << 4) | im->vs) : 0x11);
        bw_byte(&w, c == 0 ? 0 : 1);
    }
    /* DHT: all tables in one segment */
    int ntab = (im->ncomp == 1 || shared) ? 2 : 4;
    bw_u16(&w, 0xFFC4);
    bw_u16(&w, (unsigned)(2 + (17 + 12) + (17 + 162) + (ntab == 4 ? (17 + 12) + (17 + 162) : 0)));
    put_dht(&w, 0x00, tb[0], tv[0], 12);
    put_dht(&w, 0x10, tb[1], tv[1], 162);
    if (ntab == 4) { put_dht(&w, 0x01, tb[2], tv[2], 12); put_dht(&w, 0x11, tb[3], tv[3], 162); }
    if (restart_interval > 0) { bw_u16(&w, 0xFFDD); bw_u16(&w, 4); bw_u16(&w, (unsigned)restart_interval); }
    /* SOS */
    bw_u16(&w, 0xFFDA); bw_u16(&w, (unsigned)(6 + 2 * im->ncomp)); bw_byte(&w, (uint8_t)im->ncomp);
    for (int c = 0; c < im->ncomp; c++) { bw_byte(&w, (uint8_t)(c + (zero_ids ? 0 : 1))); bw_byte(&w, (c == 0 || shared) ? 0x00 : 0x11); }
    bw_byte(&w, 0); bw_byte(&w, 63); bw_byte(&w, 0);
    /* ECS */
    int pred[3] = { 0, 0, 0 };
    long nmcu = (long)im->mcux * im->mcuy;
    int nlum = im->hs * im->vs, rst = 0;
    for (long m = 0; m < nmcu && !w.fail; m++) {
        if (restart_interval > 0 && m > 0 && m % restart_interval == 0) {
            bw_flush_ones(&w);
            bw_byte(&w, 0xFF); bw_byte(&w, (uint8_t)(0xD0 + (rst & 7))); rst++;
            pred[0] = pred[1] = pred[2] = 0;
        }
        const int16_t* blk = im->coef + (size_t)m * im->bpm * 64;
        for (int b = 0; b < im->bpm; b++) {
            int comp = b < nlum ? 0 : b - nlum + 1;
            encode_block(&w, blk + b * 64, &pred[comp], (comp && !shared) ? &dcc : &dcl, (comp && !shared) ? &acc_ : &acl);
        }
    }
    bw_flush_ones(&w);
    bw_u16(&w, 0xFFD9);
    if (w.fail) { free(w.p); return -2; }
    *out = w.p; *out_len = w.n;
    return 0;
}

void jds_free_buffer(uint8_t* p) { free(p); }
