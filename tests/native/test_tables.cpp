// CPU check of the Huffman table packers (jd_tables.cpp): every code of a table, followed by every magnitude pattern
// that matters, must decode through each packed form to the symbol, length and value that the canonical code
// assignment of the reference gives (generateCodes, main.go:1156-1169; EXTEND, main.go:196-198).
// Build + run: tests/test_tables.py.  Exit status 0 = all good; failures print one line each.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "jd_tables.h"

struct Code { unsigned code, len, sym; };

static std::vector<Code> canonical(const uint8_t bits[16], const uint8_t* vals) {
    std::vector<Code> out;
    unsigned code = 0; int k = 0;
    for (unsigned len = 1; len <= 16; len++) { for (int i = 0; i < bits[len - 1]; i++) out.push_back({code++, len, vals[k++]}); code <<= 1; }
    return out;
}
static int extend(unsigned v, unsigned size) { return size && v < (1u << (size - 1)) ? (int)v - (int)((1u << size) - 1) : (int)v; }

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

// window = code, then magnitude bits, then filler, left-aligned in 32 bits
static uint32_t window(const Code& c, unsigned mag, unsigned size, uint32_t fill) {
    uint64_t w = ((uint64_t)c.code << size) | mag;
    const unsigned n = c.len + size;
    return (uint32_t)((w << (32 - n)) | (fill >> n));
}

static uint16_t look16(const std::vector<uint16_t>& t, uint32_t win, unsigned lut_bits) {
    uint16_t e = t[win >> (32 - lut_bits)];
    if (e & 0x8000u) e = (e & 0x7FFFu) ? t[(e & 0x7FFFu) + ((win >> (32 - lut_bits - 6)) & 63u)] : 0;
    return e;
}
static uint32_t look32(const std::vector<uint32_t>& t, uint32_t win, unsigned pb) {
    uint32_t e = t[win >> (32 - pb)];
    if (e & 0x80000000u) e = (e & 0x7FFFFFFFu) ? t[(e & 0x7FFFFFFFu) + ((win >> 16) & (0xFFFFu >> pb))] : 0;
    return e == JD_W32_NONE ? 0 : e;                              // "no such code" (jd_tables.cpp)
}

static void check_table(const char* name, const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc) {
    const std::vector<Code> codes = canonical(bits, vals);
    for (unsigned lb : {JD_LUT_BITS, 11, 12}) {
        std::vector<uint16_t> t16, tm;
        CHECK(jd_pack_lut_bits(bits, vals, nvals, is_dc, lb, t16) == 0, "%s pack16 bits %u", name, lb);
        std::vector<uint32_t> t32;
        const unsigned pb = lb == 12 ? 8 : lb == 11 ? 10 : 9;      // writer-table widths exercised: 9, 10, 8
        CHECK(jd_pack_lut32(bits, vals, nvals, is_dc, pb, 2, t32) == 0, "%s pack32", name);
        if (!is_dc) CHECK(jd_pack_lut_multi(bits, vals, nvals, lb, tm) == 0, "%s multi", name);
        for (const Code& c : codes) {
            const unsigned size = is_dc ? c.sym : (c.sym & 15u), run = c.sym >> 4;
            const bool valid = is_dc ? size <= 15 : (size != 0 || c.sym == 0x00 || c.sym == 0xF0);
            const unsigned adv = is_dc ? 1 : (c.sym == 0 ? 64 : run + 1);
            for (uint32_t fill : {0u, 0xFFFFFFFFu, 0x5A5A5A5Au}) {
                const unsigned mags[4] = {0, size ? (1u << size) - 1 : 0, size ? 1u << (size - 1) : 0, size ? (1u << (size - 1)) - 1 : 0};
                for (unsigned mag : mags) {
                    if (c.len + size > 31) continue;
                    const uint32_t win = window(c, mag, size, fill);
                    const uint16_t e = look16(t16, win, lb);
                    const uint32_t w = look32(t32, win, pb);
                    if (!valid) { CHECK(e == 0 && w == 0, "%s sym %02x should be invalid", name, c.sym); continue; }
                    CHECK((e & 31u) == c.len + size && ((e >> 5) & 15u) == size && ((e >> 9) & 63u) + 1 == adv,
                          "%s u16 sym %02x len %u: entry %04x", name, c.sym, c.len, e);
                    // the writer's arithmetic (jd_kernels.hip w32_value / advance), restated
                    const uint32_t m = w >> 16, raw = (win >> (w & 31u)) & m;
                    const int v = (int)raw - (int)(raw <= (w >> 17) ? m : 0u);
                    CHECK(32 - (w & 31u) == c.len + size, "%s u32 sym %02x: bits %u want %u", name, c.sym, 32 - (w & 31u), c.len + size);
                    CHECK(v == extend(mag, size), "%s u32 sym %02x mag %u: value %d want %d", name, c.sym, mag, v, extend(mag, size));
                    const unsigned a32 = (w >> 8) & 255u;
                    CHECK(a32 == (is_dc ? 0u : c.sym == 0 ? JD_W32_EOB_ADV : 2u * (run + 1)), "%s u32 sym %02x: advance %u", name, c.sym, a32);
                }
            }
        }
        if (is_dc) continue;
        // multi-symbol table: a walk over a pseudo-random bit string must land on the single-symbol walk's positions
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        std::vector<uint8_t> bitsv(1 << 16);
        for (auto& b : bitsv) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; b = (uint8_t)(rng & 1); }
        auto win_at = [&](size_t pos) { uint32_t w = 0; for (int i = 0; i < 32; i++) w = (w << 1) | bitsv[(pos + i) % bitsv.size()]; return w; };
        std::vector<char> boundary(bitsv.size() + 64, 0);
        size_t pos = 0; unsigned z = 1;
        std::vector<unsigned> zat(bitsv.size() + 64, 0);
        while (pos < bitsv.size() - 64) {                       // single-symbol walk; z wraps like a block stream
            boundary[pos] = 1; zat[pos] = z;
            uint16_t e = look16(t16, win_at(pos), lb); if (!e) e = 1 | (63u << 9);
            z += ((e >> 9) & 63u) + 1; if (z >= 64) z = 1;
            pos += e & 31u;
        }
        pos = 0; z = 1;
        unsigned groups = 0, multi = 0;
        while (pos < bitsv.size() - 64) {
            CHECK(boundary[pos] && zat[pos] == z, "%s multi walk left the symbol grid at bit %zu (z %u want %u)", name, pos, z, zat[pos]);
            if (!boundary[pos]) break;
            uint16_t e = look16(tm, win_at(pos), lb);
            unsigned total, zn;
            if (!e) { total = 1; zn = 64; }
            else { total = e & 31u; zn = z + ((e >> 5) & 127u); }
            bool ok = zn < 64 || zn == 64 || (zn - z >= 64 && zn < 128);
            if (!ok || !e) {                                     // kernel's fallback: one symbol from the single table
                uint16_t s1 = look16(t16, win_at(pos), lb); if (!s1) s1 = 1 | (63u << 9);
                total = s1 & 31u; zn = z + ((s1 >> 9) & 63u) + 1;
            } else if (total > (unsigned)(look16(t16, win_at(pos), lb) & 31u)) multi++;
            z = zn >= 64 ? 1 : zn;
            pos += total; groups++;
        }
        if (lb == JD_LUT_BITS) printf("%s: %u lookups, %u of them multi-symbol\n", name, groups, multi);
    }
}

// Pair table of the byte-form writer (jd_pack_lut32p): a walk over pseudo-random bits, decoded the way write_segments8
// does (both symbols of an entry, values by wp_value's arithmetic), must produce the single-symbol walk's sequence of
// (advance, value, bits) -- and a fair share of the lookups must resolve two symbols.
static int wp_value(uint32_t win, uint32_t off, uint32_t size) {
    off &= 31u; size &= 31u;
    int y = 0;                                                     // v_bfe_i32
    if (size) { const uint32_t f = (win >> off) & ((1u << size) - 1u); y = (f >> (size - 1)) ? (int)f - (int)(1u << size) : (int)f; }
    const uint32_t m = (1u << size) - 1u;
    return y - (int)(m ^ (uint32_t)(y >> 31));
}
static void check_pairs(const char* name, const uint8_t bits[16], const uint8_t* vals, int nvals) {
    for (unsigned pb : {9u, 10u}) {
        std::vector<uint32_t> tp, t1;
        CHECK(jd_pack_lut32p(bits, vals, nvals, pb, tp) == 0, "%s pack32p", name);
        CHECK(jd_pack_lut32(bits, vals, nvals, false, pb, 1, t1) == 0, "%s pack32", name);
        uint64_t rng = 0x2545F4914F6CDD1Dull;
        std::vector<uint8_t> bitsv(1 << 17);
        for (auto& b : bitsv) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; b = (uint8_t)((rng >> 11) & 1); }
        auto win_at = [&](size_t pos) { uint32_t w = 0; for (int i = 0; i < 32; i++) w = (w << 1) | bitsv[(pos + i) % bitsv.size()]; return w; };
        auto lookp = [&](uint32_t win) {
            uint32_t e = tp[win >> (32 - pb)];
            if ((int)e < 0 && ((e >> 8) & 255u) == 0) { const uint32_t so = (e >> 16) & 0xFFFu; e = so ? tp[so + ((win >> 16) & (0xFFFFu >> pb))] : JD_W32_NONE; if (((e >> 8) & 255u) == 0) e = JD_WP_SKIP; }
            return e;
        };
        size_t p1 = 0, pp = 0;
        unsigned lookups = 0, pairs = 0, symbols = 0;
        while (pp < bitsv.size() - 64 && fails < 20) {
            const uint32_t win = win_at(pp), e = lookp(win);
            const unsigned n = ((e >> 16) & 127u) ? 2 : 1;
            for (unsigned k = 0; k < n; k++) {
                // the single-symbol writer's arithmetic on the same position
                uint32_t w = look32(t1, win_at(p1), pb);
                if (!w) w = 31u | (JD_W32_EOB_ADV8 << 8);                  // invalid code: one bit, end of block (both writers)
                const uint32_t m = w >> 16, raw = (win_at(p1) >> (w & 31u)) & m;
                const int v = (int)raw - (int)(raw <= (w >> 17) ? m : 0u);
                const unsigned adv = (w >> 8) & 255u, nb = 32 - (w & 31u);
                const int vp = k == 0 ? wp_value(win, e >> 23, e >> 28) : wp_value(win, e, (e >> 5) & 7u);
                const unsigned ap = k == 0 ? (e >> 8) & 255u : (e >> 16) & 127u;
                const unsigned endp = k == 0 ? 32 - ((e >> 23) & 31u) : 32 - (e & 31u);
                CHECK(p1 - pp + nb == endp && adv == ap && v == vp, "%s pair table P=%u at bit %zu symbol %u: bits %zu/%u adv %u/%u value %d/%d", name, pb, pp, k, p1 - pp + nb, endp, adv, ap, v, vp);
                p1 += nb; symbols++;
                if (k == 0 && n == 2) CHECK(adv != JD_WP_EOB_ADV && (e >> 28) < 8, "%s pair table pairs behind EOB or a wide symbol", name);
            }
            if (n == 1) CHECK(((e >> 23) & 31u) == (e & 31u) && ((e >> 5) & 7u) == 0, "%s single entry %08x has a second half", name, e);
            pp = p1; lookups++; pairs += n == 2;
        }
        if (pb == 10) printf("%s: pair table, %u symbols in %u lookups (%u pairs) on random bits\n", name, symbols, lookups, pairs);
        CHECK(pairs * 4 > lookups, "%s: too few pairs (%u of %u)", name, pairs, lookups);
    }
}

// Annex K.3 tables (ITU T.81)
static const uint8_t dcl_bits[16] = {0,1,5,1,1,1,1,1,1,0,0,0,0,0,0,0}, dcl_vals[12] = {0,1,2,3,4,5,6,7,8,9,10,11};
static const uint8_t dcc_bits[16] = {0,3,1,1,1,1,1,1,1,1,1,0,0,0,0,0}, dcc_vals[12] = {0,1,2,3,4,5,6,7,8,9,10,11};
static const uint8_t acl_bits[16] = {0,2,1,3,3,2,4,3,5,5,4,4,0,0,1,0x7d};
static const uint8_t acl_vals[162] = {
 0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,0x22,0x71,0x14,0x32,0x81,0x91,0xa1,0x08,0x23,0x42,0xb1,0xc1,0x15,0x52,0xd1,0xf0,
 0x24,0x33,0x62,0x72,0x82,0x09,0x0a,0x16,0x17,0x18,0x19,0x1a,0x25,0x26,0x27,0x28,0x29,0x2a,0x34,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,
 0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x83,0x84,0x85,0x86,0x87,0x88,0x89,
 0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,
 0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe1,0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf1,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
 0xf9,0xfa};
static const uint8_t acc_bits[16] = {0,2,1,2,4,4,3,4,7,5,4,4,0,1,2,0x77};
static const uint8_t acc_vals[162] = {
 0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,0xa1,0xb1,0xc1,0x09,0x23,0x33,0x52,0xf0,
 0x15,0x62,0x72,0xd1,0x0a,0x16,0x24,0x34,0xe1,0x25,0xf1,0x17,0x18,0x19,0x1a,0x26,0x27,0x28,0x29,0x2a,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,
 0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x82,0x83,0x84,0x85,0x86,0x87,
 0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,
 0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
 0xf9,0xfa};

int main() {
    check_table("dc-luma", dcl_bits, dcl_vals, 12, true);
    check_table("dc-chroma", dcc_bits, dcc_vals, 12, true);
    check_table("ac-luma", acl_bits, acl_vals, 162, false);
    check_table("ac-chroma", acc_bits, acc_vals, 162, false);
    check_pairs("ac-luma", acl_bits, acl_vals, 162);
    check_pairs("ac-chroma", acc_bits, acc_vals, 162);
    // a table with a size-0 symbol that is neither EOB nor ZRL (the Go reference panics on it): entry must be invalid
    { uint8_t b[16] = {0,2,0}; uint8_t v[2] = {0x00, 0x30}; check_table("ac-bad-symbol", b, v, 2, false); }
    // malformed tables
    { std::vector<uint16_t> t; uint8_t b[16] = {3,0}; uint8_t v[3] = {0,1,2};
      CHECK(jd_pack_lut(b, v, 3, true, t) == -2, "over-subscribed table accepted");
      CHECK(jd_pack_lut(b, v, 2, true, t) == -1, "count mismatch accepted"); }
    printf("%s (%d failures)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
