// jd_tables.cpp -- host-side packing of the Huffman lookup tables that are uploaded to HBM.
// Replaces generateCodes (main.go:1156-1169): the canonical code assignment is the same (code++ per symbol,
// code <<= 1 per length); instead of a codes[] array searched bit by bit (scanSymbol, main.go:1297-1316) the
// codes index a JD_LUT_BITS-bit primary table plus 64-entry secondary tables for longer codes.
#include "jd_tables.h"

#include <string.h>

static inline uint16_t make_entry(bool is_dc, unsigned sym, unsigned len) {
    unsigned size, zinc;
    if (is_dc) {
        size = sym;                      // DC: category == symbol (main.go:193-195)
        if (size > 15) return 0;
        zinc = 1;
    } else {
        const unsigned run = sym >> 4;
        size = sym & 15u;
        if (sym == 0x00) zinc = 64;      // EOB: rest of the block is zero (main.go:211-215)
        else if (size == 0 && run != 15) return 0;   // Go: negative shift panic (Appendix C) -> invalid
        else zinc = run + 1;             // ZRL (0xF0) is run 15 / size 0: advance 16 (main.go:217-222)
    }
    const unsigned total = len + size;   // <= 31
    return (uint16_t)(total | (size << 5) | ((zinc - 1) << 9));
}

int jd_pack_lut(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, std::vector<uint16_t>& out) {
    int count = 0;
    for (int i = 0; i < 16; i++) count += bits[i];
    if (count != nvals || nvals > 256) return -1;
    const size_t base = out.size();
    out.resize(base + JD_LUT_PRIMARY, 0);
    unsigned code = 0;
    int k = 0;
    for (unsigned len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len - 1]; i++, k++) {
            if (code >= (1u << len)) return -2;                       // over-subscribed code space
            const uint16_t e = make_entry(is_dc, vals[k], len);
            if (len <= JD_LUT_BITS) {
                const unsigned first = code << (JD_LUT_BITS - len), n = 1u << (JD_LUT_BITS - len);
                for (unsigned j = 0; j < n; j++) out[base + first + j] = e;
            } else {
                const unsigned prefix = code >> (len - JD_LUT_BITS);
                uint16_t p = out[base + prefix];
                if (!(p & 0x8000u)) {
                    if (p != 0) return -2;                            // prefix already a complete code
                    const size_t off = out.size() - base;
                    if (off > 0x7FFF) return -3;
                    out.resize(out.size() + JD_LUT_SECONDARY, 0);
                    p = (uint16_t)(0x8000u | off);
                    out[base + prefix] = p;
                }
                const unsigned sub = (code << (16 - len)) & 63u, n = 1u << (16 - len);
                for (unsigned j = 0; j < n; j++) out[base + (p & 0x7FFFu) + sub + j] = e;
            }
            code++;
        }
        code <<= 1;
    }
    return 0;
}
