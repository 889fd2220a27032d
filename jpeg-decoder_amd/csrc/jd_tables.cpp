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
    return jd_pack_lut_bits(bits, vals, nvals, is_dc, JD_LUT_BITS, out);
}

int jd_pack_lut_bits(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, unsigned lut_bits, std::vector<uint16_t>& out) {
    const unsigned primary = 1u << lut_bits;
    int count = 0;
    for (int i = 0; i < 16; i++) count += bits[i];
    if (count != nvals || nvals > 256) return -1;
    const size_t base = out.size();
    out.resize(base + primary, 0);
    unsigned code = 0;
    int k = 0;
    for (unsigned len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len - 1]; i++, k++) {
            if (code >= (1u << len)) return -2;                       // over-subscribed code space
            const uint16_t e = make_entry(is_dc, vals[k], len);
            if (len <= lut_bits) {
                const unsigned first = code << (lut_bits - len), n = 1u << (lut_bits - len);
                for (unsigned j = 0; j < n; j++) out[base + first + j] = e;
            } else {
                const unsigned prefix = code >> (len - lut_bits);
                uint16_t p = out[base + prefix];
                if (!(p & 0x8000u)) {
                    if (p != 0) return -2;                            // prefix already a complete code
                    const size_t off = out.size() - base;
                    if (off > 0x7FFF) return -3;
                    out.resize(out.size() + JD_LUT_SECONDARY, 0);
                    p = (uint16_t)(0x8000u | off);
                    out[base + prefix] = p;
                }
                // secondary tables are indexed by the 6 bits after the primary ones
                const unsigned sub = (code << (lut_bits + 6 - len)) & 63u, n = 1u << (lut_bits + 6 - len);
                for (unsigned j = 0; j < n; j++) out[base + (p & 0x7FFFu) + sub + j] = e;
            }
            code++;
        }
        code <<= 1;
    }
    return 0;
}

// Sync-pass AC table.  The self-synchronising pass needs no AC values, only how far the stream and the zig-zag
// index move, so its AC entries use their own format and a primary entry covers as many WHOLE run/size symbols
// (code + magnitude bits) as fit in the JD_LUT_BITS-bit window:
//   bit 15 = 0 : [4:0] total bits of the group, [11:5] A + 64*E where A = sum of (run + 1) of the group's
//                symbols other than EOB (<= 63) and E = 1 when the group ends with EOB
//   bit 15 = 1 : secondary table offset, as in the writer's tables;  0 = invalid
// The kernel takes a group only when the block cannot have filled up (index 64 reached without EOB) inside it;
// otherwise it falls back to one symbol from the writer's table.
static inline uint16_t sync_entry(unsigned total, unsigned a, unsigned eob) { return (uint16_t)(total | ((a + 64u * eob) << 5)); }

int jd_pack_lut_multi(const uint8_t bits[16], const uint8_t* vals, int nvals, unsigned lut_bits, std::vector<uint16_t>& out) {
    const unsigned primary = 1u << lut_bits;
    const size_t base = out.size();
    const int rc = jd_pack_lut_bits(bits, vals, nvals, false, lut_bits, out);
    if (rc) return rc;
    const std::vector<uint16_t> single(out.begin() + base, out.end());
    for (size_t i = 0; i < single.size(); i++) {                  // every complete-symbol entry: new format
        const uint16_t e = single[i];
        if (e == 0 || (e & 0x8000u)) continue;
        const unsigned zinc = ((e >> 9) & 63u) + 1u;
        out[base + i] = zinc == 64 ? sync_entry(e & 31u, 0, 1) : sync_entry(e & 31u, zinc, 0);
    }
    for (unsigned w = 0; w < primary; w++) {                      // primary entries: greedy groups
        unsigned pos = 0, a = 0, eob = 0, nsym = 0;
        while (pos < lut_bits && !eob) {
            const uint16_t e = single[(w << pos) & (primary - 1)];
            if (e == 0 || (e & 0x8000u)) break;
            const unsigned total = e & 31u, zinc = ((e >> 9) & 63u) + 1u;
            if (pos + total > lut_bits) break;                    // not wholly inside the known bits
            if (zinc == 64) eob = 1;
            else if (a + zinc > 63) break;
            else a += zinc;
            pos += total; nsym++;
        }
        if (nsym >= 2) out[base + w] = sync_entry(pos, a, eob);
    }
    return 0;
}

// Writer table: the same canonical codes in the 32-bit layout of jd_internal.h.
static inline uint32_t make_entry32(bool is_dc, unsigned sym, unsigned len, unsigned unit) {
    unsigned size, adv;
    if (is_dc) {
        size = sym;
        if (size > 15) return 0;
        adv = 0;
    } else {
        const unsigned run = sym >> 4;
        size = sym & 15u;
        if (sym == 0x00) adv = unit == 2 ? JD_W32_EOB_ADV : JD_W32_EOB_ADV8;
        else if (size == 0 && run != 15) return 0;
        else adv = unit * (run + 1u);
    }
    const unsigned total = len + size;                 // 1 .. 31
    return (32u - total) | (adv << 8) | (((1u << size) - 1u) << 16);
}

int jd_pack_lut32(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, unsigned prim_bits, unsigned unit,
                  std::vector<uint32_t>& out) {
    if (unit != 1 && unit != 2) return -1;
    const unsigned sec_bits = 16u - prim_bits;                   // a secondary table covers the rest of a 16-bit code
    int count = 0;
    for (int i = 0; i < 16; i++) count += bits[i];
    if (count != nvals || nvals > 256) return -1;
    const size_t base = out.size();
    out.resize(base + (1u << prim_bits), 0);
    unsigned code = 0;
    int k = 0;
    for (unsigned len = 1; len <= 16; len++) {
        for (int i = 0; i < bits[len - 1]; i++, k++) {
            if (code >= (1u << len)) return -2;
            const uint32_t e = make_entry32(is_dc, vals[k], len, unit);
            if (len <= prim_bits) {
                const unsigned first = code << (prim_bits - len), n = 1u << (prim_bits - len);
                for (unsigned j = 0; j < n; j++) out[base + first + j] = e;
            } else {
                const unsigned prefix = code >> (len - prim_bits);
                uint32_t p = out[base + prefix];
                if (!(p & 0x80000000u)) {
                    if (p != 0) return -2;
                    const size_t off = out.size() - base;
                    out.resize(out.size() + (1u << sec_bits), 0);
                    p = 0x80000000u | (uint32_t)off;
                    out[base + prefix] = p;
                }
                const unsigned sub = (code << (16 - len)) & ((1u << sec_bits) - 1u), n = 1u << (16 - len);
                for (unsigned j = 0; j < n; j++) out[base + (p & 0x7FFFFFFFu) + sub + j] = e;
            }
            code++;
        }
        code <<= 1;
    }
    // "no such code" is a secondary-table pointer to nowhere: every entry the decode loop must look at twice (secondary
    // table, no code, magnitude category >= 8) is then >= 2^23 as an unsigned number -- one compare per symbol
    for (size_t i = base; i < out.size(); i++) if (out[i] == 0) out[i] = JD_W32_NONE;
    return 0;
}

// Pair table of the byte-form writer (write_segments8; layout in jd_internal.h, JD_WP_*): an entry resolves the symbol
// whose code starts the index bits and, when the CODE of the symbol after it also lies wholly inside them, that second
// symbol too -- its magnitude bits may reach past the index, the decoder reads them from its 32-bit window.  At quality
// 85 with the Annex-K tables two entries in three are pairs and a lookup resolves 1.4 symbols.  Never paired: a symbol
// after EOB (the next block starts with another table), a symbol of magnitude category >= 8 (the decoder keeps its
// high part through a rare path that knows one symbol per entry), and codes longer than the primary index.
int jd_pack_lut32p(const uint8_t bits[16], const uint8_t* vals, int nvals, unsigned prim_bits, std::vector<uint32_t>& out) {
    if (prim_bits < 8 || prim_bits > 12) return -1;
    const unsigned sec_bits = 16u - prim_bits;
    int count = 0;
    for (int i = 0; i < 16; i++) count += bits[i];
    if (count != nvals || nvals > 256) return -1;
    struct Code { unsigned code, len, sym; };
    std::vector<Code> codes;
    {
        unsigned code = 0; int k = 0;
        for (unsigned len = 1; len <= 16; len++) {
            for (int i = 0; i < bits[len - 1]; i++, k++) {
                if (code >= (1u << len)) return -2;
                codes.push_back({ code, len, vals[k] });
                code++;
            }
            code <<= 1;
        }
    }
    // (advance, size) of a run/size symbol; advance 0 = not a symbol the baseline decoder knows (Go: negative shift panic)
    auto adv_of = [](unsigned sym) -> unsigned {
        if (sym == 0x00) return JD_WP_EOB_ADV;
        if ((sym & 15u) == 0 && (sym >> 4) != 15u) return 0;
        return (sym >> 4) + 1u;
    };
    auto single = [&](unsigned sym, unsigned len) -> uint32_t {
        const unsigned adv = adv_of(sym), size = sym & 15u;
        if (!adv) return 0;
        const unsigned sh = 32u - (len + size);                     // 1 .. 31
        return sh | (adv << 8) | (sh << 23) | (size << 28);
    };
    const size_t base = out.size();
    out.resize(base + (1u << prim_bits), 0);
    for (const Code& c : codes) {
        const uint32_t e1 = single(c.sym, c.len);
        if (c.len <= prim_bits) {
            const unsigned first = c.code << (prim_bits - c.len), n = 1u << (prim_bits - c.len);
            const unsigned size1 = c.sym & 15u, bits1 = c.len + size1;
            for (unsigned j = 0; j < n; j++) {
                uint32_t e = e1;
                if (e1 && c.sym != 0x00 && size1 < 8 && bits1 < prim_bits) {
                    const unsigned rest = prim_bits - bits1, tail = (first + j) & ((1u << rest) - 1u);   // index bits behind symbol 1
                    for (const Code& d : codes) {
                        if (d.len > rest) break;                                   // (codes are in order of length)
                        if ((tail >> (rest - d.len)) != d.code) continue;
                        const unsigned adv2 = adv_of(d.sym), size2 = d.sym & 15u;
                        if (adv2 && size2 < 8)
                            e = (32u - (bits1 + d.len + size2)) | (size2 << 5) | (adv_of(c.sym) << 8) | (adv2 << 16) | ((32u - bits1) << 23) | (size1 << 28);
                        break;
                    }
                }
                out[base + first + j] = e;
            }
        } else {
            const unsigned prefix = c.code >> (c.len - prim_bits);
            uint32_t p = out[base + prefix];
            if (!(p & 0x80000000u)) {
                if (p != 0) return -2;
                const size_t off = out.size() - base;
                if (off >= 4096) return -3;
                out.resize(out.size() + (1u << sec_bits), 0);
                p = 0x80000000u | ((uint32_t)off << 16);                          // (adv1 byte 0: not a symbol)
                out[base + prefix] = p;
            }
            const unsigned sub = (c.code << (16 - c.len)) & ((1u << sec_bits) - 1u), n = 1u << (16 - c.len);
            for (unsigned j = 0; j < n; j++) out[base + ((p >> 16) & 0xFFFu) + sub + j] = e1;
        }
    }
    for (size_t i = base; i < out.size(); i++) if (out[i] == 0) out[i] = JD_W32_NONE;
    return 0;
}

// Symbol table for the progressive scans, laid out for a decoder that runs on the SCALAR unit and keeps the table in
// vector registers (kp_scan): 7 rows of 64 dwords, dword l of a row = what lane l holds.
//   rows 0-3  primary table indexed by the next 8 bits of the stream: code length << 8 | symbol; 0 = the code is longer
//   row 4     lane L-1: the largest code of length L (T.81 F.2.2.3 MAXCODE), -1 if there is none
//   row 5     lane L-1: index of the first symbol of length L minus its code (VALPTR - MINCODE)
//   row 6     the symbols, four to a dword
int jd_pack_lut_lane(const uint8_t bits[16], const uint8_t* vals, int nvals, std::vector<uint32_t>& out) {
    int count = 0;
    for (int i = 0; i < 16; i++) count += bits[i];
    if (count != nvals || nvals > 256) return -1;
    const size_t base = out.size();
    out.resize(base + JD_PLUT_DWORDS, 0);
    unsigned code = 0;
    int k = 0;
    for (unsigned len = 1; len <= 16; len++) {
        out[base + 256 + len - 1] = 0xFFFFFFFFu;
        out[base + 320 + len - 1] = (uint32_t)(k - (int)code);
        for (int i = 0; i < bits[len - 1]; i++, k++) {
            if (code >= (1u << len)) return -2;
            if (len <= 8) {
                const unsigned first = code << (8 - len), n = 1u << (8 - len);
                for (unsigned j = 0; j < n; j++) out[base + first + j] = (uint32_t)vals[k] | (len << 8);
            }
            out[base + 256 + len - 1] = code;
            code++;
        }
        code <<= 1;
    }
    for (int i = 0; i < nvals; i++) out[base + 384 + (i >> 2)] |= (uint32_t)vals[i] << ((i & 3) * 8);
    return 0;
}
