// jd_host.cpp -- C++ twin of the Go host code either side of the hot path (the Go toolchain is absent from
// this image, so the test harness and the CLI need one): the marker parser and the BMP writer.
//
//   jd_parse       decodeJPEG + decodeAPPN/skipMarker + decodeQuantizationTables + decodeStartOfFrame +
//                  decodeDefineRestartInterval + decodeDefineHuffmanTable + header part of decodeStartOfScan
//                  (main.go:1076-1154, 30-40, 1065-1074, 42-85, 87-180, 818-832, 834-888, 942-987)
//   jd_bmp_encode  writeBitMap / put4Int / put2Int (main.go:1171-1244)
//
// Unlike the reference's one-syscall-per-byte Buffer (main.go:12-28) this walks a memory image of the file,
// and unlike its ECS loop (main.go:993-1021) it does not copy or unstuff: it only finds where the segment
// ends, keeping FF00 and RSTn in place for the device.
#include <string.h>
#include <vector>

#include "../../include/jpeghip.h"

namespace {


struct ZZ {
    uint8_t nat[64];
    ZZ() {
        int r = 0, c = 0;
        for (int i = 0; i < 64; i++) {
            nat[i] = (uint8_t)(r * 8 + c);
            if (((r + c) & 1) == 0) { if (c == 7) r++; else if (r == 0) c++; else { r--; c++; } }
            else { if (r == 7) c++; else if (c == 0) r++; else { r++; c--; } }
        }
    }
};

struct Reader {
    const uint8_t* p; size_t n, i;
    bool ok(size_t k) const { return i + k <= n; }
    int u8() { return p[i++]; }
    int u16() { int v = (p[i] << 8) | p[i + 1]; i += 2; return v; }
};

}  // namespace

// One parser for both frame types.  scans == nullptr: the baseline form (jd_parse): SOF0 only, exactly one scan, its
// extent returned through ecs_off / ecs_len.
static int parse_impl(const uint8_t* file, size_t file_len, jd_frame* fr, jd_scan* scans, int cap, int* n_scans,
                      size_t* ecs_off, size_t* ecs_len) {
    static const ZZ zz;
    memset(fr, 0, sizeof *fr);
    Reader r{ file, file_len, 0 };
    if (!r.ok(2) || file[0] != 0xFF || file[1] != 0xD8) return JD_E_MARKER;       // SOI (main.go:1095)
    r.i = 2;
    int comp_id[3] = { 0, 0, 0 };
    bool have_sof = false, zero_based = false, progressive = false;
    int ns_found = 0;
    for (;;) {
        if (!r.ok(2)) return JD_E_TRUNCATED;
        if (r.u8() != 0xFF) return JD_E_MARKER;                                   // main.go:1103-1106
        int m = r.u8();
        while (m == 0xFF) { if (!r.ok(1)) return JD_E_TRUNCATED; m = r.u8(); }     // fill bytes (main.go:1108-1110)
        if (m == 0x01) continue;                                                  // TEM: no payload
        if (m == 0xD9) {                                                          // EOI
            if (ns_found == 0) return JD_E_MARKER;                                // EOI before SOS (main.go:1131-1134)
            if (n_scans) *n_scans = ns_found;
            return JD_OK;
        }
        if (m == 0xD8) return JD_E_MARKER;                                        // embedded SOI
        if (ns_found && !(m == 0xDD || m == 0xC4 || m == 0xDA)) return JD_E_MARKER;   // after a scan the reference only
                                                                                  // knows DRI / DHT / SOS / EOI (main.go:1041-1053)
        if (!r.ok(2)) return JD_E_TRUNCATED;
        const int seg = r.u16();
        if (seg < 2 || !r.ok((size_t)seg - 2)) return JD_E_TRUNCATED;
        const size_t end = r.i + (size_t)seg - 2;
        if (m == 0xDB) {                                                          // DQT
            while (r.i < end) {
                const int pt = r.u8(), id = pt & 15, wide = pt >> 4;
                if (id > 3) return JD_E_TABLE;                                    // main.go:55-58
                if (r.i + (wide ? 128u : 64u) > end) return JD_E_TABLE;
                if (fr->qt_present & (1u << id)) return JD_E_TABLE;               // duplicate id (main.go:77-82)
                for (int a = 0; a < 64; a++) {
                    const int v = wide ? r.u16() : r.u8();
                    fr->qt[id][zz.nat[a]] = (uint8_t)v;                           // low byte only
                }
                fr->qt_present |= (uint8_t)(1u << id);
            }
        } else if (m == 0xC0 || (m == 0xC2 && scans)) {                           // SOF0 / SOF2
            if (have_sof) return JD_E_MARKER;
            if (seg < 8) return JD_E_TRUNCATED;
            if (r.u8() != 8) return JD_E_UNSUPPORTED;                             // precision (main.go:97-100)
            fr->height = (uint16_t)r.u16();
            fr->width = (uint16_t)r.u16();
            const int nc = r.u8();
            if (nc > 3 || nc < 1) return JD_E_UNSUPPORTED;                        // main.go:112-115
            if (r.i + (size_t)nc * 3 > end) return JD_E_TRUNCATED;
            fr->ncomp = (uint8_t)nc;
            for (int a = 0; a < nc; a++) {
                comp_id[a] = r.u8();
                const int hv = r.u8();
                fr->comp[a].h = (uint8_t)(hv >> 4); fr->comp[a].v = (uint8_t)(hv & 15);
                fr->comp[a].tq = (uint8_t)r.u8();
                if (comp_id[a] == 0) zero_based = true;                           // main.go:133-135
                for (int b = 0; b < a; b++) if (comp_id[b] == comp_id[a]) return JD_E_UNSUPPORTED;
            }
            if (zero_based) for (int a = 0; a < nc; a++) comp_id[a] += 1;         // main.go:153-157
            have_sof = true; progressive = m == 0xC2;
        } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JD_E_UNSUPPORTED;                                              // progressive through jd_parse, lossless, ...
        } else if (m == 0xCC) {
            return JD_E_UNSUPPORTED;                                              // arithmetic coding
        } else if (m == 0xDD) {                                                   // DRI
            if (seg != 4) return JD_E_RESTART;                                    // main.go:824-827
            fr->restart_interval = (uint16_t)r.u16();
        } else if (m == 0xC4) {                                                   // DHT
            while (r.i < end) {
                const int tc = r.u8();
                const int cls = (tc >> 4) == 0 ? 0 : 1, id = tc & 15;             // main.go:850-851
                if (id > 3 || r.i + 16 > end) return JD_E_TABLE;
                jd_huff* h = &fr->huff[cls][id];
                memset(h, 0, sizeof *h);
                int count = 0;
                for (int a = 0; a < 16; a++) { h->bits[a] = (uint8_t)r.u8(); count += h->bits[a]; }
                if (count > 256 || r.i + (size_t)count > end) return JD_E_TABLE;
                for (int a = 0; a < count; a++) h->vals[a] = (uint8_t)r.u8();
                h->nvals = (uint16_t)count; h->present = 1;
            }
        } else if (m == 0xDA) {                                                   // SOS
            if (!have_sof) return JD_E_MARKER;
            if (r.i >= end) return JD_E_TRUNCATED;                                // empty SOS payload: nothing to read
            const int ns = r.u8();
            if (ns < 1 || ns > 3) return JD_E_UNSUPPORTED;                        // a scan of 0 or > 3 components is not truncation
            if (r.i + (size_t)ns * 2 + 3 > end) return JD_E_TRUNCATED;
            jd_scan local; jd_scan* sc = &local;
            if (scans) { if (ns_found >= cap) return JD_E_ARG; sc = &scans[ns_found]; }
            memset(sc, 0, sizeof *sc);
            sc->progressive = progressive ? 1 : 0; sc->ncomp = 0;
            int used = 0;
            for (int a = 0; a < ns; a++) {
                int id = r.u8();
                if (zero_based) id += 1;                                          // main.go:961-963
                const int t = r.u8();
                for (int c = 0; c < fr->ncomp; c++)
                    if (comp_id[c] == id && !(used & (1 << c))) {
                        fr->comp[c].td = (uint8_t)(t >> 4); fr->comp[c].ta = (uint8_t)(t & 15); used |= 1 << c;
                        sc->comp[sc->ncomp] = (uint8_t)c; sc->td[sc->ncomp] = (uint8_t)(t >> 4); sc->ta[sc->ncomp] = (uint8_t)(t & 15);
                        sc->ncomp++;
                    }
            }
            if (!progressive && used != (1 << fr->ncomp) - 1) return JD_E_UNSUPPORTED;   // baseline: single interleaved scan only
            if (sc->ncomp != ns) return JD_E_UNSUPPORTED;                         // a scan component that the frame does not have
            sc->ss = (uint8_t)r.u8(); sc->se = (uint8_t)r.u8();
            { const int aa = r.u8(); sc->ah = (uint8_t)(aa >> 4); sc->al = (uint8_t)(aa & 15); }
            if (progressive && (sc->se > 63 || sc->ss > sc->se || sc->al > 13)) return JD_E_UNSUPPORTED;
            memcpy(sc->huff, fr->huff, sizeof sc->huff);
            r.i = end;
            // extent of the entropy-coded segment: up to the first FF that is followed by neither 00, FF nor
            // RSTn (main.go:993-1021: EOI ends it; in a progressive file so do DRI, DHT and SOS; anything else is fatal)
            const size_t start = r.i;
            size_t i = start;
            uint8_t nx = 0;
            for (;;) {
                const uint8_t* q = (const uint8_t*)memchr(file + i, 0xFF, file_len - i);
                if (!q || (size_t)(q - file) + 1 >= file_len) return JD_E_TRUNCATED;   // EOF before EOI (main.go:22-25)
                i = (size_t)(q - file);
                nx = file[i + 1];
                if (nx == 0x00 || (nx >= 0xD0 && nx <= 0xD7)) { i += 2; continue; }
                if (nx == 0xFF) { i += 1; continue; }
                break;
            }
            if (!(nx == 0xD9 || (progressive && (nx == 0xDD || nx == 0xC4 || nx == 0xDA)))) return JD_E_MARKER;
            sc->ecs_off = start; sc->ecs_len = i - start;
            ns_found++;
            if (!scans) {                                                         // baseline form: done at the first scan
                *ecs_off = start; *ecs_len = i - start;
                return JD_OK;
            }
            r.i = i;                                                              // at the FF of the marker that ended the scan
            continue;
        }
        r.i = end;                                                                // APPn, COM, JPGn, DNL, DHP, EXP: skipped
    }
}

extern "C" int jd_parse(const uint8_t* file, size_t file_len, jd_frame* fr, size_t* ecs_off, size_t* ecs_len) {
    if (!file || !fr || !ecs_off || !ecs_len) return JD_E_ARG;
    return parse_impl(file, file_len, fr, nullptr, 0, nullptr, ecs_off, ecs_len);
}

extern "C" int jd_parse_scans(const uint8_t* file, size_t file_len, jd_frame* fr, jd_scan* scans, int cap, int* n_scans) {
    if (!file || !fr || !scans || cap <= 0 || !n_scans) return JD_E_ARG;
    *n_scans = 0;
    return parse_impl(file, file_len, fr, scans, cap, n_scans, nullptr, nullptr);
}

// Restart-interval-range sharding: a restart interval starts with the DC predictors at zero
// (main.go has no restart handling at all; T.81 E.2.4), so any run of whole intervals decodes on its own.  A band
// is a run of whole MCU ROWS: then it is an image of its own -- same tables, same width, fewer rows -- and every entry
// point of this library takes it as it is.
extern "C" int jd_split_bands(const jd_frame* fr, const uint8_t* ecs, size_t ecs_len, int parts, jd_band* bands, int* n_bands) {
    if (!fr || !ecs || !bands || !n_bands || parts < 1) return JD_E_ARG;
    *n_bands = 0;
    uint32_t mx = 0, my = 0, bpm = 0;
    const int rc = jd_frame_geometry(fr, &mx, &my, &bpm);
    if (rc != JD_OK) return rc;
    const uint32_t ri = fr->restart_interval;
    if (ri == 0 || mx % ri != 0) return JD_E_UNSUPPORTED;        // an MCU row must begin a restart interval
    const uint32_t ipr = mx / ri;                                // intervals per MCU row
    // RSTn numbers count modulo 8 from the start of the scan; a band's first marker must be RST0 again
    uint32_t step = 8;
    for (uint32_t g = ipr; step > 1 && g % 2 == 0; g /= 2) step /= 2;      // 8 / gcd(ipr, 8) rows
    const uint32_t units = (my + step - 1) / step;               // band boundaries fall on multiples of `step` rows
    const uint32_t nb = (uint32_t)parts < units ? (uint32_t)parts : units;
    // (a single-component frame has a 1x1 MCU whatever its SOF sampling byte says: frame_geometry, main.go:753-767)
    const uint32_t mcu_h = 8u * (fr->ncomp == 1 ? 1u : fr->comp[0].v);
    // first MCU row of every band, then the byte ranges between the markers that separate them
    std::vector<uint32_t> row0(nb + 1);
    for (uint32_t k = 0; k <= nb; k++) {
        const uint32_t r = (uint32_t)(((uint64_t)units * k / nb) * step);
        row0[k] = r < my ? r : my;
    }
    std::vector<size_t> cut(nb + 1);                             // cut[k] = offset of band k's first byte
    cut[0] = 0; cut[nb] = ecs_len;
    uint32_t seen = 0, next = 1;                                 // markers seen; next band whose start is wanted
    const uint8_t* p = ecs;
    const uint8_t* const end = ecs + ecs_len;
    while (next < nb && p < end) {
        p = (const uint8_t*)memchr(p, 0xFF, (size_t)(end - p));
        if (!p || p + 1 >= end) break;
        const uint8_t m = p[1];
        if (m >= 0xD0 && m <= 0xD7) {
            if (m != 0xD0 + (seen & 7)) return JD_E_RESTART;
            seen++;
            while (next < nb && seen == row0[next] * ipr) {      // this marker ends the last interval before band `next`
                cut[next] = (size_t)(p - ecs) + 2;
                next++;
            }
            p += 2;
        } else p += (m == 0xFF) ? 1 : 2;                         // FF00 stuffing, fill bytes; other markers are K0's to report
    }
    if (next < nb) return JD_E_RESTART;                          // fewer markers than the frame needs
    for (uint32_t k = 0; k < nb; k++) {
        jd_band& b = bands[k];
        b.frame = *fr;
        b.y0 = row0[k] * mcu_h;
        const uint32_t y1 = row0[k + 1] * mcu_h < fr->height ? row0[k + 1] * mcu_h : fr->height;
        b.frame.height = (uint16_t)(y1 - b.y0);
        b.ecs_off = cut[k];
        b.ecs_len = (k + 1 < nb ? cut[k + 1] - 2 : ecs_len) - cut[k];       // without the marker in front of the next band
    }
    *n_bands = (int)nb;
    return JD_OK;
}

extern "C" long jd_bmp_encode(const uint8_t* rgb, size_t row_stride, int width, int height, uint8_t* out, size_t cap) {
    if (width <= 0 || height <= 0) return JD_E_ARG;
    const int pad = width % 4;                                                    // main.go:1172
    const size_t size = 14 + 12 + (size_t)height * width * 3 + (size_t)pad * height;
    if (!out) return (long)size;
    if (!rgb || cap < size) return JD_E_ARG;
    uint8_t* p = out;
    auto le = [&p](unsigned v, int n) { for (int i = 0; i < n; i++) *p++ = (uint8_t)(v >> (8 * i)); };
    *p++ = 'B'; *p++ = 'M';
    le((unsigned)size, 4); le(0, 4); le(26, 4);                                   // main.go:1187-1190
    le(12, 4); le((unsigned)width, 2); le((unsigned)height, 2); le(1, 2); le(24, 2);
    for (int y = height - 1; y >= 0; y--) {                                       // bottom-up, B,G,R
        const uint8_t* s = rgb + (size_t)y * row_stride;
        for (int x = 0; x < width; x++) { p[0] = s[2]; p[1] = s[1]; p[2] = s[0]; p += 3; s += 3; }
        for (int k = 0; k < pad; k++) *p++ = 0;
    }
    return (long)(p - out);
}
