// jd_tables.h -- host-side Huffman LUT packing (see jd_tables.cpp).
#pragma once
#include <stdint.h>

#include <vector>

#include "jd_internal.h"

// Appends one packed table (JD_LUT_PRIMARY + 64 * n_secondary entries) to `out`.
// Returns 0, -1 (counts do not match), -2 (not a prefix code), -3 (too large).
int jd_pack_lut(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, std::vector<uint16_t>& out);
// AC table for the self-synchronising pass: several whole symbols per primary entry (see jd_tables.cpp).
int jd_pack_lut_multi(const uint8_t bits[16], const uint8_t* vals, int nvals, unsigned lut_bits, std::vector<uint16_t>& out);
// jd_pack_lut with a primary table of 2^lut_bits entries (the sync pass's tables are wider than the fallback ones)
int jd_pack_lut_bits(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, unsigned lut_bits, std::vector<uint16_t>& out);
// Writer table (32-bit entries, 2^prim_bits primary + 2^(16 - prim_bits)-entry secondaries; layout in jd_internal.h).
// unit: advance of one coefficient in the [15:8] field -- 2 (bytes of int16 coefficients; EOB 254) or 1 (byte form; EOB 127)
int jd_pack_lut32(const uint8_t bits[16], const uint8_t* vals, int nvals, bool is_dc, unsigned prim_bits, unsigned unit,
                  std::vector<uint32_t>& out);
// AC pair table of the byte-form writer (JD_WP_* layout, jd_internal.h): up to two symbols per entry.
int jd_pack_lut32p(const uint8_t bits[16], const uint8_t* vals, int nvals, unsigned prim_bits, std::vector<uint32_t>& out);
// Symbol table for the progressive scans: the decoder needs the raw symbol there (run/size, EOBn), not a baseline
// interpretation of it.  JD_PLUT_DWORDS dwords, laid out for kp_scan's scalar-unit decoder (jd_tables.cpp).
int jd_pack_lut_lane(const uint8_t bits[16], const uint8_t* vals, int nvals, std::vector<uint32_t>& out);
