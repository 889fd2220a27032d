// jd_kernels.hip -- CDNA4 (gfx950) kernels of the baseline-JPEG decode path.
//
//   K0  k0_unstuff                         unstuff FF00, drop + index RSTn, one pass with decoupled look-back
//                                          (main.go:993-1021)
//   K1  k1_sync / k1_sync_spine            self-synchronising sub-sequence pass + DC-prediction prefix scan (no DRI)
//       k1_write_intervals / _subseq       Huffman decode -> int16 coefficients (main.go:749-816, 186-240,
//                                          1156-1169, 1252-1316)
//       k1_dc_scan                         DC-prediction prefix scan as its own pass (images whose components all
//                                          share one table pair: the block-in-MCU phase cannot self-synchronise)
//   P   kp_scan                            progressive (SOF2) scans, one wave per scan (main.go:241-428)
//   K2  k2_mcu_lanes                       dequantise + de-zigzag + fp64 truncating IDCT + chroma spread +
//                                          float32 YCbCr->RGB (main.go:638-666, 454-606, 710-747, 669-707;
//                                          zmap/map.go:3-68)
//
// Compiled with -ffp-contract=off: the IDCT and the colour conversion reproduce separately rounded
// IEEE operations.  All kernels are batched: a workgroup finds its image through an
// exclusive prefix of workgroups-per-image.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "jd_internal.h"

#define WAVE 64
typedef uint32_t jd_u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_image(const uint32_t* __restrict__ prefix, int n, uint32_t wg) {
    int lo = 0, hi = n;                       // prefix[lo] <= wg < prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= wg) lo = mid; else hi = mid;
    }
    return lo;
}

// Inclusive scan over the wave in six DPP additions (row shifts 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of
// rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3): no LDS round trips (six ds_bpermute before).
#define JD_DPP(v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t v) {      // inside each row of 16 lanes
    v += JD_DPP(v, 0x111, 0xF);                                       // row_shr:1
    v += JD_DPP(v, 0x112, 0xF);                                       // row_shr:2
    v += JD_DPP(v, 0x114, 0xF);                                       // row_shr:4
    v += JD_DPP(v, 0x118, 0xF);                                       // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v = row_incl_scan(v);
    v += JD_DPP(v, 0x142, 0xA);                                       // row_bcast:15 into rows 1 and 3
    v += JD_DPP(v, 0x143, 0xC);                                       // row_bcast:31 into rows 2 and 3
    return v;
}

// exclusive scan over a thread block (<= 1024); `tmp` has >= 16 words of LDS; returns block total in `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) tmp[wv] = inc;
    __syncthreads();
    const int nw = (int)((blockDim.x + 63u) >> 6);                    // <= 16: the wave totals fit one row
    const uint32_t t = row_incl_scan(lane < nw ? tmp[lane] : 0u);
    total = (uint32_t)__builtin_amdgcn_readlane((int)t, nw - 1);
    const uint32_t base = wv ? (uint32_t)__builtin_amdgcn_readlane((int)t, wv - 1) : 0u;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// K0: unstuff + restart index
// Byte rule (T.81 B.1.1.5, what main.go:993-1021 does byte by byte):
//   FF followed by 00      -> keep FF, drop 00
//   FF followed by D0..D7  -> drop both, restart boundary
//   FF followed by FF      -> fill byte, drop the first FF
//   FF followed by other   -> marker inside the segment: error (main.go:1013-1016)
// ------------------------------------------------------------------------------------------------
// One pass, one kernel (k0_unstuff): every workgroup (1024 lanes) takes the next 32 KiB chunk of its image (ticket counter, so
// chunks start in order), classifies its bytes four at a time with SWAR flag words, scans the kept-byte and
// marker counts, and learns its output offset by DECOUPLED LOOK-BACK over the per-chunk status words of the
// chunks before it (aggregate / inclusive-prefix in one 8-byte word, written and read with agent-scope
// relaxed atomics -- the value and its tag travel together, so no fence is involved).  The ECS is read once
// and the unstuffed stream written once.
#ifndef JD_K0_WAVES_PER_SIMD
#define JD_K0_WAVES_PER_SIMD 6
#endif
#define JD_K0_LANE_BYTES (JD_K0_CHUNK / JD_K0_THREADS)          // contiguous bytes per lane
static_assert(JD_K0_LANE_BYTES == 32, "one 32-bit flag mask per lane");

// 0x80 in every byte of x that is NOT zero (exact: no carries between bytes)
__device__ __forceinline__ uint32_t nonzero_flags(uint32_t x) {
    return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// The flag words of a lane's eight dwords -> one bit per byte.  v_dot4_u32_u8 against the byte weights 1,2,4,8 (even
// dword) / 16,32,64,128 (odd dword) sums 0x80 * weight over the flagged bytes: 128 * (8-bit mask) per dword pair.
__device__ __forceinline__ uint32_t k0_gather(const uint32_t (&f)[8]) {
    uint32_t g[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
        g[p] = __builtin_amdgcn_udot4(f[2 * p + 1], 0x80402010u, __builtin_amdgcn_udot4(f[2 * p], 0x08040201u, 0u, false), false);
    return (g[0] >> 7) | (g[1] << 1) | (g[2] << 9) | (g[3] << 17);
}
// lane i <- lane i - 1 (lane 0 <- fill) / lane i <- lane i + 1 (lane 63 <- fill): DPP wave shifts, no LDS round trip
__device__ __forceinline__ uint32_t wave_from_below(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);   // wave_shr:1
}
__device__ __forceinline__ uint32_t wave_from_above(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false);   // wave_shl:1
}
// v_perm selector that moves the kept bytes (4-bit mask m) of a dword to its low end; the rest reads as 0
__device__ __forceinline__ uint32_t k0_squeeze_selector(uint32_t m) {
    uint32_t sel = 0x0C0C0C0Cu, n = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
        if (m & (1u << j)) { sel = (sel & ~(0xFFu << (8u * n))) | (j << (8u * n)); n++; }
    return sel;
}

// status word of a chunk: [27:0] kept bytes, [55:28] RSTn markers, [63:62] 1 = this chunk alone, 2 = inclusive prefix
#define JD_K0_AGG (1ull << 62)
#define JD_K0_PFX (2ull << 62)

__global__ __launch_bounds__(JD_K0_THREADS, JD_K0_WAVES_PER_SIMD) void k0_unstuff(JdKernelArgs a, uint32_t stride) {
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t s_chunk;
    __shared__ uint2 s_base;
    __shared__ uint32_t s_sel[16];
    __shared__ __align__(16) uint8_t buf[JD_K0_CHUNK + 32];   // the compacted chunk (the raw one never touches LDS)
    // blockIdx picks the image; the chunk inside it is the next ticket of THAT image (one counter for the whole
    // batch saturates at ~90 tickets/us -- measured 80 us for 7,200 chunks), so a chunk's predecessors have
    // always started before it
    // Workgroups that are resident together should belong to different images (their tickets, status words and
    // look-back chains are per image): blockIdx is spread over the images by a stride coprime to the grid size --
    // any permutation will do, an image only needs as many workgroups as it has chunks.
    const int img = __builtin_amdgcn_readfirstlane(find_image(a.prefix, a.n_images, (uint32_t)(((unsigned long long)blockIdx.x * stride) % gridDim.x)));
    const JdImg& im = a.imgs[img];
    // The image's descriptor fields are read FIRST (plain vector loads, in flight while lane 0 takes the ticket) and go
    // through scalar registers afterwards (readfirstlane: waited for once, here, not again at every branch below); the
    // segment's address becomes a global-memory one (a generic pointer makes flat loads).
    typedef __attribute__((address_space(1))) const uint8_t g_cu8;
    typedef __attribute__((address_space(1))) const jd_u32x4 g_cq;
    const uint64_t ecs_bits = (uint64_t)(uintptr_t)im.ecs, soff_bits = im.stream_off;
    const uint32_t len_v = im.ecs_len, nch_v = im.n_k0_chunks, nint_v = im.n_intervals, ivb_v = im.interval_base;
    const uint32_t gbase = a.prefix[img];
    asm volatile("" :: "v"(len_v), "v"(nch_v) : "memory");      // (issued before the atomic below, not sunk behind it)
    if (threadIdx.x == 0) s_chunk = atomicAdd(a.k0_ticket + img, 1u);
    if (threadIdx.x < 16) s_sel[threadIdx.x] = k0_squeeze_selector(threadIdx.x);
    // (readfirstlane returns int: each half goes through uint32_t, or the low half's sign bit would fill the high one)
    const uint32_t ecs_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ecs_bits >> 32));
    const uint32_t ecs_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ecs_bits);
    g_cu8* const ecs = (g_cu8*)(uintptr_t)(((uint64_t)ecs_hi << 32) | (uint64_t)ecs_lo);
    const uint32_t len = __builtin_amdgcn_readfirstlane(len_v);
    const uint32_t n_chunks = __builtin_amdgcn_readfirstlane(nch_v), n_intervals = __builtin_amdgcn_readfirstlane(nint_v);
    const uint32_t iv_base = __builtin_amdgcn_readfirstlane(ivb_v);
    const uint64_t stream_off = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(soff_bits >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)soff_bits);
    __syncthreads();
    const uint32_t chunk = s_chunk;
    const uint32_t gchunk = gbase + chunk;
    const uint32_t c0 = chunk * JD_K0_CHUNK;
    const uint32_t lane = threadIdx.x & 63u;
    // ---- this lane's 32 contiguous bytes, straight into registers (0xD9 past the end: behaves like EOI).  Two 16-byte
    // loads per lane, 32 bytes apart between lanes: each 128-byte line is touched by both and fetched once.  The byte
    // before the wave's first and the byte after its last come with the same batch of loads (lanes 0-31 read the one,
    // lanes 32-63 the other; only lanes 0 and 63 use them); inside the wave the neighbours' flags travel by DPP.
    const uint32_t g0 = c0 + threadIdx.x * JD_K0_LANE_BYTES;     // offset of this lane's bytes in the segment
    uint32_t w[8];
    jd_u32x4 stg[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t g = g0 + 16u * (uint32_t)r;
        stg[r] = jd_u32x4{0xD9D9D9D9u, 0xD9D9D9D9u, 0xD9D9D9D9u, 0xD9D9D9D9u};
        if (g + 16u <= len) stg[r] = *(g_cq*)(ecs + g);
    }
    const uint32_t wave0 = g0 - lane * JD_K0_LANE_BYTES;         // offset of the wave's first byte
    const uint32_t eaddr = lane < 32u ? wave0 - 1u : wave0 + 64u * JD_K0_LANE_BYTES;
    uint32_t edge = lane < 32u ? 0u : 0xD9u;                     // before the segment: not FF; past its end: not 00
    if (lane < 32u ? wave0 != 0u && wave0 <= len : eaddr < len) edge = ecs[eaddr];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t g = g0 + 16u * (uint32_t)r;
        uint32_t t[4] = { stg[r].x, stg[r].y, stg[r].z, stg[r].w };
        if (__builtin_expect(g < len && g + 16u > len, 0)) {      // the segment's ragged end: byte by byte (one lane per image)
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (g + (uint32_t)j < len) t[j >> 2] = (t[j >> 2] & ~(0xFFu << ((j & 3) * 8))) | ((uint32_t)ecs[g + j] << ((j & 3) * 8));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) w[4 * r + k] = t[k];
    }
    // ---- classify: one bit per byte ------------------------------------------------------------------------
    uint32_t fl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) fl[k] = nonzero_flags(~w[k]);
    const uint32_t notF = k0_gather(fl);                         // byte is not FF
#pragma unroll
    for (int k = 0; k < 8; k++) fl[k] = nonzero_flags(w[k]);
    const uint32_t notZ = k0_gather(fl);                         // byte is not 00
#pragma unroll
    for (int k = 0; k < 8; k++) fl[k] = nonzero_flags((w[k] & 0xF8F8F8F8u) ^ 0xD0D0D0D0u);
    const uint32_t notR = k0_gather(fl);                         // byte is not D0..D7
    const uint32_t F = ~notF, Z = ~notZ;
    const uint32_t below = wave_from_below(F, edge == 0xFFu ? 0x80000000u : 0u);    // bit 31: the byte before mine is FF
    const uint32_t above = wave_from_above(Z, edge == 0u ? 1u : 0u);                // bit 0: the byte after mine is 00
    const uint32_t PF = __builtin_amdgcn_alignbit(F, below, 31);                    // previous byte is FF
    const uint32_t NZ = __builtin_amdgcn_alignbit(above, Z, 1);                     // next byte is 00
    const uint32_t n = g0 < len ? (len - g0 < 32u ? len - g0 : 32u) : 0u;           // bytes of mine inside the segment
    const uint32_t V = n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u);
    const uint32_t after = notF & PF & V;                                            // non-FF byte right after an FF
    const uint32_t keep = ((F & NZ) | (notF & ~PF)) & V;
    const uint32_t rst = after & ~notR;
    const uint32_t bad = after & notZ & notR;
    const uint32_t nk = __popc(keep), nr = __popc(rst);
    if (__any(bad != 0) && lane == 0) atomicOr(&a.status[img], JD_ST_MARKER);
    uint32_t tot;
    const uint32_t ex = block_excl_scan(nk | (nr << 16), tmp, tot);   // nk <= 64 * 256 < 2^16 per workgroup
    const uint32_t ek = ex & 0xFFFFu, er = ex >> 16, tk = tot & 0xFFFFu, tr = tot >> 16;
    // ---- publish this chunk's totals, compact, then look back ------------------------------------------------
    // The compaction does not need the chunk's place in the output (the copy-out below shifts by the output's 16-byte
    // phase), so it runs between publishing the totals and reading the predecessors': by the time a workgroup looks
    // back, the chunks before it have had the same few microseconds to publish, and the look-back rarely waits.
    unsigned long long* st = a.k0_state + gchunk;
    const unsigned long long mine = (unsigned long long)tk | ((unsigned long long)tr << 28);
    if (threadIdx.x == 0) __hip_atomic_store(st, mine | (chunk ? JD_K0_AGG : JD_K0_PFX), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Compaction: a dword's kept bytes are squeezed to its low end by one v_perm (selector from a 16-entry table) and
    // the dword is stored at its BYTE offset in the compacted chunk (LDS takes unaligned dword writes on gfx950:
    // profiles/microbench/lds_unaligned.hip).  The bytes above the kept ones are rubbish; the lane's next dword overwrites them (same
    // wave, LDS executes in order).  A dword whose rubbish would reach past the lane's last kept byte -- into bytes the
    // next lane wrote in an EARLIER instruction -- is written byte by byte instead: only ever one of the lane's last
    // dwords, and only when fewer than four kept bytes follow its first.
    {
        const uint32_t buf_b = (uint32_t)(uintptr_t)buf + ek, sel_b = (uint32_t)(uintptr_t)s_sel;
        uint32_t sel[8];
#pragma unroll
        for (int k = 0; k < 8; k++)                              // all eight table reads first: one LDS latency, not eight
            sel[k] = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(sel_b + ((keep >> (4 * k)) & 15u) * 4u);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t c = __builtin_amdgcn_perm(0u, w[k], sel[k]);
            const uint32_t at = buf_b + (k ? __popc(keep & ((1u << (4 * k)) - 1u)) : 0u);
            const uint32_t rem = __popc(keep >> (4 * k));        // kept bytes from this dword on
            if (rem >= 4u) asm volatile("ds_write_b32 %0, %1" :: "v"(at), "v"(c) : "memory");
            else {
                const uint32_t cnt = __popc((keep >> (4 * k)) & 15u);
                if (cnt > 0u) asm volatile("ds_write_b8 %0, %1" :: "v"(at), "v"(c) : "memory");
                if (cnt > 1u) asm volatile("ds_write_b8 %0, %1 offset:1" :: "v"(at), "v"(c >> 8) : "memory");
                if (cnt > 2u) asm volatile("ds_write_b8 %0, %1 offset:2" :: "v"(at), "v"(c >> 16) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (threadIdx.x < 64) {                                      // wave 0
        unsigned long long run = 0;                              // sum of the chunks before this one
        int back = (int)chunk - 1;                               // nearest chunk not yet summed
        while (back >= 0) {
            const int idx = back - (int)lane;
            unsigned long long v = JD_K0_PFX;                    // before the image's first chunk: prefix 0
            if (idx >= 0) {
                do { v = __hip_atomic_load(st - (chunk - (uint32_t)idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                while ((v >> 62) == 0);                          // the chunk took its ticket before us: it will publish
            }
            const unsigned long long pf = __builtin_amdgcn_ballot_w64((v >> 62) == 2);
            const uint32_t stop = pf ? (uint32_t)__builtin_ctzll(pf) : 64u;   // nearest chunk with an inclusive prefix
            unsigned long long part = lane <= stop ? (v & ~(3ull << 62)) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
            run += part;
            if (pf) break;
            back -= 64;
        }
        if (lane == 0) {
            if (chunk) __hip_atomic_store(st, (run + mine) | JD_K0_PFX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = make_uint2((uint32_t)(run & 0xFFFFFFFull), (uint32_t)((run >> 28) & 0xFFFFFFFull));
            if (chunk + 1 == n_chunks) {                   // last chunk of the image: totals
                const uint32_t total = (uint32_t)((run + mine) & 0xFFFFFFFull), nrst = (uint32_t)(((run + mine) >> 28) & 0xFFFFFFFull);
                a.stream_len[img] = total;
                a.interval_start[iv_base] = 0;
                a.interval_start[iv_base + n_intervals] = total;
                if (nrst != n_intervals - 1) atomicOr(&a.status[img], JD_ST_RST_COUNT);
            }
        }
    }
    __syncthreads();
    const uint2 base = s_base;                                   // kept bytes / markers before this chunk
    // ---- index the restart markers --------------------------------------------------------------------------------
    const uint32_t al = base.x & 15u;
    if (nr) {                                                    // markers are rare: walk the set flags only
        uint32_t f = rst, ridx = base.y + er;                    // markers before the current one
        const uint32_t kb = base.x + ek;                         // kept bytes before this lane's
        while (f) {
            const uint32_t pos = (uint32_t)__builtin_ctz(f);     // the Dn byte
            f &= f - 1u;
            const uint32_t i = pos >> 2;
            const uint32_t a0 = i & 1u ? w[1] : w[0], a1 = i & 1u ? w[3] : w[2], a2 = i & 1u ? w[5] : w[4], a3 = i & 1u ? w[7] : w[6];
            const uint32_t b0 = i & 2u ? a1 : a0, b1 = i & 2u ? a3 : a2;
            const uint32_t digit = ((i & 4u ? b1 : b0) >> ((pos & 3u) * 8u)) & 7u;
            const uint32_t nxt = ridx + 1;                       // interval that starts after this marker
            if (nxt < n_intervals) a.interval_start[iv_base + nxt] = kb + __popc(keep & ((1u << pos) - 1u));
            if (digit != (ridx & 7u)) atomicOr(&a.status[img], JD_ST_RST_SEQ);
            ridx++;
        }
    }
    // (the barrier after the look-back also orders the compaction's LDS writes before the reads below)
    // ---- LDS -> stream: whole 16-byte quads, bytes at the two ragged ends.  Output byte i of the aligned window
    // (i = 0 at the 16-byte boundary below the chunk's first output byte) is buf[i - al] -------------------------------
    uint8_t* dst = a.stream + stream_off + JD_STREAM_FRONT + (base.x - al);   // 16-byte aligned
    const uint32_t lo = al, hi = al + tk;                        // valid byte range of the window
    const uint32_t qlo = (lo + 15u) >> 4, qhi = hi >> 4;         // whole quads [qlo, qhi)
    // (LDS takes a 16-byte read at any byte address on gfx950 -- profiles/microbench/lds_unaligned128.hip; through asm, so that the
    // reads of both rounds are in flight together)
    const uint32_t src_b = (uint32_t)(uintptr_t)buf - al;
    static_assert(JD_K0_CHUNK / 16 + 1 <= 3 * JD_K0_THREADS, "copy-out rounds");
    {
        const uint32_t q0 = qlo + threadIdx.x, q1 = q0 + JD_K0_THREADS, q2 = q1 + JD_K0_THREADS;
        jd_u32x4 v0 = {}, v1 = {}, v2 = {};
        if (q0 < qhi) asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(src_b + q0 * 16u) : "memory");
        if (q1 < qhi) asm volatile("ds_read_b128 %0, %1" : "=v"(v1) : "v"(src_b + q1 * 16u) : "memory");
        if (q2 < qhi) asm volatile("ds_read_b128 %0, %1" : "=v"(v2) : "v"(src_b + q2 * 16u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2) :: "memory");
        if (q0 < qhi) reinterpret_cast<jd_u32x4*>(dst)[q0] = v0;
        if (q1 < qhi) reinterpret_cast<jd_u32x4*>(dst)[q1] = v1;
        if (q2 < qhi) reinterpret_cast<jd_u32x4*>(dst)[q2] = v2;
    }
    if (threadIdx.x < 32) {
        uint32_t i;
        if (threadIdx.x < 16) { i = lo + threadIdx.x; if (i < hi && i < (qlo << 4)) dst[i] = buf[i - al]; }
        else {
            uint32_t t1 = qhi << 4; if (t1 < (qlo << 4)) t1 = qlo << 4;
            i = t1 + (threadIdx.x - 16);
            if (i >= lo && i < hi) dst[i] = buf[i - al];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: Huffman decode core
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

struct ImgK {              // per-image constants of the decode loop, read once (wave-uniform)
    uint32_t bpm, nl, dc0, dc1, dc2, ac0, ac1, ac2;
};
__device__ __forceinline__ ImgK load_imgk(const JdImg& im) {
    ImgK k;
    k.bpm = im.bpm; k.nl = im.hs * im.vs;
    k.dc0 = im.dc_lut[0]; k.dc1 = im.dc_lut[1]; k.dc2 = im.dc_lut[2];
    k.ac0 = im.ac_lut[0]; k.ac1 = im.ac_lut[1]; k.ac2 = im.ac_lut[2];
    return k;
}

__device__ __forceinline__ uint32_t comp_of_block(const JdImg& im, uint32_t c) {
    const uint32_t nl = im.hs * im.vs;
    return c < nl ? 0u : c - nl + 1u;
}
// three-way select kept as two v_cndmask: the empty asm stops hipcc from turning the select chain into a
// dynamically indexed table in scratch memory (scratch traffic shares vmcnt with the coefficient stores)
__device__ __forceinline__ uint32_t sel3(uint32_t comp, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t t = comp == 1 ? b : c;
    asm volatile("" : "+v"(t));
    return comp == 0 ? a : t;
}
__device__ __forceinline__ void load_lut(uint16_t* lds, const uint16_t* __restrict__ g, uint32_t n) {
    // n is padded to a multiple of 8 entries by the host
    const uint4* s = reinterpret_cast<const uint4*>(g);
    uint4* d = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < (n >> 3); i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

__device__ __forceinline__ uint32_t pack_state(uint32_t rel, uint32_t c, uint32_t z) {
    rel = rel > 255u ? 255u : rel;
    return JD_STATE_VALID | rel | (c << 8) | (z << 16);
}

extern __shared__ __align__(16) uint8_t dyn_lds[];

__device__ __forceinline__ int extend_bits(uint32_t win, uint32_t total, uint32_t size) {
    // EXTEND, main.go:196-198 / 234-236: v < 2^(size-1)  ->  v - (2^size - 1)
    const int v = (int)__builtin_amdgcn_ubfe(win, 32u - total, size);
    const int m = (1 << size) - 1;
    return (2 * v <= m) ? v - m : v;
}

// ---- whole-block writer ------------------------------------------------------------------------------------
// One lane per SEGMENT of the stream: a run of whole blocks whose first bit, first block index and DC
// predictors are known -- a restart interval (DC predictors reset, data byte aligned), or the blocks that
// start inside one sub-sequence of the self-synchronising pass.  Each wave walks its 64 segments
// BLOCK-SYNCHRONOUSLY: every lane finishes block b of its segment before any lane starts b+1.
//   * a block goes out in two halves (coefficients 0-31, then 32-63): each lane assembles a half in a private
//     64-byte LDS slot and the wave stores the 64 finished halves as 64-byte runs (4 lanes x 16 B each), zeros
//     included, so no fill pass is needed.  Halving the slot is what lets 16 waves fit on a CU; the symbol loop
//     is bound by how fast ONE wave can issue (about 600 cycles per step measured at 2 waves per SIMD);
//   * each lane reads the stream through a private 64-byte LDS ring topped up at both half-block boundaries
//     with 16-byte loads requested one half-step ahead; three stream dwords are held in registers and the ring
//     word that replaces the oldest is read beside the table lookup, so the symbol loop's only dependent LDS
//     access is the table lookup;
//   * JD_W_WAVES waves share one copy of the lookup tables but are otherwise independent (no workgroup
//     barrier after the table load): LDS is 128 B per lane + 256 B per wave + the tables, two workgroups per CU.
// Slot and ring are swizzled by lane (16-byte chunk index rotated / xor-ed with lane & 3) instead of padded.

struct Segment {
    uint32_t start_bit;      // first bit of the first block
    uint32_t end_bit;        // decoding past this bit = truncated data
    uint32_t nblk;           // whole blocks to decode
    uint32_t blk0;           // index (scan order) of the first block
    int p0, p1, p2;          // DC predictors at the first block
};

__device__ __forceinline__ uint4 be32x4(uint4 v) { return make_uint4(be32(v.x), be32(v.y), be32(v.z), be32(v.w)); }
__device__ __forceinline__ uint32_t lds_u16(uint32_t off) { return *reinterpret_cast<const uint16_t*>(dyn_lds + off); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *reinterpret_cast<const uint32_t*>(dyn_lds + off); }
// LDS operations of one wave execute in order; this only stops the compiler from moving them
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

#define JD_E_SKIP ((63u << 9) | 1u)   // stand-in for an invalid code: one bit, ends the block (the error is flagged)

// table lookup: 10-bit primary entry, 64-entry secondary tables for longer codes; 0 (or an escape with offset 0)
// = no such code
__device__ __forceinline__ uint32_t lut_lookup(uint32_t tb, uint32_t win, uint32_t& bad) {
    uint32_t e = lds_u16(tb + ((win >> (32 - JD_LUT_BITS)) << 1));
    if (__builtin_expect((e & 0x8000u) != 0 || e == 0, 0)) {
        const uint32_t o = e & 0x7FFFu;
        e = (e && o) ? lds_u16(tb + (o << 1) + (((win >> (32 - JD_LUT_BITS - 6)) & 63u) << 1)) : 0u;
        if (e == 0) { bad = 1; e = JD_E_SKIP; }
    }
    return e;
}

// The writer addresses LDS by absolute byte address (the base of dyn_lds is folded into its per-lane bases once),
// so that no access pays an add of the array's link-time address.
typedef __attribute__((address_space(3))) uint8_t* jd_lds_ptr;
__device__ __forceinline__ uint32_t lds_abs_base() { return (uint32_t)(uintptr_t)(jd_lds_ptr)dyn_lds; }
__device__ __forceinline__ uint32_t la_u32(uint32_t addr) { return *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)addr; }
__device__ __forceinline__ uint32_t la_u16(uint32_t addr) { return *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)addr; }
__device__ __forceinline__ uint4 la_u4(uint32_t addr) {
    const jd_u32x4 v = *(__attribute__((address_space(3))) const jd_u32x4*)(uintptr_t)addr;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void la_st16(uint32_t addr, int v) { *(__attribute__((address_space(3))) int16_t*)(uintptr_t)addr = (int16_t)v; }
__device__ __forceinline__ void la_st8(uint32_t addr, int v) { *(__attribute__((address_space(3))) int8_t*)(uintptr_t)addr = (int8_t)v; }
__device__ __forceinline__ int la_i8(uint32_t addr) { return *(__attribute__((address_space(3))) const int8_t*)(uintptr_t)addr; }
__device__ __forceinline__ void la_st4(uint32_t addr, uint4 v) {
    jd_u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(__attribute__((address_space(3))) jd_u32x4*)(uintptr_t)addr = t;
}

// writer tables (32-bit entries, jd_internal.h): an invalid code reads as "one bit, end of block"
#define JD_W32_SKIP (31u | (JD_W32_EOB_ADV << 8))
__device__ __forceinline__ uint32_t w32_lookup(uint32_t tb, uint32_t win, uint32_t bits, uint32_t& bad) {   // bits: wave-uniform
    uint32_t e = la_u32(tb + ((win >> (30u - bits)) & (((1u << bits) - 1u) << 2)));
    if (__builtin_expect((int)e <= 0, 0)) {                        // secondary table, or no such code
        const uint32_t o = e & 0x7FFFFFFFu;
        e = o ? la_u32(tb + (o << 2) + ((win >> 14) & ((0xFFFFu >> bits) << 2))) : 0u;   // the 16 - bits following bits
        if ((int)e <= 0) { bad = 1; e = JD_W32_SKIP; }
    }
    return e;
}
// magnitude bits + EXTEND (main.go:196-198 / 234-236) from a writer-table entry: v < 2^(size-1) -> v - (2^size - 1)
__device__ __forceinline__ int w32_value(uint32_t win, uint32_t e) {
    const uint32_t m = e >> 16;
    const uint32_t raw = (win >> (e & 31u)) & m;
    return (int)raw - (int)(raw <= (e >> 17) ? m : 0u);            // e >> 17 = 2^(size-1) - 1
}

// pair tables (JD_WP_*, jd_internal.h): magnitude bits + EXTEND of one symbol.  `off` / `size`: only their low five bits
// count (v_bfe_i32 reads no more), so the entry shifted right by the field's position is operand enough.  The signed field y
// is raw - 2^size when the first magnitude bit is set (a positive value: v = raw) and raw otherwise (v = raw - (2^size - 1));
// with m = 2^size - 1 and u = y >> 31 both cases are v = y - (m ^ u).
__device__ __forceinline__ int wp_value(uint32_t win, uint32_t off, uint32_t size) {
    const int y = __builtin_amdgcn_sbfe((int)win, off, size);
    const uint32_t m = (1u << (size & 31u)) - 1u;
    return y - (int)(m ^ (uint32_t)(y >> 31));
}

// raw_dc (wave-uniform): store each block's DC DIFFERENCE instead of the predicted value -- k1_dc_scan turns the
// differences into predictions afterwards (images whose block-in-MCU phase cannot self-synchronise).
template <bool NT>
__device__ __forceinline__ void write_segments(const JdKernelArgs& a, const JdImg& im, int img, uint32_t lut_bytes,
                                               const Segment& sg, uint32_t slen, uint32_t err, bool raw_dc = false) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lbase = lds_abs_base();                               // multiple of 128 (dyn_lds is the kernel's only LDS)
    const uint32_t wbase = lbase + lut_bytes + wv * JD_WAVE_LDS;         // lut_bytes: multiple of 128
    const uint32_t slot_b = wbase + lane * 64u;                          // byte offsets inside dyn_lds
    const uint32_t ring_b = wbase + 4096u + lane * 64u;
    uint32_t dump_b = wbase + 8192u;                                     // where out-of-phase symbols "store"
    uint32_t four = 4u;
    asm volatile("" : "+v"(dump_b), "+v"(four));                         // in VGPRs: selects against them stay two-operand
    const uint32_t rot = (lane & 3u) << 4;                               // slot swizzle and ring rotation (4 chunks each)
#pragma unroll
    for (int i = 0; i < 4; i++) la_st4(slot_b + i * 16, make_uint4(0, 0, 0, 0));
    const uint32_t bpm = im.bpm, nl = im.hs * im.vs;
    const uint32_t ac_bits = __builtin_amdgcn_readfirstlane(im.w_ac_bits);
    const uint32_t dc0 = lbase + im.wdc[0] * 4u, dc1 = lbase + im.wdc[1] * 4u, dc2 = lbase + im.wdc[2] * 4u;   // table addresses
    const uint32_t ac0 = lbase + im.wac[0] * 4u, ac1 = lbase + im.wac[1] * 4u, ac2 = lbase + im.wac[2] * 4u;

    // ---- prime the ring: four quads, synchronously -----------------------------------------------------
    const uint4* Q = reinterpret_cast<const uint4*>(a.stream + im.stream_off);
    const uint32_t qmax = (JD_STREAM_FRONT + slen + JD_STREAM_TAIL) / 16u - 1u;
    const uint32_t o0 = ((sg.start_bit - 1u) & 31u) + 1u;
    const uint32_t g = ((sg.start_bit + 32u - o0) >> 5) + 3u;            // dword (front pad included) that becomes w0
    const uint32_t qbase = (g >> 2) < qmax ? (g >> 2) : qmax;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint4 v = Q[qbase + k < qmax ? qbase + k : qmax];
        la_st4(ring_b | ((k * 16u + rot) & 63u), be32x4(v));
    }
    // Ring bookkeeping in byte units with the lane's rotation folded in:
    //   rpb = 4 * (dwords taken from the ring) + rot      wqb = 16 * (quads written to the ring) + rot
    // (both monotonic; the ring address of a byte index x is ring_b | (x & 63), ring_b being 64-aligned)
    uint32_t wqb = 64u + rot;
    uint32_t rpb = (g & 3u) * 4u + rot;
    wave_lds_order();
    auto ring_at = [&](uint32_t xb) { return la_u32(ring_b | (xb & 63u)); };
    uint32_t w0 = ring_at(rpb), w1 = ring_at(rpb + 4), nxt = ring_at(rpb + 8);
    rpb += 12;
    uint32_t sh = 32u - o0;                                              // window = alignbit(w0, w1, sh), sh in [0,31]
    uint4 pq0 = make_uint4(0, 0, 0, 0), pq1 = pq0;                       // quads requested, not yet in the ring
    uint32_t npend = 0;

    int p0 = sg.p0, p1 = sg.p1, p2 = sg.p2;
    const int keep = raw_dc ? 0 : -1;                                    // raw_dc: the predictors stay 0
    uint32_t c = sg.nblk ? sg.blk0 % bpm : 0u;
    uint32_t bad = 0;
    int16_t* const gbase = a.coef + im.coef_off;
    uint32_t nmax = sg.nblk;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = __shfl_xor(nmax, d, 64); nmax = t > nmax ? t : nmax; }

    // flush plan: a block leaves as ONE 128-byte line written by 8 lanes (16 B each) of one store instruction -- a
    // 64-byte store into a line that is not in L2 makes L2 fetch the line, and half-block stores a phase apart cost the
    // restart-interval writer a quarter of its time.  In pass q lanes 8t'..8t'+7 move the block of lane
    // t = 8q + (lane >> 3): lanes with bit 2 clear pick up their piece of the first half when that half is complete
    // and hold it in registers; lanes with bit 2 set pick up the second half; then all eight store.
    const uint32_t part = lane & 3u, f_hi = (lane >> 2) & 1u;
    uint32_t blk_first = sg.nblk ? sg.blk0 : 0xFFFFFFFFu;                // the wave's first block: store offsets fit 32 bits
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = __shfl_xor(blk_first, d, 64); blk_first = t < blk_first ? t : blk_first; }
    blk_first = __builtin_amdgcn_readfirstlane(blk_first == 0xFFFFFFFFu ? 0u : blk_first);
    uint8_t* const wave_dst = reinterpret_cast<uint8_t*>(gbase + (uint64_t)blk_first * 64u);
    uint32_t f_nblk[8], f_off[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
        const uint32_t t = q * 8u + (lane >> 3);
        f_nblk[q] = __shfl(sg.nblk, t, 64);
        const uint32_t tb0 = __shfl(sg.blk0, t, 64);                     // (unconditional: a shuffle reads active lanes only)
        f_off[q] = f_nblk[q] ? (tb0 - blk_first) * 128u + (lane & 7u) * 16u : 0u;
    }
    const uint32_t f_src = wbase + (lane >> 3) * 64u + ((part ^ ((lane >> 3) & 3u)) << 4);   // + q * 512
    uint4 f_hold[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) f_hold[q] = make_uint4(0, 0, 0, 0);

    auto quad_index = [&](uint32_t xb) { const uint32_t q = qbase + ((xb - rot) >> 4); return q < qmax ? q : qmax; };
    auto request = [&](bool want) {                                      // up to two quads into the free part of the ring
        const uint32_t used = (wqb >> 4) - (rpb >> 4);                   // quads holding unread data (<= 4); rot cancels
        const uint32_t freeq = 4u - used;
        npend = want ? (freeq < 2u ? freeq : 2u) : 0u;
        if (npend >= 1) pq0 = Q[quad_index(wqb)];
        if (npend >= 2) pq1 = Q[quad_index(wqb + 16u)];
    };
    // Requested quads are committed just before the block stores of the same half step are issued: vmcnt counts in
    // order, so a wait placed after those stores would also wait for them to reach memory.
    auto commit = [&]() {
        if (npend >= 1) la_st4(ring_b | (wqb & 63u), be32x4(pq0));
        if (npend >= 2) la_st4(ring_b | ((wqb + 16u) & 63u), be32x4(pq1));
        wqb += npend * 16u; npend = 0;
    };
    // The ring word that the next refill needs is read at the top of every symbol step, beside the table
    // lookup (`cand`).  A lane whose read pointer caught up with its write pointer gets one more quad on the
    // spot; rare, so the test is one wave-wide branch.
    auto top_up_now = [&]() {
        if (rpb >= wqb) {
            commit();
            if (rpb >= wqb) {
                const uint4 v = Q[quad_index(wqb)];
                la_st4(ring_b | (wqb & 63u), be32x4(v));
                wqb += 16u;
            }
        }
        wave_lds_order();
    };
    // consume the entry's bits (e[4:0] = 32 - bits): sh + that is < 32 exactly when the window has moved past w0
    auto advance = [&](uint32_t e, uint32_t cand) {
        const uint32_t t = sh + (e & 31u);
        const bool r = t < 32u;
        sh = t & 31u;
        w0 = r ? w1 : w0; w1 = r ? nxt : w1; nxt = r ? cand : nxt;
        rpb += r ? four : 0u;
    };
    auto flush = [&](uint32_t b, uint32_t half) {
        if (f_hi == half) {                                              // this lane's half is complete: take it, clear the slot
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                f_hold[q] = la_u4(f_src + q * 512u);
                la_st4(f_src + q * 512u, make_uint4(0, 0, 0, 0));
            }
        }
        if (half == 1u) {
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                if (b < f_nblk[q]) {
                    jd_u32x4 t; t.x = f_hold[q].x; t.y = f_hold[q].y; t.z = f_hold[q].z; t.w = f_hold[q].w;
                    jd_u32x4* at = reinterpret_cast<jd_u32x4*>(wave_dst + f_off[q] + (uint64_t)b * 128u);
                    // non-temporal: a plain store makes L2 fetch the line it is about to overwrite (FETCH_SIZE showed
                    // one line read per line written -- 1.6 GB per 8 images); both writers use the streaming form
                    if (NT) __builtin_nontemporal_store(t, at); else *at = t;
                }
            }
        }
    };

    // `o` = byte offset, inside the current half block, of the coefficient the last symbol produced.  A symbol adds
    // its advance (2 * (run + 1); 254 for EOB, far outside either half) and stores its value at the new offset when
    // that lies in the half being assembled (else to the dump word); ZRL stores a zero on a zero.
    const uint32_t idle = 1u << 20;
#pragma unroll 1
    for (uint32_t b = 0; b < nmax; b++) {
        request(b < sg.nblk);
        wave_lds_order();
        // ---- tables of this block's component -----------------------------------------------------------
        const uint32_t comp = c < nl ? 0u : c - nl + 1u;
        const uint32_t dcb = sel3(comp, dc0, dc1, dc2);
        const uint32_t acb = sel3(comp, ac0, ac1, ac2);
        uint32_t o = idle, e = 0;
        int v = 0;
        if (b < sg.nblk) {
            // DC: category, magnitude bits, EXTEND, prediction (main.go:188-201)
            if (__builtin_amdgcn_ballot_w64(rpb >= wqb)) top_up_now();
            const uint32_t cand = ring_at(rpb);
            const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
            e = w32_lookup(dcb, win, JD_W32_DC_BITS, bad);
            const int diff = w32_value(win, e);
            advance(e, cand);
            const int pred = (int)sel3(comp, (uint32_t)p0, (uint32_t)p1, (uint32_t)p2) + diff;
            const int pk = pred & keep;
            p0 = comp == 0 ? pk : p0; p1 = comp == 1 ? pk : p1; p2 = comp >= 2 ? pk : p2;
            // coefficients are held as int16 (the reference: int): a predicted DC outside it can only come from a
            // damaged stream -- flagged, not wrapped silently
            if (pred != (int)(short)pred) err |= JD_ST_COEF;
            la_st16(slot_b + rot, pred);   // coefficient 0: chunk 0 ^ (lane & 3)
            o = 0;
        }
        // ---- AC, first half: symbols until every lane has passed coefficient 31 (main.go:204-239) ----------
        while (o < 62u) {                                         // divergent loop: finished lanes drop out
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rpb >= wqb) != 0, 0)) top_up_now();
            const uint32_t cand = ring_at(rpb);
            const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
            e = w32_lookup(acb, win, ac_bits, bad);
            v = w32_value(win, e);
            advance(e, cand);
            o += (e >> 8) & 255u;
            const uint32_t at = o < 64u ? slot_b + (o ^ rot) : dump_b;
            la_st16(at, v);
        }
        commit();
        wave_lds_order();
        flush(b, 0);
        wave_lds_order();
        o -= 64u;                                                 // offsets of the second half (wraps below zero for 62)
        request((int)o < 62);
        // the symbol that crossed into the second half left its coefficient for now
        if (o < 64u) la_st16(slot_b + (o ^ rot), v);
        // ---- AC, second half ----------------------------------------------------------------------------------
        while ((int)o < 62) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rpb >= wqb) != 0, 0)) top_up_now();
            const uint32_t cand = ring_at(rpb);
            const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
            e = w32_lookup(acb, win, ac_bits, bad);
            v = w32_value(win, e);
            advance(e, cand);
            o += (e >> 8) & 255u;
            const uint32_t at = o < 64u ? slot_b + (o ^ rot) : dump_b;
            la_st16(at, v);
        }
        // a block ends at offset 62 exactly (coefficient 63) or by EOB (>= 254 - 64 - 2); in between = a run that
        // walked past coefficient 63
        if ((int)o > 62 && o < 124u) err |= JD_ST_COEF;
        commit();
        wave_lds_order();
        flush(b, 1);
        wave_lds_order();
        c = (c + 1u == bpm) ? 0u : c + 1u;
    }
    if (sg.nblk) {
        if (bad) err |= JD_ST_HUFF;
        // bit position = 32 * (dwords consumed before w0) + bits of w0 consumed; w0 is ring dword (rp - 3) and
        // the front pad holds 4 dwords
        const uint32_t rp = (rpb - rot) >> 2;
        const uint32_t pos = ((qbase * 4u + rp - 3u) - 4u) * 32u + (32u - sh);
        if (pos > sg.end_bit) err |= JD_ST_TRUNC;                 // ran past the segment: data ended early
    }
    if (err) atomicOr(&a.status[img], err);
}

// ---- whole-block writer, byte form (images flagged JD_IMG_COEF8) ------------------------------------------------
// Same walk as write_segments, but a coefficient is stored as ONE byte, so a lane's 64-byte slot holds a whole block: one
// symbol loop per block instead of two half-block phases (a wave waits for its slowest lane once per block: 25 % over the
// mean symbol count on the benchmark images instead of 37 %), one flush per block, half the coefficient bytes to HBM.
// Block in the byte plane (64 B, zig-zag order):
//   byte 0       DC, low 8 bits                    byte 63 [3:0]  DC, bits 11..8
//   bytes 1..62  AC coefficients as int8           byte 63 [7:4]  AC coefficient 63 in -7 .. 7, or 8 = WIDE block
// A block with an AC magnitude category >= 8 (|v| >= 128), |AC 63| > 7 or a predicted DC outside 12 bits (no 8-bit
// encoder produces one) is WIDE: its lane (from the high parts it kept in registers; with more than four such coefficients
// by decoding the block once more, straight from the unstuffed stream in global memory)
// writes the block's 64 bytes of the HIGH plane (second half of the image's coefficient region): byte 0 = DC bits 15..8,
// bytes 1..62 = (v - int8(v)) >> 8 modulo 256 (zero where the low byte says it all), byte 63 = AC 63's low byte -- whose bits 11..8
// take the DC's place in byte 63 [3:0] of the low plane.  K2 reads the high plane only for blocks that carry the mark.
// Limits (JD_ST_COEF beyond them, as the int16 form does at 16 bits): DC 16 bits, AC 1..62 16 bits, AC 63 12 bits.
// (Images whose quantisers make wide blocks common keep the int16 form: jd_api.cpp picks the form per image.)
// A block leaves as ONE 64-byte store instruction's worth (4 lanes x 16 B, non-temporal) as soon as it is complete.  (A
// first version held the even block of a 128-byte line in registers until its partner was ready, as the int16 form does
// with half blocks: 32 more VGPRs, spills at four waves per SIMD, and 12 % slower than two half-line stores.)
// FUSED (the single-decode path of streams without restart markers, jd_internal.h): the segment is a sub-sequence given by
// BITS -- the lane decodes the blocks that start in [sg.start_bit, sg.end_bit), however many they are -- the blocks go to
// staging slots (chunks of JD_F_CHUNK, handed out as the walk reaches them), DC values are relative to the segment's start
// and errors are recorded with the block they happened in instead of raised (a speculative decode of garbage is not an
// error of the image): the lane's results are its JdSub record.
struct FusedSeg {
    bool valid;              // the lane has a sub-sequence
    uint32_t c0;             // block-in-MCU index of its first block
    uint32_t entry;          // the entry state it decodes from (recorded for the next check)
    uint32_t old_chunks;     // overflow chunks an earlier decode of the sub-sequence already owns
    uint32_t gsub;           // index of its record and chunk row (subseq_base + s)
    uint32_t slot0;          // first of the slots it owns (s * f_ch0)
    bool cutoff;             // (wave-uniform) first decode: a wave does not wait for its last few lanes -- they are most likely
                             // the ones that started from a wrong guess and walk through garbage; their record says "decode
                             // me again" and the next round takes them (a decode from a checked entry state is never cut)
    uint32_t guess_exit;     // ... and what the record of a lane that was cut hands its successor: the blind guess it used anyway
};
template <bool NT, bool FUSED = false>
__device__ __forceinline__ void write_segments8(const JdKernelArgs& a, const JdImg& im, int img, uint32_t lut_bytes,
                                                const Segment& sg, uint32_t slen, uint32_t err, const FusedSeg* fs = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lbase = lds_abs_base();
    const uint32_t wbase = lbase + lut_bytes + wv * JD_WAVE_LDS;
    const uint32_t slot_b = wbase + lane * 64u;
    const uint32_t ring_b = wbase + 4096u + lane * 64u;
    uint32_t dump_b = wbase + 8192u;
    uint32_t four = 4u;
    asm volatile("" : "+v"(dump_b), "+v"(four));
    const uint32_t rot = (lane & 3u) << 4;
#pragma unroll
    for (int i = 0; i < 4; i++) la_st4(slot_b + i * 16, make_uint4(0, 0, 0, 0));
    const uint32_t bpm = im.bpm, nl = im.hs * im.vs;
    const uint32_t ac_bits = __builtin_amdgcn_readfirstlane(im.w_ac_bits);
    const uint32_t dc0 = lbase + im.wdc[0] * 4u, dc1 = lbase + im.wdc[1] * 4u, dc2 = lbase + im.wdc[2] * 4u;
    const uint32_t ac0 = lbase + im.wac[0] * 4u, ac1 = lbase + im.wac[1] * 4u, ac2 = lbase + im.wac[2] * 4u;

    // ---- prime the ring (as write_segments) -------------------------------------------------------------------
    const uint4* Q = reinterpret_cast<const uint4*>(a.stream + im.stream_off);
    const uint32_t qmax = (JD_STREAM_FRONT + slen + JD_STREAM_TAIL) / 16u - 1u;
    const uint32_t o0 = ((sg.start_bit - 1u) & 31u) + 1u;
    const uint32_t g = ((sg.start_bit + 32u - o0) >> 5) + 3u;
    const uint32_t qbase = (g >> 2) < qmax ? (g >> 2) : qmax;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint4 v = Q[qbase + k < qmax ? qbase + k : qmax];
        la_st4(ring_b | ((k * 16u + rot) & 63u), be32x4(v));
    }
    uint32_t wqb = 64u + rot;
    uint32_t rpb = (g & 3u) * 4u + rot;
    wave_lds_order();
    auto ring_at = [&](uint32_t xb) { return la_u32(ring_b | (xb & 63u)); };
    uint32_t w0 = ring_at(rpb), w1 = ring_at(rpb + 4), nxt = ring_at(rpb + 8);
    rpb += 12;
    // `sh` (how far the window's first bit sits below the top of w0) is kept with every bit above its five set: a
    // shift operand reads the low five anyway, and adding an entry's 32 - bits then CARRIES exactly when the window
    // stays in w0 -- v_add_co gives the new count and the select mask in one instruction (no separate compare)
    uint32_t sh = (32u - o0) | ~31u;
    uint4 pq0 = make_uint4(0, 0, 0, 0), pq1 = pq0;
    uint32_t npend = 0;

    int p0 = sg.p0, p1 = sg.p1, p2 = sg.p2;
    uint32_t c = FUSED ? fs->c0 : (sg.nblk ? sg.blk0 % bpm : 0u);
    uint32_t bad = 0;
    uint32_t nsteps = sg.nblk;
    if constexpr (!FUSED) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = __shfl_xor(nsteps, d, 64); nsteps = t > nsteps ? t : nsteps; }
    }
    const uint32_t nblocks = im.n_mcus * bpm;
    uint8_t* const lo_plane = reinterpret_cast<uint8_t*>(a.coef + im.coef_off);
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    g_u8* const hi_plane = (g_u8*)(lo_plane + (FUSED ? JD_F_HIPOOL_OFF(im.slot_cap) : (uint64_t)nblocks * 64u));
    // FUSED: first slot of the chunk the walk is in (this lane's, and those of the four lanes whose blocks it helps to
    // flush), blocks decoded, step of the first error, "ran out of staging"; wave-uniform: steps into the chunk, its size,
    // overflow chunks so far
    uint32_t cur_base = FUSED ? fs->slot0 : 0u, f_base[4] = { 0, 0, 0, 0 }, my_nblk = 0, err_st = 0xFFFFu, err_first = 0;
    bool stopped = false, cut = false;
    uint32_t n_first = 0;
    // A lane whose first block is an odd one of its MCU starts at the second slot of every run (a run holds two blocks less
    // than it has slots): blocks 2k and 2k + 1 of an MCU then share a 128-byte line, as in the dense layout -- K2 reads them
    // in consecutive steps and the second read finds the line in L2 (a pair split over two lines is read five steps apart).
    uint32_t sc = 0, lim = FUSED ? __builtin_amdgcn_readfirstlane(im.f_ch0) - 2u : 0u, kq = 0;
    const uint32_t par = FUSED ? (fs->c0 & 1u) : 0u;
    if constexpr (FUSED) cur_base += par;
    uint32_t* const ctab = FUSED ? a.chunk + (uint64_t)fs->gsub * a.fkmax : nullptr;
    if constexpr (FUSED) {
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) f_base[q] = (uint32_t)__shfl((int)cur_base, (int)(q * 16u + (lane >> 2)), 64);
    }

    // flush plan: a block leaves as 64 bytes written by 4 lanes (16 B each) of one store instruction; in pass q lanes
    // 4t' .. 4t'+3 move the block of lane t = 16q + (lane >> 2)
    const uint32_t part = lane & 3u;
    uint32_t blk_first = sg.nblk ? sg.blk0 : 0xFFFFFFFFu;
    uint32_t f_end[4] = { 0, 0, 0, 0 }, f_off[4] = { 0, 0, 0, 0 };
    if constexpr (!FUSED) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t t = __shfl_xor(blk_first, d, 64); blk_first = t < blk_first ? t : blk_first; }
        blk_first = __builtin_amdgcn_readfirstlane(blk_first == 0xFFFFFFFFu ? 0u : blk_first);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t t = q * 16u + (lane >> 2);
            f_end[q] = __shfl(sg.nblk, t, 64);
            const uint32_t tb = __shfl(sg.blk0, t, 64);
            f_off[q] = f_end[q] ? (tb - blk_first) * 64u + part * 16u : 0u;
        }
    } else blk_first = 0;
    uint8_t* const wave_dst = lo_plane + (uint64_t)blk_first * 64u;
    const uint32_t f_src = wbase + (lane >> 2) * 64u + ((part ^ ((lane >> 2) & 3u)) << 4);   // + q * 1024
    auto quad_index = [&](uint32_t xb) { const uint32_t q = qbase + ((xb - rot) >> 4); return q < qmax ? q : qmax; };
    auto request = [&](bool want) {
        const uint32_t used = (wqb >> 4) - (rpb >> 4);
        const uint32_t freeq = 4u - used;
        npend = want ? (freeq < 2u ? freeq : 2u) : 0u;
        if (npend >= 1) pq0 = Q[quad_index(wqb)];
        if (npend >= 2) pq1 = Q[quad_index(wqb + 16u)];
    };
    auto commit = [&]() {
        if (npend >= 1) la_st4(ring_b | (wqb & 63u), be32x4(pq0));
        if (npend >= 2) la_st4(ring_b | ((wqb + 16u) & 63u), be32x4(pq1));
        wqb += npend * 16u; npend = 0;
    };
    // A lane whose read pointer caught up with its write pointer gets quads on the spot (rare).  Its pending request is
    // DROPPED, not committed: with the commit's byte swap of the requested quads visible inside the symbol loop, hipcc
    // hoists it -- and the wait for the loads -- in front of the loop that is there to hide them.
    auto top_up_now = [&]() {
        if (rpb >= wqb) {
            npend = 0;
            const uint4 v = Q[quad_index(wqb)];
            la_st4(ring_b | (wqb & 63u), be32x4(v));
            wqb += 16u;
        }
        wave_lds_order();
    };
    auto advance = [&](uint32_t e, uint32_t cand) {
        uint32_t t;
        const bool stay = __builtin_add_overflow(sh, e & 31u, &t);
        sh = t | ~31u;
        w0 = stay ? w0 : w1; w1 = stay ? w1 : nxt; nxt = stay ? nxt : cand;
        rpb += stay ? 0u : four;
    };
    auto flush = [&](uint32_t st, uint64_t am) {                          // am (FUSED): the lanes that decoded a block in this step
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint4 h = la_u4(f_src + q * 1024u);
            la_st4(f_src + q * 1024u, make_uint4(0, 0, 0, 0));
            if (FUSED ? ((am >> (q * 16u + (lane >> 2))) & 1ull) != 0 : st < f_end[q]) {
                jd_u32x4 t; t.x = h.x; t.y = h.y; t.z = h.z; t.w = h.w;
                jd_u32x4* at = FUSED ? reinterpret_cast<jd_u32x4*>(lo_plane + (uint64_t)(f_base[q] + sc) * 64u + part * 16u)
                                     : reinterpret_cast<jd_u32x4*>(wave_dst + f_off[q] + (uint64_t)st * 64u);
                if (NT) __builtin_nontemporal_store(t, at); else *at = t;
            }
        }
    };
    auto cur_pos = [&]() { const uint32_t rp = (rpb - rot) >> 2; return ((qbase * 4u + rp - 3u) - 4u) * 32u + (32u - (sh & 31u)); };
    // The high plane of a wide block.  The symbol loop keeps the high parts of up to four coefficients of magnitude
    // category >= 8 (position << 8 | high part, 16 bits each, in wl0 / wl1); with more than four the block is decoded ...
    auto wide_from_list = [&](uint32_t blk, int pred, int b63, uint32_t l0, uint32_t l1, uint32_t cnt) -> uint32_t {
        g_u8* hib = hi_plane + (uint64_t)blk * 64u;
        typedef __attribute__((address_space(1))) jd_u32x4 g_u4;
        jd_u32x4 z; z.x = z.y = z.z = z.w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) ((g_u4*)hib)[i] = z;
        hib[0] = (uint8_t)((uint32_t)pred >> 8);
        int hi63 = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (i < cnt) {
                const uint32_t ent = ((i < 2 ? l0 : l1) >> (16u * (i & 1u))) & 0xFFFFu, k = ent >> 8;
                if (k == 63u) hi63 = (int)(int8_t)ent; else if (k < 63u) hib[k] = (uint8_t)ent;
            }
        }
        const int ac63 = (int)(short)(b63 + hi63 * 256);
        hib[63] = (uint8_t)ac63;
        if (ac63 != ((ac63 << 20) >> 20)) err |= JD_ST_COEF;
        return ((uint32_t)ac63 >> 8) & 15u;
    };
    // ... once more: symbols straight from global memory (the ring has moved on), high-plane bytes out.
    auto wide_block = [&](uint32_t pos, uint32_t dcb, uint32_t acb, uint32_t blk, int pred) -> uint32_t {
        const uint32_t* S = reinterpret_cast<const uint32_t*>(a.stream + im.stream_off + JD_STREAM_FRONT);
        const uint32_t imax = (slen + JD_STREAM_TAIL) / 4u - 2u;
        auto window = [&](uint32_t p) {
            uint32_t i = p >> 5; i = i < imax ? i : imax;
            const uint32_t d0 = be32(S[i]), d1 = be32(S[i + 1]);
            return (p & 31u) ? __builtin_amdgcn_alignbit(d0, d1, 32u - (p & 31u)) : d0;
        };
        g_u8* hib = hi_plane + (uint64_t)blk * 64u;
        typedef __attribute__((address_space(1))) jd_u32x4 g_u4;
        jd_u32x4 z; z.x = z.y = z.z = z.w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) ((g_u4*)hib)[i] = z;
        hib[0] = (uint8_t)((uint32_t)pred >> 8);
        uint32_t junk = 0, ac63_hi = 0;
        const uint32_t ed = w32_lookup(dcb, window(pos), JD_W32_DC_BITS, junk);
        pos += 32u - (ed & 31u);
        auto high_part = [&](uint32_t k, int v) {
            if (k == 63u) {
                hib[63] = (uint8_t)v; ac63_hi = ((uint32_t)v >> 8) & 15u;
                if (v != ((v << 20) >> 20)) err |= JD_ST_COEF;
            } else {
                const int hc = (v - (int)(int8_t)v) >> 8;
                if (hc != 0) hib[k] = (uint8_t)hc;
            }
        };
        uint32_t k = 0;
        while (k < 63u) {
            const uint32_t win = window(pos);
            uint32_t e = la_u32(acb + ((win >> (30u - ac_bits)) & (((1u << ac_bits) - 1u) << 2)));
            if ((int)e < 0 && ((e >> 8) & 255u) == 0u) {
                const uint32_t so = (e >> 16) & 0xFFFu;
                e = so ? la_u32(acb + (so << 2) + ((win >> 14) & ((0xFFFFu >> ac_bits) << 2))) : JD_W32_NONE;
                if (((e >> 8) & 255u) == 0u) e = JD_WP_SKIP;
            }
            k += (e >> 8) & 255u;
            if (k > 63u) break;
            high_part(k, wp_value(win, e >> 23, e >> 28));
            const uint32_t adv2 = (e >> 16) & 127u;
            if (k == 63u || adv2 == 0u) { pos += 32u - ((e >> 23) & 31u); continue; }   // symbol 1 alone
            pos += 32u - (e & 31u);
            k += adv2;
            if (k > 63u) break;
            high_part(k, wp_value(win, e, (e >> 5) & 7u));
        }
        return ac63_hi;
    };

    // `o` = zig-zag index of the coefficient the last symbol produced.  A symbol adds its advance (run + 1; 127 for EOB)
    // and stores its value's low byte at the new index when that is inside the block (else to the dump byte).
    const uint32_t idle = 1u << 20;
#pragma unroll 1
    for (uint32_t st = 0; FUSED || st < nsteps; st++) {
        bool act = st < sg.nblk;                                          // this lane decodes block st of its segment at step st
        uint64_t act_mask = 0;
        if constexpr (FUSED) {
            act = fs->valid && !stopped && cur_pos() < sg.end_bit;        // a block that STARTS inside the sub-sequence is its own
            act_mask = __builtin_amdgcn_ballot_w64(act);
            if (act_mask == 0) break;
            if (st == 0) n_first = (uint32_t)__popcll(act_mask);
            if (fs->cutoff && n_first >= 32u && st >= 8u && (uint32_t)__popcll(act_mask) * JD_F_CUT_DIV <= n_first) { cut = act; break; }
            if (sc == lim) {                                              // (wave-uniform) the lanes still walking need an overflow chunk
                if (act && kq >= a.fkmax) { stopped = true; act = false; atomicOr(&a.status[img], (uint32_t)JD_ST_FOVF); }
                const bool have = kq < fs->old_chunks;
                uint32_t ch = (act && have) ? ctab[kq] : 0u;
                const uint64_t need = __builtin_amdgcn_ballot_w64(act && !have);
                if (need) {                                               // one atomic per wave
                    const uint32_t first = (uint32_t)__builtin_ctzll(need);
                    uint32_t base = 0;
                    if (lane == first) base = atomicAdd(a.falloc + img, (uint32_t)__popcll(need));
                    base = (uint32_t)__shfl((int)base, (int)first, 64);
                    if (act && !have) {
                        ch = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                        if ((uint64_t)im.f_ov_base + ((uint64_t)ch + 1u) * JD_F_CHUNK > im.slot_cap) { ch = 0; stopped = true; act = false; atomicOr(&a.status[img], (uint32_t)JD_ST_FOVF); }
                        else ctab[kq] = ch;
                    }
                }
                cur_base = im.f_ov_base + ch * JD_F_CHUNK + par;
                act_mask = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) f_base[q] = (uint32_t)__shfl((int)cur_base, (int)(q * 16u + (lane >> 2)), 64);
                sc = 0; lim = JD_F_CHUNK - 2u; kq++;
            }
            my_nblk += act ? 1u : 0u;
        }
        request(act);
        wave_lds_order();
        const uint32_t comp = c < nl ? 0u : c - nl + 1u;
        const uint32_t dcb = sel3(comp, dc0, dc1, dc2);
        const uint32_t acb = sel3(comp, ac0, ac1, ac2);
        uint32_t o = idle, e = 0, wl0 = 0, wl1 = 0, wcnt = 0, dchi = 0;
        bool wide_dc = false;
        int dcv = 0;
        const uint32_t rpb0 = rpb, sh0 = sh & 31u;                        // where this block starts (for wide_block)
        if (act) {
            if (__builtin_amdgcn_ballot_w64(rpb >= wqb)) top_up_now();
            const uint32_t cand = ring_at(rpb);
            const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
            e = w32_lookup(dcb, win, JD_W32_DC_BITS, bad);
            const int diff = w32_value(win, e);
            advance(e, cand);
            const int pred = (int)sel3(comp, (uint32_t)p0, (uint32_t)p1, (uint32_t)p2) + diff;
            p0 = comp == 0 ? pred : p0; p1 = comp == 1 ? pred : p1; p2 = comp >= 2 ? pred : p2;
            if (pred != (int)(short)pred) err |= JD_ST_COEF;             // as the int16 form
            wide_dc = pred != ((pred << 20) >> 20);                      // outside 12 bits: a wide block
            la_st8(slot_b + rot, pred);
            dchi = ((uint32_t)pred >> 8) & 15u;
            dcv = pred;
            o = 0;
        }
        int v = 0;
        // One lookup resolves up to two symbols (pair tables, JD_WP_*): the steps of the dependent chain -- window, table
        // look-up, length, window -- are what the loop waits for, and at quality 85 a lookup now covers 1.4 symbols.
        while (o < 63u) {                                                 // divergent loop: finished lanes drop out
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(rpb >= wqb) != 0, 0)) top_up_now();
            const uint32_t cand = ring_at(rpb);
            const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
            e = la_u32(acb + ((win >> (30u - ac_bits)) & (((1u << ac_bits) - 1u) << 2)));
            // ONE test for everything rare: bit 31 marks a secondary-table pointer, "no such code" and a first symbol of
            // magnitude category >= 8 (always a single)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64((int)e < 0) != 0, 0) && (int)e < 0) {   // (one wave-wide branch in the common case)
                if (((e >> 8) & 255u) == 0u) {
                    const uint32_t so = (e >> 16) & 0xFFFu;
                    e = so ? la_u32(acb + (so << 2) + ((win >> 14) & ((0xFFFFu >> ac_bits) << 2))) : JD_W32_NONE;
                    if (((e >> 8) & 255u) == 0u) { bad = 1; e = JD_WP_SKIP; }
                }
                if ((int)e < 0) {                                         // keep the high part (the low byte is stored below)
                    const int wv = wp_value(win, e >> 23, e >> 28);
                    const uint32_t wo = o + ((e >> 8) & 255u);
                    wl1 = (wl1 << 16) | (wl0 >> 16);
                    wl0 = (wl0 << 16) | ((wo & 255u) << 8) | (((uint32_t)(wv - (int)(int8_t)wv) >> 8) & 0xFFu);
                    wcnt++;
                }
            }
            const int v1 = wp_value(win, e >> 23, e >> 28);
            const uint32_t o1 = o + ((e >> 8) & 255u);
            const int v2 = wp_value(win, e, (e >> 5) & 7u);
            const uint32_t o2 = o1 + ((e >> 16) & 127u);
            // symbol 1 ends the block (coefficient 63, EOB, or a run past the end): symbol 2, if the entry has one, is not
            // this block's -- the window moves past symbol 1 only.  The lanes that stay in the loop are those whose LAST
            // coefficient of this step lies inside the block; a coefficient that completes the block is stored below, once.
            const bool d1 = o1 >= 63u;
            const uint32_t a1 = d1 ? dump_b : slot_b + (o1 ^ rot);
            const uint32_t a2 = (!d1 && o2 < 63u) ? slot_b + (o2 ^ rot) : dump_b;
            la_st8(a2, v2);                                               // (an empty second half lands on symbol 1's byte first ...
            la_st8(a1, v1);                                               //  ... LDS operations of a wave execute in order)
            v = d1 ? v1 : v2;
            o = d1 ? o1 : o2;
            advance(d1 ? (e >> 23) : e, cand);
        }
        if (o == 63u) la_st8(slot_b + (63u ^ rot), v);
        if (act) {
            // a block ends at index 63 exactly or by EOB (>= JD_WP_EOB_ADV); in between = a run that walked past coefficient 63
            if (o > 63u && o < JD_WP_EOB_ADV) err |= JD_ST_COEF;
            wave_lds_order();
            const int b63 = la_i8(slot_b + (63u ^ rot));
            const bool wide = wcnt != 0 || wide_dc || (uint32_t)(b63 + 7) > 14u;
            uint32_t low4 = dchi;
            if (__builtin_expect(wide, 0)) {
                uint32_t wblk = sg.blk0 + st;
                if constexpr (FUSED) {                                    // a high-part slot of its own, noted under the block's slot
                    wblk = atomicAdd(a.fhalloc + img, 1u);
                    if (wblk >= im.hi_cap) { wblk = 0; atomicOr(&a.status[img], (uint32_t)JD_ST_FOVF); }
                    reinterpret_cast<uint32_t*>(lo_plane + JD_F_HIMAP_OFF(im.slot_cap))[cur_base + sc] = wblk;
                }
                if (wcnt <= 4u) low4 = wide_from_list(wblk, dcv, b63, wl0, wl1, wcnt);
                else {
                    const uint32_t rp = (rpb0 - rot) >> 2;
                    const uint32_t pos = ((qbase * 4u + rp - 3u) - 4u) * 32u + (32u - sh0);
                    low4 = wide_block(pos, dcb, acb, wblk, dcv);
                }
            }
            la_st8(slot_b + (63u ^ rot), (int)((wide ? 0x80u : ((uint32_t)b63 << 4)) | low4));
        }
        commit();
        wave_lds_order();
        flush(st, act_mask);
        wave_lds_order();
        if (act) c = (c + 1u == bpm) ? 0u : c + 1u;
        if constexpr (FUSED) {
            sc++;
            const uint32_t now = err | (bad ? (uint32_t)JD_ST_HUFF : 0u);
            if (now != 0u && err_st == 0xFFFFu) { err_st = st; err_first = now; }
        }
    }
    if constexpr (FUSED) {
        if (fs->valid) {
            JdSub r;
            const uint32_t pos = cur_pos();
            r.entry = cut ? 0u : fs->entry;                               // (0 is no state: the next check lists the lane)
            r.exit = cut ? fs->guess_exit : JD_FSTATE(pos > sg.end_bit ? pos - sg.end_bit : 0u, c);
            r.nstart = my_nblk;
            // what went wrong: in all its blocks [7:0], in the first block that had an error [15:8], which block that was
            // [31:16] (the blocks behind the frame's last are garbage decoded from padding: their errors are nobody's)
            r.first = ((err | (bad ? (uint32_t)JD_ST_HUFF : 0u)) & 0xFFu) | ((err_first & 0xFFu) << 8) | (err_st << 16);
            r.dc[0] = p0; r.dc[1] = p1; r.dc[2] = p2;
            const uint32_t own = im.f_ch0 - 2u;
            const uint32_t nch = my_nblk > own ? (my_nblk - own + JD_F_CHUNK - 3u) / (JD_F_CHUNK - 2u) : 0u;   // overflow chunks it owns
            r.pad = nch > fs->old_chunks ? nch : fs->old_chunks;
            a.sub[fs->gsub] = r;
        }
        return;
    }
    if (sg.nblk) {
        if (bad) err |= JD_ST_HUFF;
        const uint32_t rp = (rpb - rot) >> 2;
        const uint32_t pos = ((qbase * 4u + rp - 3u) - 4u) * 32u + (32u - (sh & 31u));
        if (pos > sg.end_bit) err |= JD_ST_TRUNC;
    }
    if (err) atomicOr(&a.status[img], err);
}

// ---- restart-interval path: segment = restart interval ------------------------------------------------
// C8: the launch serves the images whose coefficients are stored in the byte form (JD_IMG_COEF8); the other launch the rest
template <bool C8>
__global__ __launch_bounds__(64 * JD_W_WAVES, 4) void k1_write_intervals(JdKernelArgs a, uint32_t lut_bytes) {
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (((im.flags & JD_IMG_COEF8) != 0) != C8) return;         // workgroup-uniform
    load_lut(reinterpret_cast<uint16_t*>(dyn_lds), a.lut + im.w_off, im.w_len * 2u);
    const uint32_t ri = im.restart_interval;
    const uint32_t iv = (blockIdx.x - a.prefix[img]) * (64u * JD_W_WAVES) + threadIdx.x;
    const uint32_t slen = a.stream_len[img];
    Segment sg; sg.start_bit = 0; sg.end_bit = 0; sg.nblk = 0; sg.blk0 = 0; sg.p0 = sg.p1 = sg.p2 = 0;
    uint32_t err = 0;
    if (iv < im.n_intervals) {
        const uint32_t b0 = a.interval_start[im.interval_base + iv], b1 = a.interval_start[im.interval_base + iv + 1];
        if (b0 > slen || b1 > slen || b1 < b0) err = JD_ST_RST_COUNT;
        else {
            uint32_t nm = im.n_mcus - iv * ri; nm = nm > ri ? ri : nm;
            sg.start_bit = b0 * 8u; sg.end_bit = b1 * 8u; sg.nblk = nm * im.bpm; sg.blk0 = iv * ri * im.bpm;
        }
    }
    if constexpr (C8) write_segments8<true>(a, im, img, lut_bytes, sg, slen, err);
    else write_segments<true>(a, im, img, lut_bytes, sg, slen, err);
}

// ---- no-DRI path, pass 1: self-synchronising sub-sequences --------------------------------------------
// Lane s decodes bits [s*S, (s+1)*S) of the stream (S = JD_SUBSEQ_BITS) starting from a guessed state; then
// entry states are propagated lane to lane (entry[s] = exit[s-1]) and changed lanes re-decode until nothing
// changes inside the workgroup.  fixup = 1 re-runs the propagation with lane 0 of each workgroup fed from
// the previous workgroup's last exit state.  (Klein & Wiseman 2003; Weissenberger & Schmidt 2021.)
//
// The decode is FREE-RUNNING: every lane walks its own symbols (DC and AC alike) in one loop, so a wave
// takes as long as its longest sub-sequence, not the sum of per-block maxima.  Each lane reads the stream
// through a private 32-dword LDS ring laid out [dword][lane] (conflict-free whatever the lanes' positions);
// when any lane of the wave runs dry, all lanes top their rings up at once, so the global loads come in
// bursts a couple of hundred symbols apart.  Per sub-sequence it records the exit state, where the first
// block starts, how many blocks start inside it and the sum of their DC differences per component -- what
// the spine scan turns into block indices and DC predictors for the write pass (the DC prefix scan).
#ifndef JD_SPEC_RING_DW
#define JD_SPEC_RING_DW 8u                                    // ring depth in dwords (2 quads; 4 quads cost a workgroup per CU)
#endif
#define JD_SPEC_RING_BYTES (JD_SPEC_RING_DW * 256u)           // per wave: [dword][lane]

// LDS map of k1_sync (byte offsets in dyn_lds)
#define JD_SY_RINGS 0u
#define JD_SY_DCSUM ((JD_SYNC_WG / 64u) * JD_SPEC_RING_BYTES)  // [3 components][JD_SYNC_WG lanes] ints
#define JD_SY_CTAB (JD_SY_DCSUM + 3u * JD_SYNC_WG * 4u)         // per block-in-MCU: dcb/4 | acb/4 << 15 | comp << 30
#define JD_SY_LUT (JD_SY_CTAB + 64u)

__device__ __forceinline__ int* lds_int(uint32_t off) { return reinterpret_cast<int*>(__builtin_assume_aligned(dyn_lds + off, 4)); }

// COLD = the first, blind decode: it starts JD_COLD_BITS before the END of the sub-sequence (its only product is
// a guess of the exit state, and a guess made from the last stretch is as good as one made from all of it)
// and skips the block-start bookkeeping.
#ifndef JD_SYNC_GATHER
#define JD_SYNC_GATHER 128u                                      // most lanes a gathered re-decode round takes
#endif
#ifndef JD_COLD_BITS
#define JD_COLD_BITS 2048u
#endif
// MODE 0 = COLD (above); 1 = a sub-sequence's first full decode: records its checkpoints (JdSubCk, taken at block
// starts); 2 = a later decode from a corrected entry state: stops at the first checkpoint whose state it reproduces and
// completes its counts from the record.  3 = the blind decode of the single-decode path (k1f_cold): as 0, but it walks on to
// the first block START past the sub-sequence's end and reports that (JD_FSTATE).
__device__ __forceinline__ uint32_t ck_state(int left, uint32_t c, uint32_t z) { return ((uint32_t)left << 12) | (c << 8) | z; }
// one checkpoint = two whole 16-byte non-temporal stores (stores of parts of a line make L2 fetch the line first)
__device__ __forceinline__ void ck_store(JdCkEntry* e, uint32_t state, uint32_t nstart, int d0, int d1, int d2) {
    jd_u32x4 lo, hi;
    lo.x = state; lo.y = nstart; lo.z = (uint32_t)d0; lo.w = (uint32_t)d1;
    hi.x = (uint32_t)d2; hi.y = 0; hi.z = 0; hi.w = 0;
    __builtin_nontemporal_store(lo, reinterpret_cast<jd_u32x4*>(e));
    __builtin_nontemporal_store(hi, reinterpret_cast<jd_u32x4*>(e) + 1);
}
template <int MODE>
__device__ __forceinline__ void spec_decode(const ImgK& ik, const uint4* Q, uint32_t qmax, const uint16_t* __restrict__ lutg,
                                            uint32_t s, uint32_t entry, uint32_t sbits, bool run, JdSub& rec,
                                            JdSubCk* __restrict__ ck = nullptr, bool use_ck = false, uint32_t fsub_bits = 0) {
    constexpr bool COLD = MODE == 0 || MODE == 3;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ring_l = JD_SY_RINGS + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * JD_SPEC_RING_BYTES + lane * 4u;
    auto ring_at = [&](uint32_t xb) { return lds_u32(ring_l | (xb & (JD_SPEC_RING_BYTES - 1u))); };
    const uint32_t dcs_l = JD_SY_DCSUM + threadIdx.x * 4u;
    // stream position: bit `start` of the unstuffed stream; data begins at quad 1 (front pad)
    const uint32_t kSubBits = MODE == 3 ? fsub_bits : JD_SUBSEQ_BITS;   // (the single-decode path has its own size, set per run)
    uint32_t end = (s + 1u) * kSubBits; if (end > sbits) end = sbits;
    uint32_t start = s * kSubBits + (entry & 0xFFu);
    if (COLD && end - start > JD_COLD_BITS) start = end - JD_COLD_BITS;
    const uint32_t o0 = ((start - 1u) & 31u) + 1u;
    const uint32_t g = ((start + 32u - o0) >> 5) + 3u;                    // dword (front pad included) that becomes w0
    const uint32_t qbase = (g >> 2) < qmax ? (g >> 2) : qmax;
    // ring bookkeeping in units of 256 bytes per dword (= the [dword][lane] row pitch)
    uint32_t rpb = (g & 3u) * 256u, wqb = 0;
    auto top_up = [&]() {                                                 // every lane fills every free quad
        constexpr uint32_t NQ = JD_SPEC_RING_DW / 4u;                     // quads in the ring
        const uint32_t used = (wqb >> 10) - (rpb >> 10);                  // quads holding unread dwords (<= NQ)
        // the loads are unconditional: under per-lane branches hipcc waits for each load before it issues the next, NQ
        // memory latencies instead of one.  A lane with no room for quad k reads the stream's first quad instead (one
        // line for everybody: always a cache hit), not a quad it would have to fetch again later.
        uint4 v[NQ];
#pragma unroll
        for (uint32_t k = 0; k < NQ; k++) {
            const uint32_t q = (run && used + k < NQ) ? qbase + (wqb >> 10) + k : 0u;
            v[k] = Q[q < qmax ? q : qmax];
        }
#pragma unroll
        for (uint32_t k = 0; k < NQ; k++) {
            if (run && used + k < NQ) {
                const uint32_t x = wqb + k * 1024u;
                *reinterpret_cast<uint32_t*>(dyn_lds + (ring_l | (x & (JD_SPEC_RING_BYTES - 1u)))) = be32(v[k].x);
                *reinterpret_cast<uint32_t*>(dyn_lds + (ring_l | ((x + 256u) & (JD_SPEC_RING_BYTES - 1u)))) = be32(v[k].y);
                *reinterpret_cast<uint32_t*>(dyn_lds + (ring_l | ((x + 512u) & (JD_SPEC_RING_BYTES - 1u)))) = be32(v[k].z);
                *reinterpret_cast<uint32_t*>(dyn_lds + (ring_l | ((x + 768u) & (JD_SPEC_RING_BYTES - 1u)))) = be32(v[k].w);
            }
        }
        wqb += run ? (NQ - used) * 1024u : 0u;
        wave_lds_order();
    };
    top_up();
    uint32_t w0 = ring_at(rpb), w1 = ring_at(rpb + 256u), nxt = ring_at(rpb + 512u);
    rpb += 768u;
    uint32_t sh = 32u - o0;
    int left = run ? (int)(end - start) : 0;                              // bits of the sub-sequence still to decode
    uint32_t c = (entry >> 8) & 0xFFu, z = (entry >> 16) & 0x7Fu;
    if (c >= ik.bpm) c = 0;
    uint32_t t = lds_u32(JD_SY_CTAB + c * 4u);
    // table bases as absolute LDS addresses, and the one the NEXT symbol reads kept in a register of its own (switched
    // where a block starts and where its DC symbol has been read: both rare paths run anyway) -- no select, no base
    // addition in the per-symbol path
    const uint32_t lut0 = lds_abs_base();
    uint32_t dcb = lut0 + ((t & 0x7FFFu) << 2), acb = lut0 + (((t >> 15) & 0x7FFFu) << 2), csum = dcs_l + ((t >> 30) * (JD_SYNC_WG * 4u));
    uint32_t tb = z ? acb : dcb;
    uint32_t nstart = 0, first = 0;
    *lds_int(dcs_l) = 0; *lds_int(dcs_l + JD_SYNC_WG * 4u) = 0; *lds_int(dcs_l + 2u * JD_SYNC_WG * 4u) = 0;
    wave_lds_order();
    // use_ck (per lane): the sub-sequence is one this workgroup owns.  A warm-up lane decodes a sub-sequence of the
    // workgroup before -- whose record that workgroup is writing at this very moment -- so it neither records nor stops early.
    //
    //
    // The decode is FREE-RUNNING: every lane walks its own symbols (DC and AC alike) in one loop.  (A block-synchronous
    // walk -- DC bookkeeping and table switch once per block for all lanes -- measured 2 % slower: waiting for the
    // wave's longest block costs what the leaner loop saves.)
    constexpr int kCkStep = (int)(JD_SUBSEQ_BITS / (JD_CK_N + 1));       // MODE 1: a checkpoint at the first block start past
    int thr = (int)JD_SUBSEQ_BITS - kCkStep;                             //         every kCkStep bits of the sub-sequence
    uint32_t ckj = 0;
    uint32_t tgt = 0, tj = 0;                                             // MODE 2: pending checkpoint's state and index
    int matched = -1;
    if (MODE == 2 && run && use_ck) tgt = ck->e[0].state;
    // (MODE 3 gives up a sub-sequence's length past the end: garbage that never completes a block is just a wrong guess)
    while (left > 0 || (MODE == 3 && z != 0u && left > -(int)JD_SUBSEQ_BITS)) {   // (a block is < 2000 bits)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rpb >= wqb) != 0, 0)) top_up();
        const uint32_t cand = ring_at(rpb);
        const uint32_t win = __builtin_amdgcn_alignbit(w0, w1, sh);
        uint32_t e = la_u16(tb + ((win >> (32 - JD_SLUT_BITS)) << 1));
        const bool special = (e & 0x8000u) != 0 || e == 0;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0) && special) {   // (one wave-wide branch in the common case)
            const uint32_t o = e & 0x7FFFu;
            e = (e && o) ? la_u16(tb + (o << 1) + (((win >> (32 - JD_SLUT_BITS - 6)) & 63u) << 1)) : 0u;
            if (e == 0) e = 1;                                            // speculative: step over one bit
        }
        uint32_t total = e & 31u;
        uint32_t zn = z + ((e >> 5) & 127u);                              // AC entries: A + 64 * E (jd_tables.cpp)
        if (z == 0) {                                                     // a block starts here: DC difference
            if (MODE == 1 && use_ck && left <= thr) {                     // (per lane, a few times per sub-sequence)
                if (ckj < JD_CK_N) {
                    wave_lds_order();
                    ck_store(&ck->e[ckj], ck_state(left, c, 0), nstart, *lds_int(dcs_l), *lds_int(dcs_l + JD_SYNC_WG * 4u),
                             *lds_int(dcs_l + 2u * JD_SYNC_WG * 4u));
                    ckj++;
                }
                thr -= kCkStep;
            }
            if (MODE == 2) {
                const uint32_t cur = ck_state(left, c, 0);
                while (__builtin_expect(cur <= tgt, 0)) {                 // at or past the pending checkpoint
                    if (cur == tgt) { matched = (int)tj; break; }
                    tj++;                                                 // passed it: wait for the next one
                    tgt = tj < JD_CK_N ? ck->e[tj < JD_CK_N ? tj : 0].state : 0u;
                }
                if (matched >= 0) {                                       // from here on this decode IS the recorded one
                    if (nstart == 0) first = (uint32_t)left;
                    break;
                }
            }
            if (!COLD) {
                if (nstart == 0) first = (uint32_t)left;                  // kept as "bits left" at that point
                nstart++;
                atomicAdd(lds_int(csum), extend_bits(win, total, (e >> 5) & 15u));   // ds_add, no return
            }
            zn = 1;
            tb = acb;
        } else if (zn >= 64u) {
            // the block ends in this group: fine if it ends with the group (index 64 reached by its last symbol, or
            // EOB before index 64); if the block filled up INSIDE the group, the rest of the group belongs to the
            // next block -- redo the step with one symbol from the single-symbol table
            const bool ok = zn == 64u || (zn - z >= 64u && zn < 128u);
            if (!ok) {
                const uint16_t* gt = lutg + sel3(t >> 30, ik.ac0, ik.ac1, ik.ac2);   // (global memory)
                uint32_t e1 = gt[win >> (32 - JD_LUT_BITS)];
                if (e1 & 0x8000u) e1 = (e1 & 0x7FFFu) ? gt[(e1 & 0x7FFFu) + ((win >> (32 - JD_LUT_BITS - 6)) & 63u)] : 0u;
                if (e1 == 0) e1 = 1;
                total = e1 & 31u;
                zn = z + ((e1 >> 9) & 63u) + 1u;
            }
        }
        // sh - total borrows exactly when the window moves past w0 (v_sub_co: the new count and the select mask at once)
        uint32_t sh2;
        const bool r = __builtin_sub_overflow(sh, total, &sh2);
        sh = sh2 & 31u; left -= (int)total;
        w0 = r ? w1 : w0; w1 = r ? nxt : w1; nxt = r ? cand : nxt;
        rpb += r ? 256u : 0u;
        z = zn;
        if (z >= 64u) {                                                   // block complete: next block's tables
            z = 0;
            c = (c + 1u == ik.bpm) ? 0u : c + 1u;
            t = lds_u32(JD_SY_CTAB + c * 4u);
            dcb = lut0 + ((t & 0x7FFFu) << 2); acb = lut0 + (((t >> 15) & 0x7FFFu) << 2); csum = dcs_l + ((t >> 30) * (JD_SYNC_WG * 4u));
            tb = dcb;
        }
    }
    wave_lds_order();
    if (run) {
        rec.entry = entry;
        rec.exit = MODE == 3 ? JD_FSTATE((uint32_t)(-left), c) : pack_state((uint32_t)(-left), c, z);
        rec.nstart = nstart;
        if (MODE == 2 && matched >= 0) rec.nstart += ck->e[JD_CK_N].nstart - ck->e[matched].nstart;   // (the recorded decode's block starts from here on)
        rec.first = rec.nstart ? (end - start) - first + (entry & 0xFFu) : 0u;   // bit offset from the sub-sequence start
        rec.dc[0] = *lds_int(dcs_l); rec.dc[1] = *lds_int(dcs_l + JD_SYNC_WG * 4u); rec.dc[2] = *lds_int(dcs_l + 2u * JD_SYNC_WG * 4u);
        if (MODE == 1 && use_ck) {
            // checkpoints the decode did not reach stay invalid; the totals close the record
            for (uint32_t j = ckj; j < JD_CK_N; j++) ck_store(&ck->e[j], 0u, 0u, 0, 0, 0);
            ck_store(&ck->e[JD_CK_N], rec.exit, rec.nstart, rec.dc[0], rec.dc[1], rec.dc[2]);
        }
        if (MODE == 2 && matched >= 0) {
            // from the checkpoint on this decode IS the recorded one: its exit state, its remaining counts
            const JdCkEntry tot = ck->e[JD_CK_N], at = ck->e[matched];
            rec.exit = tot.state;
            rec.dc[0] += tot.dc[0] - at.dc[0]; rec.dc[1] += tot.dc[1] - at.dc[1]; rec.dc[2] += tot.dc[2] - at.dc[2];
        }
    }
}

// Workgroup w owns sub-sequences [w * JD_SYNC_OWN, (w + 1) * JD_SYNC_OWN); its lanes 0 .. JD_SYNC_WARM-1 warm up
// on the JD_SYNC_WARM sub-sequences before them (results discarded).
// warm_off (debug knob, JD_DEBUG_SYNC_NO_WARM): the warm-up lanes stay idle, so that every workgroup boundary depends on
// the fix-up launches -- lets a test drive the host's extra fix-up loop.
__global__ __launch_bounds__(JD_SYNC_WG, 6) void k1_sync(JdKernelArgs a, int fixup, int flag_slot, int warm_off) {
    __shared__ uint32_t ex[JD_SYNC_WG];
    __shared__ uint32_t tmp[16];
    // re-decode list and its results (gathered rounds): they live in the stream rings of the waves that idle while the
    // first JD_SYNC_GATHER / 64 waves decode
    constexpr uint32_t kGatherWaves = (JD_SYNC_GATHER + 63u) / 64u;
    uint32_t* const re_src = reinterpret_cast<uint32_t*>(dyn_lds + JD_SY_RINGS + kGatherWaves * JD_SPEC_RING_BYTES);
    uint32_t* const re_ent = re_src + JD_SYNC_GATHER;
    uint32_t (*const re_out)[6] = reinterpret_cast<uint32_t (*)[6]>(re_src + 2u * JD_SYNC_GATHER);
    static_assert(JD_SYNC_GATHER * 32u <= (JD_SYNC_WG / 64u - kGatherWaves) * JD_SPEC_RING_BYTES, "gather lists must fit the idle rings");
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    const uint32_t lw = blockIdx.x - a.prefix[img];
    const uint32_t slen = a.stream_len[img];
    const uint32_t sbits = slen * 8u;
    const uint32_t nsub = (sbits + JD_SUBSEQ_BITS - 1u) / JD_SUBSEQ_BITS;
    if (lw * JD_SYNC_OWN >= nsub) return;                       // whole workgroup past the stream (uniform)
    ImgK ik = load_imgk(im);
    // every component decodes through the same table pair: the stream's path does not depend on the block-in-MCU
    // phase, which therefore never self-synchronises -- the pass runs without it (one "component", c = 0) and the
    // DC predictions come from k1_dc_scan instead of the per-component sums below
    if (im.flags & JD_IMG_UNIFORM_TABLES) { ik.bpm = 1; ik.nl = 1; }
    if (threadIdx.x < 16u) {
        const uint32_t cc = threadIdx.x < ik.bpm ? threadIdx.x : 0u;
        const uint32_t comp = cc < ik.nl ? 0u : cc - ik.nl + 1u;
        const uint32_t d = JD_SY_LUT + sel3(comp, im.sdc[0], im.sdc[1], im.sdc[2]) * 2u, ac = JD_SY_LUT + sel3(comp, im.sac[0], im.sac[1], im.sac[2]) * 2u;
        *reinterpret_cast<uint32_t*>(dyn_lds + JD_SY_CTAB + threadIdx.x * 4u) = (d >> 2) | ((ac >> 2) << 15) | (comp << 30);
    }
    load_lut(reinterpret_cast<uint16_t*>(dyn_lds + JD_SY_LUT), a.lut + im.lut_off + im.lut_len, im.s_len);   // sync twin set
    const int si = (int)(lw * JD_SYNC_OWN + threadIdx.x) - JD_SYNC_WARM;
    const uint32_t s = si < 0 ? 0u : (uint32_t)si;
    const bool owned = threadIdx.x >= JD_SYNC_WARM && s < nsub;
    const bool active = (fixup || warm_off) ? owned : (si >= 0 && s < nsub);
    const uint4* Q = reinterpret_cast<const uint4*>(a.stream + im.stream_off);
    const uint32_t qmax = (JD_STREAM_FRONT + slen + JD_STREAM_TAIL) / 16u - 1u;
    const uint32_t start_state = pack_state(0, 0, 0);
    JdSub rec; rec.entry = rec.exit = 0; rec.nstart = 0; rec.first = 0; rec.dc[0] = rec.dc[1] = rec.dc[2] = 0; rec.pad = 0;
    JdSub* gsub = a.sub + im.subseq_base;
    if (fixup) { if (active) rec = gsub[s]; }
    else {
        // blind decode: only its exit state is kept; entry 0 (not a valid state) makes every lane decode again
        // from its neighbour's exit in the first round of the loop below
        spec_decode<0>(ik, Q, qmax, a.lut + im.lut_off, s, start_state, sbits, active && s != 0, rec);
        rec.entry = 0;
    }
    const uint32_t exit_before = rec.exit;
    uint32_t prev_wg_exit = 0;
    if (threadIdx.x == JD_SYNC_WARM && s > 0 && fixup) prev_wg_exit = gsub[s - 1].exit;
    for (int it = 0; it <= JD_SYNC_WG; it++) {
        ex[threadIdx.x] = rec.exit;
        __syncthreads();
        uint32_t ne = rec.entry;
        if (active) {
            if (s == 0) ne = start_state;
            else if (fixup && threadIdx.x == JD_SYNC_WARM) ne = prev_wg_exit;
            else if (threadIdx.x > 0 && si > 0) ne = ex[threadIdx.x - 1];
        }
        const bool changed = active && ne != rec.entry;
        const uint64_t bm = __builtin_amdgcn_ballot_w64(changed);
        if ((threadIdx.x & 63u) == 0u) tmp[threadIdx.x >> 6] = (uint32_t)__popcll(bm);
        __syncthreads();
        static_assert(JD_SYNC_WG == 256, "k1_sync sums the changed-lane counts of exactly four waves (n0 .. n3) and places gathered lanes by them");
        const uint32_t n0 = tmp[0], n1 = tmp[1], n2 = tmp[2], n3 = tmp[3];
        const uint32_t n = n0 + n1 + n2 + n3;
        if (n == 0) break;
        // After the first round only a few lanes per workgroup (the ones whose blind guess was wrong, then the
        // ones after them) decode again: gathered into the first wave(s) they cost one wave's pass, not one per wave that
        // holds any of them.
        const bool gather = n <= JD_SYNC_GATHER;                    // workgroup-uniform
        const uint32_t wv = threadIdx.x >> 6;
        const uint32_t pos = (wv > 0 ? n0 : 0u) + (wv > 1 ? n1 : 0u) + (wv > 2 ? n2 : 0u) +
                             (uint32_t)__popcll(bm & ((1ull << (threadIdx.x & 63u)) - 1ull));
        // ck_x: the sub-sequence HAS checkpoints of this run (JdSub::pad).  Not every first full decode is a MODE 1 one:
        // sub-sequence 1 sits behind sub-sequence 0, which has no blind decode, so its entry state only changes in round 1
        // (and with JD_DEBUG_SYNC_NO_WARM every workgroup's first lane waits for a fix-up launch).  A MODE 2 decode that
        // trusted the checkpoint memory there compared its state with whatever an earlier batch had left in it -- and once
        // in a few thousand images found a match (tests/test_gpu_sync.py::test_stale_checkpoints_are_not_trusted).
        uint32_t s_x = s, e_x = ne; bool run_x = changed, own_x = owned, ck_x = owned && rec.pad != 0u;
        if (gather) {
            if (changed) { re_src[pos] = s | (owned ? 0x80000000u : 0u) | (ck_x ? 0x40000000u : 0u); re_ent[pos] = ne; }
            __syncthreads();
            run_x = threadIdx.x < n;
            s_x = run_x ? re_src[threadIdx.x] : 0u;
            e_x = run_x ? re_ent[threadIdx.x] : 0u;
            own_x = (s_x >> 31) != 0; ck_x = ((s_x >> 30) & 1u) != 0; s_x &= 0x3FFFFFFFu;
        }
        JdSub r2 = rec;
        JdSubCk* const ckp = a.subck + im.subseq_base + s_x;
        if (__builtin_amdgcn_ballot_w64(run_x)) {                // wave-uniform: skip waves with nothing to redo
            // the first round of the first launch is every sub-sequence's first full decode: it leaves the checkpoints
            // that every later decode (later rounds, gathered or not, and the fix-up launches) stops at
            if (!fixup && it == 0) spec_decode<1>(ik, Q, qmax, a.lut + im.lut_off, s_x, e_x, sbits, run_x, r2, ckp, own_x);
            else spec_decode<2>(ik, Q, qmax, a.lut + im.lut_off, s_x, e_x, sbits, run_x, r2, ckp, ck_x);
        }
        const bool recorded = !fixup && it == 0 && changed && owned;   // (this lane's sub-sequence got its checkpoints just now)
        if (gather) {
            if (run_x) {
                re_out[threadIdx.x][0] = r2.exit; re_out[threadIdx.x][1] = r2.nstart; re_out[threadIdx.x][2] = r2.first;
                re_out[threadIdx.x][3] = (uint32_t)r2.dc[0]; re_out[threadIdx.x][4] = (uint32_t)r2.dc[1]; re_out[threadIdx.x][5] = (uint32_t)r2.dc[2];
            }
            __syncthreads();
            if (changed) {
                rec.entry = ne; rec.exit = re_out[pos][0]; rec.nstart = re_out[pos][1]; rec.first = re_out[pos][2];
                rec.dc[0] = (int)re_out[pos][3]; rec.dc[1] = (int)re_out[pos][4]; rec.dc[2] = (int)re_out[pos][5];
            }
        } else if (changed) rec = r2;
        if (recorded) rec.pad = 1u;
    }
    if (owned) gsub[s] = rec;
    // workgroup totals for the spine scan
    // ... and every owned lane's exclusive prefix inside the workgroup, for the write pass
    uint32_t t0, t1, t2, t3;
    JdSubPfx pf;
    pf.blk_off = block_excl_scan(owned ? rec.nstart : 0u, tmp, t0);
    pf.dcp[0] = (int)block_excl_scan(owned ? (uint32_t)rec.dc[0] : 0u, tmp, t1);
    pf.dcp[1] = (int)block_excl_scan(owned ? (uint32_t)rec.dc[1] : 0u, tmp, t2);
    pf.dcp[2] = (int)block_excl_scan(owned ? (uint32_t)rec.dc[2] : 0u, tmp, t3);
    if (owned) a.subpfx[im.subseq_base + s] = pf;
    if (threadIdx.x == 0) a.sync_tot[im.sync_wg_base + lw] = make_uint4(t0, t1, t2, t3);
    const uint32_t rem = nsub - lw * JD_SYNC_OWN;
    const uint32_t last = JD_SYNC_WARM + (rem < JD_SYNC_OWN ? rem : JD_SYNC_OWN) - 1u;
    if (fixup && threadIdx.x == last && rec.exit != exit_before) atomicOr(&a.changed[flag_slot], 1u);
}

// one workgroup per image without DRI: exclusive prefix of the sync workgroup totals
__global__ __launch_bounds__(256) void k1_sync_spine(JdKernelArgs a) {
    __shared__ uint32_t tmp[16];
    const int img = blockIdx.x;
    const JdImg& im = a.imgs[img];
    if (im.restart_interval != 0 || (im.flags & JD_IMG_FUSED)) return;
    const uint32_t sbits = a.stream_len[img] * 8u;
    const uint32_t nsub = (sbits + JD_SUBSEQ_BITS - 1u) / JD_SUBSEQ_BITS;
    uint32_t nwg = (nsub + JD_SYNC_OWN - 1u) / JD_SYNC_OWN;
    if (nwg > im.n_sync_wg) nwg = im.n_sync_wg;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (uint32_t c0 = 0; c0 < nwg; c0 += 256) {
        const uint32_t c = c0 + threadIdx.x;
        uint4 v = c < nwg ? a.sync_tot[im.sync_wg_base + c] : make_uint4(0, 0, 0, 0);
        uint32_t t0, t1, t2, t3;
        const uint32_t e0 = block_excl_scan(v.x, tmp, t0), e1 = block_excl_scan(v.y, tmp, t1);
        const uint32_t e2 = block_excl_scan(v.z, tmp, t2), e3 = block_excl_scan(v.w, tmp, t3);
        if (c < nwg) a.sync_tot[im.sync_wg_base + c] = make_uint4(r0 + e0, r1 + e1, r2 + e2, r3 + e3);
        r0 += t0; r1 += t1; r2 += t2; r3 += t3;
    }
    // fewer block starts than the frame has blocks = data ended early; surplus (bits after the last MCU) is
    // ignored, as the reference stops reading once every MCU is decoded (main.go:769-770)
    if (threadIdx.x == 0 && r0 < im.n_mcus * im.bpm) atomicOr(&a.status[img], JD_ST_TRUNC);
}

// ---- no-DRI path, pass 2: segment = the blocks that start inside one sub-sequence ------------------------
// Lane L of the image handles owned lane L % JD_SYNC_WG of sync workgroup L / JD_SYNC_WG.
template <bool C8>
__global__ __launch_bounds__(64 * JD_W_WAVES, 4) void k1_write_subseq(JdKernelArgs a, uint32_t lut_bytes) {
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (((im.flags & JD_IMG_COEF8) != 0) != C8) return;         // workgroup-uniform
    const uint32_t lw = blockIdx.x - a.prefix[img];
    const uint32_t slen = a.stream_len[img];
    const uint32_t sbits = slen * 8u;
    const uint32_t nsub = (sbits + JD_SUBSEQ_BITS - 1u) / JD_SUBSEQ_BITS;
    const uint32_t L = lw * (64u * JD_W_WAVES) + threadIdx.x;
    const uint32_t swg = L / JD_SYNC_WG, sl = L % JD_SYNC_WG;
    const bool uniform = (im.flags & JD_IMG_UNIFORM_TABLES) != 0;
    if ((lw * (64u * JD_W_WAVES) / JD_SYNC_WG) * JD_SYNC_OWN >= nsub) return;   // whole workgroup past the stream
    load_lut(reinterpret_cast<uint16_t*>(dyn_lds), a.lut + im.w_off, im.w_len * 2u);
    const uint32_t s = swg * JD_SYNC_OWN + sl;
    const bool active = sl < JD_SYNC_OWN && s < nsub && swg < im.n_sync_wg;
    Segment sg; sg.start_bit = 0; sg.end_bit = sbits; sg.nblk = 0; sg.blk0 = 0; sg.p0 = sg.p1 = sg.p2 = 0;
    uint32_t err = 0;
    if (active) {
        const JdSub rec = a.sub[im.subseq_base + s];
        const JdSubPfx pf = a.subpfx[im.subseq_base + s];
        const uint4 base = a.sync_tot[im.sync_wg_base + swg];
        const uint32_t total_blocks = im.n_mcus * im.bpm;
        const uint32_t blk0 = base.x + pf.blk_off;
        if (!uniform) { sg.p0 = (int)base.y + pf.dcp[0]; sg.p1 = (int)base.z + pf.dcp[1]; sg.p2 = (int)base.w + pf.dcp[2]; }
        sg.start_bit = s * JD_SUBSEQ_BITS + rec.first; sg.blk0 = blk0;
        sg.nblk = blk0 < total_blocks ? (rec.nstart < total_blocks - blk0 ? rec.nstart : total_blocks - blk0) : 0u;
        if (sg.nblk && !uniform) {
            // consistency of the scan with the sub-sequence's own state: the first block that starts here must
            // be the block-in-MCU the entry state implies
            const uint32_t c = (rec.entry >> 8) & 0xFFu, z = (rec.entry >> 16) & 0x7Fu;
            const uint32_t cf = z == 0 ? c : (c + 1u == im.bpm ? 0u : c + 1u);
            if (blk0 % im.bpm != cf) { err = JD_ST_SYNC; sg.nblk = 0; }
        }
        if (!sg.nblk) { sg.start_bit = 0; sg.blk0 = 0; }
    }
    if constexpr (C8) write_segments8<true>(a, im, img, lut_bytes, sg, slen, err);   // (uniform-table images keep int16)
    else write_segments<true>(a, im, img, lut_bytes, sg, slen, err, uniform);
}

// ================================================================================================================
// Single-decode path of streams without restart markers (JD_IMG_FUSED; the scheme is described in jd_internal.h).
//   k1f_cold    blind decode of the last JD_COLD_BITS of every sub-sequence -> guess[s]: where the first block past its end
//               starts and which block of the MCU it is (11 guesses in 12 are right)
//   k1f_write   round < 0: every sub-sequence decodes and WRITES from guess[s - 1];  round >= 0: the sub-sequences on
//               the round's list decode again from their predecessor's recorded exit state
//   k1f_check   lists the sub-sequences whose entry state is not their predecessor's exit state
//   k1f_scan    exclusive prefix of the block counts and DC sums over an image's sub-sequences
//   k1f_index   per block: its staging slot and the DC base of its component; raises the errors of blocks that exist
// Lane L of image-relative workgroup w (64 * JD_W_WAVES lanes) is sub-sequence w * 64 * JD_W_WAVES + L.
// ================================================================================================================
#define JD_F_LANES (64u * JD_W_WAVES)

__global__ __launch_bounds__(JD_SYNC_WG, 6) void k1f_cold(JdKernelArgs a) {
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_FUSED)) return;
    const uint32_t lw = blockIdx.x - a.prefix[img];
    const uint32_t slen = a.stream_len[img];
    const uint32_t sbits = slen * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    if (lw * JD_F_LANES + 1u >= nsub) return;                  // (the last sub-sequence's exit is nobody's entry)
    const ImgK ik = load_imgk(im);
    if (threadIdx.x < 16u) {
        const uint32_t cc = threadIdx.x < ik.bpm ? threadIdx.x : 0u;
        const uint32_t comp = cc < ik.nl ? 0u : cc - ik.nl + 1u;
        const uint32_t d = JD_SY_LUT + sel3(comp, im.sdc[0], im.sdc[1], im.sdc[2]) * 2u, ac = JD_SY_LUT + sel3(comp, im.sac[0], im.sac[1], im.sac[2]) * 2u;
        *reinterpret_cast<uint32_t*>(dyn_lds + JD_SY_CTAB + threadIdx.x * 4u) = (d >> 2) | ((ac >> 2) << 15) | (comp << 30);
    }
    load_lut(reinterpret_cast<uint16_t*>(dyn_lds + JD_SY_LUT), a.lut + im.lut_off + im.lut_len, im.s_len);   // sync twin set
    const uint4* Q = reinterpret_cast<const uint4*>(a.stream + im.stream_off);
    const uint32_t qmax = (JD_STREAM_FRONT + slen + JD_STREAM_TAIL) / 16u - 1u;
#pragma unroll 1
    for (uint32_t h = 0; h < JD_F_LANES / JD_SYNC_WG; h++) {
        const uint32_t s = lw * JD_F_LANES + h * JD_SYNC_WG + threadIdx.x;
        if (__builtin_amdgcn_readfirstlane(lw * JD_F_LANES + h * JD_SYNC_WG + (threadIdx.x & ~63u)) + 1u >= nsub) continue;   // wave-uniform
        const bool run = s + 1u < nsub;
        JdSub rec; rec.entry = rec.exit = 0; rec.nstart = 0; rec.first = 0; rec.dc[0] = rec.dc[1] = rec.dc[2] = 0; rec.pad = 0;
        spec_decode<3>(ik, Q, qmax, a.lut + im.lut_off, run ? s : 0u, pack_state(0, 0, 0), sbits, run, rec, nullptr, false, a.fsub_bits);
        if (run) a.guess[im.subseq_base + s] = rec.exit;
    }
}

// share_w = 0: a workgroup serves one image (image-relative workgroup lw of the launch's prefix).
// share_w = 1, 2, 4 (re-decode rounds of a batch whose single-decode images all use ONE table set, lists of a wave or two):
// a workgroup of eight waves loads the tables once and every share_w waves of it walk the list of an image of their own
// -- workgroup g serves images g * 8 / share_w ... .  A list of a handful of entries needs one wave, and one-wave
// workgroups each with the tables to themselves fit eight to a CU; this way sixteen such walks run on a CU at once.
__global__ __launch_bounds__(64 * JD_W_WAVES, 4) void k1f_write(JdKernelArgs a, uint32_t lut_bytes, int round, uint32_t share_w,
                                                                 uint32_t share_w_off, uint32_t share_w_len) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int img;
    uint32_t first_tile, tile_step, tile_lanes, in_tile;
    if (share_w) {
        const uint32_t per_wg = (blockDim.x >> 6) / share_w;
        img = (int)(blockIdx.x * per_wg + wave / share_w);
        first_tile = wave % share_w; tile_step = share_w; tile_lanes = 64u; in_tile = threadIdx.x & 63u;
    } else {
        img = find_image(a.prefix, a.n_images, blockIdx.x);
        first_tile = blockIdx.x - a.prefix[img]; tile_step = a.prefix[img + 1] - a.prefix[img];
        tile_lanes = blockDim.x;                                // 64 * JD_W_WAVES for the first decode, 64 .. JD_F_REDO_LANES for the lists
        in_tile = threadIdx.x;
    }
    const bool mine = img < a.n_images && (a.imgs[img < a.n_images ? img : 0].flags & JD_IMG_FUSED) != 0;
    const JdImg& im = a.imgs[mine ? img : 0];
    const uint32_t slen = mine ? a.stream_len[img] : 0u;
    const uint32_t sbits = slen * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    const uint32_t n_todo = !mine ? 0u : round < 0 ? nsub : a.fcount[(uint32_t)round * (uint32_t)a.n_images + (uint32_t)img];
    if (share_w) {
        if (!__syncthreads_or(first_tile * tile_lanes < n_todo)) return;   // none of the workgroup's images has a list
        load_lut(reinterpret_cast<uint16_t*>(dyn_lds), a.lut + share_w_off, share_w_len * 2u);
    } else {
        if (!mine || first_tile * tile_lanes >= n_todo) return;          // whole workgroup idle (uniform)
        load_lut(reinterpret_cast<uint16_t*>(dyn_lds), a.lut + im.w_off, im.w_len * 2u);
    }
    // (a re-decode launch has fewer workgroups than the list may have tiles: each walks its share.  Its workgroups are small:
    // a list is short, and a big workgroup with one busy wave would still hold its whole share of the CU's LDS)
#pragma unroll 1
    for (uint32_t tile = first_tile; tile * tile_lanes < n_todo; tile += tile_step) {
        const uint32_t L = tile * tile_lanes + in_tile;
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(tile * tile_lanes + (in_tile & ~63u))) >= n_todo) continue;   // wave-uniform
        FusedSeg fs; fs.valid = L < n_todo; fs.c0 = 0; fs.entry = JD_FSTATE(0u, 0u); fs.old_chunks = 0; fs.gsub = im.subseq_base; fs.slot0 = 0;
        fs.cutoff = round < 0; fs.guess_exit = 0;
        Segment sg; sg.start_bit = 0; sg.end_bit = 0; sg.nblk = 0; sg.blk0 = 0; sg.p0 = sg.p1 = sg.p2 = 0;
        if (fs.valid) {
            uint32_t s = L;
            if (round >= 0) {
                s = a.wl[im.subseq_base + L];
                fs.old_chunks = a.sub[im.subseq_base + s].pad;
                if (s) fs.entry = a.sub[im.subseq_base + s - 1u].exit;   // (sub-sequence 0 starts the stream: listed only if it was cut)
            } else {
                if (s) fs.entry = a.guess[im.subseq_base + s - 1u];
                if (s + 1u < nsub) fs.guess_exit = a.guess[im.subseq_base + s];
            }
            fs.gsub = im.subseq_base + s;
            fs.slot0 = s * im.f_ch0;
            const uint32_t c0 = (fs.entry >> 16) & 0xFFu;
            fs.c0 = c0 < im.bpm ? c0 : 0u;
            sg.end_bit = (s + 1u) * a.fsub_bits < sbits ? (s + 1u) * a.fsub_bits : sbits;
            sg.start_bit = s * a.fsub_bits + (fs.entry & 0xFFFFu);
            if (sg.start_bit > sbits) sg.start_bit = sbits;          // (a block cannot start past the data; the walk then has nothing to do)
        }
        write_segments8<true, true>(a, im, img, lut_bytes, sg, slen, 0u, &fs);
    }
}

// round = the list to fill (0 .. JD_F_ROUNDS; after the last one a.changed[32] tells the host that blind rounds were not
// enough).  One atomic per workgroup: the lists of an image share one counter.
__global__ __launch_bounds__(64 * JD_W_WAVES) void k1f_check(JdKernelArgs a, int round) {
    __shared__ uint32_t wcount[JD_W_WAVES];
    __shared__ uint32_t wg_base;
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_FUSED)) return;
    const uint32_t lw = blockIdx.x - a.prefix[img];
    const uint32_t sbits = a.stream_len[img] * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    const uint32_t s = lw * JD_F_LANES + threadIdx.x;
    bool redo = false;
    if (s < nsub) redo = a.sub[im.subseq_base + s].entry != (s ? a.sub[im.subseq_base + s - 1u].exit : JD_FSTATE(0u, 0u));
    const uint64_t m = __builtin_amdgcn_ballot_w64(redo);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 0) wcount[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < JD_W_WAVES; k++) { const uint32_t n = wcount[k]; before += k < wv ? n : 0u; total += n; }
    if (total == 0) return;                                    // (uniform)
    if (threadIdx.x == 0) {
        wg_base = atomicAdd(a.fcount + (uint32_t)round * (uint32_t)a.n_images + (uint32_t)img, total);
        if (round >= JD_F_ROUNDS) atomicOr(&a.changed[32], 1u);
    }
    __syncthreads();
    if (redo) a.wl[im.subseq_base + wg_base + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
}

// exclusive scan of four values at once over a thread block (<= 1024 threads; `tmp`: 16 uint4 of LDS)
__device__ __forceinline__ uint4 block_excl_scan4(uint4 v, uint4* tmp, uint4& total) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint4 inc = make_uint4(wave_incl_scan(v.x), wave_incl_scan(v.y), wave_incl_scan(v.z), wave_incl_scan(v.w));
    __syncthreads();
    if (lane == 63) tmp[wv] = inc;
    __syncthreads();
    const int nw = (int)((blockDim.x + 63u) >> 6);
    uint4 t = lane < nw ? tmp[lane] : make_uint4(0, 0, 0, 0);
    t = make_uint4(row_incl_scan(t.x), row_incl_scan(t.y), row_incl_scan(t.z), row_incl_scan(t.w));
    auto rl = [&](uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); };
    total = make_uint4(rl(t.x, nw - 1), rl(t.y, nw - 1), rl(t.z, nw - 1), rl(t.w, nw - 1));
    uint4 base = make_uint4(0, 0, 0, 0);
    if (wv) base = make_uint4(rl(t.x, wv - 1), rl(t.y, wv - 1), rl(t.z, wv - 1), rl(t.w, wv - 1));
    return make_uint4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z, base.w + inc.w - v.w);
}

// Prefix scan over an image's sub-sequences in two launches on the grid of the first decode (workgroups of JD_F_LANES
// lanes; every 4th one works, on a TILE of 4 * JD_F_LANES sub-sequences, 4 per thread: 128 contiguous bytes of records): PASS 0 leaves each tile's
// totals in sync_tot, PASS 1 adds up the tiles before its own and scans inside.  (One workgroup per image walking all of
// them was a 100-130 us serial step of every batch with a large image in it.)
#define JD_F_SCAN_PER 4u
template <int PASS>
__global__ __launch_bounds__(64 * JD_W_WAVES) void k1f_scan(JdKernelArgs a) {
    __shared__ uint4 tmp[16];
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_FUSED)) return;
    const uint32_t lw = blockIdx.x - a.prefix[img];
    if (lw % JD_F_SCAN_PER) return;
    const uint32_t tile = lw / JD_F_SCAN_PER;
    const uint32_t sbits = a.stream_len[img] * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    const uint32_t tile_subs = JD_F_LANES * JD_F_SCAN_PER;
    if (tile * tile_subs >= nsub && tile) return;               // (tile 0 always runs: it reports an empty stream)
    uint4* const tile_tot = a.sync_tot + im.sync_wg_base;
    const uint32_t first = tile * tile_subs + threadIdx.x * JD_F_SCAN_PER;
    uint4 v[JD_F_SCAN_PER];
    uint4 mine = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t e = 0; e < JD_F_SCAN_PER; e++) {
        v[e] = make_uint4(0, 0, 0, 0);
        if (first + e < nsub) { const JdSub r = a.sub[im.subseq_base + first + e]; v[e] = make_uint4(r.nstart, (uint32_t)r.dc[0], (uint32_t)r.dc[1], (uint32_t)r.dc[2]); }
        mine.x += v[e].x; mine.y += v[e].y; mine.z += v[e].z; mine.w += v[e].w;
    }
    uint4 tot;
    uint4 at = block_excl_scan4(mine, tmp, tot);
    if constexpr (PASS == 0) {
        if (threadIdx.x == 0) tile_tot[tile] = tot;
        return;
    }
    // the tiles before this one (an image has at most a few hundred)
    uint4 before = make_uint4(0, 0, 0, 0);
    for (uint32_t t = threadIdx.x; t < tile; t += JD_F_LANES) { const uint4 x = tile_tot[t]; before.x += x.x; before.y += x.y; before.z += x.z; before.w += x.w; }
    uint4 base;
    (void)block_excl_scan4(before, tmp, base);
    at.x += base.x; at.y += base.y; at.z += base.z; at.w += base.w;
#pragma unroll
    for (uint32_t e = 0; e < JD_F_SCAN_PER; e++) {
        if (first + e < nsub) {
            JdSubPfx pf; pf.blk_off = at.x; pf.dcp[0] = (int)at.y; pf.dcp[1] = (int)at.z; pf.dcp[2] = (int)at.w;
            a.subpfx[im.subseq_base + first + e] = pf;
        }
        at.x += v[e].x; at.y += v[e].y; at.z += v[e].z; at.w += v[e].w;
    }
    if ((tile + 1u) * tile_subs >= nsub) {                       // the image's last tile knows how many blocks there are
        // fewer blocks than the frame has = data ended early; surplus (bits after the last MCU) is ignored, as the
        // reference stops reading once every MCU is decoded (main.go:769-770)
        const uint32_t r0 = base.x + tot.x, total = im.n_mcus * im.bpm;
        if (r0 < total) {
            if (threadIdx.x == 0) atomicOr(&a.status[img], (uint32_t)JD_ST_TRUNC);
            // the blocks that never came: K2 reads SOME slot for them (the picture is reported as damaged anyway)
            for (uint32_t j = r0 + threadIdx.x; j < total; j += JD_F_LANES) a.loc[im.loc_off + j] = make_uint2(0u, 0u);
        }
    }
}

// The same scan with one workgroup per image (1024 sub-sequences per trip): what a batch of many small images wants -- no
// search for the image, no idle workgroups (10,000 images of 1800 sub-sequences: 0.09 ms against 0.6 ms for the tiled form).
__global__ __launch_bounds__(1024) void k1f_scan_img(JdKernelArgs a) {
    __shared__ uint4 tmp[16];
    const int img = blockIdx.x;
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_FUSED)) return;
    const uint32_t sbits = a.stream_len[img] * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    uint4 run = make_uint4(0, 0, 0, 0);
    for (uint32_t s0 = 0; s0 < nsub; s0 += 1024u) {
        const uint32_t s = s0 + threadIdx.x;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (s < nsub) { const JdSub r = a.sub[im.subseq_base + s]; v = make_uint4(r.nstart, (uint32_t)r.dc[0], (uint32_t)r.dc[1], (uint32_t)r.dc[2]); }
        uint4 tot;
        const uint4 e = block_excl_scan4(v, tmp, tot);
        if (s < nsub) {
            JdSubPfx pf; pf.blk_off = run.x + e.x; pf.dcp[0] = (int)(run.y + e.y); pf.dcp[1] = (int)(run.z + e.z); pf.dcp[2] = (int)(run.w + e.w);
            a.subpfx[im.subseq_base + s] = pf;
        }
        run.x += tot.x; run.y += tot.y; run.z += tot.z; run.w += tot.w;
    }
    const uint32_t r0 = run.x, total = im.n_mcus * im.bpm;        // (as the tiled form's last tile)
    if (r0 < total) {
        if (threadIdx.x == 0) atomicOr(&a.status[img], (uint32_t)JD_ST_TRUNC);
        for (uint32_t j = r0 + threadIdx.x; j < total; j += 1024u) a.loc[im.loc_off + j] = make_uint2(0u, 0u);
    }
}

// One wave per 64 sub-sequences, which it takes one at a time: lane j writes the entry of the sub-sequence's block j (and
// j + 64, ...), so a sub-sequence's entries leave as one contiguous store.
__global__ __launch_bounds__(64 * JD_W_WAVES) void k1f_index(JdKernelArgs a) {
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_FUSED)) return;
    const uint32_t lw = blockIdx.x - a.prefix[img];
    const uint32_t sbits = a.stream_len[img] * 8u;
    const uint32_t nsub = (sbits + a.fsub_bits - 1u) / a.fsub_bits;
    const uint32_t s = lw * JD_F_LANES + threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t total = im.n_mcus * im.bpm, bpm = im.bpm, nl = im.hs * im.vs, ch0 = im.f_ch0, ovb = im.f_ov_base;
    // x mod bpm as x - bpm * mulhi(x, M), M = ceil(2^32 / bpm): exact while x * (M * bpm - 2^32) < 2^32, i.e. for every
    // x < 2^32 / bpm (bpm = 1: M does not fit, and the answer is 0 anyway)
    const uint32_t mod_m = bpm > 1u ? (uint32_t)((0x100000000ull + bpm - 1u) / bpm) : 0u;
    uint32_t nb = 0, c = 0, blk0 = 0, par = 0;
    int d0 = 0, d1 = 0, d2 = 0;
    if (s < nsub) {
        const JdSub r = a.sub[im.subseq_base + s];
        const JdSubPfx pf = a.subpfx[im.subseq_base + s];
        if (pf.blk_off < total && r.nstart != 0u) {
            nb = r.nstart < total - pf.blk_off ? r.nstart : total - pf.blk_off;
            blk0 = pf.blk_off; d0 = pf.dcp[0]; d1 = pf.dcp[1]; d2 = pf.dcp[2];
            uint32_t err = 0;
            c = (r.entry >> 16) & 0xFFu;
            par = (c < bpm ? c : 0u) & 1u;                                 // (what the writer placed its blocks by)
            if (c >= bpm || blk0 % bpm != c) { err |= JD_ST_SYNC; c = blk0 % bpm; }
            if (nb == r.nstart) err |= r.first & 0xFFu;                   // errors in blocks the frame has
            else if ((r.first >> 16) < nb) err |= (r.first >> 8) & 0xFFu;
            // a block the frame needs ran past the end of the data
            const uint32_t end_s = (s + 1u) * a.fsub_bits < sbits ? (s + 1u) * a.fsub_bits : sbits;
            if (nb == r.nstart && end_s + (r.exit & 0xFFFFu) > sbits) err |= JD_ST_TRUNC;
            if (err) atomicOr(&a.status[img], err);
        }
    }
    uint64_t todo = __builtin_amdgcn_ballot_w64(nb != 0u);
    while (todo) {                                                         // (wave-uniform)
        const int i = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t nb_i = (uint32_t)__builtin_amdgcn_readlane((int)nb, i), c_i = (uint32_t)__builtin_amdgcn_readlane((int)c, i);
        const uint32_t blk_i = (uint32_t)__builtin_amdgcn_readlane((int)blk0, i), par_i = (uint32_t)__builtin_amdgcn_readlane((int)par, i);
        const int e0 = __builtin_amdgcn_readlane(d0, i), e1 = __builtin_amdgcn_readlane(d1, i), e2 = __builtin_amdgcn_readlane(d2, i);
        const uint32_t s_i = __builtin_amdgcn_readfirstlane(s - lane) + (uint32_t)i;
        const uint32_t* ctab = a.chunk + (uint64_t)(im.subseq_base + s_i) * a.fkmax;
        uint2* out = a.loc + im.loc_off + blk_i;
        for (uint32_t j = lane; j < nb_i; j += 64u) {
            uint32_t slot;
            if (j < ch0 - 2u) slot = s_i * ch0 + par_i + j;
            else { const uint32_t o = j - (ch0 - 2u), k = o / (JD_F_CHUNK - 2u); slot = ovb + ctab[k] * JD_F_CHUNK + par_i + (o - k * (JD_F_CHUNK - 2u)); }
            const uint32_t xj = c_i + j, cb = bpm > 1u ? xj - bpm * __umulhi(xj, mod_m) : 0u;
            const uint32_t comp = cb < nl ? 0u : cb - nl + 1u;
            out[j] = make_uint2(slot, (uint32_t)(comp == 0u ? e0 : comp == 1u ? e1 : e2));
        }
    }
}

// ---- DC-prediction prefix scan as its own pass --------------------------------------------------------------
// For images flagged JD_IMG_UNIFORM_TABLES (no DRI) the writer stored DC differences; one workgroup per (image,
// component) walks that component's blocks in scan order and replaces each difference by the running sum
// (prevDC, main.go:199-201, 750).
__global__ __launch_bounds__(1024) void k1_dc_scan(JdKernelArgs a) {
    __shared__ uint32_t tmp[16];
    const int img = blockIdx.x / 3;
    const uint32_t comp = blockIdx.x % 3;
    const JdImg& im = a.imgs[img];
    if (!(im.flags & JD_IMG_UNIFORM_TABLES) || im.restart_interval != 0 || comp >= im.ncomp) return;
    const uint32_t nl = im.hs * im.vs, per = comp == 0 ? nl : 1u, n = im.n_mcus * per;
    int16_t* const cf = a.coef + im.coef_off;
    int carry = 0;
    uint32_t bad = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 1024u) {
        const uint32_t j = j0 + threadIdx.x;
        const uint64_t blk = (uint64_t)(j / per) * im.bpm + (comp == 0 ? j % per : nl + comp - 1u);
        const int d = j < n ? (int)cf[blk * 64u] : 0;
        uint32_t tot;
        const int v = carry + (int)block_excl_scan((uint32_t)d, tmp, tot) + d;
        if (j < n) { cf[blk * 64u] = (int16_t)v; if (v != (int)(short)v) bad = 1; }
        carry += (int)tot;
    }
    if (bad) atomicOr(&a.status[img], JD_ST_COEF);
}

// ------------------------------------------------------------------------------------------------
// Progressive (SOF2) scans, SURVEY row N4: decodeBandCoeffecients' SOF2 branch (main.go:241-428) under decodeHuffmanData's
// block walk (main.go:749-816).  A progressive scan has no restart structure the reference acts on and every scan refines
// what earlier scans of the same band left, so the unit of parallel work is the SCAN: one wave per scan, scans that do not
// depend on each other (different components or disjoint bands) in one launch, dependent ones in later launches (the
// host orders them in levels).  Coefficients accumulate in the same int16 buffer, same layout, as the baseline writer's;
// K2 then runs unchanged.
//
// Inside a scan the symbols are a serial chain, and a serial chain is SCALAR work: the walk runs on the wave's scalar unit
// (bit buffer, positions, EOB-run counter, 64-bit coefficient masks all live in scalar registers; s_lshl_b64, s_ff1_i32_b64,
// s_bcnt1_i32_b64 are single instructions there), and the data it looks things up in sits in VECTOR registers, read with
// v_readlane (a few cycles, no memory round trip):
//   * the scan's symbol tables (jd_pack_lut_lane: 7 registers per table -- a 256-entry primary table indexed by the next 8
//     bits, MAXCODE / VALPTR per code length for the rare longer codes, the symbols);
//   * a 512-byte window of the stream in two registers (lane l = dword l; the second is the prefetch of the next 256
//     bytes, issued 2048 bits before it is needed).
// (Round 2 walked the chain with lane 0 out of LDS: about 900 cycles per symbol, most of it LDS round trips of a lone wave.)
// What does not depend on the chain runs wave-wide:
//   * the 64 lanes hold the current block, one coefficient each: a first-visit AC coefficient is placed with v_writelane and
//     the lanes that received one store it; DC values of 64 blocks are collected the same way and stored together;
//   * DC refinement (main.go:312-320) is one bit per block at a known position: lane l handles block 64 i + l;
//   * AC refinement (main.go:321-427) reads a correction bit for every coefficient that is already non-zero: a wave ballot
//     gives the scalar walk the block's non-zero mask; for every symbol it works out WHICH coefficients take correction
//     bits (mask arithmetic: the (run+1)-th zero at or after the current index) and reads that many bits at once; each lane
//     finds its own bit by its rank in the mask (v_mbcnt) and corrects its coefficient.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lane `l` of v <- val (both wave-uniform): a compare against the lane number and a select -- what v_writelane_b32 does,
// spelt in operations the compiler can schedule (this hipcc has no builtin for the instruction)
__device__ __forceinline__ uint32_t wrlane(uint32_t val, uint32_t l, uint32_t v) { return threadIdx.x == l ? val : v; }

struct PTab { uint32_t p0, p1, p2, p3, mx, off, vals; };       // one symbol table: what THIS lane holds of its seven rows
__device__ __forceinline__ PTab ptab_load(const uint32_t* __restrict__ g, uint32_t lane, bool have) {
    PTab t;
    t.p0 = t.p1 = t.p2 = t.p3 = t.off = t.vals = 0; t.mx = 0xFFFFFFFFu;
    if (have) { t.p0 = g[lane]; t.p1 = g[64 + lane]; t.p2 = g[128 + lane]; t.p3 = g[192 + lane]; t.mx = g[256 + lane]; t.off = g[320 + lane]; t.vals = g[384 + lane]; }
    return t;
}

struct PBits {                 // MSB-first reader (BitReader, main.go:1252-1295); every member is wave-uniform
    // the scan's unstuffed bytes, read through the CONSTANT address space: s_load_dword, straight into scalar registers
    // (a dword ahead: the load issued at one refill is waited for at the next, some five symbols later)
    const __attribute__((address_space(4))) uint32_t* stream;
    uint32_t ndw;              // dwords of data; dword ndw and the ones after it are zero padding (the host leaves 16 bytes)
    uint32_t next;             // dwords taken into `acc` so far
    uint32_t ahead;            // dword `next`, already loaded (memory byte order)
    uint64_t acc;              // unread bits, left aligned
    int n;                     // how many
    __device__ __forceinline__ uint32_t load(uint32_t i) const { return stream[i < ndw ? i : ndw]; }   // past the end: zeros (readBit -1 is reported at the end)
    __device__ __forceinline__ void init(const uint32_t* s, uint32_t nd) {
        stream = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)s; ndw = nd; next = 0; acc = 0; n = 0;
        ahead = load(0);
    }
    __device__ __forceinline__ void fill() {                  // afterwards n >= 33
        if (n <= 32) {
            const uint32_t d = be32(ahead);
            next++;
            ahead = load(next);
            acc |= (uint64_t)d << (32 - n); n += 32;
        }
    }
    __device__ __forceinline__ uint32_t get(uint32_t k) {     // readBits, k <= 32 (0 bits = 0)
        fill();
        const uint32_t v = k ? (uint32_t)(acc >> (64u - k)) : 0u;
        acc <<= k; n -= (int)k;
        return v;
    }
    // ... without the refill test: up to 16 bits right behind a symbol (pscan_symbol refilled to >= 33 and took <= 16)
    __device__ __forceinline__ uint32_t take(uint32_t k) {
        const uint32_t v = k ? (uint32_t)(acc >> (64u - k)) : 0u;
        acc <<= k; n -= (int)k;
        return v;
    }
    __device__ __forceinline__ uint32_t used() const { return next * 32u - (uint32_t)n; }   // bits consumed
};
// scanSymbol (main.go:1297-1316) by table lookup; returns 0x100 for a code the table does not have
__device__ __forceinline__ uint32_t pscan_symbol(PBits& br, const PTab& t) {
    br.fill();
    const uint32_t idx = (uint32_t)(br.acc >> 56), l = idx & 63u, h = idx >> 6;
    const uint32_t a = rdlane(t.p0, l), b = rdlane(t.p1, l), c = rdlane(t.p2, l), d = rdlane(t.p3, l);
    uint32_t e = h == 0u ? a : h == 1u ? b : h == 2u ? c : d;
    if (__builtin_expect(e == 0u, 0)) {                       // longer than 8 bits: T.81 F.2.2.3, lengths 9 .. 16
#pragma unroll 1
        for (uint32_t L = 9; L <= 16u; L++) {
            const uint32_t code = (uint32_t)(br.acc >> (64u - L));
            if ((int)code <= (int)rdlane(t.mx, L - 1u)) {
                const uint32_t i = (code + rdlane(t.off, L - 1u)) & 255u;
                e = (L << 8) | ((rdlane(t.vals, i >> 2) >> ((i & 3u) * 8u)) & 0xFFu);
                break;
            }
        }
        if (e == 0u) return 0x100u;
    }
    const uint32_t len = e >> 8;
    br.acc <<= len; br.n -= (int)len;
    return e & 0xFFu;
}
__device__ __forceinline__ int pextend(uint32_t v, uint32_t len) {   // main.go:248-250 / 292-294
    return (len != 0 && v < (1u << (len - 1))) ? (int)v - (int)((1u << len) - 1u) : (int)v;
}

__global__ __launch_bounds__(64) void kp_scan(JdKernelArgs a, const JdPScan* __restrict__ scans, const uint32_t* __restrict__ list,
                                               const uint8_t* __restrict__ pstream, const uint32_t* __restrict__ plut,
                                               const uint32_t* __restrict__ plen, const uint32_t* __restrict__ pstatus) {
    const uint32_t scan_id = list[blockIdx.x];
    const JdPScan sc = scans[scan_id];
    const JdImg& im = a.imgs[sc.img];
    const uint32_t lane = threadIdx.x;
    // K0 unstuffed the scan (to it an image of one restart interval): its verdict on markers other than RSTn counts,
    // its restart bookkeeping does not (the reference drops RSTn in a progressive scan, main.go:999-1000)
    if (pstatus[scan_id] & JD_ST_MARKER) { if (lane == 0) atomicOr(&a.status[sc.img], (uint32_t)JD_ST_MARKER); return; }
    const uint32_t* __restrict__ stream = reinterpret_cast<const uint32_t*>(pstream + sc.stream_off);
    const uint32_t slen = plen[scan_id];                        // unstuffed bytes
    const uint32_t ndw = (slen + 3u) >> 2;
    int16_t* const cf = a.coef + im.coef_off;
    const uint32_t hs = im.hs, vs = im.vs, nl = hs * vs;
    const uint32_t hsh = hs - 1u, vsh = vs - 1u;                // sampling factors are 1 or 2: divide and modulo by shift and mask
    const uint32_t bw = (im.width + 7u) >> 3, bh = (im.height + 7u) >> 3;          // main.go:160-161
    // decodeHuffmanData, main.go:753-770: a scan of the first component alone walks its blocks in raster order, anything
    // else walks the MCU grid
    const bool lum_only = sc.ncomp == 1 && sc.comp[0] == 0;
    const uint32_t xstep = lum_only ? 1u : hs, ystep = lum_only ? 1u : vs;
    const uint32_t ss = sc.ss, se = sc.se, al = sc.al;
    const uint32_t mcux = im.mcux, bpm = im.bpm;
    const uint32_t cpA = sc.comp[0], cpB = sc.comp[1], cpC = sc.comp[2];
    uint32_t err = 0;
    // element offset (in int16) of block (bx, by) of component cp
    auto block_off = [&](uint32_t cp, uint32_t bx, uint32_t by) -> uint32_t {
        const uint32_t mcu = (by >> vsh) * mcux + (bx >> hsh);
        const uint32_t b = cp ? nl + cp - 1u : (by & vsh) * hs + (bx & hsh);
        return (mcu * bpm + b) * 64u;
    };
    // blocks per step of the walk (one (x, y) of the loops at main.go:769-770) and its steps per row
    const uint32_t nA = (lum_only || cpA) ? 1u : nl, nB = sc.ncomp > 1 ? ((lum_only || cpB) ? 1u : nl) : 0u,
                   nC = sc.ncomp > 2 ? ((lum_only || cpC) ? 1u : nl) : 0u;
    const uint32_t per_step = nA + nB + nC;
    const uint32_t steps_x = (bw + xstep - 1u) / xstep, steps_y = (bh + ystep - 1u) / ystep;
    // block number n of the scan's walk -> its element offset (vector code: every lane its own n)
    auto nth_block = [&](uint32_t n) -> uint32_t {
        const uint32_t step = n / per_step;
        uint32_t j = n - step * per_step;
        const uint32_t sy = step / steps_x, sx = step - sy * steps_x;
        uint32_t cp = cpA, cnt = nA;
        if (j >= nA) { j -= nA; cp = cpB; cnt = nB; if (j >= nB) { j -= nB; cp = cpC; cnt = nC; } }
        const uint32_t u = cnt > 1u ? j / hs : 0u, v = cnt > 1u ? j - u * hs : 0u;       // (u outer, v inner: main.go:785-786)
        return block_off(cp, sx * xstep + v, sy * ystep + u);
    };

    // ---- DC refinement: block n of the scan takes bit n; lane l handles blocks 64 i + l ---------------------------
    if (ss == 0 && sc.ah != 0) {
        const uint32_t total = steps_x * steps_y * per_step;
        for (uint32_t n = lane; n < total; n += 64u) {
            const uint32_t d = (n >> 5) < ndw ? be32(stream[n >> 5]) : 0u;
            int16_t* p = cf + nth_block(n);
            *p = (int16_t)(*p | (int)(((d >> (31u - (n & 31u))) & 1u) << al));
        }
        if (lane == 0 && total > slen * 8u) atomicOr(&a.status[sc.img], (uint32_t)JD_ST_TRUNC);   // readBit -1 (main.go:1273)
        return;
    }

    // ---- everything else: a serial chain of symbols, walked on the scalar unit.  An error ends the scan (`goto done`:
    // loops that also test an error flag cost the walk a dozen scalar instructions per symbol). -----------------------
    const PTab tA = ptab_load(plut + sc.lut_off[0], lane, true);
    PBits br; br.init(stream, ndw);
    const int positive = 1 << al, negative = -(1 << al);

    if (ss == 0) {
        // ---- DC first visit, main.go:243-253: the blocks of each component of every step, in walk order.  Values and
        // block offsets of 64 blocks are collected in two vector registers and stored together.
        const PTab tB = ptab_load(plut + sc.lut_off[1], lane, sc.ncomp > 1);
        const PTab tC = ptab_load(plut + sc.lut_off[2], lane, sc.ncomp > 2);
        int predA = 0, predB = 0, predC = 0;                   // prevDC of the scan's components (main.go:750)
        uint32_t vval = 0, voff = 0, slot = 0;
#define JD_P_DC_FIRST(cp, cnt, tab, pred)                                                                   \
        for (uint32_t j = 0; j < (cnt); j++) {                                                              \
            const uint32_t u = (cnt) > 1u ? j >> hsh : 0u, v = (cnt) > 1u ? j & hsh : 0u;                   \
            const uint32_t sym = pscan_symbol(br, tab);                                                     \
            if (sym > 16u) { err |= JD_ST_HUFF; goto dc_done; }                                             \
            const int val = pextend(br.take(sym), sym) + pred;                                               \
            pred = val;                                                                                     \
            const int out = val << al;                                                                      \
            if (out != (int)(short)out) { err |= JD_ST_COEF; goto dc_done; }                                \
            vval = wrlane((uint32_t)out, slot, vval);                                                       \
            voff = wrlane(block_off(cp, x + v, y + u), slot, voff);                                         \
            if (++slot == 64u) { cf[voff] = (int16_t)vval; slot = 0; }                                      \
        }
        for (uint32_t y = 0; y < bh; y += ystep)
            for (uint32_t x = 0; x < bw; x += xstep) {
                JD_P_DC_FIRST(cpA, nA, tA, predA)
                JD_P_DC_FIRST(cpB, nB, tB, predB)
                JD_P_DC_FIRST(cpC, nC, tC, predC)
            }
#undef JD_P_DC_FIRST
    dc_done:
        if (lane < slot) cf[voff] = (int16_t)vval;
    } else if (sc.ah == 0) {
        // ---- AC first visit, main.go:254-311 (one component: checked by the host).  The band is zero before this scan: the
        // lanes start from zero, a coded coefficient is dropped into its lane, and only those lanes store (another scan of
        // the same level may be writing the block's other band at this very moment).
        uint32_t skips = 0;
        for (uint32_t y = 0; y < bh; y += ystep)
            for (uint32_t x = 0; x < bw; x += xstep) {
                if (skips > 0) { skips -= 1; continue; }       // inside an end-of-band run
                uint32_t v = 0;
                uint64_t wrote = 0;
                uint32_t index = ss;
                while (index <= se) {
                    const uint32_t sym = pscan_symbol(br, tA);
                    if (sym > 0xFFu) { err |= JD_ST_HUFF; goto done; }
                    if (sym == 0xF0u) {                        // sixteen zeros (already there: first visit)
                        if (index + 16u > 64u) { err |= JD_ST_COEF; goto done; }
                        index += 16u;
                        continue;
                    }
                    const uint32_t zeros = sym >> 4, len = sym & 15u;
                    if (len == 0) { skips = ((1u << zeros) - 1u) + br.take(zeros); break; }   // EOBn
                    index += zeros;
                    if (index > 63u) { err |= JD_ST_COEF; goto done; }
                    const int out = pextend(br.take(len), len) << al;
                    if (out != (int)(short)out) { err |= JD_ST_COEF; goto done; }
                    v = wrlane((uint32_t)out, index, v);
                    wrote |= 1ull << index;
                    index++;
                }
                if (wrote && ((wrote >> lane) & 1ull)) cf[block_off(cpA, x, y) + lane] = (int16_t)v;
            }
    } else {
        // ---- AC refinement, main.go:321-427 ------------------------------------------------------------------------------
        uint32_t skips = 0;
        int cur = cf[block_off(cpA, 0, 0) + lane];             // this lane's coefficient of the current block
        for (uint32_t y = 0; y < bh; y += ystep)
            for (uint32_t x = 0; x < bw; x += xstep) {
                // next block in walk order: its coefficients start loading now
                uint32_t nx = x + xstep, ny = y;
                if (nx >= bw) { nx = 0; ny = y + ystep; }
                int nxt = 0;
                if (ny < bh) nxt = cf[block_off(cpA, nx, ny) + lane];
                const uint64_t nz = __builtin_amdgcn_ballot_w64(cur != 0);
                int val = cur;
                // a correction bit for each coefficient of m, ascending (the Go loop looks at one coefficient at a time,
                // main.go:369-394): read them all at once; the lane of rank r in m owns the r-th bit read
                auto take_bits = [&](uint64_t m) {
                    const uint32_t k = (uint32_t)__popcll(m);
                    if (k == 0) return;
                    const uint32_t k1 = k > 32u ? 32u : k, k2 = k - k1;
                    uint64_t bits = br.get(k1);
                    if (k2) bits = (bits << k2) | br.get(k2);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (((m >> lane) & 1ull) && ((bits >> (k - 1u - rank)) & 1ull)) val = cur >= 0 ? cur + positive : cur + negative;
                };
                uint32_t index = ss;
                if (skips == 0) {
                    while (index <= se) {
                        const uint32_t sym = pscan_symbol(br, tA);
                        if (sym > 0xFFu) { err |= JD_ST_HUFF; goto done; }
                        const uint32_t zeroes = sym >> 4, len = sym & 15u;
                        bool neg = false;
                        if (len != 0) {
                            if (len != 1) { err |= JD_ST_COEF; goto done; }
                            neg = br.take(1) == 0;
                        }
                        if (len == 0 && sym != 0xF0u) { skips = (1u << zeroes) + br.take(zeroes); break; }
                        const uint64_t from = ~0ull << index;
                        uint64_t z = ~nz & from;               // zero coefficients at or after index
                        for (uint32_t k = 0; k < zeroes && z; k++) z &= z - 1ull;
                        if (!z) { err |= JD_ST_COEF; goto done; }   // the run walks past coefficient 63
                        const uint32_t p = (uint32_t)__builtin_ctzll(z);
                        take_bits(nz & from & ((1ull << p) - 1ull));
                        if (len != 0 && lane == p) val = neg ? negative : positive;   // (ZRL: the sixteenth zero stays zero)
                        index = p + 1u;
                    }
                }
                if (skips > 0) {                               // inside an end-of-band run: the rest of the band is only corrected
                    if (index <= se) take_bits(nz & (~0ull << index) & (se == 63u ? ~0ull : (1ull << (se + 1u)) - 1ull));
                    skips -= 1;
                }
                if (val != cur) cf[block_off(cpA, x, y) + lane] = (int16_t)val;
                if (__builtin_amdgcn_ballot_w64(val != (int)(short)val)) { err |= JD_ST_COEF; goto done; }
                cur = nxt;
            }
    }
done:
    if (br.used() > slen * 8u) err |= JD_ST_TRUNC;     // readBit -1 (main.go:1273)
    if (err && lane == 0) atomicOr(&a.status[sc.img], err);
}

// ------------------------------------------------------------------------------------------------
// K2: dequantise + de-zigzag + IDCT + chroma spread + colour conversion
// ------------------------------------------------------------------------------------------------
struct ZZTables { int n2z[64]; };
__host__ __device__ constexpr ZZTables make_zz() {
    ZZTables t{};
    int r = 0, c = 0;
    for (int i = 0; i < 64; i++) {
        t.n2z[r * 8 + c] = i;
        if (((r + c) & 1) == 0) { if (c == 7) r++; else if (r == 0) c++; else { r--; c++; } }
        else { if (r == 7) c++; else if (c == 0) r++; else { r++; c--; } }
    }
    return t;
}
__device__ constexpr ZZTables kZZ = make_zz();     // natural index -> zig-zag index (inverse of zmap.Map1)

// One 1-D pass, main.go:458-528 (columns) / 534-604 (rows): separately rounded fp64 ops, truncating output.
__device__ __forceinline__ void idct_1d(const JdIdctConst& K, int x0, int x1, int x2, int x3, int x4, int x5, int x6,
                                        int x7, int (&o)[8]) {
    const double g0 = (double)x0 * K.S0, g1 = (double)x4 * K.S4, g2 = (double)x2 * K.S2, g3 = (double)x6 * K.S6;
    const double g4 = (double)x5 * K.S5, g5 = (double)x1 * K.S1, g6 = (double)x7 * K.S7, g7 = (double)x3 * K.S3;
    const double f4 = g4 - g7, f5 = g5 + g6, f6 = g5 - g6, f7 = g4 + g7;
    const double e2 = g2 - g3, e3 = g2 + g3, e5 = f5 - f7, e7 = f5 + f7, e8 = f4 + f6;
    const double d2 = e2 * K.M1, d4 = f4 * K.M2, d5 = e5 * K.M3, d6 = f6 * K.M4, d8 = e8 * K.M5;
    const double c0 = g0 + g1, c1 = g0 - g1, c2 = d2 - e3, c4 = d4 + d8, c5 = d5 + e7, c6 = d6 - d8;
    const double c8 = c5 - c6;
    const double b0 = c0 + e3, b1 = c1 + c2, b2 = c1 - c2, b3 = c0 - e3, b4 = c4 - c8, b6 = c6 - e7;
    o[0] = (int)(b0 + e7); o[1] = (int)(b1 + b6); o[2] = (int)(b2 + c8); o[3] = (int)(b3 + b4);
    o[4] = (int)(b3 - b4); o[5] = (int)(b2 - c8); o[6] = (int)(b1 - b6); o[7] = (int)(b0 - e7);
}

// convertColorSpace, main.go:679-703: float32, constants 1.402/0.344/0.714/1.772, +128 last, clamp to
// [0,255], truncate.  The four chroma products are shared by the pixels that read the same chroma sample.
// v_cvt_pk_u8_f32 saturates to [0,255] (measured on gfx950: -5.5 -> 0, 300 -> 255, 1e9 -> 255) and rounds as
// MODE.FP_ROUND says (measured: 2.9999998 -> 3 under round-to-nearest, -> 2 under round-toward-zero).  The 24
// conversions of a pixel row therefore run between two s_setreg that switch the f32 rounding mode to
// toward-zero and back: trunc(v) wherever the reference's clamp leaves v alone, 0 / 255 outside -- with no
// v_floor_f32 in front.  Nothing else may execute in that window (f32 adds and multiplies round by the same
// field): the conversions and both mode writes are volatile asm statements, which keep their order, and the
// values they read come out of a volatile fence placed before the first mode write (k2_rtz_*).
template <int BYTE>
__device__ __forceinline__ void k2_put_byte(uint32_t (&d)[6], float v) {
    if constexpr ((BYTE & 3) == 0) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(d[BYTE >> 2]) : "v"(v));
    else asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(d[BYTE >> 2]) : "v"(v), "n"(BYTE & 3));
}
typedef float jd_f32x2 __attribute__((ext_vector_type(2)));
// byte BYTE of a group of NK pixel pairs (6 * NK bytes of a pixel row, starting at byte BASE): pixel BYTE / 3,
// channel BYTE % 3
template <int BASE, int BYTE, int NK>
__device__ __forceinline__ void k2_put_group_byte(uint32_t (&d)[6], const jd_f32x2 (&R)[NK], const jd_f32x2 (&G)[NK],
                                                  const jd_f32x2 (&B)[NK]) {
    constexpr int px = BYTE / 3, ch = BYTE % 3;
    const jd_f32x2& t = ch == 0 ? R[px >> 1] : ch == 1 ? G[px >> 1] : B[px >> 1];
    k2_put_byte<BASE + BYTE>(d, t[px & 1]);
}
// the group's conversions, ordered so that consecutive ones write different dwords (back-to-back asm statements on
// one register make hipcc put a wait state between them)
template <int BASE, int NK, int... I>
__device__ __forceinline__ void k2_put_group(uint32_t (&d)[6], const jd_f32x2 (&R)[NK], const jd_f32x2 (&G)[NK],
                                             const jd_f32x2 (&B)[NK], std::integer_sequence<int, I...>) {
    constexpr int ND = NK * 6 / 4;                                 // dwords the group fills
    (k2_put_group_byte<BASE, (I % ND) * 4 + I / ND, NK>(d, R, G, B), ...);
}
#ifndef JD_K2_WIN_PAIRS
#define JD_K2_WIN_PAIRS 4                                          // pixel pairs per conversion window (2 or 4)
#endif
__device__ __forceinline__ void k2_rtz_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3"); }
__device__ __forceinline__ void k2_rtz_off() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"); }

// ---- K2, MCU-per-lane form -----------------------------------------------------------------------------
// One lane owns one whole MCU, one wave owns 64 consecutive MCUs; waves never talk to each other: no LDS, no
// barrier.  The lane runs the chroma blocks first and keeps their samples in registers (packed int16), then
// each luma block: load 128 B of coefficients, dequantise + de-zigzag + both IDCT passes in registers, colour-
// convert the 8 x 8 pixels against the chroma quadrant (PY>>1, PX>>1) and store 24-byte rows.  The block loop is
// a real loop (the IDCT and colour code exist once); only the register shuffles that pick the chroma quadrant
// are specialised per block.
// C8: the image's coefficients are in the byte form (write_segments8): 64 bytes per block in the low plane, the high
// plane only for blocks that carry the wide mark.
// ST (with C8): the image went through the single-decode path -- block b of the frame lives in the staging slot loc[b].x and
// its DC is relative to loc[b].y (jd_internal.h).
template <int HS, int VS, int NC, bool C8, bool ST = false>
__device__ __forceinline__ void k2m_body(const JdKernelArgs& a, const JdImg& im, int img, uint32_t tile, const JdIdctConst& K) {
    constexpr uint32_t NL = HS * VS, BPM = NL + NC - 1;
    constexpr int NP = C8 ? 4 : 8;                                 // 16-byte pieces per block
    const uint32_t M = tile * 64u + (threadIdx.x & 63u);
    if (M >= im.n_mcus) return;
    const uint32_t my = M / im.mcux, mx = M - my * im.mcux;
    const uint32_t X0 = mx * 8u * HS, Y0 = my * 8u * VS;
    const uint8_t* const cplane = reinterpret_cast<const uint8_t*>(a.coef + im.coef_off);
    const uint4* __restrict__ cbase = reinterpret_cast<const uint4*>(cplane + (uint64_t)M * BPM * (NP * 16u));
    typedef __attribute__((address_space(1))) const jd_u32x4 g_cu4;
    g_cu4* const hbase = ST ? (g_cu4*)(cplane + JD_F_HIPOOL_OFF(im.slot_cap))                        // ST: the high parts, by their own slots
                            : (g_cu4*)(cplane + ((uint64_t)im.n_mcus * BPM + (uint64_t)M * BPM) * 64u);   // C8: this MCU in the high plane
    const uint32_t rs = im.row_stride, W = im.width, H = im.height;
    // a copy (stores through it must not force reloads of `im`), in the GLOBAL address space: a generic-pointer
    // store could alias LDS, so hipcc would drain the LDS-DMA queue (and every earlier store) before each one
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u32x2 g_u2;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) u32x4 g_u4;
    g_u8* const rgbp = (g_u8*)im.rgb;
    const bool al8 = ((rs & 7u) == 0) && ((reinterpret_cast<uintptr_t>(im.rgb) & 7u) == 0);
    const bool al16 = ((rs & 15u) == 0) && ((reinterpret_cast<uintptr_t>(im.rgb) & 15u) == 0);
    uint32_t chroma[2][32];                                        // [Cb, Cr][row * 4 + col / 2]: int16 pairs
#pragma unroll
    for (int i = 0; i < 32; i++) { chroma[0][i] = 0; chroma[1][i] = 0; }
    // Coefficients come in through LDS-DMA (global_load_lds, no VGPRs): piece i of every lane's block lands at
    // dyn_lds + i * 1024 + lane * 16, and the next block's eight pieces are in flight while this one is computed.
    // dyn_lds + 8192 .. : the left luma block's RGB rows wait here (24 B per lane and row) for the right one.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stage_base = lds_abs_base() + 8192u + lane * 24u;  // LDS byte address of this lane's parked row 0
    auto block_of = [&](uint32_t st) { return st < (uint32_t)(NC - 1) ? NL + st : st - (uint32_t)(NC - 1); };   // chroma first
    auto fetch_from = [&](const uint4* src) {
#pragma unroll
        for (int i = 0; i < NP; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                             (__attribute__((address_space(3))) void*)(dyn_lds + i * 1024),
                                             16, 0, 0);
    };
    auto fetch = [&](uint32_t st) { fetch_from(cbase + block_of(st) * (uint32_t)NP); };
    // ST: where the blocks are -- the MCU's BPM entries (48 contiguous bytes per lane) are read once, up front, and picked
    // by the wave-uniform step.  (Read one step ahead inside the loop they cost 10 % of the kernel: the loop-carried copy of
    // a loaded value puts a vmcnt(0) at the loop's end.)  A slot number is clamped to the staging (a truncated stream
    // leaves entries unwritten).
    typedef uint32_t locv __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) const locv g_cloc;
    g_cloc* const locp = ST ? (g_cloc*)(a.loc + im.loc_off + (uint64_t)M * BPM) : nullptr;
    const uint32_t slot_max = ST ? im.slot_cap - 1u : 0u;
    auto slot_src = [&](uint32_t slot) { return reinterpret_cast<const uint4*>(cplane + (uint64_t)(slot < slot_max ? slot : slot_max) * 64u); };
    locv lcs[ST ? BPM : 1u];
    auto loc_of = [&](uint32_t blk) {                            // blk: wave-uniform
        locv r = lcs[0];
        if constexpr (ST) {
#pragma unroll
            for (uint32_t k = 1; k < BPM; k++) {
                r = blk == k ? lcs[k] : r;
                asm volatile("" : "+v"(r));                      // (a select chain, not a table in scratch memory: see sel3)
            }
        }
        return r;
    };
    locv lc_cur = { 0u, 0u };
    if constexpr (ST) {
#pragma unroll
        for (uint32_t k = 0; k < BPM; k++) lcs[k] = locp[k];
        lc_cur = loc_of(block_of(0));
        fetch_from(slot_src(lc_cur.x));
    } else {
        lcs[0] = lc_cur;
        fetch(0);
    }

    // (grey: one block per lane.  Left as straight-line code hipcc needs more than 256 registers for it -- it fits the
    // six-block loop of 4:2:0 into 218 -- so the trip count is hidden from it)
    uint32_t n_steps = BPM;
    if constexpr (NC == 1) asm volatile("" : "+s"(n_steps));
#pragma unroll 1
    for (uint32_t st = 0; st < n_steps; st++) {
        const uint32_t b = __builtin_amdgcn_readfirstlane(block_of(st));
        const uint32_t comp = b < NL ? 0u : b - NL + 1u;
        // the quantisation table is wave-uniform: read through the constant address space so that it IS a scalar load
        // (s_load_dwordx16).  As a plain global read hipcc made four vector loads + readfirstlane, and their vmcnt(0)
        // wait also drained the LDS-DMA queue at the top of every block.
        uint32_t qw[16];
        {
            typedef __attribute__((address_space(4))) const uint32_t c_u32;
            c_u32* qp = (c_u32*)(uintptr_t)(im.q[comp]);
#pragma unroll
            for (int i = 0; i < 16; i++) qw[i] = qp[i];
        }
        uint32_t cw[NP * 4];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(dyn_lds + i * 1024 + lane * 16u);
            cw[i * 4 + 0] = v.x; cw[i * 4 + 1] = v.y; cw[i * 4 + 2] = v.z; cw[i * 4 + 3] = v.w;
        }
        // the DMA of the next block reuses the buffer: the reads above must have returned first (LDS-DMA writes
        // arrive through the vector-memory path and are not ordered behind ds_read)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // byte form: a block marked wide (high nibble of byte 63 = 8) has its high parts in the high plane; they are
        // fetched before the next block's DMA is queued behind them
        uint32_t hw[C8 ? 16 : 1];
        bool wflag = false, anyw = false;
        if constexpr (C8) {
#pragma unroll
            for (int i = 0; i < 16; i++) hw[i] = 0;
            wflag = (cw[15] >> 28) == 8u;
            anyw = __builtin_amdgcn_ballot_w64(wflag) != 0;    // wave-uniform, rare
            if (anyw && wflag) {
                uint32_t hslot = b;
                if constexpr (ST) {
                    typedef __attribute__((address_space(1))) const uint32_t g_cw;
                    hslot = ((g_cw*)(cplane + JD_F_HIMAP_OFF(im.slot_cap)))[lc_cur.x < slot_max ? lc_cur.x : slot_max];
                    hslot = hslot < im.hi_cap ? hslot : 0u;
                }
                g_cu4* hp = hbase + hslot * 4u;
#pragma unroll
                for (int i = 0; i < 4; i++) { const jd_u32x4 v = hp[i]; hw[i * 4] = v.x; hw[i * 4 + 1] = v.y; hw[i * 4 + 2] = v.z; hw[i * 4 + 3] = v.w; }
                // "use" them here: hipcc then waits for these loads inside this rare branch.  Left pending they made it
                // put a vmcnt(0) in front of the column pass of EVERY block (their registers are reused there), which also
                // waited for the next block's DMA, issued just below -- the prefetch never overlapped anything.
                asm volatile("" : "+v"(hw[0]), "+v"(hw[1]), "+v"(hw[2]), "+v"(hw[3]), "+v"(hw[4]), "+v"(hw[5]), "+v"(hw[6]), "+v"(hw[7]));
                asm volatile("" : "+v"(hw[8]), "+v"(hw[9]), "+v"(hw[10]), "+v"(hw[11]), "+v"(hw[12]), "+v"(hw[13]), "+v"(hw[14]), "+v"(hw[15]));
            }
        }
        const int dc_base = ST ? (int)lc_cur.y : 0;
        if constexpr (ST) {
            if (st + 1u < n_steps) {
                lc_cur = loc_of((uint32_t)__builtin_amdgcn_readfirstlane((int)block_of(st + 1u)));
                fetch_from(slot_src(lc_cur.x));
            }
        } else if (st + 1u < n_steps) fetch(st + 1u);
        // column pass; its outputs are truncated to int as the reference does (main.go:521-528).  (Keeping them in fp64
        // with trunc() would save the 64 conversions back, but 128 registers of intermediates spill: 162 VGPRs spilled.)
        int mid[64];
        if constexpr (C8) {
            // dequantise the whole block first (main.go:660-662), then run the column pass in place
            auto qv_of = [&](int n) { return (int)((qw[n >> 2] >> ((n & 3) * 8)) & 0xFFu); };
            auto lo8 = [&](int zi) { return (int)(int8_t)(cw[zi >> 2] >> ((zi & 3) * 8)); };
            const int low4 = (int)(cw[15] << 4) >> 28;                  // byte 63 [3:0], sign-extended
            int dc = (int)((low4 << 8) | (int)(cw[0] & 0xFFu));
            if (anyw && wflag) dc = (int)(((int)(int8_t)(hw[0] & 0xFFu) << 8) | (int)(cw[0] & 0xFFu));
            if constexpr (ST) {
                dc += dc_base;                                          // the writer's DC is relative to its sub-sequence
                if (__builtin_expect(dc != (int)(short)dc, 0)) atomicOr(&a.status[img], (uint32_t)JD_ST_COEF);   // as the writers do
            }
            if (anyw) {
                auto hi8 = [&](int zi) { return (int)(int8_t)(hw[zi >> 2] >> ((zi & 3) * 8)); };
#pragma unroll
                for (int n = 0; n < 64; n++) {
                    const int zi = kZZ.n2z[n];
                    int cv;
                    if (zi == 0) cv = dc;
                    else if (zi == 63) cv = wflag ? (int)((low4 << 8) | (int)(hw[15] >> 24)) : ((int)cw[15] >> 28);
                    else cv = (int)(short)(lo8(zi) + (hi8(zi) << 8));   // (the high part is modulo 256: 32767 = -1 + (-128 << 8))
                    mid[n] = __mul24(cv, qv_of(n));
                }
            } else {
#pragma unroll
                for (int n = 0; n < 64; n++) {
                    const int zi = kZZ.n2z[n];
                    const int cv = zi == 0 ? dc : zi == 63 ? ((int)cw[15] >> 28) : lo8(zi);
                    mid[n] = __mul24(cv, qv_of(n));
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int o[8];
                idct_1d(K, mid[0 * 8 + j], mid[1 * 8 + j], mid[2 * 8 + j], mid[3 * 8 + j], mid[4 * 8 + j], mid[5 * 8 + j],
                        mid[6 * 8 + j], mid[7 * 8 + j], o);
#pragma unroll
                for (int r = 0; r < 8; r++) mid[r * 8 + j] = o[r];
            }
        } else
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int x[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int zi = kZZ.n2z[r * 8 + j];
                const int cv = (zi & 1) ? ((int)cw[zi >> 1] >> 16) : (int)(short)(cw[zi >> 1] & 0xFFFFu);
                const int qv = (int)((qw[(r * 8 + j) >> 2] >> (((r * 8 + j) & 3) * 8)) & 0xFFu);
                x[r] = __mul24(cv, qv);                             // dequantize, main.go:660-662
            }
            int o[8];
            idct_1d(K, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], o);
#pragma unroll
            for (int r = 0; r < 8; r++) mid[r * 8 + j] = o[r];
        }
        // a row of the row pass is consumed at once (packed as a chroma sample row, or colour-converted as a luma row)
        auto row_pass = [&](int i, int (&o)[8]) {
            idct_1d(K, mid[i * 8 + 0], mid[i * 8 + 1], mid[i * 8 + 2], mid[i * 8 + 3], mid[i * 8 + 4], mid[i * 8 + 5],
                    mid[i * 8 + 6], mid[i * 8 + 7], o);
        };
        if (b >= NL) {                                             // wave-uniform: a chroma block's samples wait in registers
            int smax = 0, smin = 0;                                // as int16 pairs (exact for every stream an 8-bit encoder
#pragma unroll                                                     // can make; a sample outside int16 raises JD_ST_SAMPLE below)
            for (int i = 0; i < 8; i++) {
                int o[8];
                row_pass(i, o);
                smax = max(max(max(o[0], o[1]), o[2]), max(max(max(o[3], o[4]), o[5]), max(max(o[6], o[7]), smax)));
                smin = min(min(min(o[0], o[1]), o[2]), min(min(min(o[3], o[4]), o[5]), min(min(o[6], o[7]), smin)));
                typedef short short2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    union { short2v s; uint32_t u; } p;
                    p.s = __builtin_amdgcn_cvt_pk_i16(o[2 * k], o[2 * k + 1]);
                    if (b == NL) chroma[0][i * 4 + k] = p.u; else chroma[1][i * 4 + k] = p.u;
                }
            }
            // VS = 1: the luma blocks only read chroma rows 0-3, hipcc drops the other four row passes -- and
            // then allocates the 4:2:2 kernel so badly that it spills 47 registers (K2 + 35 %).  Kept "in use", the kernel
            // has the shape of the 4:2:0 one, which fits: four wasted row passes per chroma block are the cheaper evil.
            // only a crafted stream gets here (|sample| of an 8-bit encoder's data stays below ~1100): the packing above
            // saturated, so the pixels would differ from the reference's -- say so instead
            if (__builtin_expect(smax > 32767 || smin < -32768, 0)) atomicOr(&a.status[img], (uint32_t)JD_ST_SAMPLE);
            if constexpr (VS == 1 && HS == 2) {
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int i = 16; i < 32; i += 8)
                        asm volatile("" :: "v"(chroma[c][i]), "v"(chroma[c][i + 1]), "v"(chroma[c][i + 2]), "v"(chroma[c][i + 3]),
                                     "v"(chroma[c][i + 4]), "v"(chroma[c][i + 5]), "v"(chroma[c][i + 6]), "v"(chroma[c][i + 7]));
            }
            continue;
        }
        // chroma quadrant of this luma block: rows by*4 .. by*4+3, columns bx*4 .. bx*4+3
        uint32_t cq[2][8];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 8; i++) cq[c][i] = 0;
#pragma unroll
        for (uint32_t yb = 0; yb < NL; yb++) {
            if (b == yb) {                                         // wave-uniform; static register indices inside
                const uint32_t bx = yb % HS, by = yb / HS;
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        cq[c][r * 2 + 0] = chroma[c][(by * 4 + r) * 4 + bx * 2 + 0];
                        cq[c][r * 2 + 1] = chroma[c][(by * 4 + r) * 4 + bx * 2 + 1];
                    }
            }
        }
        const uint32_t bx = b % HS;
        uint32_t by = b / HS;
        // (VS = 1: by is always 0, the eight row addresses become loop invariants, hipcc hoists them and the 4:2:2
        // kernel spills 79 registers.  Laundered through a scalar register they are computed where they are used.)
        asm volatile("" : "+s"(by));
        const uint32_t X = X0 + bx * 8u;
        if (X >= W) continue;
        const bool full = al8 && X + 8u <= W;
        // HS = 2: the two luma blocks of a row band are stored together, 48 contiguous bytes per lane and row (a
        // wave then writes 3 KiB runs); stores of 24 B at a 48-B pitch ran at 3 TB/s instead of 4.5
        const bool pair = HS == 2 && al16 && X0 + 16u <= W;
#pragma unroll
        for (int r = 0; r < 4; r++) {                              // chroma row r serves pixel rows 2r, 2r+1
            float rc[4], gb[4], gr[4], bc[4];
            // fence: the products below stay behind the previous row's k2_rtz_off
            asm volatile("" : "+v"(cq[0][r * 2]), "+v"(cq[0][r * 2 + 1]), "+v"(cq[1][r * 2]), "+v"(cq[1][r * 2 + 1]));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t cbw = cq[0][r * 2 + (k >> 1)], crw = cq[1][r * 2 + (k >> 1)];
                const int cbi = (k & 1) ? ((int)cbw >> 16) : (int)(short)(cbw & 0xFFFFu);
                const int cri = (k & 1) ? ((int)crw >> 16) : (int)(short)(crw & 0xFFFFu);
                const float cb = (float)cbi, cr = (float)cri;
                rc[k] = __fmul_rn(1.402f, cr); gb[k] = __fmul_rn(0.344f, cb); gr[k] = __fmul_rn(0.714f, cr);
                bc[k] = __fmul_rn(1.772f, cb);
            }
            // fence: the products exist before the first k2_rtz_on below (the scheduler would otherwise sink the ones
            // only later pixels use into an earlier conversion window)
            asm volatile("" : "+v"(rc[0]), "+v"(rc[1]), "+v"(rc[2]), "+v"(rc[3]), "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3]));
            asm volatile("" : "+v"(gr[0]), "+v"(gr[1]), "+v"(gr[2]), "+v"(gr[3]), "+v"(bc[0]), "+v"(bc[1]), "+v"(bc[2]), "+v"(bc[3]));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = 2 * r + h;
                const uint32_t Yp = Y0 + by * 8u + (uint32_t)i;
                int ys[8];                                         // row i of the luma block, straight from the row pass
                row_pass(i, ys);
                // the two pixels that share chroma sample k, as one packed-f32 pair: the same additions in the same
                // order as main.go:679-703, two at a time
                typedef jd_f32x2 f32x2;
                // fence: the row's samples are final before any f32 arithmetic of this row (keeps that arithmetic
                // behind the previous row's k2_rtz_off)
                asm volatile("" : "+v"(ys[0]), "+v"(ys[1]), "+v"(ys[2]), "+v"(ys[3]), "+v"(ys[4]), "+v"(ys[5]), "+v"(ys[6]), "+v"(ys[7]));
                uint32_t d[6];
                auto window = [&](auto hh_c) {                     // JD_K2_WIN_PAIRS pixel pairs: 6 values each in flight
                    constexpr int hh = decltype(hh_c)::value, NK = JD_K2_WIN_PAIRS;
                    f32x2 R[NK], G[NK], B[NK];
#pragma unroll
                    for (int kk = 0; kk < NK; kk++) {
                        const int k = NK * hh + kk;
                        // (the sums with a chroma product are plain v_add_f32 / v_sub_f32: as packed operations each
                        // needed its product copied into both halves of a register pair first -- a v_mov per product and
                        // pixel row, 512 per MCU -- and two plain adds cost no more issue cycles than one packed add)
                        const float y0 = (float)ys[2 * k], y1 = (float)ys[2 * k + 1];
                        const f32x2 off = { 128.0f, 128.0f };
                        float r0 = __fadd_rn(y0, rc[k]), r1 = __fadd_rn(y1, rc[k]);
                        float g0 = __fsub_rn(y0, gb[k]), g1 = __fsub_rn(y1, gb[k]);
                        float b0 = __fadd_rn(y0, bc[k]), b1 = __fadd_rn(y1, bc[k]);
                        asm("" : "+v"(r0), "+v"(r1), "+v"(g0), "+v"(g1), "+v"(b0), "+v"(b1));   // (or hipcc packs them again)
                        g0 = __fsub_rn(g0, gr[k]); g1 = __fsub_rn(g1, gr[k]);
                        asm("" : "+v"(g0), "+v"(g1));
                        R[kk] = f32x2{ r0, r1 } + off;
                        G[kk] = f32x2{ g0, g1 } + off;
                        B[kk] = f32x2{ b0, b1 } + off;
                    }
                    // fence: the values exist before the rounding mode changes
#pragma unroll
                    for (int kk = 0; kk < NK; kk += 2)
                        asm volatile("" : "+v"(R[kk]), "+v"(R[kk + 1]), "+v"(G[kk]), "+v"(G[kk + 1]), "+v"(B[kk]), "+v"(B[kk + 1]));
                    k2_rtz_on();
                    k2_put_group<hh * NK * 6, NK>(d, R, G, B, std::make_integer_sequence<int, NK * 6>{});
                    k2_rtz_off();
                };
                window(std::integral_constant<int, 0>{});
#if JD_K2_WIN_PAIRS == 2
                // fence: the second half's arithmetic starts behind the first half's k2_rtz_off
                asm volatile("" : "+v"(ys[4]), "+v"(ys[5]), "+v"(ys[6]), "+v"(ys[7]));
                window(std::integral_constant<int, 1>{});
#endif
                if (Yp < H && pair) {
                    // The parked rows go to LDS and back through asm: hipcc cannot tell this area from the one the LDS-DMA
                    // writes, so before every ds access of its own it waits for vmcnt(0) -- the next block's DMA AND every
                    // pixel store issued so far, once per pixel row.  (LDS operations of one wave execute in order.)
                    // (row i at stage_base + i * 1536: the row offset rides in the instruction's offset field -- as eight
                    // separate addresses hipcc keeps eight loop-invariant registers for them)
                    if (bx == 0) {
                        const uint64_t p0 = d[0] | ((uint64_t)d[1] << 32), p1 = d[2] | ((uint64_t)d[3] << 32), p2 = d[4] | ((uint64_t)d[5] << 32);
                        asm volatile("ds_write_b64 %0, %1 offset:%4\n\tds_write_b64 %0, %2 offset:%5\n\tds_write_b64 %0, %3 offset:%6"
                                     :: "v"(stage_base), "v"(p0), "v"(p1), "v"(p2), "n"(i * 1536), "n"(i * 1536 + 8), "n"(i * 1536 + 16) : "memory");
                    } else {
                        uint64_t l0, l1, l2;
                        asm volatile("ds_read_b64 %0, %3 offset:%4\n\tds_read_b64 %1, %3 offset:%5\n\tds_read_b64 %2, %3 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(l0), "=&v"(l1), "=&v"(l2) : "v"(stage_base), "n"(i * 1536), "n"(i * 1536 + 8), "n"(i * 1536 + 16) : "memory");
                        g_u4* o = (g_u4*)(rgbp + (uint64_t)Yp * rs + X0 * 3u);
                        o[0] = u32x4{ (uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)l1, (uint32_t)(l1 >> 32) };
                        o[1] = u32x4{ (uint32_t)l2, (uint32_t)(l2 >> 32), d[0], d[1] };
                        o[2] = u32x4{ d[2], d[3], d[4], d[5] };
                    }
                } else if (Yp < H) {
                    g_u8* dst = rgbp + (uint64_t)Yp * rs + X * 3u;
                    if (full) {
                        g_u2* o = (g_u2*)dst;
                        o[0] = u32x2{ d[0], d[1] }; o[1] = u32x2{ d[2], d[3] }; o[2] = u32x2{ d[4], d[5] };
                    } else {
                        const uint32_t n = W - X < 8u ? W - X : 8u;
#pragma unroll
                        for (uint32_t k = 0; k < 24; k++)
                            if (k < n * 3u) dst[k] = (uint8_t)(d[k >> 2] >> ((k & 3u) * 8u));
                    }
                }
            }
        }
    }
}

// One kernel per sampling mode (0: grey, 1: 1x1, 2: 2x1, 3: 1x2, 4: 2x2): compiled together the five bodies share one
// register allocation and spill (55 VGPRs; 113 with a whole pixel row per conversion window), alone the 4:2:0 body
// needs 216.  A launch covers the batch's whole tile grid; tiles of images in another mode leave at once, and the
// launcher only starts the modes the batch holds (jd_batch: k2_modes).
__device__ __forceinline__ uint32_t k2_mode_of(const JdImg& im) { return im.ncomp == 1 ? 0u : im.hs + 2u * im.vs - 2u; }
template <int MODE, bool C8, bool ST = false>
__global__ __launch_bounds__(64, 2) void k2_mcu_lanes(JdKernelArgs a, JdIdctConst K) {
    const int img = find_image(a.prefix, a.n_images, blockIdx.x);
    const JdImg& im = a.imgs[img];
    if (k2_mode_of(im) != (uint32_t)MODE || ((im.flags & JD_IMG_COEF8) != 0) != C8 || ((im.flags & JD_IMG_FUSED) != 0) != ST) return;   // wave-uniform
    const uint32_t tile = blockIdx.x - a.prefix[img];
    if constexpr (MODE == 0) k2m_body<1, 1, 1, C8, ST>(a, im, img, tile, K);
    else if constexpr (MODE == 1) k2m_body<1, 1, 3, C8, ST>(a, im, img, tile, K);
    else if constexpr (MODE == 2) k2m_body<2, 1, 3, C8, ST>(a, im, img, tile, K);
    else if constexpr (MODE == 3) k2m_body<1, 2, 3, C8, ST>(a, im, img, tile, K);
    else k2m_body<2, 2, 3, C8, ST>(a, im, img, tile, K);
}

// byte form -> int16, zig-zag order: what jd_batch_coefficients hands out (one lane per block)
__global__ __launch_bounds__(256) void k_coef8_expand(JdKernelArgs a, int img, int16_t* __restrict__ out) {
    const JdImg& im = a.imgs[img];
    const uint32_t nblocks = im.n_mcus * im.bpm;
    const uint32_t blk = blockIdx.x * 256u + threadIdx.x;
    if (blk >= nblocks) return;
    const uint8_t* lo = reinterpret_cast<const uint8_t*>(a.coef + im.coef_off) + (uint64_t)blk * 64u;
    const uint8_t* hi = reinterpret_cast<const uint8_t*>(a.coef + im.coef_off) + ((uint64_t)nblocks + blk) * 64u;
    int dc_base = 0;
    if (im.flags & JD_IMG_FUSED) {                                // staged: the block's slot and DC base
        const uint2 l = a.loc[im.loc_off + blk];
        const uint32_t slot = l.x < im.slot_cap ? l.x : im.slot_cap - 1u;
        lo = reinterpret_cast<const uint8_t*>(a.coef + im.coef_off) + (uint64_t)slot * 64u;
        const uint8_t* base = reinterpret_cast<const uint8_t*>(a.coef + im.coef_off);
        const uint32_t hs = (lo[63] >> 4) == 8u ? reinterpret_cast<const uint32_t*>(base + JD_F_HIMAP_OFF(im.slot_cap))[slot] : 0u;
        hi = base + JD_F_HIPOOL_OFF(im.slot_cap) + (uint64_t)(hs < im.hi_cap ? hs : 0u) * 64u;
        dc_base = (int)l.y;
    }
    int16_t* o = out + (uint64_t)blk * 64u;
    const uint32_t b63 = lo[63];
    const bool wide = (b63 >> 4) == 8u;
    const int low4 = (int)((int32_t)(b63 << 28) >> 28);
    o[0] = (int16_t)((((wide ? (int)(int8_t)hi[0] : low4) << 8) | (int)lo[0]) + dc_base);
    for (int k = 1; k < 63; k++) o[k] = (int16_t)((int)(int8_t)lo[k] + (wide ? ((int)(int8_t)hi[k] << 8) : 0));
    o[63] = wide ? (int16_t)((low4 << 8) | (int)hi[63]) : (int16_t)((int)(int8_t)b63 >> 4);
}

// ------------------------------------------------------------------------------------------------
// Fixed workspace of jd_decode_resident (include/jpeghip.h): callers that cannot allocate get the whole decode
// workspace -- descriptors, tables, unstuffed stream, restart index, sync records, coefficients -- out of this array.
// Each device the library runs on holds its own copy until the process ends.  The name is a host-side stand-in, not a
// device address: host code asks for the address once per device (jdk_static_scratch, from jd_init with that device
// current) and hands THAT pointer to memsets, copies and kernels.  (No __attribute__((used)) here: with it hipcc keeps the
// device symbol LOCAL instead of externalising it for the host's use, and hipGetSymbolAddress aborts the process.)
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) uint8_t scratch[JD_RESIDENT_SCRATCH_BYTES];

extern "C" int jdk_static_scratch(void** ptr, size_t* bytes) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) return e != hipSuccess ? (int)e : -1;
    size_t n = 0;
    if (hipGetSymbolSize(&n, HIP_SYMBOL(scratch)) != hipSuccess || n != JD_RESIDENT_SCRATCH_BYTES) return -1;
    *ptr = p; *bytes = n;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
extern "C" {

void jdk_launch_k0(const JdKernelArgs& a, const uint32_t* d_prefix_k0, uint32_t wgs, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix_k0;
    // ticket counter + per-chunk status words start at zero every run (memsets of >= 256 bytes on the stream)
    (void)hipMemsetAsync(a.k0_ticket, 0, (((size_t)a.n_images * 4) + 255) & ~(size_t)255, s);
    (void)hipMemsetAsync(a.k0_state, 0, ((size_t)wgs * 8 + 255) & ~(size_t)255, s);
    uint32_t stride = a.n_images > 1 ? wgs / (uint32_t)a.n_images : 1u;   // about one image's chunks
    if (stride == 0) stride = 1;
    auto gcd = [](uint32_t p, uint32_t q) { while (q) { const uint32_t t = p % q; p = q; q = t; } return p; };
    while (gcd(stride, wgs) != 1u) stride++;
    hipLaunchKernelGGL(k0_unstuff, dim3(wgs), dim3(JD_K0_THREADS), 0, s, x, stride);
}

// fmts: bit 0 = some image keeps int16 coefficients, bit 1 = some image uses the byte form
void jdk_launch_intervals(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, uint32_t fmts, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    const uint32_t lb = (lut_bytes + 127u) & ~127u;
    const uint32_t lds = lb + JD_W_WAVES * JD_WAVE_LDS;
    if (lds > 64u * 1024u) {                                     // > 64 KiB of dynamic LDS needs the attribute (per device)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1_write_intervals<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1_write_intervals<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (fmts & 1u) hipLaunchKernelGGL(k1_write_intervals<false>, dim3(wgs), dim3(64 * JD_W_WAVES), lds, s, x, lb);
    if (fmts & 2u) hipLaunchKernelGGL(k1_write_intervals<true>, dim3(wgs), dim3(64 * JD_W_WAVES), lds, s, x, lb);
}

void jdk_launch_sync(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, int fixup,
                     int flag_slot, int warm_off, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    const uint32_t lb = (lut_bytes + 15u) & ~15u;
    const uint32_t lds = JD_SY_LUT + lb;
    hipLaunchKernelGGL(k1_sync, dim3(wgs), dim3(JD_SYNC_WG), lds, s, x, fixup, flag_slot, warm_off);
}

void jdk_launch_sync_spine(const JdKernelArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k1_sync_spine, dim3(a.n_images), dim3(256), 0, s, a);
}

void jdk_launch_dc_scan(const JdKernelArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k1_dc_scan, dim3(a.n_images * 3), dim3(1024), 0, s, a);
}

void jdk_launch_pscan(const JdKernelArgs& a, const JdPScan* d_scans, const uint32_t* d_list, int n, const uint8_t* d_pstream,
                      const uint32_t* d_plut, const uint32_t* d_plen, const uint32_t* d_pstatus, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(kp_scan, dim3(n), dim3(64), 0, s, a, d_scans, d_list, d_pstream, d_plut, d_plen, d_pstatus);
}

void jdk_launch_write_subseq(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, uint32_t fmts, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    const uint32_t lb = (lut_bytes + 127u) & ~127u;
    const uint32_t lds = lb + JD_W_WAVES * JD_WAVE_LDS;
    if (lds > 64u * 1024u) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1_write_subseq<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1_write_subseq<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (fmts & 1u) hipLaunchKernelGGL(k1_write_subseq<false>, dim3(wgs), dim3(64 * JD_W_WAVES), lds, s, x, lb);
    if (fmts & 2u) hipLaunchKernelGGL(k1_write_subseq<true>, dim3(wgs), dim3(64 * JD_W_WAVES), lds, s, x, lb);
}

// modes: bit (3 * m + f): some image has sampling mode m and coefficient form f (0 = int16, 1 = bytes, 2 = bytes in staging)
void jdk_launch_idct_colour(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t modes, const JdIdctConst& K,
                            hipStream_t s) {
    JdKernelArgs x = a; x.prefix = d_prefix;
    if (!wgs) return;
    // dynamic LDS per wave: 8 KiB for the coefficient DMA; the modes with two luma blocks side by side (2 and 4) also park
    // the left block's pixel rows (8 x 1536 B) -- the others must not pay for that in occupancy
#define JD_K2_LAUNCH(M) \
    { const uint32_t lds = 8192u + ((M) == 2 || (M) == 4 ? 8u * 1536u : 0u); \
      if (modes & (1u << (3 * M))) hipLaunchKernelGGL((k2_mcu_lanes<M, false>), dim3(wgs), dim3(64), lds, s, x, K); \
      if (modes & (2u << (3 * M))) hipLaunchKernelGGL((k2_mcu_lanes<M, true>), dim3(wgs), dim3(64), lds, s, x, K); \
      if (modes & (4u << (3 * M))) hipLaunchKernelGGL((k2_mcu_lanes<M, true, true>), dim3(wgs), dim3(64), lds, s, x, K); }
    JD_K2_LAUNCH(0) JD_K2_LAUNCH(1) JD_K2_LAUNCH(2) JD_K2_LAUNCH(3) JD_K2_LAUNCH(4)
#undef JD_K2_LAUNCH
}

// ---- single-decode path: d_prefix = workgroups of JD_F_LANES lanes per image -------------------------------------------
void jdk_launch_fcold(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    const uint32_t lb = (lut_bytes + 15u) & ~15u;
    hipLaunchKernelGGL(k1f_cold, dim3(wgs), dim3(JD_SYNC_WG), JD_SY_LUT + lb, s, x);
}
// round < 0: every sub-sequence; round >= 0: the sub-sequences k1f_check listed for that round
// lanes0: workgroup size of round 0 (64 / 128 / JD_F_REDO_LANES, by how long an image's first list will be)
// share: 0, or the waves an image gets in workgroups of eight waves on one copy of the tables (k1f_write's share_w: 1, 2, 4)
void jdk_launch_fwrite(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, int round, uint32_t lanes0,
                       uint32_t share, uint32_t share_w_off, uint32_t share_w_len, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    const uint32_t lb = (lut_bytes + 127u) & ~127u;
    if (lb + JD_W_WAVES * JD_WAVE_LDS > 64u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1f_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lb + JD_W_WAVES * JD_WAVE_LDS));
    // lanes per workgroup: the first decode fills eight waves; round 0's lists (one sub-sequence in twelve) fill one to
    // four; later lists are a handful of entries per image -- one wave, so that a CU holds eight such workgroups, not
    // three (a workgroup's LDS is the tables + 8 KB per wave, and these launches are bound by how many walks run at once)
    const uint32_t lanes = round < 0 ? 64u * JD_W_WAVES : round == 0 ? lanes0 : 64u;
    if (share == 1u || share == 2u || share == 4u) {             // one table set for all: eight waves share it, `share` waves per image
        const uint32_t per_wg = JD_W_WAVES / share;
        hipLaunchKernelGGL(k1f_write, dim3(((uint32_t)a.n_images + per_wg - 1u) / per_wg), dim3(64u * JD_W_WAVES), lb + JD_W_WAVES * JD_WAVE_LDS, s,
                           x, lb, round, share, share_w_off, share_w_len);
        return;
    }
    const uint32_t lds = lb + (lanes / 64u) * JD_WAVE_LDS;
    hipLaunchKernelGGL(k1f_write, dim3(wgs), dim3(lanes), lds, s, x, lb, round, 0u, 0u, 0u);
}
void jdk_launch_fcheck(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, int round, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    hipLaunchKernelGGL(k1f_check, dim3(wgs), dim3(64 * JD_W_WAVES), 0, s, x, round);
}
// tiled: some image is large (its sub-sequences are many trips of one workgroup)
void jdk_launch_fscan_index(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, int tiled, hipStream_t s) {
    if (!wgs) return;
    JdKernelArgs x = a; x.prefix = d_prefix;
    if (tiled) {
        hipLaunchKernelGGL(k1f_scan<0>, dim3(wgs), dim3(64 * JD_W_WAVES), 0, s, x);
        hipLaunchKernelGGL(k1f_scan<1>, dim3(wgs), dim3(64 * JD_W_WAVES), 0, s, x);
    } else hipLaunchKernelGGL(k1f_scan_img, dim3(a.n_images), dim3(1024), 0, s, x);
    hipLaunchKernelGGL(k1f_index, dim3(wgs), dim3(64 * JD_W_WAVES), 0, s, x);
}

void jdk_launch_coef8_expand(const JdKernelArgs& a, int img, uint32_t nblocks, int16_t* d_out, hipStream_t s) {
    if (!nblocks) return;
    hipLaunchKernelGGL(k_coef8_expand, dim3((nblocks + 255u) / 256u), dim3(256), 0, s, a, img, d_out);
}

}  // extern "C"
