// jd_api.cpp -- the C-ABI of include/jpeghip.h: frame validation, workspace planning, table upload, the kernel
// sequence and the pinned-memory I/O pipeline.  Host code only; the kernels are in jd_kernels.hip.
//
// There is no CPU decode path in this library: every entry point that produces coefficients or pixels runs
// the HIP kernels and fails with JD_E_HIP / JD_E_NOT_INIT when no device is usable.
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jpeghip.h"
#include "jd_internal.h"
#include "jd_tables.h"

// writer-table bytes that leave room for two writer workgroups on a CU (160 KiB of LDS)
static const size_t kWriterLutTwoPerCu = (160 * 1024 / 2 - JD_W_WAVES * JD_WAVE_LDS) & ~(size_t)127;

extern "C" {
void jdk_launch_k0(const JdKernelArgs& a, const uint32_t* d_prefix_k0, uint32_t wgs, hipStream_t s);
void jdk_launch_intervals(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, uint32_t fmts, hipStream_t s);
void jdk_launch_sync(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, int fixup,
                     int flag_slot, int warm_off, hipStream_t s);
void jdk_launch_sync_spine(const JdKernelArgs& a, hipStream_t s);
void jdk_launch_write_subseq(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, uint32_t fmts, hipStream_t s);
void jdk_launch_coef8_expand(const JdKernelArgs& a, int img, uint32_t nblocks, int16_t* d_out, hipStream_t s);
void jdk_launch_fcold(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, hipStream_t s);
void jdk_launch_fwrite(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t lut_bytes, int round, uint32_t lanes0,
                       uint32_t share, uint32_t share_w_off, uint32_t share_w_len, hipStream_t s);
void jdk_launch_fcheck(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, int round, hipStream_t s);
void jdk_launch_fscan_index(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, int tiled, hipStream_t s);
void jdk_launch_dc_scan(const JdKernelArgs& a, hipStream_t s);
void jdk_launch_idct_colour(const JdKernelArgs& a, const uint32_t* d_prefix, uint32_t wgs, uint32_t modes, const JdIdctConst& K, hipStream_t s);
void jdk_launch_pscan(const JdKernelArgs& a, const JdPScan* d_scans, const uint32_t* d_list, int n, const uint8_t* d_pstream,
                      const uint32_t* d_plut, const uint32_t* d_plen, const uint32_t* d_pstatus, hipStream_t s);
int jdk_static_scratch(void** ptr, size_t* bytes);
}

namespace {

thread_local int g_last_hip = 0;

#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return JD_E_HIP; } \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- per-device state set up by jd_init ----------------------------------------------------------------
// `scratch` is the device address of the library's file-scope scratch array on that device (jd_decode_resident works
// inside it and allocates nothing); `pinned` is the host staging block its descriptors, tables and status words go
// through.
struct DevCtx {
    int device = -1;
    uint8_t* scratch = nullptr; size_t scratch_bytes = 0;
    uint8_t* pinned = nullptr; size_t pinned_bytes = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;                            // host-side: one jd_decode_resident at a time per device
    // the plan of the last jd_decode_resident on this device (descriptors, packed tables already in the scratch): a call
    // with the same frame and an entropy-coded segment that fits its capacity skips the planning and the table upload
    ::jd_batch* plan = nullptr;
    jd_frame plan_frame;
    size_t plan_cap = 0;
};
const size_t kStaticPinnedBytes = 1u << 20;

std::mutex g_mu;
bool g_init = false;
std::vector<std::unique_ptr<DevCtx>> g_ctx;

DevCtx* find_ctx(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return nullptr;
    for (auto& c : g_ctx) if (c->device == device) return c.get();
    return nullptr;
}

void drop_plan(DevCtx* c);

void release_ctx_locked() {
    for (auto& c : g_ctx) {
        if (hipSetDevice(c->device) != hipSuccess) continue;
        drop_plan(c.get());
        if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
        if (c->pinned) (void)hipHostFree(c->pinned);
    }
    g_ctx.clear();
    g_init = false;
}

// ---- IDCT constants, main.go:1339-1354 ---------------------------------------------------------------
// The Go binary computes them at init with its pure-Go math.Cos; the values below are those results (bit
// patterns listed in SURVEY.md Appendix A.2, Go flavour).  tests/test_abi.py checks them against the oracle's
// independent derivation.
inline double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
JdIdctConst idct_constants() {
    JdIdctConst K;
    K.M1 = from_bits(0x3ff6a09e667f3bcdull); K.M2 = from_bits(0x3ff1517a7bdb3894ull);
    K.M3 = from_bits(0x3ff6a09e667f3bcdull); K.M4 = from_bits(0x4004e7ae9144f0fcull);
    K.M5 = from_bits(0x3fe87de2a6aea964ull);
    K.S0 = from_bits(0x3fd6a09e667f3bccull); K.S1 = from_bits(0x3fdf6297cff75cb0ull);
    K.S2 = from_bits(0x3fdd906bcf328d46ull); K.S3 = from_bits(0x3fda9b66290ea1a4ull);
    K.S4 = from_bits(0x3fd6a09e667f3bcdull); K.S5 = from_bits(0x3fd1c73b39ae68c9ull);
    K.S6 = from_bits(0x3fc87de2a6aea964ull); K.S7 = from_bits(0x3fb8f8b83c69a60cull);
    return K;
}

int status_to_error(uint32_t st) {
    if (st & JD_ST_MARKER) return JD_E_MARKER;
    if (st & (JD_ST_RST_COUNT | JD_ST_RST_SEQ)) return JD_E_RESTART;
    if (st & JD_ST_HUFF) return JD_E_HUFFMAN;
    if (st & JD_ST_COEF) return JD_E_COEF;
    if (st & JD_ST_TRUNC) return JD_E_TRUNCATED;
    if (st & (JD_ST_SYNC | JD_ST_COUNT)) return JD_E_SYNC;
    if (st & JD_ST_SAMPLE) return JD_E_COEF;
    return JD_OK;
}

struct Geometry { uint32_t mcux, mcuy, bpm, hs, vs; };

int frame_geometry(const jd_frame* f, Geometry* g) {
    if (!f) return JD_E_ARG;
    if (f->width == 0 || f->height == 0) return JD_E_ARG;
    if (!(f->ncomp == 1 || f->ncomp == 3)) return JD_E_UNSUPPORTED;
    uint32_t hs = f->comp[0].h, vs = f->comp[0].v;
    if (f->ncomp == 1) { hs = 1; vs = 1; }   // single-component scan: 1x1 MCU (luminanceOnlyScan, main.go:753-767)
    if (!(hs == 1 || hs == 2) || !(vs == 1 || vs == 2)) return JD_E_UNSUPPORTED;
    for (int c = 1; c < f->ncomp; c++)
        if (f->comp[c].h != 1 || f->comp[c].v != 1) return JD_E_UNSUPPORTED;   // spreadCoeffecients assumes 1x1
    g->hs = hs; g->vs = vs;
    const uint32_t bw = (f->width + 7u) / 8u, bh = (f->height + 7u) / 8u;      // main.go:160-161
    g->mcux = (bw + hs - 1) / hs; g->mcuy = (bh + vs - 1) / vs;                 // loops at main.go:769-770
    g->bpm = hs * vs + (f->ncomp - 1u);
    return JD_OK;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" const char* jd_strerror(int code) {
    switch (code) {
    case JD_OK: return "ok";
    case JD_E_ARG: return "invalid argument";
    case JD_E_UNSUPPORTED: return "Error! unsupported frame (8-bit, 1 or 3 components -- the reference's loops would take 2, main.go:112-115, "
                                  "but no encoder writes such files and no kernel is built for them --, luma 1-2 x 1-2, chroma 1x1; scans of 1-3 components)";
    case JD_E_TABLE: return "Error! missing or malformed quantisation / Huffman table";
    case JD_E_HIP: return "HIP runtime error";
    case JD_E_NOMEM: return "out of memory";
    case JD_E_MARKER: return "Invalid marker found in the bitsteam";
    case JD_E_RESTART: return "Error! restart markers do not match the restart interval";
    case JD_E_HUFFMAN: return "Error! invalid DC Symbols";
    case JD_E_COEF: return "Error! run/size walks past coefficient 63, or a coefficient (AC 63 in the byte form: 12 bits) or a chroma sample leaves 16 bits";
    case JD_E_TRUNCATED: return "Error! EOF";
    case JD_E_SYNC: return "internal error: self-synchronising decode did not converge";
    case JD_E_NOT_INIT: return "jd_init has not been called or found no device";
    case JD_E_STATE: return "batch call out of order";
    case JD_E_TOO_LARGE: return "image exceeds the fixed scratch of jd_decode_resident (JD_RESIDENT_MAX_ECS / JD_RESIDENT_MAX_COEF)";
    default: return "unknown error";
    }
}
extern "C" int jd_last_hip_error(void) { return g_last_hip; }

extern "C" int jd_init(const int* device_ids, int n_devices) {
    std::lock_guard<std::mutex> lk(g_mu);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { g_last_hip = (int)e; return JD_E_HIP; }
    std::vector<int> devs;
    if (!device_ids || n_devices <= 0) for (int i = 0; i < count; i++) devs.push_back(i);
    else for (int i = 0; i < n_devices; i++) { if (device_ids[i] < 0 || device_ids[i] >= count) return JD_E_ARG; devs.push_back(device_ids[i]); }
    if (g_init) {                                              // idempotent for the same device set
        bool same = devs.size() == g_ctx.size();
        for (size_t i = 0; same && i < devs.size(); i++) same = g_ctx[i]->device == devs[i];
        if (same) return JD_OK;
        release_ctx_locked();
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = JD_OK;
    for (int d : devs) {
        std::unique_ptr<DevCtx> c(new DevCtx());
        c->device = d;
        hipError_t he = hipSetDevice(d);
        // each device holds its own copy of the scratch array at its own address: ask with that device current
        void* p = nullptr; size_t bytes = 0;
        if (he == hipSuccess && jdk_static_scratch(&p, &bytes) != 0) he = hipErrorInvalidSymbol;
        if (he == hipSuccess) he = hipHostMalloc((void**)&c->pinned, kStaticPinnedBytes, hipHostMallocDefault);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        g_ctx.push_back(std::move(c));
        if (he != hipSuccess) { g_last_hip = (int)he; rc = JD_E_HIP; break; }
        g_ctx.back()->scratch = (uint8_t*)p; g_ctx.back()->scratch_bytes = bytes; g_ctx.back()->pinned_bytes = kStaticPinnedBytes;
    }
    g_init = true;
    if (rc != JD_OK) release_ctx_locked();
    (void)hipSetDevice(prev);
    return rc;
}
extern "C" void jd_shutdown(void) { std::lock_guard<std::mutex> lk(g_mu); int prev = 0; (void)hipGetDevice(&prev); release_ctx_locked(); (void)hipSetDevice(prev); }
extern "C" int jd_device_count(void) { std::lock_guard<std::mutex> lk(g_mu); return g_init ? (int)g_ctx.size() : 0; }

extern "C" int jd_frame_geometry(const jd_frame* f, uint32_t* mx, uint32_t* my, uint32_t* bpm) {
    Geometry g; int rc = frame_geometry(f, &g);
    if (rc) return rc;
    if (mx) *mx = g.mcux; if (my) *my = g.mcuy; if (bpm) *bpm = g.bpm;
    return JD_OK;
}

extern "C" int jd_frame_check(const jd_frame* f) {
    Geometry g; int rc = frame_geometry(f, &g);
    if (rc) return rc;
    for (int c = 0; c < f->ncomp; c++) {
        const jd_component& k = f->comp[c];
        if (k.tq > 3 || !(f->qt_present & (1u << k.tq))) return JD_E_TABLE;       // getQuantizationTable nil (main.go:443-452)
        if (k.td > 3 || k.ta > 3) return JD_E_TABLE;
        if (!f->huff[0][k.td].present || !f->huff[1][k.ta].present) return JD_E_TABLE;   // getTable nil (main.go:432-440)
    }
    std::vector<uint16_t> scratch;
    for (int c = 0; c < f->ncomp; c++) {
        const jd_huff& d = f->huff[0][f->comp[c].td]; const jd_huff& a = f->huff[1][f->comp[c].ta];
        if (jd_pack_lut(d.bits, d.vals, d.nvals, true, scratch) || jd_pack_lut(a.bits, a.vals, a.nvals, false, scratch)) return JD_E_TABLE;
    }
    return JD_OK;
}

extern "C" int jd_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return JD_E_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { g_last_hip = (int)e; return e == hipErrorOutOfMemory ? JD_E_NOMEM : JD_E_HIP; }
    return JD_OK;
}
extern "C" void jd_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---------------------------------------------------------------------------------------------------
// Where a batch's device workspace comes from: pieces of its own from hipMalloc, or slices of a fixed block (nothing
// is allocated then; a request that does not fit fails).
struct Arena {
    uint8_t* base = nullptr; size_t cap = 0, used = 0;
    std::vector<void*> owned;
    size_t total = 0;
    int poison = -1;                          // JD_DEBUG_POISON=byte: every block starts filled with it (tests: nothing may depend
                                              // on what freshly allocated device memory happens to hold)
    int poison_only = -1, count = 0;          // JD_DEBUG_POISON_ONLY=k: ... only the k-th block taken
    bool fixed() const { return base != nullptr; }
    int take(void** p, size_t bytes) {
        bytes = std::max<size_t>(align_up(bytes, 256), 256);
        if (fixed()) {
            if (used + bytes > cap) return JD_E_TOO_LARGE;
            *p = base + used; used += bytes;
        } else {
            HIPCHK(hipMalloc(p, bytes));
            owned.push_back(*p);
        }
        if (poison >= 0 && (poison_only < 0 || poison_only == count)) {
            // (a device memset may return before it has run, and the batch's streams do not wait for the null stream:
            // without the wait the fill can land on top of the tables uploaded next)
            HIPCHK(hipMemset(*p, poison, bytes));
            HIPCHK(hipStreamSynchronize(nullptr));
        }
        count++;
        total += bytes;
        return JD_OK;
    }
};

struct jd_batch {
    int device = 0, n = 0;
    unsigned flags = 0;
    std::vector<jd_frame> frames;
    std::vector<JdImg> img;                   // host mirror of the device descriptors
    std::vector<size_t> ecs_cap, own_ecs_off, own_rgb_off;
    std::vector<uint8_t> rgb_bound;
    std::vector<uint32_t> status;
    Arena arena;
    // device memory
    uint8_t* d_desc = nullptr;                // JdImg[n] + five workgroup prefixes, one block (one copy per refresh)
    JdImg* d_img = nullptr;
    uint8_t* d_ecs_own = nullptr; uint8_t* d_rgb_own = nullptr; uint8_t* d_stream = nullptr;
    int16_t* d_coef = nullptr; uint16_t* d_lut = nullptr;
    uint32_t *d_status = nullptr, *d_stream_len = nullptr, *d_interval = nullptr, *d_changed = nullptr;
    size_t clear_bytes = 0;                   // d_status .. d_changed + 64 words
    unsigned long long* d_k0 = nullptr; uint32_t* d_ticket = nullptr; JdSub* d_sub = nullptr; JdSubPfx* d_subpfx = nullptr; JdSubCk* d_subck = nullptr; uint4* d_sync_tot = nullptr;
    uint32_t *d_pre_k0 = nullptr, *d_pre_iv = nullptr, *d_pre_sync = nullptr, *d_pre_wr = nullptr, *d_pre_tile = nullptr, *d_pre_f = nullptr, *d_pre_fr = nullptr;
    // single-decode path of the streams without restart markers (JD_IMG_FUSED, jd_internal.h)
    uint2* d_loc = nullptr; uint32_t *d_chunk = nullptr, *d_guess = nullptr, *d_wl = nullptr, *d_falloc = nullptr, *d_fhalloc = nullptr, *d_fcount = nullptr;
    uint32_t fkmax = 0, wg_f = 0, wg_fr = 0;
    int fscan_tiled = 0;                        // some image has more than 8192 sub-sequences: prefix scan in tiles
    uint32_t fsub_bits = JD_SUBSEQ_BITS;        // sub-sequence size of the single-decode path in this run (by the batch's size)
    uint32_t redo0_lanes = JD_F_REDO_LANES;     // workgroup size of the first re-decode round (by the images' list lengths)
    bool f_small_class = false;               // this run: many small images on one table set (refresh)
    uint32_t f_share = 0, f_share_first = 0, f_share_off = 0, f_share_len = 0;   // re-decode launches of a batch on one table set: eight waves per copy of the tables (refresh; bit 0: round 0, bit 1: later rounds)
    bool fused = true;                         // JD_NODRI_FUSED=0, or a run whose staging overflowed, turns it off for the batch
    bool fused_always = false;                 // JD_NODRI_FUSED=2: whatever the batch's size (tests)
    std::vector<uint8_t> fz_ok;                // per image: planned for the single-decode path (staging allocated)
    int fused_rounds = JD_F_ROUNDS;            // JD_DEBUG_FUSED_ROUNDS (0 .. JD_F_ROUNDS): fewer blind rounds, so that a test can
    int fused_rounds_env = -1;                 // drive jd_batch_wait's extra rounds; unset: by the batch's size
    // pinned host mirror of the descriptor block, followed by the read-back words (status per image, 64 sync flags)
    uint8_t* h_desc = nullptr; size_t desc_bytes = 0, pre_stride = 0;
    uint32_t *h_status = nullptr, *h_changed = nullptr;
    bool own_pinned = false;
    hipEvent_t ev_desc = nullptr; bool desc_in_flight = false;
    size_t ecs_own_bytes = 0, rgb_own_bytes = 0, stream_bytes = 0, coef_elems = 0;
    uint32_t max_lut_bytes = 0, max_w_bytes = 0;
    uint32_t wg_k0 = 0, wg_iv = 0, wg_sync = 0, wg_wr = 0, wg_tile = 0;
    bool dirty = true, ran = false, any_uniform = false;
    uint32_t k2_modes = 0;                                       // bit 3m + f: some image has sampling mode m and coefficient form f (2 = staged)
    uint32_t coef_fmts = 0;                                      // bit 0: some image keeps int16 coefficients, bit 1: some use bytes
    int coef_form = 0;                                           // JD_COEF_FORM: 0 = choose per image, 8 / 16 = force
    hipStream_t stream = nullptr, last_stream = nullptr;
    bool own_stream = false;
    JdIdctConst K;
    // pinned-memory I/O pipeline (JD_BATCH_PINNED_IO): library-owned staging for the stuffed ECS of every image and for
    // the decoded RGB, copy streams and the events that chain H2D -> kernels -> D2H
    uint8_t* h_ecs = nullptr; uint8_t* h_rgb = nullptr;
    size_t staged_bytes = 0; int staged_count = 0; bool stage_open = false, io_pending = false;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_run = nullptr, ev_d2h = nullptr;
    // optional per-stage hipEvent timing of jd_batch_run (jd_batch_timing / jd_batch_stage_times)
    static const int kTimedRuns = 64;
    std::vector<hipEvent_t> events;           // kTimedRuns x (stages + 1)
    int timed_runs = 0;
    bool timing = false;
    int sync_fixups = 2;                      // fix-up launches enqueued blindly; jd_batch_wait adds more if the flag says so
    int sync_warm_off = 0;                    // debug knob (JD_DEBUG_SYNC_NO_WARM)
};

// internal jd_batch_create flag (not in jpeghip.h): every image keeps int16 coefficients -- the progressive scans refine
// coefficients in place
#define JD_BATCH_COEF16_ 0x100u

namespace {

JdKernelArgs kernel_args(const jd_batch* b) {
    JdKernelArgs a;
    a.imgs = b->d_img; a.n_images = b->n; a.prefix = nullptr; a.stream = b->d_stream; a.coef = b->d_coef;
    a.lut = b->d_lut; a.status = b->d_status; a.stream_len = b->d_stream_len; a.interval_start = b->d_interval;
    a.k0_state = b->d_k0; a.k0_ticket = b->d_ticket; a.sub = b->d_sub; a.subpfx = b->d_subpfx; a.subck = b->d_subck; a.sync_tot = b->d_sync_tot; a.changed = b->d_changed;
    a.loc = b->d_loc; a.chunk = b->d_chunk; a.guess = b->d_guess; a.wl = b->d_wl; a.falloc = b->d_falloc; a.fhalloc = b->d_fhalloc; a.fcount = b->d_fcount; a.fkmax = b->fkmax; a.fsub_bits = b->fsub_bits;
    return a;
}

// single-decode path: staging slots set aside for the sub-sequences' own runs (a quarter more than the blocks + 16 each)
uint64_t fused_own_budget(uint64_t nb, uint64_t nsub_cap) { return nb + nb / 4 + 16 * nsub_cap; }

// recompute the per-kernel workgroup prefixes from the current ECS lengths and push the descriptor block: host
// mirror in pinned memory, one asynchronous copy on the stream the kernels will run on
int refresh(jd_batch* b, hipStream_t s) {
    const int n = b->n;
    // the pinned mirror is about to be rewritten: the copy that last read it must have run
    if (b->desc_in_flight) {
        if (b->ev_desc) HIPCHK(hipEventSynchronize(b->ev_desc)); else HIPCHK(hipStreamSynchronize(b->last_stream));
        b->desc_in_flight = false;
    }
    // ... and a run still in flight on ANOTHER stream must not see its descriptors change under it
    if (b->ran && b->last_stream != s) HIPCHK(hipStreamSynchronize(b->last_stream));
    JdImg* himg = reinterpret_cast<JdImg*>(b->h_desc);
    uint32_t* pre[7];
    for (int k = 0; k < 7; k++) pre[k] = reinterpret_cast<uint32_t*>(b->h_desc + align_up((size_t)n * sizeof(JdImg), 256) + (size_t)k * b->pre_stride);
    for (int k = 0; k < 7; k++) pre[k][0] = 0;
    b->k2_modes = 0; b->fscan_tiled = 0;
    // Which no-DRI path this run takes.  The single-decode path is the THROUGHPUT path: it decodes every bit 1.6 times
    // instead of 2.6, but its critical path is longer (blind decode, writing decode and two or three re-decode rounds are
    // each one walk through a sub-sequence, ~200 us for a lone wave; the two-pass path has 2.5 such walks), so it only
    // pays once the writing decode fills the machine several times over.  Measured cross-over (MI355X, q85 images,
    // sub-sequences = stream bits / 4096): 1080p from ~200 images per batch (+17 % slower at 64 images = 77,000
    // sub-sequences, +9 % at 128, -10 % at 256 = 307,000 and beyond), 8192 x 8192 from ~20 (+41 % for one image = 58,000,
    // -2 % at 16 = 925,000, -7 % at 24, -11 % at 64), 4096 x 4096 from ~30 (+12 % at 16 = 231,000, -6 % at 48): the larger
    // the images the more stream it takes (their per-image steps -- lists, index -- are serial), hence the three-step rule.
    bool fz_now = b->fused;
    b->f_small_class = false;
    if (fz_now && !b->fused_always) {
        uint64_t subs = 0, nfz = 0, most = 0;
        bool one_set = true;                                       // every candidate image on the same packed tables
        uint32_t set_off = 0, set_len = 0;
        for (int i = 0; i < n; i++)
            if (b->fz_ok[i]) {
                const uint64_t k = ((uint64_t)b->img[i].ecs_len * 8 + JD_SUBSEQ_BITS - 1) / JD_SUBSEQ_BITS;
                subs += k; nfz++; most = std::max(most, k);
                if (!set_len) { set_off = b->img[i].w_off; set_len = b->img[i].w_len; }
                else if (b->img[i].w_off != set_off || b->img[i].w_len != set_len) one_set = false;
            }
        const uint64_t avg = nfz ? subs / nfz : 0;
        fz_now = nfz && subs >= (avg <= 4096 ? 250000u : avg <= 20000 ? 450000u : 800000u);
        // Many SMALL images on one table set (thumbnails; at most 256 sub-sequences each): the path's launches then run
        // eight waves per copy of the tables (f_share below), and it wins from ~600 images whatever the total -- 320x240:
        // 500 images +-0, 1000 +25 %, 3000 +65 % against the two-pass path; 640x360: 300 +-0, 600 +10 %, 1200 +13 %.
        b->f_small_class = nfz >= 600 && one_set && most <= 256 && env_int("JD_DEBUG_FUSED_SHARE", 1) != 0;
        if (b->f_small_class) fz_now = true;
        // ... and with long sub-sequences once the batch is larger still (8192 bits beat 4096 from ~900,000: 512 x 1080p
        // -13 % against -11 %, 16 x 8192^2 -7 % against -2 %; below, 48 x 4096^2 +1 % against -7 %; at C4 / C5 size -17 % against -11 %)
        b->fsub_bits = subs >= (uint64_t)env_int("JD_DEBUG_FUSED_LONG_FROM", 900000) ? JD_F_SUBSEQ_BITS_MAX : JD_SUBSEQ_BITS;
    } else b->fsub_bits = env_int("JD_DEBUG_FUSED_LONG_FROM", 1) == 0 ? JD_F_SUBSEQ_BITS_MAX : JD_SUBSEQ_BITS;
    for (int i = 0; i < n; i++) {
        JdImg& im = b->img[i];
        const bool stuffed = !(im.flags & JD_IMG_UNSTUFFED_INPUT);
        im.n_k0_chunks = stuffed ? (uint32_t)((im.ecs_len + JD_K0_CHUNK - 1) / JD_K0_CHUNK) : 0;
        const uint32_t nsub = (uint32_t)(((uint64_t)im.ecs_len * 8 + JD_SUBSEQ_BITS - 1) / JD_SUBSEQ_BITS);
        const uint32_t nsub_f = (uint32_t)(((uint64_t)im.ecs_len * 8 + b->fsub_bits - 1) / b->fsub_bits);   // single-decode path
        if (fz_now && b->fz_ok[i]) im.flags |= JD_IMG_FUSED; else im.flags &= ~(uint32_t)JD_IMG_FUSED;
        const bool fz = (im.flags & JD_IMG_FUSED) != 0;
        const uint32_t nsw = (im.restart_interval || fz) ? 0 : std::min(im.n_sync_wg, (nsub + JD_SYNC_OWN - 1) / JD_SYNC_OWN);
        b->k2_modes |= (fz ? 4u : (im.flags & JD_IMG_COEF8) ? 2u : 1u) << (3 * (im.ncomp == 1 ? 0 : im.hs + 2 * im.vs - 2));
        pre[0][i + 1] = pre[0][i] + im.n_k0_chunks;
        pre[1][i + 1] = pre[1][i] + (im.restart_interval ? (im.n_intervals + 64 * JD_W_WAVES - 1) / (64 * JD_W_WAVES) : 0);
        pre[2][i + 1] = pre[2][i] + nsw;
        pre[3][i + 1] = pre[3][i] + (nsw * JD_SYNC_WG + 64 * JD_W_WAVES - 1) / (64 * JD_W_WAVES);
        pre[4][i + 1] = pre[4][i] + im.n_tiles;
        const uint32_t fwg = fz ? std::min(im.n_fwg, (nsub_f + 64 * JD_W_WAVES - 1) / (64 * JD_W_WAVES)) : 0;
        if (fz && nsub_f > 8192) b->fscan_tiled = 1;
        pre[5][i + 1] = pre[5][i] + fwg;
        pre[6][i + 1] = pre[6][i] + (fz ? (nsub_f + JD_F_REDO_LANES * JD_F_REDO_DIV - 1) / (JD_F_REDO_LANES * JD_F_REDO_DIV) : 0);
        if (fz) {
            // slots every sub-sequence owns: a quarter more than the average it will hold (the stuffed length bounds the
            // number of sub-sequences from above); the overflow chunks start behind them
            const uint64_t nb = (uint64_t)im.n_mcus * im.bpm, ns = std::max<uint32_t>(nsub_f, 1);
            uint64_t ch0 = ((nb + nb / 4) / ns + 8 + 7) & ~(uint64_t)7;
            { const int dbg = env_int("JD_DEBUG_FUSED_CH0", 0); if (dbg > 0) ch0 = std::max<uint64_t>((uint64_t)dbg & ~(uint64_t)1, 4); }   // tests: small runs, many overflow chunks
            if (ch0 * ns > fused_own_budget(nb, im.n_subseq_cap)) ch0 = 8;   // (cannot happen: ns <= n_subseq_cap)
            im.f_ch0 = (uint32_t)ch0; im.f_ov_base = (uint32_t)(ch0 * ns);
        }
        himg[i] = im;
    }
    b->wg_k0 = pre[0][n]; b->wg_iv = pre[1][n]; b->wg_sync = pre[2][n]; b->wg_wr = pre[3][n]; b->wg_tile = pre[4][n]; b->wg_f = pre[5][n]; b->wg_fr = pre[6][n];
    // blind re-decode rounds: what a round leaves shrinks by an order of magnitude, so a small batch needs fewer (each is
    // two launches on the critical path; jd_batch_wait finishes what a last check still finds)
    {   // the first round's list holds about a twelfth of an image's sub-sequences
        uint64_t subs_f = 0, nfz2 = 0;
        for (int i = 0; i < n; i++) if (b->img[i].flags & JD_IMG_FUSED) { subs_f += ((uint64_t)b->img[i].ecs_len * 8 + b->fsub_bits - 1) / b->fsub_bits; nfz2++; }
        const uint64_t est = nfz2 ? subs_f / nfz2 / 12 : 0;
        b->redo0_lanes = est <= 48 ? 64u : est <= 110 ? 128u : JD_F_REDO_LANES;
        // Many small images on ONE table set (a batch of photographs of one camera setting): the re-decode launches
        // put eight waves behind one copy of the tables, a wave or two per image (k1f_write, share_w), instead of a
        // workgroup per image -- twice as many walks run at once.  Not with a long list in the batch: its image's one or
        // two waves would walk it tile after tile.
        uint32_t nsub_max = 0; bool one_set = nfz2 > 0; b->f_share_off = b->f_share_len = 0;
        for (int i = 0; i < n; i++) {
            const JdImg& im = b->img[i];
            if (!(im.flags & JD_IMG_FUSED)) continue;
            nsub_max = std::max<uint32_t>(nsub_max, (uint32_t)(((uint64_t)im.ecs_len * 8 + b->fsub_bits - 1) / b->fsub_bits));
            if (!b->f_share_len) { b->f_share_off = im.w_off; b->f_share_len = im.w_len; }
            else if (im.w_off != b->f_share_off || im.w_len != b->f_share_len) one_set = false;
        }
        // (bit 0: round 0, whose lists hold a twelfth of an image's sub-sequences; bit 1: the later rounds, a 150th and less)
        if (env_int("JD_DEBUG_FUSED_SHARE", 1) == 0) one_set = false;
        // Only where a workgroup per image would not fit the machine at once (eight one-wave workgroups per CU: 2048): with
        // fewer images the walks are better spread over all CUs (1080p images: 512 - 2 % with sharing, 1024 + 0.5 %, 2048 + 5 %, 5000 + 4 %).
        const bool many = nfz2 >= (uint64_t)env_int("JD_DEBUG_FUSED_SHARE_FROM", b->f_small_class ? 600 : 1200);
        b->f_share = (one_set && many && b->redo0_lanes <= 128u && nsub_max <= 2048u ? 1u : 0u) | (one_set && many && nsub_max <= 8192u ? 2u : 0u);
        // ... and the FIRST decode of a batch of many tiny images (thumbnails: every image a wave or a few): a 512-lane
        // workgroup per image holds 78 KB of LDS for its one busy wave -- two images per CU at a time instead of sixteen
        b->f_share_first = (one_set && many && nsub_max <= 256u) ? (nsub_max <= 64u ? 1u : nsub_max <= 128u ? 2u : 4u) : 0u;
    }
    if (b->fused_rounds_env < 0) b->fused_rounds = (uint64_t)b->wg_f * 64 * JD_W_WAVES < 200000 ? 3 : JD_F_ROUNDS;
    HIPCHK(hipMemcpyAsync(b->d_desc, b->h_desc, b->desc_bytes, hipMemcpyHostToDevice, s));
    if (b->ev_desc) HIPCHK(hipEventRecord(b->ev_desc, s));
    b->desc_in_flight = true; b->last_stream = s;
    b->dirty = false;
    return JD_OK;
}

enum Stage { ST_CLEAR = 0, ST_UNSTUFF, ST_HUFF_DRI, ST_HUFF_SYNC, ST_HUFF_WRITE, ST_IDCT_COLOUR, ST_COUNT_ };
const char* kStageNames[ST_COUNT_] = { "clear", "unstuff", "huffman_intervals", "huffman_sync", "huffman_write", "idct_colour" };

// enqueue one stage (used by run and by profile)
int enqueue_stage(jd_batch* b, int stage, hipStream_t s) {
    const JdKernelArgs a = kernel_args(b);
    switch (stage) {
    case ST_CLEAR:
        HIPCHK(hipMemsetAsync(b->d_status, 0, b->clear_bytes, s));   // status words + changed flags
        break;
    case ST_UNSTUFF:
        jdk_launch_k0(a, b->d_pre_k0, b->wg_k0, s);
        break;
    case ST_HUFF_DRI:
        jdk_launch_intervals(a, b->d_pre_iv, b->wg_iv, b->max_w_bytes, b->coef_fmts, s);
        break;
    case ST_HUFF_SYNC:
        if (b->wg_sync) {
            jdk_launch_sync(a, b->d_pre_sync, b->wg_sync, b->max_lut_bytes, 0, 0, b->sync_warm_off, s);
            for (int k = 1; k <= b->sync_fixups; k++)
                jdk_launch_sync(a, b->d_pre_sync, b->wg_sync, b->max_lut_bytes, 1, k, 0, s);
            jdk_launch_sync_spine(a, s);
        }
        jdk_launch_fcold(a, b->d_pre_f, b->wg_f, b->max_lut_bytes, s);
        break;
    case ST_HUFF_WRITE:
        jdk_launch_write_subseq(a, b->d_pre_wr, b->wg_wr, b->max_w_bytes, b->coef_fmts, s);
        if (b->wg_wr && b->any_uniform) jdk_launch_dc_scan(a, s);
        if (b->wg_f) {
            // the writing decode from the blind guesses, then JD_F_ROUNDS rounds of "list the sub-sequences whose entry is
            // not their predecessor's exit, decode those again" (the first redoes one in twelve, the rest next to nothing),
            // a last check for jd_batch_wait, the prefix scan and the block index
            jdk_launch_fwrite(a, b->d_pre_f, b->wg_f, b->max_w_bytes, -1, 0, b->f_share_first, b->f_share_off, b->f_share_len, s);
            for (int r = 0; r < b->fused_rounds; r++) {
                jdk_launch_fcheck(a, b->d_pre_f, b->wg_f, r, s);
                jdk_launch_fwrite(a, b->d_pre_fr, b->wg_fr, b->max_w_bytes, r, b->redo0_lanes, r == 0 ? ((b->f_share & 1u) ? b->redo0_lanes / 64u : 0u) : (b->f_share >> 1), b->f_share_off, b->f_share_len, s);
            }
            jdk_launch_fcheck(a, b->d_pre_f, b->wg_f, JD_F_ROUNDS, s);
            jdk_launch_fscan_index(a, b->d_pre_f, b->wg_f, b->fscan_tiled, s);
        }
        break;
    case ST_IDCT_COLOUR:
        jdk_launch_idct_colour(a, b->d_pre_tile, b->wg_tile, b->k2_modes, b->K, s);
        break;
    }
    HIPCHK(hipGetLastError());
    return JD_OK;
}

bool stage_is_empty(const jd_batch* b, int stage) {
    return (stage == ST_HUFF_DRI && b->wg_iv == 0) || (stage == ST_HUFF_SYNC && b->wg_sync == 0 && b->wg_f == 0) ||
           (stage == ST_HUFF_WRITE && b->wg_wr == 0 && b->wg_f == 0);
}

void batch_release(jd_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->own_stream && b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    for (hipStream_t s : { b->s_h2d, b->s_d2h }) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : { b->ev_desc, b->ev_h2d, b->ev_run, b->ev_d2h }) if (e) (void)hipEventDestroy(e);
    for (void* p : b->arena.owned) (void)hipFree(p);
    if (b->own_pinned && b->h_desc) (void)hipHostFree(b->h_desc);
    if (b->h_ecs) (void)hipHostFree(b->h_ecs);
    if (b->h_rgb) (void)hipHostFree(b->h_rgb);
    for (auto& e : b->events) (void)hipEventDestroy(e);
    delete b;
}

// Plans a batch and lays its workspace out.  dev == nullptr: a batch of its own (hipMalloc, pinned mirror, stream and
// events created here).  dev != nullptr: everything lives in that context's fixed device and pinned blocks, the tables
// go up asynchronously on `s`, and no allocating HIP call is made.
int batch_build(int device, int n_images, const jd_frame* frames, const size_t* ecs_cap, unsigned flags, DevCtx* dev,
                hipStream_t s, jd_batch** out) {
    *out = nullptr;
    for (int i = 0; i < n_images; i++) {
        int rc = jd_frame_check(&frames[i]);
        if (rc) return rc;
        if (ecs_cap[i] == 0 || ecs_cap[i] >= (size_t)1 << 28) return JD_E_ARG;
    }
    HIPCHK(hipSetDevice(device));
    jd_batch* b = new (std::nothrow) jd_batch();
    if (!b) return JD_E_NOMEM;
    b->device = device; b->n = n_images; b->flags = flags; b->K = idct_constants();
    b->frames.assign(frames, frames + n_images);
    b->img.resize(n_images); b->ecs_cap.assign(ecs_cap, ecs_cap + n_images);
    b->own_ecs_off.resize(n_images); b->own_rgb_off.resize(n_images); b->rgb_bound.assign(n_images, 0);
    b->status.assign(n_images, 0);
    b->sync_fixups = std::min(std::max(env_int("JD_SYNC_FIXUPS", 2), 0), 60);
    b->sync_warm_off = env_int("JD_DEBUG_SYNC_NO_WARM", 0) ? 1 : 0;
    b->coef_form = env_int("JD_COEF_FORM", 0);                    // 8 / 16: force the coefficient form (tests); else per image
    if (b->coef_form != 8 && b->coef_form != 16) b->coef_form = 0;
    if (dev) { b->arena.base = dev->scratch; b->arena.cap = dev->scratch_bytes; }
    b->arena.poison = env_int("JD_DEBUG_POISON", -1);
    if (b->arena.poison > 255) b->arena.poison = -1;
    b->arena.poison_only = env_int("JD_DEBUG_POISON_ONLY", -1);
    // the single-decode path needs staging beyond the fixed block of jd_decode_resident: batches of their own only
    b->fused = !dev && env_int("JD_NODRI_FUSED", 1) != 0;
    b->fused_always = env_int("JD_NODRI_FUSED", 1) == 2;
    b->fz_ok.assign(n_images, 0);
    b->fused_rounds_env = env_int("JD_DEBUG_FUSED_ROUNDS", -1);
    if (b->fused_rounds_env >= 0) b->fused_rounds = std::min(b->fused_rounds_env, JD_F_ROUNDS);

    // ---- Huffman table sets, de-duplicated across the batch -----------------------------------------
    std::vector<uint16_t> lut;
    std::map<std::string, std::pair<uint32_t, uint32_t>> seen;   // serialised tables -> (offset, length)
    size_t ecs_own = 0, rgb_own = 0, stream = 0, coef = 0;
    uint32_t k0c = 0, ivb = 0, subb = 0, swb = 0;
    size_t locn = 0;
    for (int i = 0; i < n_images; i++) {
        const jd_frame& f = frames[i];
        JdImg& im = b->img[i];
        memset(&im, 0, sizeof im);
        Geometry g; frame_geometry(&f, &g);
        im.width = f.width; im.height = f.height; im.mcux = g.mcux; im.mcuy = g.mcuy; im.n_mcus = g.mcux * g.mcuy;
        im.hs = g.hs; im.vs = g.vs; im.ncomp = f.ncomp; im.bpm = g.bpm;
        im.restart_interval = f.restart_interval;
        im.n_intervals = f.restart_interval ? (im.n_mcus + f.restart_interval - 1) / f.restart_interval : 1;
        // table set: per component (td, ta); identical (class,id) pairs share one packed table
        std::string key;
        std::vector<uint16_t> set, sset;                          // single-symbol tables / sync-pass twin (same layout)
        std::vector<uint32_t> wset;                               // writer tables (32-bit entries)
        std::map<int, uint32_t> placed, wplaced, splaced;         // class*4+id -> entry offset in set / wset / sset
        // All components on one table pair (legal, rare): the entropy decoder's path through the stream does not depend
        // on the block-in-MCU phase, so that phase never self-synchronises.  The no-DRI path then runs its sync pass
        // without the phase and predicts DC in a pass of its own (k1_dc_scan).
        bool uniform = f.ncomp > 1;
        for (int c = 0; c < f.ncomp; c++)
            if (f.comp[c].td != f.comp[0].td || f.comp[c].ta != f.comp[0].ta) uniform = false;
        if (uniform && f.restart_interval == 0) { im.flags |= JD_IMG_UNIFORM_TABLES; b->any_uniform = true; }
        // Coefficient form.  Bytes (low plane + marked wide blocks in the high plane, write_segments8) halve the traffic
        // between the Huffman writer and K2 and let the writer finish a block in one loop, but a wide block (an AC
        // coefficient outside int8, |AC 63| > 7) costs its wave extra work in both kernels.  An AC coefficient needs
        // |DCT| >= 128 * q to leave int8.  Measured on C3-shaped images (MI355X, Annex-K tables scaled to a quality): with
        // every AC quantiser >= 2 (quality <= 92: no or very few wide blocks) bytes win by 10-15 %; with quantisers of 1
        // (quality 95: 0.7 % of the blocks wide; 98: 2 %) int16 wins by 3 % / 20 %.  Uniform-table images keep int16
        // (k1_dc_scan rewrites DC in place), and so do progressive files (JD_BATCH_COEF16_: scans refine in place).
        bool bytes = !(im.flags & JD_IMG_UNIFORM_TABLES) && !(flags & JD_BATCH_COEF16_);
        if (bytes && b->coef_form == 0) {
            for (int c = 0; c < f.ncomp && bytes; c++)
                for (int k = 1; k < 64; k++) if (f.qt[f.comp[c].tq][k] < 2) { bytes = false; break; }
        } else if (b->coef_form == 16) bytes = false;
        if (bytes) im.flags |= JD_IMG_COEF8;
        const bool fz = b->fused && bytes && f.restart_interval == 0;
        if (fz) { im.flags |= JD_IMG_FUSED; b->fz_ok[i] = 1; }
        b->coef_fmts |= bytes ? 2u : 1u;
        b->k2_modes |= (fz ? 4u : bytes ? 2u : 1u) << (3 * (f.ncomp == 1 ? 0 : g.hs + 2 * g.vs - 2));
        const unsigned unit = bytes ? 1u : 2u;
        key.push_back((char)unit);
        for (int c = 0; c < f.ncomp; c++) {
            for (int cls = 0; cls < 2; cls++) {
                const int id = cls ? f.comp[c].ta : f.comp[c].td;
                const jd_huff& h = f.huff[cls][id];
                auto it = placed.find(cls * 4 + id);
                if (it == placed.end()) {
                    const uint32_t off = (uint32_t)set.size();
                    if (jd_pack_lut(h.bits, h.vals, h.nvals, cls == 0, set)) { batch_release(b); return JD_E_TABLE; }
                    splaced.emplace(cls * 4 + id, (uint32_t)sset.size());
                    if (cls == 0) jd_pack_lut_bits(h.bits, h.vals, h.nvals, true, JD_SLUT_BITS, sset);
                    else jd_pack_lut_multi(h.bits, h.vals, h.nvals, JD_SLUT_BITS, sset);
                    it = placed.emplace(cls * 4 + id, off).first;
                    wplaced.emplace(cls * 4 + id, (uint32_t)wset.size());
                    if (bytes && cls ? jd_pack_lut32p(h.bits, h.vals, h.nvals, JD_W32_AC_BITS_WIDE, wset)
                                     : jd_pack_lut32(h.bits, h.vals, h.nvals, cls == 0, cls ? JD_W32_AC_BITS_WIDE : JD_W32_DC_BITS, unit, wset)) { batch_release(b); return JD_E_TABLE; }
                    key.append((const char*)h.bits, 16); key.append((const char*)h.vals, h.nvals); key.push_back((char)(cls * 4 + id));
                }
                (cls ? im.ac_lut : im.dc_lut)[c] = it->second;
                (cls ? im.wac : im.wdc)[c] = wplaced[cls * 4 + id];
                (cls ? im.sac : im.sdc)[c] = splaced[cls * 4 + id];
            }
            key.push_back((char)(0x40 | (f.comp[c].td << 2) | f.comp[c].ta));
            memcpy(im.q[c], f.qt[f.comp[c].tq], 64);
        }
        while (set.size() % 8) set.push_back(0);
        while (sset.size() % 8) sset.push_back(0);
        im.w_ac_bits = JD_W32_AC_BITS_WIDE;
        if (wset.size() * 4 > kWriterLutTwoPerCu) {               // too big for two writer workgroups per CU: narrow AC tables
            wset.clear(); wplaced.clear();
            for (int c = 0; c < f.ncomp; c++)
                for (int cls = 0; cls < 2; cls++) {
                    const int id = cls ? f.comp[c].ta : f.comp[c].td;
                    if (!wplaced.count(cls * 4 + id)) {
                        wplaced.emplace(cls * 4 + id, (uint32_t)wset.size());
                        const jd_huff& h = f.huff[cls][id];
                        if (bytes && cls ? jd_pack_lut32p(h.bits, h.vals, h.nvals, JD_W32_AC_BITS_NARROW, wset)
                                         : jd_pack_lut32(h.bits, h.vals, h.nvals, cls == 0, cls ? JD_W32_AC_BITS_NARROW : JD_W32_DC_BITS, unit, wset)) { batch_release(b); return JD_E_TABLE; }
                    }
                    (cls ? im.wac : im.wdc)[c] = wplaced[cls * 4 + id];
                }
            im.w_ac_bits = JD_W32_AC_BITS_NARROW;
        }
        while (wset.size() % 4) wset.push_back(0);
        auto sit = seen.find(key);
        if (sit == seen.end()) {
            sit = seen.emplace(key, std::make_pair((uint32_t)lut.size(), (uint32_t)set.size())).first;
            lut.insert(lut.end(), set.begin(), set.end());
            lut.insert(lut.end(), sset.begin(), sset.end());
            const uint16_t* w16 = reinterpret_cast<const uint16_t*>(wset.data());
            lut.insert(lut.end(), w16, w16 + wset.size() * 2);
        }
        im.lut_off = sit->second.first; im.lut_len = sit->second.second;
        im.s_len = (uint32_t)sset.size();
        im.w_off = im.lut_off + im.lut_len + im.s_len; im.w_len = (uint32_t)wset.size();
        b->max_lut_bytes = std::max<uint32_t>(b->max_lut_bytes, im.s_len * 2u);
        b->max_w_bytes = std::max<uint32_t>(b->max_w_bytes, im.w_len * 4u);
        // workspace slices
        b->own_ecs_off[i] = ecs_own; ecs_own += align_up(ecs_cap[i] + 16, 256);
        b->own_rgb_off[i] = rgb_own; rgb_own += align_up((size_t)f.width * 3 * f.height, 256);
        im.stream_off = stream; stream += align_up(JD_STREAM_FRONT + ecs_cap[i] + JD_STREAM_TAIL, 256);
        im.n_subseq_cap = (uint32_t)(((uint64_t)ecs_cap[i] * 8 + JD_SUBSEQ_BITS - 1) / JD_SUBSEQ_BITS);
        im.coef_off = coef;
        if (fz) {
            // staging: the sub-sequences' own runs + overflow chunks for as many blocks again, a part-filled one for every
            // second sub-sequence (if that is ever not enough the batch falls back to the two-pass path, JD_ST_FOVF)
            const uint64_t nb = (uint64_t)im.n_mcus * im.bpm;
            uint64_t cap = fused_own_budget(nb, im.n_subseq_cap) + nb + (uint64_t)(JD_F_CHUNK / 2) * im.n_subseq_cap + 64 * JD_F_CHUNK;
            const int pct = env_int("JD_DEBUG_FUSED_CAP_PERCENT", 0);   // tests: staging of pct % of the blocks, to see the fallback
            if (pct > 0) cap = fused_own_budget(nb, im.n_subseq_cap) + std::max<uint64_t>(nb * (uint64_t)pct / 100, 4 * JD_F_CHUNK);
            if (cap >= (1ull << 31)) { batch_release(b); return JD_E_TOO_LARGE; }
            im.slot_cap = (uint32_t)cap;
            im.loc_off = locn; locn += (size_t)nb;
            im.n_fwg = (im.n_subseq_cap + 64 * JD_W_WAVES - 1) / (64 * JD_W_WAVES);
            im.hi_cap = (uint32_t)(nb / 8 + 256);
            // (never less than the two-pass layout needs: the fallback uses the region)
            const uint64_t bytes_f = JD_F_HIPOOL_OFF(cap) + (uint64_t)im.hi_cap * 64, bytes_c = nb * 128;
            coef += (size_t)((std::max(bytes_f, bytes_c) + 255) & ~(uint64_t)255) / 2;
            // chunk-table entries a sub-sequence can need: its shortest possible block is the shortest DC code plus the
            // shortest AC code (an EOB, or whatever the table offers)
            uint32_t minbits = 32;
            for (int c = 0; c < f.ncomp; c++) {
                uint32_t ld = 16, la = 16;
                for (uint32_t l = 16; l >= 1; l--) { if (f.huff[0][f.comp[c].td].bits[l - 1]) ld = l; if (f.huff[1][f.comp[c].ta].bits[l - 1]) la = l; }
                minbits = std::min(minbits, ld + la);
            }
            const uint32_t kmax = (JD_F_SUBSEQ_BITS_MAX / minbits + 1 + JD_F_CHUNK - 3) / (JD_F_CHUNK - 2) + 1;
            b->fkmax = std::max(b->fkmax, std::min<uint32_t>(kmax, JD_F_KMAX_LIMIT));
        } else coef += (size_t)im.n_mcus * im.bpm * 64;
        k0c += (uint32_t)((ecs_cap[i] + JD_K0_CHUNK - 1) / JD_K0_CHUNK);
        im.interval_base = ivb; ivb += im.n_intervals + 1;
        im.n_sync_wg = f.restart_interval ? 0 : (im.n_subseq_cap + JD_SYNC_OWN - 1) / JD_SYNC_OWN;
        im.subseq_base = subb; subb += f.restart_interval ? 0 : im.n_sync_wg * JD_SYNC_OWN;
        im.sync_wg_base = swb; swb += im.n_sync_wg;
        im.n_tiles = (im.n_mcus + JD_TILE_MCUS - 1) / JD_TILE_MCUS;
        im.row_stride = (uint32_t)f.width * 3;
        im.ecs_len = 0;
    }
    if (b->max_w_bytes > 160 * 1024 - JD_W_WAVES * JD_WAVE_LDS - 1024 || b->max_lut_bytes > 44 * 1024 /* sync pass: 64 KiB of LDS less its rings */) { batch_release(b); return JD_E_TABLE; }
    b->ecs_own_bytes = ecs_own; b->rgb_own_bytes = rgb_own; b->stream_bytes = stream; b->coef_elems = coef;

    // ---- descriptor block: JdImg[n], then five prefixes of n + 1 words; read-back words follow in the pinned mirror ----
    const size_t img_bytes = align_up((size_t)n_images * sizeof(JdImg), 256);
    b->pre_stride = align_up((size_t)(n_images + 1) * sizeof(uint32_t), 256);
    b->desc_bytes = img_bytes + 7 * b->pre_stride;
    const size_t lut_bytes = lut.size() * sizeof(uint16_t);
    const size_t pinned_need = b->desc_bytes + align_up((size_t)n_images * 4, 256) + 256 + (dev ? align_up(lut_bytes, 256) : 0);

    int rc = JD_OK;
    auto A = [&](void* pp, size_t bytes) { if (rc == JD_OK) rc = b->arena.take((void**)pp, bytes); };
    A(&b->d_desc, b->desc_bytes);
    if (!dev && !(flags & JD_BATCH_EXTERNAL_IO)) {                // (a fixed-block batch reads and writes the caller's buffers only)
        A(&b->d_ecs_own, ecs_own);
        A(&b->d_rgb_own, rgb_own);
    }
    A(&b->d_stream, stream);
    A(&b->d_coef, coef * sizeof(int16_t));
    A(&b->d_lut, lut_bytes);
    // status words and the sync pass's "changed" flags sit in one block: one clear per run instead of two
    // (and the single-decode path's chunk counters and list lengths: cleared by the same memset)
    const size_t st_words = align_up((size_t)n_images * 4, 256) / 4, fw = b->fkmax ? st_words : 0;
    b->clear_bytes = (st_words + 64 + fw * (3 + JD_F_ROUNDS)) * 4;
    A(&b->d_status, b->clear_bytes);
    if (rc == JD_OK) {
        b->d_changed = b->d_status + st_words;
        if (b->fkmax) { b->d_falloc = b->d_changed + 64; b->d_fhalloc = b->d_falloc + fw; b->d_fcount = b->d_fhalloc + fw; }
    }
    if (b->fkmax) {
        A(&b->d_loc, std::max<size_t>(locn, 1) * sizeof(uint2));
        A(&b->d_chunk, (size_t)subb * b->fkmax * 4);
        A(&b->d_guess, (size_t)subb * 4);
        A(&b->d_wl, (size_t)subb * 4);
    }
    A(&b->d_stream_len, (size_t)n_images * 4);
    A(&b->d_interval, (size_t)ivb * 4);
    A(&b->d_k0, ((size_t)k0c + 32) * 8);
    A(&b->d_ticket, ((size_t)n_images + 64) * 4);
    A(&b->d_sub, (size_t)subb * sizeof(JdSub));
    A(&b->d_subpfx, (size_t)subb * sizeof(JdSubPfx));
    A(&b->d_subck, (size_t)subb * sizeof(JdSubCk));
    A(&b->d_sync_tot, (size_t)swb * sizeof(uint4));
    if (rc == JD_OK) {
        b->d_img = reinterpret_cast<JdImg*>(b->d_desc);
        uint32_t** pre[7] = { &b->d_pre_k0, &b->d_pre_iv, &b->d_pre_sync, &b->d_pre_wr, &b->d_pre_tile, &b->d_pre_f, &b->d_pre_fr };
        for (int k = 0; k < 7; k++) *pre[k] = reinterpret_cast<uint32_t*>(b->d_desc + img_bytes + (size_t)k * b->pre_stride);
    }
    if (rc == JD_OK) {
        if (dev) {
            if (pinned_need > dev->pinned_bytes) rc = JD_E_TOO_LARGE;
            else { b->h_desc = dev->pinned; b->stream = s; }
        } else {
            if (hipHostMalloc((void**)&b->h_desc, pinned_need, hipHostMallocDefault) != hipSuccess) rc = JD_E_NOMEM;
            else b->own_pinned = true;
            if (rc == JD_OK && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) rc = JD_E_HIP;
            else b->own_stream = true;
            if (rc == JD_OK && hipEventCreateWithFlags(&b->ev_desc, hipEventDisableTiming) != hipSuccess) rc = JD_E_HIP;
        }
    }
    if (rc == JD_OK) {
        b->h_status = reinterpret_cast<uint32_t*>(b->h_desc + b->desc_bytes);
        b->h_changed = reinterpret_cast<uint32_t*>(b->h_desc + b->desc_bytes + align_up((size_t)n_images * 4, 256));
        memset(b->h_status, 0, (size_t)n_images * 4); memset(b->h_changed, 0, 256);
    }
    // ---- tables up, stream pads cleared.  Everything here is ordered on the batch's stream: the clear of the unstuffed-
    // stream buffer used to be a hipMemset on the legacy stream, which the batch's non-blocking stream does not wait
    // for, so a jd_batch_upload_unstuffed issued right after create could be overwritten by it (round 1's stale uploads).
    if (rc == JD_OK) {
        hipError_t he;
        if (dev) {
            uint8_t* hl = b->h_desc + pinned_need - align_up(lut_bytes, 256);
            memcpy(hl, lut.data(), lut_bytes);
            he = hipMemcpyAsync(b->d_lut, hl, lut_bytes, hipMemcpyHostToDevice, b->stream);
        } else {
            he = hipMemcpyAsync(b->d_lut, lut.data(), lut_bytes, hipMemcpyHostToDevice, b->stream);
        }
        if (he == hipSuccess) he = hipMemsetAsync(b->d_stream, 0, std::max<size_t>(stream, 256), b->stream);
        if (he == hipSuccess && !dev) he = hipStreamSynchronize(b->stream);   // `lut` is pageable and goes out of scope
        if (he != hipSuccess) { g_last_hip = (int)he; rc = JD_E_HIP; }
    }
    // ---- pinned I/O staging + copy streams -------------------------------------------------------------
    if (rc == JD_OK && (flags & JD_BATCH_PINNED_IO)) {
        if (dev || (flags & JD_BATCH_EXTERNAL_IO)) rc = JD_E_ARG;
        else {
            hipError_t he = hipHostMalloc((void**)&b->h_ecs, std::max<size_t>(ecs_own, 256), hipHostMallocDefault);
            if (he == hipSuccess) he = hipHostMalloc((void**)&b->h_rgb, std::max<size_t>(rgb_own, 256), hipHostMallocDefault);
            if (he == hipSuccess) he = hipStreamCreateWithFlags(&b->s_h2d, hipStreamNonBlocking);
            if (he == hipSuccess) he = hipStreamCreateWithFlags(&b->s_d2h, hipStreamNonBlocking);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&b->ev_h2d, hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&b->ev_run, hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&b->ev_d2h, hipEventDisableTiming);
            if (he != hipSuccess) { g_last_hip = (int)he; rc = he == hipErrorOutOfMemory ? JD_E_NOMEM : JD_E_HIP; }
        }
    }
    if (rc != JD_OK) {
        const bool oom = rc == JD_E_HIP && g_last_hip == (int)hipErrorOutOfMemory;
        batch_release(b);
        return oom ? JD_E_NOMEM : rc;
    }
    for (int i = 0; i < n_images; i++) {
        b->img[i].ecs = b->d_ecs_own ? b->d_ecs_own + b->own_ecs_off[i] : nullptr;
        b->img[i].rgb = b->d_rgb_own ? b->d_rgb_own + b->own_rgb_off[i] : nullptr;
    }
    *out = b;
    return JD_OK;
}

int run_on(jd_batch* b, hipStream_t s) {
    for (int i = 0; i < b->n; i++) if (b->img[i].ecs_len == 0 || !b->img[i].rgb) return JD_E_STATE;
    if (b->dirty) { int rc = refresh(b, s); if (rc) return rc; }
    else if (b->ran && b->last_stream != s) HIPCHK(hipStreamSynchronize(b->last_stream));   // one run at a time per workspace
    const bool timed = b->timing && b->timed_runs < jd_batch::kTimedRuns;
    hipEvent_t* ev = timed ? &b->events[(size_t)b->timed_runs * (ST_COUNT_ + 1)] : nullptr;
    if (timed) HIPCHK(hipEventRecord(ev[0], s));
    for (int st = 0; st < ST_COUNT_; st++) {
        int rc = enqueue_stage(b, st, s); if (rc) return rc;
        // (a stage with nothing to launch gets no event: an event costs a few microseconds of stream time)
        if (timed && !stage_is_empty(b, st)) HIPCHK(hipEventRecord(ev[st + 1], s));
    }
    if (timed) b->timed_runs++;
    b->last_stream = s; b->ran = true;
    return JD_OK;
}

int enqueue_rgb_download(jd_batch* b) {          // pipeline form: whole RGB staging in one copy behind the kernels
    HIPCHK(hipEventRecord(b->ev_run, b->last_stream));
    HIPCHK(hipStreamWaitEvent(b->s_d2h, b->ev_run, 0));
    HIPCHK(hipMemcpyAsync(b->h_rgb, b->d_rgb_own, b->rgb_own_bytes, hipMemcpyDeviceToHost, b->s_d2h));
    HIPCHK(hipEventRecord(b->ev_d2h, b->s_d2h));
    return JD_OK;
}

// Blocks until the last run is complete and returns the first per-image error.  The self-synchronising pass carries
// states across its workgroup boundaries with `sync_fixups` blind fix-up launches; if the last of them still moved a
// boundary state (never seen with JD_SYNC_WARM warm-up sub-sequences; forced by tests/test_gpu_sync.py), the host keeps
// launching fix-ups until nothing moves and re-runs the stages that consumed the result.
int wait_on(jd_batch* b) {
    hipStream_t s = b->last_stream;
    if (b->io_pending) HIPCHK(hipEventSynchronize(b->ev_d2h));
    HIPCHK(hipMemcpyAsync(b->h_changed, b->d_changed, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(b->h_status, b->d_status, (size_t)b->n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    b->desc_in_flight = false;
    // The single-decode path ran out of staging (speculative decodes of a hostile stream): the batch goes through the
    // two-pass path, from the top, and stays there.
    auto staging_overflowed = [&]() {
        bool ovf = false;
        for (int i = 0; i < b->n; i++) ovf = ovf || (b->h_status[i] & JD_ST_FOVF) != 0;
        return ovf;
    };
    auto two_pass_from_the_top = [&]() -> int {
        b->fused = false; b->dirty = true;
        int rc = run_on(b, s); if (rc) return rc;
        if (b->io_pending) { rc = enqueue_rgb_download(b); if (rc) return rc; }
        return wait_on(b);
    };
    if (b->wg_f && staging_overflowed()) return two_pass_from_the_top();
    const bool more = b->wg_sync && (b->sync_fixups == 0 || b->h_changed[b->sync_fixups]);
    const bool more_f = b->wg_f && b->h_changed[32];
    if (more || more_f) {
        const JdKernelArgs a = kernel_args(b);
        int guard = 0;
        if (more) do {
            HIPCHK(hipMemsetAsync(b->d_changed, 0, 64 * sizeof(uint32_t), s));
            jdk_launch_sync(a, b->d_pre_sync, b->wg_sync, b->max_lut_bytes, 1, 1, 0, s);
            HIPCHK(hipMemcpyAsync(b->h_changed, b->d_changed, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        } while (b->h_changed[1] && ++guard < 100000);
        // single-decode path: the last check left a list; decode it, check again, until a check finds nothing
        guard = 0;
        if (more_f) do {
            jdk_launch_fwrite(a, b->d_pre_fr, b->wg_fr, b->max_w_bytes, JD_F_ROUNDS, 64, b->f_share >> 1, b->f_share_off, b->f_share_len, s);
            HIPCHK(hipMemsetAsync(b->d_changed, 0, 64 * sizeof(uint32_t), s));
            HIPCHK(hipMemsetAsync(b->d_fcount + (size_t)JD_F_ROUNDS * b->n, 0, (size_t)b->n * sizeof(uint32_t), s));
            jdk_launch_fcheck(a, b->d_pre_f, b->wg_f, JD_F_ROUNDS, s);
            HIPCHK(hipMemcpyAsync(b->h_changed, b->d_changed, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(b->h_status, b->d_status, (size_t)b->n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (staging_overflowed()) return two_pass_from_the_top();   // (these rounds take staging too)
        } while (b->h_changed[32] && ++guard < 100000);
        // status bits and everything downstream of the sync records are recomputed
        HIPCHK(hipMemsetAsync(b->d_status, 0, (size_t)b->n * sizeof(uint32_t), s));
        jdk_launch_k0(a, b->d_pre_k0, b->wg_k0, s);             // restores K0's status bits (idempotent)
        jdk_launch_intervals(a, b->d_pre_iv, b->wg_iv, b->max_w_bytes, b->coef_fmts, s);
        if (b->wg_sync) {
            // (the spine turns the sync workgroups' totals into prefixes IN PLACE: it may only follow a sync launch, which
            // writes the totals.  When only the single-decode images needed more rounds, none has run since the last spine.)
            if (!more) jdk_launch_sync(a, b->d_pre_sync, b->wg_sync, b->max_lut_bytes, 1, 1, 0, s);
            jdk_launch_sync_spine(a, s);
            jdk_launch_write_subseq(a, b->d_pre_wr, b->wg_wr, b->max_w_bytes, b->coef_fmts, s);
            if (b->wg_wr && b->any_uniform) jdk_launch_dc_scan(a, s);
        }
        jdk_launch_fscan_index(a, b->d_pre_f, b->wg_f, b->fscan_tiled, s);
        int rc = enqueue_stage(b, ST_IDCT_COLOUR, s); if (rc) return rc;
        if (b->io_pending) { rc = enqueue_rgb_download(b); if (rc) return rc; }
        HIPCHK(hipMemcpyAsync(b->h_status, b->d_status, (size_t)b->n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (b->io_pending) HIPCHK(hipEventSynchronize(b->ev_d2h));
    }
    b->io_pending = false;
    for (int i = 0; i < b->n; i++) b->status[i] = b->h_status[i];
    for (int i = 0; i < b->n; i++) { int e = status_to_error(b->status[i]); if (e) return e; }
    return JD_OK;
}

}  // namespace

extern "C" int jd_batch_create(int device, int n_images, const jd_frame* frames, const size_t* ecs_cap,
                               unsigned flags, jd_batch** out) {
    if (!out) return JD_E_ARG;
    *out = nullptr;
    if (!frames || !ecs_cap || n_images <= 0) return JD_E_ARG;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_init) return JD_E_NOT_INIT;
    }
    if (!find_ctx(device)) return JD_E_ARG;
    return batch_build(device, n_images, frames, ecs_cap, flags, nullptr, nullptr, out);
}

extern "C" void jd_batch_destroy(jd_batch* b) { batch_release(b); }

static int check_image(const jd_batch* b, int image) { return (!b || image < 0 || image >= b->n) ? JD_E_ARG : JD_OK; }

// Host-memory uploads: the source may be pageable, and the caller's buffer is free again on return (cgo rule), so these
// enqueue on the batch's own stream and wait for it.
extern "C" int jd_batch_upload(jd_batch* b, int image, const uint8_t* ecs, size_t ecs_len) {
    if (check_image(b, image) || !ecs) return JD_E_ARG;
    if (ecs_len == 0 || ecs_len > b->ecs_cap[image]) return JD_E_ARG;
    if (!b->d_ecs_own) return JD_E_STATE;
    HIPCHK(hipSetDevice(b->device));
    uint8_t* dst = b->d_ecs_own + b->own_ecs_off[image];
    HIPCHK(hipMemcpyAsync(dst, ecs, ecs_len, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    JdImg& im = b->img[image];
    im.ecs = dst; im.ecs_len = (uint32_t)ecs_len; im.flags &= ~JD_IMG_UNSTUFFED_INPUT;
    b->dirty = true;
    return JD_OK;
}

extern "C" int jd_batch_upload_unstuffed(jd_batch* b, int image, const uint8_t* bs, size_t len,
                                         const uint32_t* rst_offsets, size_t n_rst) {
    if (check_image(b, image) || !bs) return JD_E_ARG;
    if (len == 0 || len > b->ecs_cap[image]) return JD_E_ARG;
    JdImg& im = b->img[image];
    if (n_rst != (size_t)im.n_intervals - 1) return JD_E_RESTART;
    if (n_rst && !rst_offsets) return JD_E_ARG;
    std::vector<uint32_t> iv(im.n_intervals + 1);
    iv[0] = 0; iv[im.n_intervals] = (uint32_t)len;
    for (size_t k = 0; k < n_rst; k++) {
        if (rst_offsets[k] > len || (k && rst_offsets[k] < rst_offsets[k - 1])) return JD_E_RESTART;
        iv[k + 1] = rst_offsets[k];
    }
    HIPCHK(hipSetDevice(b->device));
    const uint32_t l32 = (uint32_t)len;
    HIPCHK(hipMemcpyAsync(b->d_stream + im.stream_off + JD_STREAM_FRONT, bs, len, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_interval + im.interval_base, iv.data(), iv.size() * 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_stream_len + image, &l32, 4, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    im.ecs_len = l32; im.flags |= JD_IMG_UNSTUFFED_INPUT;
    b->dirty = true;
    return JD_OK;
}

extern "C" int jd_batch_bind_ecs(jd_batch* b, int image, const void* d_ecs, size_t ecs_len) {
    if (check_image(b, image) || !d_ecs) return JD_E_ARG;
    if (ecs_len == 0 || ecs_len > b->ecs_cap[image] || ((uintptr_t)d_ecs & 15)) return JD_E_ARG;
    JdImg& im = b->img[image];
    im.ecs = (const uint8_t*)d_ecs; im.ecs_len = (uint32_t)ecs_len; im.flags &= ~JD_IMG_UNSTUFFED_INPUT;
    b->dirty = true;
    return JD_OK;
}

extern "C" int jd_batch_bind_rgb(jd_batch* b, int image, void* d_rgb, size_t row_stride) {
    if (check_image(b, image) || !d_rgb) return JD_E_ARG;
    JdImg& im = b->img[image];
    if (row_stride < (size_t)im.width * 3 || row_stride > 0xFFFFFFFFu) return JD_E_ARG;
    im.rgb = (uint8_t*)d_rgb; im.row_stride = (uint32_t)row_stride; b->rgb_bound[image] = 1;
    b->dirty = true;
    return JD_OK;
}

extern "C" int jd_batch_run(jd_batch* b, void* hip_stream) {
    if (!b) return JD_E_ARG;
    HIPCHK(hipSetDevice(b->device));
    return run_on(b, hip_stream ? (hipStream_t)hip_stream : b->stream);
}

extern "C" int jd_batch_timing(jd_batch* b, int enable) {
    if (!b) return JD_E_ARG;
    HIPCHK(hipSetDevice(b->device));
    if (enable && b->events.empty()) {
        b->events.resize((size_t)jd_batch::kTimedRuns * (ST_COUNT_ + 1));
        for (auto& e : b->events) HIPCHK(hipEventCreate(&e));
    }
    b->timing = enable != 0; b->timed_runs = 0;
    return JD_OK;
}

extern "C" int jd_batch_stage_times(jd_batch* b, float* stage_us, const char** names, int cap) {
    if (!b || !stage_us || cap < ST_COUNT_) return JD_E_ARG;
    if (b->timed_runs == 0) return JD_E_STATE;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->last_stream));
    std::vector<double> acc(ST_COUNT_, 0.0);
    for (int r = 0; r < b->timed_runs; r++) {
        hipEvent_t* ev = &b->events[(size_t)r * (ST_COUNT_ + 1)];
        hipEvent_t last = ev[0];
        for (int st = 0; st < ST_COUNT_; st++) {
            if (stage_is_empty(b, st)) continue;                 // no event was recorded for it: 0
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, last, ev[st + 1])); acc[st] += ms * 1000.0;
            last = ev[st + 1];
        }
    }
    for (int st = 0; st < ST_COUNT_; st++) { stage_us[st] = (float)(acc[st] / b->timed_runs); if (names) names[st] = kStageNames[st]; }
    const int n = b->timed_runs; b->timed_runs = 0;
    return n;
}

extern "C" int jd_batch_wait(jd_batch* b) {
    if (!b) return JD_E_ARG;
    if (!b->ran) return JD_E_STATE;
    HIPCHK(hipSetDevice(b->device));
    return wait_on(b);
}

extern "C" int jd_batch_status(const jd_batch* b, int image) {
    if (check_image(b, image)) return JD_E_ARG;
    return status_to_error(b->status[image]);
}

extern "C" int jd_batch_download(jd_batch* b, int image, uint8_t* rgb_out, size_t row_stride) {
    if (check_image(b, image) || !rgb_out) return JD_E_ARG;
    const JdImg& im = b->img[image];
    if (row_stride < (size_t)im.width * 3 || !im.rgb) return JD_E_ARG;
    HIPCHK(hipSetDevice(b->device));
    if (row_stride == im.row_stride)                         // one copy (the 2-D form goes row by row)
        HIPCHK(hipMemcpyAsync(rgb_out, im.rgb, (size_t)(im.height - 1) * row_stride + (size_t)im.width * 3, hipMemcpyDeviceToHost, b->stream));
    else
        HIPCHK(hipMemcpy2DAsync(rgb_out, row_stride, im.rgb, im.row_stride, (size_t)im.width * 3, im.height, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return JD_OK;
}

extern "C" int jd_batch_coefficients(jd_batch* b, int image, int16_t* out, size_t cap) {
    if (check_image(b, image) || !out) return JD_E_ARG;
    const JdImg& im = b->img[image];
    const size_t n = (size_t)im.n_mcus * im.bpm * 64;
    if (cap < n) return JD_E_ARG;
    HIPCHK(hipSetDevice(b->device));
    if (im.flags & JD_IMG_COEF8) {                                // byte form: expand on the device into a temporary
        int16_t* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, n * sizeof(int16_t)));
        jdk_launch_coef8_expand(kernel_args(b), image, im.n_mcus * im.bpm, tmp, b->stream);
        hipError_t he = hipMemcpyAsync(out, tmp, n * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(b->stream);
        (void)hipFree(tmp);
        if (he != hipSuccess) { g_last_hip = (int)he; return JD_E_HIP; }
        return JD_OK;
    }
    HIPCHK(hipMemcpyAsync(out, b->d_coef + im.coef_off, n * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return JD_OK;
}

extern "C" size_t jd_batch_workspace_bytes(const jd_batch* b) { return b ? b->arena.total : 0; }

extern "C" int jd_batch_unstuffed_len(jd_batch* b, int image, size_t* len) {
    if (check_image(b, image) || !len) return JD_E_ARG;
    uint32_t v = 0;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipMemcpyAsync(&v, b->d_stream_len + image, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    *len = v;
    return JD_OK;
}

extern "C" int jd_batch_profile(jd_batch* b, int iters, float* stage_us, const char** names, int cap) {
    if (!b || !stage_us || iters <= 0 || cap < ST_COUNT_) return JD_E_ARG;
    for (int i = 0; i < b->n; i++) if (b->img[i].ecs_len == 0) return JD_E_STATE;
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = b->stream;
    if (b->dirty) { int rc = refresh(b, s); if (rc) return rc; }
    hipEvent_t ev[ST_COUNT_ + 1];
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    std::vector<double> acc(ST_COUNT_, 0.0);
    for (int it = 0; it < iters; it++) {
        HIPCHK(hipEventRecord(ev[0], s));
        for (int st = 0; st < ST_COUNT_; st++) {
            int rc = enqueue_stage(b, st, s); if (rc) return rc;
            HIPCHK(hipEventRecord(ev[st + 1], s));
        }
        HIPCHK(hipStreamSynchronize(s));
        hipEvent_t last = ev[0];
        for (int st = 0; st < ST_COUNT_; st++) {
            if (stage_is_empty(b, st)) continue;                 // no event was recorded for it: 0
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, last, ev[st + 1])); acc[st] += ms * 1000.0;
            last = ev[st + 1];
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int st = 0; st < ST_COUNT_; st++) { stage_us[st] = (float)(acc[st] / iters); if (names) names[st] = kStageNames[st]; }
    b->last_stream = s; b->ran = true;
    return ST_COUNT_;
}

// ---- pinned-memory I/O pipeline ---------------------------------------------------------------
// The reference reads one file, decodes it and writes one BMP, in turn (main.go:1509-1512).  In throughput form the
// host keeps two or three JD_BATCH_PINNED_IO batches in flight: it fills one batch's pinned ECS staging (straight from
// the files -- the staging IS the read buffer) while the device copies / decodes / copies back the others.
static int stage_check(jd_batch* b) {
    if (!b) return JD_E_ARG;
    if (!(b->flags & JD_BATCH_PINNED_IO)) return JD_E_STATE;
    if (b->io_pending) return JD_E_STATE;                    // a submit is in flight: jd_batch_wait first
    if (!b->stage_open) { b->stage_open = true; b->staged_bytes = 0; b->staged_count = 0; for (auto& im : b->img) im.ecs_len = 0; }
    return JD_OK;
}

extern "C" int jd_batch_stage_reserve(jd_batch* b, int image, size_t ecs_len, uint8_t** dst) {
    int rc = stage_check(b); if (rc) return rc;
    if (check_image(b, image) || !dst) return JD_E_ARG;
    if (ecs_len == 0 || ecs_len > b->ecs_cap[image]) return JD_E_ARG;
    if (b->img[image].ecs_len) return JD_E_STATE;            // staged twice
    const size_t off = align_up(b->staged_bytes, 16);
    if (off + ecs_len > b->ecs_own_bytes) return JD_E_ARG;
    JdImg& im = b->img[image];
    im.ecs = b->d_ecs_own + off; im.ecs_len = (uint32_t)ecs_len; im.flags &= ~JD_IMG_UNSTUFFED_INPUT;
    im.rgb = b->d_rgb_own + b->own_rgb_off[image]; im.row_stride = im.width * 3; b->rgb_bound[image] = 0;
    b->staged_bytes = off + ecs_len; b->staged_count++; b->dirty = true;
    *dst = b->h_ecs + off;
    return JD_OK;
}

extern "C" int jd_batch_stage_ecs(jd_batch* b, int image, const uint8_t* ecs, size_t ecs_len) {
    if (!ecs) return JD_E_ARG;
    uint8_t* dst = nullptr;
    int rc = jd_batch_stage_reserve(b, image, ecs_len, &dst); if (rc) return rc;
    memcpy(dst, ecs, ecs_len);
    return JD_OK;
}

// Parse `count` JPEG files held in memory (jd_parse, the twin of decodeJPEG's header work), check that file k matches
// the frame image first + k was created with, and copy its entropy-coded segment into the staging; `threads` host
// threads share the parsing and the copying.
extern "C" int jd_batch_stage_jpegs(jd_batch* b, int first, int count, const uint8_t* const* files, const size_t* lens,
                                    int threads) {
    int rc = stage_check(b); if (rc) return rc;
    if (!files || !lens || count <= 0 || first < 0 || first + count > b->n) return JD_E_ARG;
    threads = std::max(1, std::min(threads, std::min(count, 64)));
    std::vector<size_t> off(count), len(count);
    std::vector<int> err(threads, JD_OK);
    auto each = [&](auto&& fn) {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back([&, t] { for (int k = t; k < count; k += threads) fn(k, t); });
        for (int k = 0; k < count; k += threads) fn(k, 0);
        for (auto& th : pool) th.join();
    };
    each([&](int k, int t) {
        jd_frame fr;
        int r = files[k] ? jd_parse(files[k], lens[k], &fr, &off[k], &len[k]) : JD_E_ARG;
        if (r == JD_OK && memcmp(&fr, &b->frames[first + k], sizeof fr) != 0) r = JD_E_ARG;   // not the frame the slot was planned for
        if (r != JD_OK && err[t] == JD_OK) err[t] = r;
    });
    for (int e : err) if (e) return e;
    std::vector<uint8_t*> dst(count);
    for (int k = 0; k < count; k++) { rc = jd_batch_stage_reserve(b, first + k, len[k], &dst[k]); if (rc) return rc; }
    each([&](int k, int) { memcpy(dst[k], files[k] + off[k], len[k]); });
    return JD_OK;
}

extern "C" int jd_batch_submit(jd_batch* b, void* hip_stream) {
    if (!b) return JD_E_ARG;
    if (!(b->flags & JD_BATCH_PINNED_IO) || b->io_pending) return JD_E_STATE;
    if (b->staged_count != b->n) return JD_E_STATE;          // every image staged (a finished staging may be re-submitted)
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
    HIPCHK(hipMemcpyAsync(b->d_ecs_own, b->h_ecs, b->staged_bytes, hipMemcpyHostToDevice, b->s_h2d));
    HIPCHK(hipEventRecord(b->ev_h2d, b->s_h2d));
    HIPCHK(hipStreamWaitEvent(s, b->ev_h2d, 0));
    int rc = run_on(b, s); if (rc) return rc;
    rc = enqueue_rgb_download(b); if (rc) return rc;
    b->io_pending = true; b->stage_open = false;
    return JD_OK;
}

extern "C" int jd_batch_host_rgb(jd_batch* b, int image, const uint8_t** rgb, size_t* row_stride) {
    if (check_image(b, image) || !rgb) return JD_E_ARG;
    if (!(b->flags & JD_BATCH_PINNED_IO) || b->io_pending) return JD_E_STATE;
    *rgb = b->h_rgb + b->own_rgb_off[image];
    if (row_stride) *row_stride = (size_t)b->img[image].width * 3;
    return JD_OK;
}

// ---- one-shot forms ----------------------------------------------------------------------------------
static int one_shot(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int stuffed, const uint32_t* rst,
                    size_t n_rst, int device, unsigned flags, jd_batch** out) {
    if (!frame || !ecs || ecs_len == 0) return JD_E_ARG;
    jd_batch* b = nullptr;
    int rc = jd_batch_create(device, 1, frame, &ecs_len, flags, &b);
    if (rc) return rc;
    rc = stuffed ? jd_batch_upload(b, 0, ecs, ecs_len) : jd_batch_upload_unstuffed(b, 0, ecs, ecs_len, rst, n_rst);
    if (!rc) rc = jd_batch_run(b, nullptr);
    if (!rc) rc = jd_batch_wait(b);
    if (rc) { jd_batch_destroy(b); return rc; }
    *out = b;
    return JD_OK;
}

extern "C" int jd_decode(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int ecs_is_stuffed,
                         const uint32_t* rst_offsets, size_t n_rst, uint8_t* rgb_out, size_t row_stride, int device) {
    if (!rgb_out) return JD_E_ARG;
    jd_batch* b = nullptr;
    int rc = one_shot(frame, ecs, ecs_len, ecs_is_stuffed, rst_offsets, n_rst, device, 0, &b);
    if (rc) return rc;
    rc = jd_batch_download(b, 0, rgb_out, row_stride);
    jd_batch_destroy(b);
    return rc;
}

extern "C" int jd_decode_coefficients(const jd_frame* frame, const uint8_t* ecs, size_t ecs_len, int ecs_is_stuffed,
                                      const uint32_t* rst_offsets, size_t n_rst, int16_t* coef_out, size_t coef_cap,
                                      int device) {
    if (!coef_out) return JD_E_ARG;
    jd_batch* b = nullptr;
    int rc = one_shot(frame, ecs, ecs_len, ecs_is_stuffed, rst_offsets, n_rst, device, JD_BATCH_KEEP_COEF, &b);
    if (rc) return rc;
    rc = jd_batch_coefficients(b, 0, coef_out, coef_cap);
    jd_batch_destroy(b);
    return rc;
}

// ---- allocation-free entry point ---------------------------------------------------------------------------
// jd_batch_{create, bind_ecs, bind_rgb, run, wait} for ONE image, with the workspace taken from the library's file-scope
// device array instead of hipMalloc: no hipMalloc / hipFree / hipHostMalloc / stream or event creation happens here
// (the array's address, the pinned block and the stream were set up per device by jd_init).  Calls on one device share
// that device's array: they are serialised on the host by a mutex, and each call clears what must start at zero (stream
// pads, status and sync flags, K0 tickets and look-back words) on the stream before its kernels.
namespace {
void drop_plan(DevCtx* c) {
    if (!c->plan) return;
    if (c->plan->last_stream) (void)hipStreamSynchronize(c->plan->last_stream);
    batch_release(c->plan);
    c->plan = nullptr; c->plan_cap = 0;
}
}  // namespace

extern "C" int jd_decode_resident(const jd_frame* frame, const void* d_ecs, size_t ecs_len, void* d_rgb, size_t row_stride,
                                  int device, void* hip_stream) {
    if (!frame || !d_ecs || !d_rgb || ecs_len == 0) return JD_E_ARG;
    DevCtx* ctx = find_ctx(device);
    if (!ctx) return jd_device_count() ? JD_E_ARG : JD_E_NOT_INIT;
    Geometry g; int rc = frame_geometry(frame, &g);
    if (rc) return rc;
    if (ecs_len > JD_RESIDENT_MAX_ECS || (uint64_t)g.mcux * g.mcuy * g.bpm * 64 > JD_RESIDENT_MAX_COEF) return JD_E_TOO_LARGE;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    // Repeated calls with one frame (a serving loop on one camera / one encoder setting) reuse the plan: no table
    // packing, no host allocation, no table upload.  Anything else plans afresh (host work only; the device side
    // is the same fixed array either way).
    if (ctx->plan && (memcmp(&ctx->plan_frame, frame, sizeof *frame) != 0 || ecs_len > ctx->plan_cap)) drop_plan(ctx);
    if (!ctx->plan) {
        size_t cap = std::min<size_t>(JD_RESIDENT_MAX_ECS, align_up(ecs_len + ecs_len / 4, 65536));   // room for the next, larger one
        jd_batch* nb = nullptr;
        rc = batch_build(device, 1, frame, &cap, 0, ctx, s, &nb);
        if (rc == JD_E_TOO_LARGE && cap != ecs_len) { cap = ecs_len; rc = batch_build(device, 1, frame, &cap, 0, ctx, s, &nb); }
        if (rc) { (void)hipStreamSynchronize(s); return rc; }   // (a table upload may already be in flight from the pinned block)
        ctx->plan = nb; ctx->plan_frame = *frame; ctx->plan_cap = cap;
    }
    jd_batch* b = ctx->plan;
    b->stream = s;
    rc = jd_batch_bind_ecs(b, 0, d_ecs, ecs_len);
    if (!rc) rc = jd_batch_bind_rgb(b, 0, d_rgb, row_stride);
    if (!rc) rc = run_on(b, s);
    if (!rc) rc = wait_on(b);
    if (rc == JD_E_HIP || rc == JD_E_STATE || rc == JD_E_ARG) {
        // not a verdict on the image: something may still be in flight on `s` that reads the shared scratch and the
        // pinned block -- drain it before the lock is released, and do not trust the plan any more
        (void)hipStreamSynchronize(s);
        drop_plan(ctx);
    }
    return rc;
}

// ---- progressive files ---------------------------------------------------------------------
// Host side of jd_decode_scans: copy every scan's raw bytes (16-byte aligned) into one upload, pack the symbol tables each
// scan names, order the scans in dependency levels; on the device K0 unstuffs all scans in one launch the way
// main.go:993-1021 does (FF00 -> FF, RSTn dropped -- the reference never acts on restart markers, least of all in
// progressive scans -- FF fill bytes skipped: to K0 a scan is an image of one restart interval), kp_scan runs level by
// level, then K2 as for any frame.
namespace {

struct DevBuf {                       // hipMalloc'ed scratch of one call
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { HIPCHK(hipMalloc(&p, std::max<size_t>(bytes, 256))); return JD_OK; }
};

// n_images progressive files in one go: the scans of all of them are dealt to waves level by level, so a batch keeps the
// chip busy where one file cannot (a scan is a serial chain).
int decode_scans_impl(int n_images, const jd_frame* frames, const jd_scan* const* scans, const int* n_scans,
                      const uint8_t* const* files, const size_t* file_lens, uint8_t* const* rgb_out, const size_t* row_strides,
                      int16_t* coef_out, size_t coef_cap, int* status_out, int device) {
    if (n_images <= 0 || !frames || !scans || !n_scans || !files || !file_lens) return JD_E_ARG;
    if (coef_out && n_images != 1) return JD_E_ARG;
    const bool tt = env_int("JD_DEBUG_SCAN_TIMES", 0) != 0;    // phase times of the call on stderr (experiments)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_0 = now();
    std::vector<uint8_t> stream;                               // raw (stuffed) bytes of every scan
    std::vector<uint32_t> raw_off;                             // per scan: where they lie in `stream`
    size_t un_bytes = 0;                                       // device buffer of the unstuffed scans
    std::vector<uint32_t> lut(1, 0);                           // entry 0 stays unused
    std::map<std::string, std::pair<uint32_t, uint32_t>> packed;   // table bytes -> (offset, length) in `lut`
    std::vector<JdPScan> ps;
    std::vector<int> level;
    int n_levels = 0;
    std::vector<jd_frame> fr(frames, frames + n_images);
    for (int im = 0; im < n_images; im++) {
        const jd_frame* frame = &frames[im];
        const jd_scan* sc_im = scans[im];
        const uint8_t* file = files[im];
        const size_t file_len = file_lens[im];
        if (!sc_im || n_scans[im] <= 0 || !file) return JD_E_ARG;
        Geometry g; int rc = frame_geometry(frame, &g);
        if (rc) return rc;
        for (int c = 0; c < frame->ncomp; c++)
            if (frame->comp[c].tq > 3 || !(frame->qt_present & (1u << frame->comp[c].tq))) return JD_E_TABLE;
        const size_t first = ps.size();
        for (int k = 0; k < n_scans[im]; k++) {
            const jd_scan& sc = sc_im[k];
            if (!sc.progressive) return n_images == 1 ? JD_E_ARG : JD_E_UNSUPPORTED;   // baseline files: the jd_batch path
            if (sc.ncomp < 1 || sc.ncomp > frame->ncomp || sc.se > 63 || sc.ss > sc.se || sc.al > 13) return JD_E_ARG;
            if (sc.ecs_off > file_len || sc.ecs_len > file_len - sc.ecs_off) return JD_E_ARG;
            if (sc.ss == 0 && sc.se != 0) return JD_E_UNSUPPORTED; // T.81 G.1.1.1.1: a DC scan holds DC only (the reference would mix the cases up)
            if (sc.ss != 0 && sc.ncomp != 1) return JD_E_UNSUPPORTED;   // AC scans are never interleaved (the reference's skips counter assumes it)
            JdPScan d;
            memset(&d, 0, sizeof d);
            d.img = (uint32_t)im; d.ss = sc.ss; d.se = sc.se; d.ah = sc.ah; d.al = sc.al; d.ncomp = sc.ncomp;
            // the reference walks the FRAME's components and decodes those the scan uses (main.go:771-775): ascending order
            int order[3] = { 0, 1, 2 };
            std::sort(order, order + sc.ncomp, [&](int p, int q) { return sc.comp[p] < sc.comp[q]; });
            for (int j = 0; j < sc.ncomp; j++) {
                const int src = order[j];
                if (sc.comp[src] >= frame->ncomp || (j && sc.comp[src] == sc.comp[order[j - 1]])) return JD_E_ARG;
                d.comp[j] = sc.comp[src]; d.lut_off[j] = d.lut_len[j] = 0;
                const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = sc.ss != 0;
                if (need_dc || need_ac) {
                    const int id = need_dc ? sc.td[src] : sc.ta[src];
                    if (id > 3 || !sc.huff[need_dc ? 0 : 1][id].present) return JD_E_TABLE;
                    const jd_huff& h = sc.huff[need_dc ? 0 : 1][id];
                    // (files of one encoder share their tables, and so do most scans of a file: pack each once per call)
                    std::string key((const char*)h.bits, 16); key.append((const char*)h.vals, h.nvals);
                    auto it = packed.find(key);
                    if (it == packed.end()) {
                        const uint32_t off = (uint32_t)lut.size();
                        if (jd_pack_lut_lane(h.bits, h.vals, h.nvals, lut)) return JD_E_TABLE;
                        it = packed.emplace(std::move(key), std::make_pair(off, (uint32_t)lut.size() - off)).first;
                    }
                    d.lut_off[j] = it->second.first; d.lut_len[j] = it->second.second;
                }
            }
            // the scan's raw bytes, 16-byte aligned in the upload (K0 reads whole quads); its unstuffed bytes will lie at
            // un_off + JD_STREAM_FRONT of the device's stream buffer, zeros behind them (kp_scan reads past the end as zeros)
            while (stream.size() % 16) stream.push_back(0);
            if (stream.size() + sc.ecs_len + 64 >= (size_t)1 << 31 || un_bytes + sc.ecs_len + 512 >= (size_t)1 << 31) return JD_E_ARG;
            raw_off.push_back((uint32_t)stream.size());
            stream.insert(stream.end(), file + sc.ecs_off, file + sc.ecs_off + sc.ecs_len);
            d.stream_off = (uint32_t)un_bytes + JD_STREAM_FRONT;
            d.stream_len = (uint32_t)sc.ecs_len;                  // raw length: the unstuffed one is K0's to tell (stream_len[] on the device)
            un_bytes += align_up(JD_STREAM_FRONT + sc.ecs_len + JD_STREAM_TAIL + 64, 256);
            // dependency level: after every earlier scan of the file that touches one of its components in an overlapping band
            int lv = 0;
            for (int i = 0; i < k; i++) {
                bool share = false;
                for (int p = 0; p < sc.ncomp; p++) for (int q = 0; q < sc_im[i].ncomp; q++) share |= sc.comp[p] == sc_im[i].comp[q];
                if (share && !(sc.se < sc_im[i].ss || sc_im[i].se < sc.ss)) lv = std::max(lv, level[first + i] + 1);
            }
            ps.push_back(d); level.push_back(lv);
            n_levels = std::max(n_levels, lv + 1);
        }
        // the batch needs SOME valid table pair per component; K0 / K1 never run on these images
        memset(fr[im].huff, 0, sizeof fr[im].huff);
        fr[im].huff[0][0].bits[0] = 1; fr[im].huff[0][0].nvals = 1; fr[im].huff[0][0].present = 1;
        fr[im].huff[1][0] = fr[im].huff[0][0];
        for (int c = 0; c < fr[im].ncomp; c++) fr[im].comp[c].td = fr[im].comp[c].ta = 0;
        fr[im].restart_interval = 0;
    }
    // ---- device: a batch supplies geometry, quantisation tables, coefficient buffers and K2 ----
    jd_batch* b = nullptr;
    std::vector<size_t> caps(n_images, 16);
    const auto t_1 = now();
    int rc = jd_batch_create(device, n_images, fr.data(), caps.data(), JD_BATCH_KEEP_COEF | JD_BATCH_COEF16_, &b);
    if (rc) return rc;
    const auto t_2 = now();
    struct Guard { jd_batch* b; ~Guard() { batch_release(b); } } guard{ b };
    DevBuf d_stream, d_lut, d_scans, d_list, d_un, d_k0;
    std::vector<uint32_t> list;
    std::vector<int> level_start(n_levels + 1, 0);
    for (int l = 0; l < n_levels; l++) {
        for (size_t k = 0; k < ps.size(); k++) if (level[k] == l) list.push_back((uint32_t)k);
        level_start[l + 1] = (int)list.size();
    }
    if ((rc = d_stream.alloc(stream.size())) || (rc = d_lut.alloc(lut.size() * 4)) || (rc = d_scans.alloc(ps.size() * sizeof(JdPScan))) ||
        (rc = d_list.alloc(list.size() * 4))) return rc;
    hipStream_t s = b->stream;
    // ---- K0 over the scans: one pseudo-image of one restart interval per scan -------------------------------------
    const size_t nsc = ps.size();
    std::vector<JdImg> pim(nsc);
    std::vector<uint32_t> ppre(nsc + 1, 0);
    for (size_t k = 0; k < nsc; k++) {
        JdImg& q = pim[k];
        memset(&q, 0, sizeof q);
        q.ecs = (const uint8_t*)d_stream.p + raw_off[k]; q.ecs_len = ps[k].stream_len;
        q.n_k0_chunks = (uint32_t)((q.ecs_len + JD_K0_CHUNK - 1) / JD_K0_CHUNK);
        q.n_intervals = 1; q.interval_base = (uint32_t)(2 * k);
        q.stream_off = ps[k].stream_off - JD_STREAM_FRONT;
        ppre[k + 1] = ppre[k] + q.n_k0_chunks;
    }
    // one block: JdImg[nsc] | prefix[nsc + 1] | status[nsc] | stream_len[nsc] | interval_start[2 nsc] | tickets | look-back words
    const size_t o_img = 0, o_pre = align_up(nsc * sizeof(JdImg), 256), o_st = o_pre + align_up((nsc + 1) * 4, 256),
                 o_len = o_st + align_up(nsc * 4, 256), o_iv = o_len + align_up(nsc * 4, 256), o_tk = o_iv + align_up(nsc * 8, 256),
                 o_lb = o_tk + align_up((nsc + 64) * 4, 256), k0_bytes = o_lb + align_up(((size_t)ppre[nsc] + 32) * 8, 256);
    if ((rc = d_un.alloc(un_bytes)) || (rc = d_k0.alloc(k0_bytes))) return rc;
    uint8_t* const k0b = (uint8_t*)d_k0.p;
    HIPCHK(hipMemsetAsync(d_un.p, 0, std::max<size_t>(un_bytes, 256), s));
    HIPCHK(hipMemsetAsync(k0b + o_st, 0, o_iv - o_st, s));                     // scan status words and unstuffed lengths
    HIPCHK(hipMemcpyAsync(k0b + o_img, pim.data(), nsc * sizeof(JdImg), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(k0b + o_pre, ppre.data(), (nsc + 1) * 4, hipMemcpyHostToDevice, s));
    JdKernelArgs ka;
    memset(&ka, 0, sizeof ka);
    ka.imgs = (const JdImg*)(k0b + o_img); ka.n_images = (int)nsc; ka.stream = (uint8_t*)d_un.p;
    ka.status = (uint32_t*)(k0b + o_st); ka.stream_len = (uint32_t*)(k0b + o_len); ka.interval_start = (uint32_t*)(k0b + o_iv);
    ka.k0_ticket = (uint32_t*)(k0b + o_tk); ka.k0_state = (unsigned long long*)(k0b + o_lb);
    HIPCHK(hipMemcpyAsync(d_stream.p, stream.data(), stream.size(), hipMemcpyHostToDevice, s));
    jdk_launch_k0(ka, (const uint32_t*)(k0b + o_pre), ppre[nsc], s);
    HIPCHK(hipMemcpyAsync(d_lut.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_scans.p, ps.data(), ps.size() * sizeof(JdPScan), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(b->d_coef, 0, b->coef_elems * sizeof(int16_t), s));   // scans only write what they code
    rc = refresh(b, s); if (rc) return rc;
    rc = enqueue_stage(b, ST_CLEAR, s); if (rc) return rc;
    const JdKernelArgs a = kernel_args(b);
    for (int l = 0; l < n_levels; l++)
        jdk_launch_pscan(a, (const JdPScan*)d_scans.p, (const uint32_t*)d_list.p + level_start[l], level_start[l + 1] - level_start[l],
                         (const uint8_t*)d_un.p, (const uint32_t*)d_lut.p, ka.stream_len, ka.status, s);
    HIPCHK(hipGetLastError());
    rc = enqueue_stage(b, ST_IDCT_COLOUR, s); if (rc) return rc;
    b->ran = true; b->last_stream = s;
    const auto t_3 = now();
    const int first_err = wait_on(b);                          // synchronises: the pageable sources above are free again
    const auto t_4 = now();
    if (first_err == JD_E_HIP) return first_err;
    if (coef_out) return first_err ? first_err : jd_batch_coefficients(b, 0, coef_out, coef_cap);
    // The pictures go to pageable host memory: such a copy is staged by the calling thread, so a call with many pictures
    // shares them out over a few threads, each with a stream of its own (1024 pictures, 544 MB: 116 ms on one thread).
    std::vector<int> todo;
    for (int im = 0; im < n_images; im++) {
        const int st = status_to_error(b->status[im]);
        if (status_out) status_out[im] = st;
        if (st != JD_OK || !rgb_out || !rgb_out[im]) continue;
        const size_t rs = row_strides ? row_strides[im] : (size_t)frames[im].width * 3;
        if (rs < (size_t)b->img[im].width * 3 || !b->img[im].rgb) return JD_E_ARG;
        todo.push_back(im);
    }
    const int nthr = (int)std::min<size_t>(8, (todo.size() + 31) / 32);
    std::vector<int> terr(std::max(nthr, 1), (int)hipSuccess);
    auto copy_share = [&](int t, hipStream_t cs) {
        for (size_t k = (size_t)t; k < todo.size(); k += (size_t)std::max(nthr, 1)) {
            const int im = todo[k];
            const JdImg& q = b->img[im];
            const size_t rs = row_strides ? row_strides[im] : (size_t)frames[im].width * 3;
            hipError_t he = rs == q.row_stride
                ? hipMemcpyAsync(rgb_out[im], q.rgb, (size_t)(q.height - 1) * rs + (size_t)q.width * 3, hipMemcpyDeviceToHost, cs)
                : hipMemcpy2DAsync(rgb_out[im], rs, q.rgb, q.row_stride, (size_t)q.width * 3, q.height, hipMemcpyDeviceToHost, cs);
            if (he != hipSuccess) { terr[t] = (int)he; break; }
        }
        const hipError_t he = hipStreamSynchronize(cs);
        if (he != hipSuccess && terr[t] == (int)hipSuccess) terr[t] = (int)he;
    };
    if (nthr <= 1) copy_share(0, s);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; t++)
            pool.emplace_back([&, t] {
                hipStream_t cs = nullptr;
                if (hipSetDevice(b->device) != hipSuccess || hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { terr[t] = (int)hipErrorUnknown; return; }
                copy_share(t, cs);
                (void)hipStreamDestroy(cs);
            });
        for (auto& th : pool) th.join();
    }
    for (int e : terr) if (e != (int)hipSuccess) { g_last_hip = e; return JD_E_HIP; }
    if (tt) fprintf(stderr, "jd_decode_scans: %d files: lists + tables %.1f ms, batch %.1f, alloc + enqueue %.1f, device %.1f, download %.1f\n",
                    n_images, ms(t_0, t_1), ms(t_1, t_2), ms(t_2, t_3), ms(t_3, t_4), ms(t_4, now()));
    return first_err;
}

}  // namespace

extern "C" int jd_decode_scans(const jd_frame* frame, const jd_scan* scans, int n_scans, const uint8_t* file, size_t file_len,
                               uint8_t* rgb_out, size_t row_stride, int device) {
    if (!rgb_out || !frame || !scans || n_scans <= 0 || !file) return JD_E_ARG;
    if (!scans[0].progressive) {                               // an SOF0 file: the baseline path
        if (n_scans != 1 || scans[0].ecs_off > file_len || scans[0].ecs_len > file_len - scans[0].ecs_off) return JD_E_ARG;
        return jd_decode(frame, file + scans[0].ecs_off, scans[0].ecs_len, 1, nullptr, 0, rgb_out, row_stride, device);
    }
    return decode_scans_impl(1, frame, &scans, &n_scans, &file, &file_len, &rgb_out, &row_stride, nullptr, 0, nullptr, device);
}

extern "C" int jd_decode_scans_coefficients(const jd_frame* frame, const jd_scan* scans, int n_scans, const uint8_t* file,
                                            size_t file_len, int16_t* coef_out, size_t coef_cap, int device) {
    if (!coef_out || !frame || !scans || n_scans <= 0 || !file) return JD_E_ARG;
    if (!scans[0].progressive) {
        if (n_scans != 1 || scans[0].ecs_off > file_len || scans[0].ecs_len > file_len - scans[0].ecs_off) return JD_E_ARG;
        return jd_decode_coefficients(frame, file + scans[0].ecs_off, scans[0].ecs_len, 1, nullptr, 0, coef_out, coef_cap, device);
    }
    return decode_scans_impl(1, frame, &scans, &n_scans, &file, &file_len, nullptr, nullptr, coef_out, coef_cap, nullptr, device);
}

extern "C" int jd_decode_scans_batch(int n_images, const jd_frame* frames, const jd_scan* const* scans, const int* n_scans,
                                     const uint8_t* const* files, const size_t* file_lens, uint8_t* const* rgb_out,
                                     const size_t* row_strides, int* status, int device) {
    if (!rgb_out) return JD_E_ARG;
    return decode_scans_impl(n_images, frames, scans, n_scans, files, file_lens, rgb_out, row_strides, nullptr, 0, status, device);
}
