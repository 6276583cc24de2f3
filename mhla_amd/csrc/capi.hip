// C ABI of libmhla_hip.so (see include/mhla_hip.h): library-wide entry points and the block-mixing operator.  Validates
// arguments, carves the caller's workspace, picks the path and enqueues kernels on the caller's stream.  The bf16 fast paths
// (fused.hpp, fused_tile16.hpp, smalln.hpp) are launched from here; the generic / split-operand launches live in the per-dtype
// units (capi_bm_typed.hpp), the causal operator in capi_causal.hip, prologues / epilogues / LePE in capi_misc.hip.
#include "capi_common.hpp"
#include "fused.hpp"
#include "fused_tile16.hpp"
#include "smalln.hpp"
#include "smalln_f32.hpp"

using namespace mhla;
using namespace mhla::capi;

namespace mhla {
namespace capi {
extern template int bm_fwd_typed<float, false>(const BmCall&);
extern template int bm_fwd_typed<bf16_t, true>(const BmCall&);
extern template int bm_fwd_typed<bf16_t, false>(const BmCall&);
extern template int bm_fwd_typed<f16_t, false>(const BmCall&);
extern template int bm_bwd_typed<float, false>(const BmCall&);
extern template int bm_bwd_typed<bf16_t, true>(const BmCall&);
extern template int bm_bwd_typed<bf16_t, false>(const BmCall&);
extern template int bm_bwd_typed<f16_t, false>(const BmCall&);
}  // namespace capi
}  // namespace mhla

namespace {

// ---- fast path (bf16, D = 64, M <= 64, q_den aliasing q_num): see fused.hpp ----
// summaries of single-chunk blocks (S <= 64): the straight-line kernel, instantiated on (gather map, normaliser)
template <int MODE>
static int launch_state1c(const fast::FsStateArgs& sa, int njg, int BH, hipStream_t st, const char* name) {
    const dim3 g(njg, BH), t(fast::FT8);
    const bool idx = sa.idx != nullptr, norm = sa.normalize != 0;
    if (idx && norm) return launch(fast::k_fs_state1c<MODE, true, true>, g, t, fast::FS_STATE1C_SMEM, st, name, sa);
    if (idx) return launch(fast::k_fs_state1c<MODE, true, false>, g, t, fast::FS_STATE1C_SMEM, st, name, sa);
    if (norm) return launch(fast::k_fs_state1c<MODE, false, true>, g, t, fast::FS_STATE1C_SMEM, st, name, sa);
    return launch(fast::k_fs_state1c<MODE, false, false>, g, t, fast::FS_STATE1C_SMEM, st, name, sa);
}

// small-sequence kernels (smalln.hpp): instantiated on (head-dim tiles, gather map, exactly 16 blocks, hi + lo score tiles)
template <int DT, bool HL>
static auto sn_fwd_kernel(bool gather, int M) {
    return gather ? fast::k_sn_fwd<DT, true, false, HL> : (M == 16 ? fast::k_sn_fwd<DT, false, true, HL> : fast::k_sn_fwd<DT, false, false, HL>);
}
#ifndef SN_BWD_NB
#define SN_BWD_NB 1   // blocks per wave of the default-arithmetic small-sequence backward (smalln.hpp k_sn_bwd): sixteen waves with one block each
                      // beat eight with two by 3.5 % at C3 (107.0 -> 103.0 us, twice each; workgroup life 91 400 -> 80 000 cycles in tools/trace_smalln.py)
#endif
template <int DT, bool HL>
static auto sn_bwd_kernel(bool gather, int M) {
    // (HL: the run-time-guarded block loops also for exactly 16 blocks -- fully unrolled, pass A becomes one basic block whose
    // schedule needs 100-135 registers more than the 256 there are; the guarded form allocates 187 / 230 without a spill)
    if constexpr (HL) return gather ? fast::k_sn_bwd<DT, true, false, true, SN_BWD_NB> : fast::k_sn_bwd<DT, false, false, true, SN_BWD_NB>;
    else return gather ? fast::k_sn_bwd<DT, true, false, false> : (M == 16 ? fast::k_sn_bwd<DT, false, true, false> : fast::k_sn_bwd<DT, false, false, false>);
}

struct FastWs {
    fast::u16 *state, *dstate;
    float *z, *ksum, *ninv, *dn, *dz, *dwp, *dksum;
    int* done;   // per 16-block tile: dQ done (k_tile_bwd)
    int* err;    // error word of the token-gradient launch (a hand-over flag that did not arrive)
    size_t total_fwd, total_bwd;
    int njg, ntt;
    int cs;      // workgroups per tile and role for blocks of several 64-token chunks (S > 64), 1 otherwise
};
FastWs fast_carve(void* ws, int B, int H, int M, int S) {
    const size_t bh = (size_t)B * H;
    FastWs w;
    w.njg = (M + fast::IT - 1) / fast::IT;
    const size_t st = bh * w.njg * fast::FE * fast::IT * 2;   // bytes, multiple of 16
    char* p = (char*)ws;
    w.state = (fast::u16*)p; p += st;
    w.z = (float*)p; p += al4(bh * M * S) * 4;
    w.ksum = (float*)p; p += al4(bh * M * 64) * 4;
    w.ninv = (float*)p; p += al4(bh * M * S) * 4;
    w.total_fwd = (size_t)(p - (char*)ws);
    w.dstate = (fast::u16*)p; p += st;
    w.dn = (float*)p; p += al4(bh * M * S) * 4;
    w.dz = (float*)p; p += al4(bh * M * S) * 4;
    w.cs = S > 64 ? std::min((S + 63) / 64, 4) : 1;
    w.dksum = (float*)p; p += al4(bh * M * 64) * 4 * w.cs;   // one share per chunk part
    w.dwp = (float*)p; p += bh * fast::DW_SPLIT * 4096 * 4;
    w.ntt = fast::tiles_per_bh(w.njg, 16);
    w.done = (int*)p; p += al4(bh * w.ntt * w.cs + 1) * 4;   // per-tile (and part) flags
    w.err = nullptr;                                  // (the error word lives at the tail of the workspace: bwd_err_word)
    w.total_bwd = (size_t)(p - (char*)ws);
    return w;
}
// ---- process-wide options of the fused token-gradient launch (read ONCE, at library load, from the environment; changed afterwards
// only through mhla_set_option) ----
std::atomic<int> g_two_launches{[] { const char* e = getenv("MHLA_BWD_TWO_LAUNCHES"); return (e && e[0] == '1') ? 1 : 0; }()};
std::atomic<int> g_drop_signal{getenv("MHLA_DEBUG_DROP_SIGNAL") != nullptr ? 1 : 0};

// Self-healing of the in-launch hand-over (fused.hpp, tile_wait): every fused launch is followed by an asynchronous copy of its
// error word into a pinned host slot (+ an event on the same stream); the NEXT backward of this process looks at the slots whose
// event has completed -- no synchronisation anywhere -- and the first raised word latches the process to the two-launch form
// (the kernel boundary then orders the hand-over, whatever the dispatch order is) and says so once on stderr.
struct HandoverWatch {
    static constexpr int SLOTS = 64, MAXDEV = 16;
    std::mutex mu;
    int* host = nullptr;              // pinned (portable: visible to every device's streams), MAXDEV x SLOTS words
    hipEvent_t ev[MAXDEV][SLOTS] = {};   // an event belongs to the device it was created on: one set per device
    bool pending[MAXDEV][SLOTS] = {};
    int next[MAXDEV] = {};
    bool warned = false, unarmed_said = false;
    void unarmed(const char* why) {   // (mu held) say once that the watch cannot arm; the fused launch still runs, only the self-healing is off
        if (!unarmed_said) {
            unarmed_said = true;
            fprintf(stderr, "[mhla] the hand-over watch of the fused token-gradient launch could not be armed (%s): an expired hand-over will only "
                            "show through mhla_blockmix_bwd_status\n", why);
        }
        (void)hipGetLastError();
    }
    void poll(int dev) {   // (mu held) event queries are allowed while another thread captures a stream in global mode only in relaxed mode
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        for (int i = 0; i < SLOTS; ++i) {
            if (!pending[dev][i] || hipEventQuery(ev[dev][i]) != hipSuccess) continue;
            pending[dev][i] = false;
            if (host[dev * SLOTS + i] != 0) {
                g_two_launches.store(1);
                if (!warned) {
                    warned = true;
                    fprintf(stderr, "[mhla] a dK/dV tile of the fused token-gradient launch gave up waiting for its dQ tile's hand-over (that "
                                    "backward's dk is NaN); this process runs the two roles as two launches from now on\n");
                }
            }
        }
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        (void)hipGetLastError();   // (hipEventQuery's hipErrorNotReady is not an error of the caller)
    }
    static int device_of() { int d = 0; (void)hipGetDevice(&d); return d; }   // (the caller's current device: the one its stream and workspace live on)
    void watch(const int* err_word, hipStream_t st) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }   // a graph cannot adapt anyway
        const int dev = device_of();
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= MAXDEV) { unarmed("device index beyond the watch's table"); return; }
        if (!host && hipHostMalloc((void**)&host, MAXDEV * SLOTS * sizeof(int), hipHostMallocPortable) != hipSuccess) { host = nullptr; unarmed("no pinned host memory"); return; }
        poll(dev);
        int i = next[dev];
        for (int t = 0; t < SLOTS && pending[dev][i]; ++t) i = (i + 1) % SLOTS;
        if (pending[dev][i]) return;       // every slot still in flight: skip this one
        next[dev] = (i + 1) % SLOTS;
        if (!ev[dev][i] && hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming) != hipSuccess) { ev[dev][i] = nullptr; unarmed("hipEventCreate failed"); return; }
        host[dev * SLOTS + i] = 0;
        if (hipMemcpyAsync(&host[dev * SLOTS + i], err_word, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[dev][i], st) == hipSuccess)
            pending[dev][i] = true;
        else
            unarmed("copy / event record on the launch stream failed");
    }
    void check(hipStream_t st) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // (no event queries while the caller's stream is being captured)
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
        const int dev = device_of();
        std::lock_guard<std::mutex> lk(mu);
        if (host && dev >= 0 && dev < MAXDEV) poll(dev);
    }
};
HandoverWatch g_watch;

// Error word of the fused token-gradient launch: the LAST 16 bytes of the backward workspace (mhla_blockmix_bwd_ws_bytes), behind
// the carve of every path a problem may take.  Every backward of a shape that can take the fast path leaves a defined word there:
// the fast path clears it in k_fs_dw and raises it in k_tile_bwd; when such a shape falls back to another path (misaligned views,
// rotary prologue) the dispatcher clears it with a 4-byte memset -- so mhla_blockmix_bwd_status never reads an unrelated word
// (round-3 ADVICE: it used to infer the path from the shape alone).
size_t bwd_ws_body_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    const size_t gen = bm_carve(nullptr, B, H, M, S, D, bm_sumfmt(M, S, D, dtype, flags), sp_shape_ok(D, flags), bm_olo(M, S, D, dtype, flags)).total_bwd;
    const bool fast = fast_shape_ok(M, D, dtype, split != 0, flags) && !(flags & MHLA_FLAG_FORCE_GENERIC);
    return (std::max(gen, fast ? fast_carve(nullptr, B, H, M, S).total_bwd : (size_t)0) + 15) & ~(size_t)15;
}
int* bwd_err_word(void* ws, int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    return (int*)((char*)ws + bwd_ws_body_bytes(B, H, M, S, D, dtype, split, flags));
}

}  // namespace

extern "C" {


int mhla_abi_version(void) { return MHLA_ABI_VERSION; }

// the device-code options this library was compiled with (mhla_amd/build.py passes them as -DMHLA_BUILD_FLAGS): the loader
// refuses a library whose flags differ from the ones the package expects (no packed-fp32 VALU code)
#ifndef MHLA_BUILD_FLAGS
#define MHLA_BUILD_FLAGS "unknown"
#endif
const char* mhla_build_flags(void) { return MHLA_BUILD_FLAGS; }

void mhla_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

// Waits for the recorded events, writes "name count total_ms" lines (one per kernel name) and clears.
int mhla_prof_report(char* buf, size_t cap) {
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_prof);
    }
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& r : recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        auto& a = agg[r.name];
        a.first += 1;
        a.second += ms;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    std::string out;
    char line[256];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (!buf || cap == 0) return (int)out.size();
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
    return (int)n;
}
const char* mhla_last_error(void) { return g_err; }

// Process-wide options (the two environment variables of the same meaning are read once, when the library is loaded):
//   "bwd_two_launches"   1: the backward's dQ and dK/dV tile roles as two launches (also latched by the library itself after a
//                        hand-over expired), 0: one fused launch
//   "debug_drop_signal"  testing aid: the dQ role does not raise its hand-over flags
//   "fp32_summaries"     1: 16-bit tensors at the default arithmetic keep their block summaries as fp32 in the workspace instead of
//                        24-bit floats (split.hpp p24) -- a measurement aid; set it BETWEEN calls, not between a forward and the
//                        backward that reuses its state
// Returns the previous value, or MHLA_EINVAL for an unknown name.
int mhla_set_option(const char* name, int value) {
    if (name && !strcmp(name, "bwd_two_launches")) return g_two_launches.exchange(value != 0);
    if (name && !strcmp(name, "debug_drop_signal")) return g_drop_signal.exchange(value != 0);
    if (name && !strcmp(name, "fp32_summaries")) return g_no_p24.exchange(value != 0);
    if (name && !strcmp(name, "recut_kernels")) return g_recut.exchange(value != 0);
    return fail(MHLA_EINVAL, "mhla_set_option: unknown option '%s'", name ? name : "(null)");
}

void mhla_debug_set_trace(void* buf) { g_trace = (unsigned long long*)buf; }

// 1 when mhla_blockmix_fwd leaves reusable block summaries in its workspace for this problem (pass it as fwd_ws to the
// backward), 0 when the forward is stateless (generic recompute / small-sequence path).
int mhla_blockmix_fwd_keeps_state(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    (void)B; (void)H;
    if (flags & MHLA_FLAG_FORCE_GENERIC) return 0;
    if ((sn_shape_ok(M, S, D, dtype, split != 0) || snf_shape_ok(M, S, D, dtype, split != 0)) && !(flags & MHLA_FLAG_NO_SMALLN)) return 0;
    if (fast_shape_ok(M, D, dtype, split != 0, flags)) return 1;
    return sp_shape_ok(D, flags) ? 1 : 0;   // split-operand path: KV, G, z, ksum, 1/n (fp32)
}

// Upper bound over the paths the library may take for this problem (the fast path needs less).
// (the fast path needs less than the split-operand path, but which one runs also depends on the alignment of the views, which
// these queries do not see: the bound covers both)
// (MHLA_FLAG_NO_BWD_STATE: the forward neither writes nor carves the O-residual region)
static bool fwd_olo(int M, int S, int D, int dtype, unsigned flags) { return bm_olo(M, S, D, dtype, flags) && !(flags & MHLA_FLAG_NO_BWD_STATE); }
size_t mhla_blockmix_fwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    const size_t gen = bm_carve(nullptr, B, H, M, S, D, bm_sumfmt(M, S, D, dtype, flags), sp_shape_ok(D, flags), fwd_olo(M, S, D, dtype, flags)).total_fwd;
    if (fast_shape_ok(M, D, dtype, split != 0, flags) && !(flags & MHLA_FLAG_FORCE_GENERIC)) return std::max(gen, fast_carve(nullptr, B, H, M, S).total_fwd);
    return gen;
}
size_t mhla_blockmix_bwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    return bwd_ws_body_bytes(B, H, M, S, D, dtype, split, flags) + 16;   // + the error word of the fused token-gradient launch
}

static int bm_fwd_impl(mhla_view q_num, mhla_view k_num, mhla_view v, mhla_view q_den, mhla_view k_den, const float* W,
                       int ldw, mhla_mview out, const int32_t* block_index, void* ws, size_t ws_bytes, int B, int H,
                       int M, int S, int D, int dtype, float eps, unsigned flags, void* stream, const float* rcos,
                       const float* rsin, long ldr, bool epi = false, const float* nw = nullptr, float neps = 0.f,
                       mhla_view gate = mhla_view{nullptr, 0, 0, 0}, int out_dtype = 0) {
    const bool normalize = q_den.ptr != nullptr;
    const bool split = normalize && (q_den.ptr != q_num.ptr || k_den.ptr != k_num.ptr);
    RC(bm_check(B, H, M, S, D, dtype, flags, normalize, split));
    CHECK_VIEW(q_num); CHECK_VIEW(k_num); CHECK_VIEW(v); CHECK_VIEW(out);
    if (normalize) { CHECK_VIEW(q_den); CHECK_VIEW(k_den); }
    if (!W || ldw < M) return fail(MHLA_EINVAL, "W null or ldw=%d < M=%d", ldw, M);
    if (!ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace null or not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (!normalize) { q_den = q_num; k_den = k_num; }
    const int relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0;
    const mhla_view outv{out.ptr, out.sb, out.sn, out.sh};
    if (epi && (!sp_shape_ok(D, flags) || dtype != MHLA_F32))
        return fail(MHLA_ENOTSUP, "fused norm/gate epilogue needs fp32 q, k, v, D %% 8 == 0 and the split-operand kernels (D=%d, dtype=%d)", D, dtype);
    if (rcos && !sp_shape_ok(D, flags)) return fail(MHLA_ENOTSUP, "fused rotary prologue needs D %% 8 == 0 and the split-operand kernels (D=%d, flags=%u)", D, flags);
    if (!rcos && !epi && sn_shape_ok(M, S, D, dtype, split) && !(flags & (MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) && view_ok16(q_num) &&
        view_ok16(k_num) && view_ok16(v) && view_ok16(outv)) {
        fast::SnArgs sa{};
        sa.q = cv(q_num); sa.k = cv(k_num); sa.v = cv(v); sa.out = cmv(out); sa.idx = block_index; sa.W = W; sa.ldw = ldw;
        sa.H = H; sa.M = M; sa.D = D; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
        const bool hl = !(flags & MHLA_FLAG_BF16_SUMMARIES), g = sa.idx != nullptr;
        if (D <= 64) RC(launch(hl ? sn_fwd_kernel<4, true>(g, M) : sn_fwd_kernel<4, false>(g, M), dim3(B * H), dim3(fast::SN_T), fast::sn_fwd_smem<4>(), st, hl ? "k_sn_fwd<4,hl>" : "k_sn_fwd<4>", sa));
        else         RC(launch(hl ? sn_fwd_kernel<5, true>(g, M) : sn_fwd_kernel<5, false>(g, M), dim3(B * H), dim3(fast::SN_T), fast::sn_fwd_smem<5>(), st, hl ? "k_sn_fwd<5,hl>" : "k_sn_fwd<5>", sa));
        return MHLA_OK;
    }
    if (!rcos && !epi && snf_shape_ok(M, S, D, dtype, split) && !(flags & (MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) && view_ok16(q_num) &&
        view_ok16(k_num) && view_ok16(v) && view_ok16(outv)) {
        // fp32 tensors in the DiT / ViT regime: the attention-form kernels with hi + lo bf16 operands (smalln_f32.hpp)
        fast::SnArgs sa{};
        sa.q = cv(q_num); sa.k = cv(k_num); sa.v = cv(v); sa.out = cmv(out); sa.idx = block_index; sa.W = W; sa.ldw = ldw;
        sa.H = H; sa.M = M; sa.D = D; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
        if (D <= 64) RC(launch(sa.idx ? fast::k_snf_fwd<4, true> : (M == 16 ? fast::k_snf_fwd<4, false, true> : fast::k_snf_fwd<4, false>), dim3(B * H), dim3(fast::SNF_T), fast::snf_smem<4>(), st, "k_snf_fwd<4>", sa));
        else         RC(launch(sa.idx ? fast::k_snf_fwd<5, true> : (M == 16 ? fast::k_snf_fwd<5, false, true> : fast::k_snf_fwd<5, false>), dim3(B * H), dim3(fast::SNF_T), fast::snf_smem<5>(), st, "k_snf_fwd<5>", sa));
        return MHLA_OK;
    }
    if (!rcos && !epi && fast_shape_ok(M, D, dtype, split, flags) && !(flags & MHLA_FLAG_FORCE_GENERIC) && view_ok16(q_num) && view_ok16(k_num) &&
        view_ok16(v) && view_ok16(outv)) {
        const FastWs f = fast_carve(ws, B, H, M, S);
        if (ws_bytes < f.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, f.total_fwd);
        fast::FsStateArgs sa{};
        sa.x = cv(k_num); sa.y = cv(v); sa.t = cv(q_num); sa.idx = block_index; sa.state = f.state; sa.ksum = f.ksum;
        sa.z_out = f.z; sa.H = H; sa.M = M; sa.S = S; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
        if (S <= 64) RC(launch_state1c<0>(sa, f.njg, B * H, st, "k_fs_state_fwd"));
        else RC(launch(fast::k_fs_state_fwd<0>, dim3(f.njg, B * H), dim3(fast::FT8), fast::FS_STATE_FWD_SMEM, st, "k_fs_state_fwd", sa));
        if (normalize)
            RC(launch(fast::k_fs_wz<0>, dim3((S + fast::WZ_C - 1) / fast::WZ_C, B * H), dim3(fast::FT), 0, st, "k_fs_wz<0>", W, ldw, (const float*)f.z, f.ninv, M, S, eps));
        fast::FsOutArgs oa{};
        oa.q = cv(q_num); oa.o = cmv(out); oa.idx = block_index; oa.W = W; oa.ldw = ldw; oa.state = f.state; oa.ninv = f.ninv;
        oa.H = H; oa.M = M; oa.S = S; oa.njg = f.njg; oa.eps = eps; oa.relu = relu; oa.normalize = normalize;
        oa.trace = g_trace.load();
        oa.cs = f.cs;
        if (f.cs > 1) RC(launch(fast::k_tile_out<16, true>, dim3(fast::tiles_per_bh(f.njg, 16) * B * H * f.cs), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_out", oa));
        else          RC(launch(fast::k_tile_out<16>, dim3(fast::tiles_per_bh(f.njg, 16) * B * H), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_out", oa));
        return MHLA_OK;
    }
    const BmWs w = bm_carve(ws, B, H, M, S, D, bm_sumfmt(M, S, D, dtype, flags), sp_shape_ok(D, flags), fwd_olo(M, S, D, dtype, flags));
    if (ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    BmCall c{};
    c.q_num = q_num; c.k_num = k_num; c.v = v; c.q_den = q_den; c.k_den = k_den; c.out = out; c.gate = gate;
    c.W = W; c.ldw = ldw; c.block_index = block_index; c.w = w; c.B = B; c.H = H; c.M = M; c.S = S; c.D = D; c.eps = eps; c.flags = flags;
    c.normalize = normalize; c.split = split; c.epi = epi; c.st = st; c.rcos = rcos; c.rsin = rsin; c.ldr = ldr;
    c.nw = nw; c.neps = neps; c.out_dtype = out_dtype;
    switch (dtype) {
        case MHLA_F32: return bm_fwd_typed<float, false>(c);
        case MHLA_BF16: return bm_sum16(D, dtype, flags) ? bm_fwd_typed<bf16_t, true>(c) : bm_fwd_typed<bf16_t, false>(c);
        case MHLA_F16: return bm_fwd_typed<f16_t, false>(c);
        default: return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    }
}

int mhla_blockmix_fwd(mhla_view q_num, mhla_view k_num, mhla_view v, mhla_view q_den, mhla_view k_den, const float* W,
                      int ldw, mhla_mview out, const int32_t* block_index, void* ws, size_t ws_bytes, int B, int H,
                      int M, int S, int D, int dtype, float eps, unsigned flags, void* stream) {
    return bm_fwd_impl(q_num, k_num, v, q_den, k_den, W, ldw, out, block_index, ws, ws_bytes, B, H, M, S, D, dtype, eps, flags,
                       stream, nullptr, nullptr, 0);
}

int mhla_blockmix_rope_fwd(mhla_view q, mhla_view k, mhla_view v, int normalize, const float* W, int ldw,
                           const float* rope_cos, const float* rope_sin, int64_t ld_rope, mhla_mview out,
                           const int32_t* block_index, void* ws, size_t ws_bytes, int B, int H, int M, int S, int D,
                           int dtype, float eps, unsigned flags, void* stream) {
    if (!rope_cos || !rope_sin) return fail(MHLA_EINVAL, "rope tables null");
    if (ld_rope < D / 2 || (ld_rope & 3) || ((uintptr_t)rope_cos | (uintptr_t)rope_sin) % 16)
        return fail(MHLA_EINVAL, "rope tables: ld=%lld must be >= D/2, a multiple of 4, and the tables 16-byte aligned", (long long)ld_rope);
    if (flags & MHLA_FLAG_RELU_EPS) return fail(MHLA_ENOTSUP, "relu prologue and rotary prologue are not combined (Wan applies relu before the norm output is roped: use mhla_qk_prologue)");
    const mhla_view none{nullptr, 0, 0, 0};
    return bm_fwd_impl(q, k, v, normalize ? q : none, normalize ? k : none, W, ldw, out, block_index, ws, ws_bytes, B, H, M, S, D,
                       dtype, eps, flags, stream, rope_cos, rope_sin, (long)ld_rope);
}

int mhla_blockmix_wan_fwd(mhla_view q, mhla_view k, mhla_view v, int normalize, const float* W, int ldw,
                          const float* rope_cos, const float* rope_sin, int64_t ld_rope, const float* norm_w, float norm_eps,
                          mhla_view gate, mhla_mview out, int out_dtype, const int32_t* block_index, void* ws,
                          size_t ws_bytes, int B, int H, int M, int S, int D, int dtype, float eps, unsigned flags,
                          void* stream) {
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return fail(MHLA_EINVAL, "rope_cos and rope_sin must be given together");
    if (rope_cos && (ld_rope < D / 2 || (ld_rope & 3) || ((uintptr_t)rope_cos | (uintptr_t)rope_sin) % 16))
        return fail(MHLA_EINVAL, "rope tables: ld=%lld must be >= D/2, a multiple of 4, and the tables 16-byte aligned", (long long)ld_rope);
    if (out_dtype < 0 || out_dtype > 2) return fail(MHLA_EINVAL, "unknown out_dtype %d", out_dtype);
    if (flags & MHLA_FLAG_RELU_EPS) return fail(MHLA_ENOTSUP, "relu prologue is not combined with the Wan prologue / epilogue (use mhla_qk_prologue)");
    if (gate.ptr) {
        if (((uintptr_t)gate.ptr) % 8 || ((gate.sb | gate.sn | gate.sh) & 3)) return fail(MHLA_EINVAL, "gate: pointer must be 8-byte aligned, strides multiples of 4");
    }
    if (norm_w && ((uintptr_t)norm_w) % 16) return fail(MHLA_EINVAL, "norm_w must be 16-byte aligned");
    const mhla_view none{nullptr, 0, 0, 0};
    return bm_fwd_impl(q, k, v, normalize ? q : none, normalize ? k : none, W, ldw, out, block_index, ws, ws_bytes, B, H, M, S, D,
                       dtype, eps, flags, stream, rope_cos, rope_sin, (long)ld_rope, true, norm_w, norm_eps, gate, out_dtype);
}

static int bm_bwd_impl(mhla_view q_num, mhla_view k_num, mhla_view v, mhla_view q_den, mhla_view k_den, const float* W,
                       int ldw, mhla_view out, mhla_view dout, mhla_mview dq_num, mhla_mview dk_num, mhla_mview dv,
                       mhla_mview dq_den, mhla_mview dk_den, float* dW, const int32_t* block_index, void* ws,
                       size_t ws_bytes, const void* fwd_ws, int B, int H, int M, int S, int D, int dtype, float eps,
                       unsigned flags, void* stream, const float* rcos = nullptr, const float* rsin = nullptr, long ldr = 0) {
    const bool normalize = q_den.ptr != nullptr;
    const bool split = normalize && (q_den.ptr != q_num.ptr || k_den.ptr != k_num.ptr);
    RC(bm_check(B, H, M, S, D, dtype, flags, normalize, split));
    CHECK_VIEW(q_num); CHECK_VIEW(k_num); CHECK_VIEW(v); CHECK_VIEW(dout);
    CHECK_VIEW(dq_num); CHECK_VIEW(dk_num); CHECK_VIEW(dv);
    if (normalize) { CHECK_VIEW(q_den); CHECK_VIEW(k_den); CHECK_VIEW(out); }
    if (split) { CHECK_VIEW(dq_den); CHECK_VIEW(dk_den); }
    if (!W || ldw < M || !dW) return fail(MHLA_EINVAL, "W/dW null or ldw=%d < M=%d", ldw, M);
    if (!ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace null or not 16-byte aligned");
    if (fwd_ws && ((uintptr_t)fwd_ws) % 16) return fail(MHLA_EINVAL, "fwd_ws not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (!normalize) { q_den = q_num; k_den = k_num; }
    const int relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0;
    {
        const mhla_view dqv{dq_num.ptr, dq_num.sb, dq_num.sn, dq_num.sh}, dkv_{dk_num.ptr, dk_num.sb, dk_num.sn, dk_num.sh},
            dvv{dv.ptr, dv.sb, dv.sn, dv.sh};
        if (!rcos && snf_shape_ok(M, S, D, dtype, split) && !(flags & (MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) && view_ok16(q_num) &&
            view_ok16(k_num) && view_ok16(v) && view_ok16(dout) && (!normalize || view_ok16(out)) && view_ok16(dqv) &&
            view_ok16(dkv_) && view_ok16(dvv)) {
            const size_t need = (size_t)B * H * M * M * 4;
            if (ws_bytes < need) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, need);
            fast::SnArgs sa{};
            sa.q = cv(q_num); sa.k = cv(k_num); sa.v = cv(v); sa.o = normalize ? cv(out) : cv(q_num); sa.dout = cv(dout);
            sa.dq = cmv(dq_num); sa.dk = cmv(dk_num); sa.dv = cmv(dv); sa.idx = block_index; sa.W = W; sa.ldw = ldw;
            sa.dwp = (float*)ws; sa.H = H; sa.M = M; sa.D = D; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
            if (D <= 64) RC(launch(sa.idx ? fast::k_snf_bwd<4, true> : (M == 16 ? fast::k_snf_bwd<4, false, true> : fast::k_snf_bwd<4, false>), dim3(B * H), dim3(fast::SNF_T), fast::snf_smem<4>(), st, "k_snf_bwd<4>", sa));
            else         RC(launch(sa.idx ? fast::k_snf_bwd<5, true> : (M == 16 ? fast::k_snf_bwd<5, false, true> : fast::k_snf_bwd<5, false>), dim3(B * H), dim3(fast::SNF_T), fast::snf_smem<5>(), st, "k_snf_bwd<5>", sa));
            RC(launch(fast::k_sn_dw_reduce, dim3(M * M), dim3(256), 0, st, "k_sn_dw_reduce", (const float*)ws, dW, M * M, B * H));
            return MHLA_OK;
        }
        if (!rcos && sn_shape_ok(M, S, D, dtype, split) && !(flags & (MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) && view_ok16(q_num) &&
            view_ok16(k_num) && view_ok16(v) && view_ok16(dout) && (!normalize || view_ok16(out)) && view_ok16(dqv) &&
            view_ok16(dkv_) && view_ok16(dvv)) {
            const size_t need = (size_t)B * H * M * M * 4;
            if (ws_bytes < need) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, need);
            fast::SnArgs sa{};
            sa.q = cv(q_num); sa.k = cv(k_num); sa.v = cv(v); sa.o = normalize ? cv(out) : cv(q_num); sa.dout = cv(dout);
            sa.dq = cmv(dq_num); sa.dk = cmv(dk_num); sa.dv = cmv(dv); sa.idx = block_index; sa.W = W; sa.ldw = ldw;
            sa.dwp = (float*)ws; sa.H = H; sa.M = M; sa.D = D; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
            sa.trace = g_trace.load();
            const bool hl = !(flags & MHLA_FLAG_BF16_SUMMARIES), g = sa.idx != nullptr;
            const int nth = hl ? fast::SN_TB * (3 - SN_BWD_NB) : fast::SN_TB;
            if (D <= 64) RC(launch(hl ? sn_bwd_kernel<4, true>(g, M) : sn_bwd_kernel<4, false>(g, M), dim3(B * H), dim3(nth), hl ? fast::sn_bwd_smem<4, SN_BWD_NB>() : fast::sn_bwd_smem<4>(), st, hl ? "k_sn_bwd<4,hl>" : "k_sn_bwd<4>", sa));
            else         RC(launch(hl ? sn_bwd_kernel<5, true>(g, M) : sn_bwd_kernel<5, false>(g, M), dim3(B * H), dim3(nth), hl ? fast::sn_bwd_smem<5, SN_BWD_NB>() : fast::sn_bwd_smem<5>(), st, hl ? "k_sn_bwd<5,hl>" : "k_sn_bwd<5>", sa));
            RC(launch(fast::k_sn_dw_reduce, dim3(M * M), dim3(256), 0, st, "k_sn_dw_reduce", (const float*)ws, dW, M * M, B * H));
            return MHLA_OK;
        }
        if (!rcos && fast_shape_ok(M, D, dtype, split, flags) && !(flags & MHLA_FLAG_FORCE_GENERIC) && view_ok16(q_num) && view_ok16(k_num) &&
            view_ok16(v) && view_ok16(dout) && (!normalize || view_ok16(out)) && view_ok16(dqv) && view_ok16(dkv_) && view_ok16(dvv)) {
            FastWs f = fast_carve(ws, B, H, M, S);
            const size_t need = mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dtype, split, flags);
            if (ws_bytes < need) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, need);
            f.err = bwd_err_word(ws, B, H, M, S, D, dtype, split, flags);
            const fast::u16* state = f.state;
            const float *z = f.z, *ksum = f.ksum, *ninv = f.ninv;
            if (fwd_ws) {   // forward workspace retained by the caller: reuse KV^T, z, ksum, 1/n
                const FastWs ff = fast_carve(const_cast<void*>(fwd_ws), B, H, M, S);
                state = ff.state; z = ff.z; ksum = ff.ksum; ninv = ff.ninv;
            } else {
                fast::FsStateArgs sa{};
                sa.x = cv(k_num); sa.y = cv(v); sa.t = cv(q_num); sa.idx = block_index; sa.state = f.state; sa.ksum = f.ksum;
                sa.z_out = f.z; sa.H = H; sa.M = M; sa.S = S; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
                if (S <= 64) RC(launch_state1c<0>(sa, f.njg, B * H, st, "k_fs_state_fwd"));
        else RC(launch(fast::k_fs_state_fwd<0>, dim3(f.njg, B * H), dim3(fast::FT8), fast::FS_STATE_FWD_SMEM, st, "k_fs_state_fwd", sa));
                if (normalize)
                    RC(launch(fast::k_fs_wz<0>, dim3((S + fast::WZ_C - 1) / fast::WZ_C, B * H), dim3(fast::FT), 0, st, "k_fs_wz<0>", W, ldw, (const float*)f.z, f.ninv, M, S, eps));
            }
            fast::FsStateArgs ga{};
            ga.x = cv(q_num); ga.y = cv(dout); ga.t = cv(out); ga.idx = block_index; ga.W = W; ga.ldw = ldw; ga.ninv = ninv;
            ga.state = f.dstate; ga.dn = f.dn; ga.H = H; ga.M = M; ga.S = S; ga.eps = eps; ga.relu = relu; ga.normalize = normalize;
            if (S <= 64) RC(launch_state1c<1>(ga, f.njg, B * H, st, "k_fs_state<1>"));
            else RC(launch(fast::k_fs_state<1>, dim3(f.njg, B * H), dim3(fast::FT8), fast::FS_STATE_SMEM, st, "k_fs_state<1>", ga));
            // dW needs only dG^T, KV^T, dn and z, all complete here; dz = W^T dn (needed by the token-gradient kernels) rides in
            // the same launch as extra workgroups
            fast::FsDwArgs da{f.dstate, state, normalize ? f.dn : nullptr, z, f.dwp, M, S, f.njg, W, ldw, f.dz, f.done, f.err, f.ntt * f.cs};
            hipStream_t sd = st;
            const int nwz = normalize ? (S + fast::WZ_C - 1) / fast::WZ_C : 0;
            RC(launch(fast::k_fs_dw<>, dim3(fast::DW_SPLIT + nwz, B * H), dim3(fast::FT8), fast::FS_DW_SMEM, sd, "k_fs_dw", da));
            fast::FsTokArgs ta{};
            ta.q = cv(q_num); ta.k = cv(k_num); ta.v = cv(v); ta.dout = cv(dout); ta.dq = cmv(dq_num); ta.dk = cmv(dk_num);
            ta.dv = cmv(dv); ta.idx = block_index; ta.W = W; ta.ldw = ldw; ta.state = state; ta.dstate = f.dstate; ta.ninv = ninv;
            ta.dz = f.dz; ta.ksum = ksum; ta.H = H; ta.M = M; ta.S = S; ta.njg = f.njg; ta.eps = eps; ta.relu = relu;
            ta.normalize = normalize;
            ta.dksum = f.dksum;
            const long ntile_wgs = (long)f.ntt * B * H * f.cs;
            ta.cs = f.cs; ta.dks_part = (long)al4((size_t)B * H * M * 64);
            unsigned long long* tr = g_trace.load();   // regions: [0] k_t16_out, [1] dQ role, [2] dK/dV role (record = ntiles + blockIdx.x)
            ta.trace = tr ? tr + ntile_wgs * fast::TRACE_SLOTS : nullptr;
            ta.dwp = f.dwp; ta.dW = dW; ta.nparts = B * H * fast::DW_SPLIT; ta.ntiles = (int)ntile_wgs; ta.done = f.done;
            ta.err = f.err;
            ta.drop_signal = g_drop_signal.load();   // testing aid for the bounded wait (mhla_set_option)
            g_watch.check(st);                        // did an earlier fused launch of this process report an expired hand-over?
            const auto tile_bwd = f.cs > 1 ? fast::k_tile_bwd<16, true> : fast::k_tile_bwd<16, false>;   // (multi-chunk blocks: cs workgroups per tile)
            if (g_two_launches.load()) {   // (the kernel boundary orders dksum and the flags: the wait returns at its first poll)
                RC(launch(tile_bwd, dim3((unsigned)ntile_wgs), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_bwd_dq", ta));
                ta.x0 = (int)ntile_wgs;
                RC(launch(tile_bwd, dim3((unsigned)ntile_wgs + fast::DWR_WGS), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_bwd_dkv", ta));
                return MHLA_OK;
            }
            RC(launch(tile_bwd, dim3((unsigned)(2 * ntile_wgs) + fast::DWR_WGS), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_bwd", ta));
            if (normalize) g_watch.watch(f.err, st);
            return MHLA_OK;
        }
    }
    BmWs w = bm_carve(ws, B, H, M, S, D, bm_sumfmt(M, S, D, dtype, flags), sp_shape_ok(D, flags), bm_olo(M, S, D, dtype, flags));
    if (ws_bytes < w.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bwd);
    if (fast_shape_ok(M, D, dtype, split, flags) && !(flags & MHLA_FLAG_FORCE_GENERIC)) {
        // a fast-path shape on another path (misaligned views, rotary prologue): leave a defined error word for the status call
        if (ws_bytes < mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dtype, split, flags))
            return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dtype, split, flags));
        if (hipMemsetAsync(bwd_err_word(ws, B, H, M, S, D, dtype, split, flags), 0, 4, st) != hipSuccess)
            return fail(MHLA_ELAUNCH, "hipMemsetAsync of the error word failed");
    }
    // the forward's KV, G, z, ksum, 1/n are still in its workspace (only when the shape cannot have taken the bf16 fast path,
    // whose workspace has another layout)
    const bool reuse = fwd_ws && sp_shape_ok(D, flags) && (rcos || !fast_shape_ok(M, D, dtype, split, flags));
    unsigned short* const olo_own = w.olo;
    if (reuse) {
        const BmWs f = bm_carve(const_cast<void*>(fwd_ws), B, H, M, S, D, bm_sumfmt(M, S, D, dtype, flags), sp_shape_ok(D, flags), fwd_olo(M, S, D, dtype, flags));
        w.kv = f.kv; w.g = f.g; w.z = f.z; w.ksum = f.ksum; w.ninv = f.ninv;
        if (f.olo) w.olo = f.olo;   // (a forward given MHLA_FLAG_NO_BWD_STATE has no residual region: the backward recomputes it into its own)
    }
    BmCall c{};
    c.olo_own = olo_own;
    c.q_num = q_num; c.k_num = k_num; c.v = v; c.q_den = q_den; c.k_den = k_den; c.outv = out; c.dout = dout;
    c.dq_num = dq_num; c.dk_num = dk_num; c.dv = dv; c.dq_den = dq_den; c.dk_den = dk_den; c.dW = dW;
    c.W = W; c.ldw = ldw; c.block_index = block_index; c.w = w; c.B = B; c.H = H; c.M = M; c.S = S; c.D = D; c.eps = eps; c.flags = flags;
    c.normalize = normalize; c.split = split; c.reuse = reuse; c.st = st; c.rcos = rcos; c.rsin = rsin; c.ldr = ldr;
    switch (dtype) {
        case MHLA_F32: return bm_bwd_typed<float, false>(c);
        case MHLA_BF16: return bm_sum16(D, dtype, flags) ? bm_bwd_typed<bf16_t, true>(c) : bm_bwd_typed<bf16_t, false>(c);
        case MHLA_F16: return bm_bwd_typed<f16_t, false>(c);
        default: return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    }
}

int mhla_blockmix_bwd(mhla_view q_num, mhla_view k_num, mhla_view v, mhla_view q_den, mhla_view k_den, const float* W,
                      int ldw, mhla_view out, mhla_view dout, mhla_mview dq_num, mhla_mview dk_num, mhla_mview dv,
                      mhla_mview dq_den, mhla_mview dk_den, float* dW, const int32_t* block_index, void* ws,
                      size_t ws_bytes, const void* fwd_ws, int B, int H, int M, int S, int D, int dtype, float eps,
                      unsigned flags, void* stream) {
    return bm_bwd_impl(q_num, k_num, v, q_den, k_den, W, ldw, out, dout, dq_num, dk_num, dv, dq_den, dk_den, dW, block_index, ws,
                       ws_bytes, fwd_ws, B, H, M, S, D, dtype, eps, flags, stream);
}

// Backward of mhla_blockmix_rope_fwd: q, k are the UN-rotated tensors the forward was given; the kernels rotate them on load
// where the rotated ones are needed (dG = Q_rot^T dP, dV = K_rot dKV) and turn the gradients of the rotated tensors back before
// the normaliser's part is added, so dq / dk are the gradients w.r.t. q / k themselves -- no rotated copies, no second
// gradient pair.  fp32 tensors, D % 8 == 0 (the split-operand kernels).
int mhla_blockmix_rope_bwd(mhla_view q, mhla_view k, mhla_view v, int normalize, const float* W, int ldw,
                           const float* rope_cos, const float* rope_sin, int64_t ld_rope, mhla_view out, mhla_view dout,
                           mhla_mview dq, mhla_mview dk, mhla_mview dv, float* dW, const int32_t* block_index, void* ws,
                           size_t ws_bytes, const void* fwd_ws, int B, int H, int M, int S, int D, int dtype, float eps,
                           unsigned flags, void* stream) {
    if (!rope_cos || !rope_sin) return fail(MHLA_EINVAL, "rope tables null");
    if (ld_rope < D / 2 || (ld_rope & 3) || ((uintptr_t)rope_cos | (uintptr_t)rope_sin) % 16)
        return fail(MHLA_EINVAL, "rope tables: ld=%lld must be >= D/2, a multiple of 4, and the tables 16-byte aligned", (long long)ld_rope);
    if (dtype != MHLA_F32 || !sp_shape_ok(D, flags))
        return fail(MHLA_ENOTSUP, "rotary backward needs fp32 tensors and D %% 8 == 0 (D=%d, dtype=%d, flags=%u)", D, dtype, flags);
    if (flags & MHLA_FLAG_RELU_EPS) return fail(MHLA_ENOTSUP, "relu prologue and rotary prologue are not combined");
    const mhla_view none{nullptr, 0, 0, 0};
    const mhla_mview mnone{nullptr, 0, 0, 0};
    return bm_bwd_impl(q, k, v, normalize ? q : none, normalize ? k : none, W, ldw, out, dout, dq, dk, dv, mnone, mnone, dW, block_index,
                       ws, ws_bytes, fwd_ws, B, H, M, S, D, dtype, eps, flags, stream, rope_cos, rope_sin, (long)ld_rope);
}

// Which kernel family, summary format and launch sequence serve a block-mix problem (16-byte aligned views assumed; a misaligned view
// of a small-sequence / fast-path shape falls through to the split-operand family): the dispatcher's own predicates, as text --
// "family=...; summaries=...; fwd=k1,k2,...; bwd=k1,k2,..." -- so that tests and tools can assert the path a BASELINE configuration
// takes instead of inferring it from timings (DESIGN.md section 0a is the same table).  Returns the length written (without the NUL).
int mhla_describe_dispatch(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags, char* buf, size_t cap) {
    RC(bm_check(B, H, M, S, D, dtype, flags, true, split != 0));
    std::string fam, sum, fwd, bwd;
    const bool gen = (flags & MHLA_FLAG_FORCE_GENERIC) != 0, nosn = gen || (flags & MHLA_FLAG_NO_SMALLN);
    if (!nosn && snf_shape_ok(M, S, D, dtype, split != 0)) {
        fam = "small-sequence fp32 (attention form, one launch per direction)";
        sum = "none (score tiles in LDS as bf16 hi + lo pairs)";
        fwd = D <= 64 ? "k_snf_fwd<4>" : "k_snf_fwd<5>"; bwd = std::string(D <= 64 ? "k_snf_bwd<4>" : "k_snf_bwd<5>") + " k_sn_dw_reduce";
    } else if (!nosn && sn_shape_ok(M, S, D, dtype, split != 0)) {
        fam = "small-sequence bf16 (attention form, one launch per direction)";
        const bool hl = !(flags & MHLA_FLAG_BF16_SUMMARIES);
        sum = hl ? "none (score tiles in LDS as bf16 hi + lo pairs)" : "none (score tiles as single bf16: reduced precision)";
        const std::string dt = D <= 64 ? "4" : "5";
        fwd = "k_sn_fwd<" + dt + (hl ? ",hl>" : ">"); bwd = "k_sn_bwd<" + dt + (hl ? ",hl>" : ">") + " k_sn_dw_reduce";
    } else if (!gen && fast_shape_ok(M, D, dtype, split != 0, flags)) {
        fam = "bf16 fast path (fused mixing + token tiles)";
        sum = "bf16 (single bf16 values, 8-block interleaved: reduced precision, opt-in)";
        fwd = "k_fs_state_fwd k_fs_wz<0> k_t16_out"; bwd = "k_fs_state<1> k_fs_dw k_t16_bwd";
    } else if (sp_shape_ok(D, flags)) {
        const int fmt = bm_sumfmt(M, S, D, dtype, flags);
        const long E = (long)D * D;
        const bool s16 = fmt == SF_BF16, mixr = (M > 32 || !s16) && M <= 256 && E % (s16 ? 128 : 64) == 0, evenS = (S % 2) == 0;
        fam = "split-operand (bf16 hi + lo MFMA operands)";
        sum = fmt == SF_H16 ? "h16 (fp16 payload x row multiplier: 11 significand bits, 2 bytes)" : fmt == SF_P24 ? "p24 (24-bit floats: 16 significand bits, 3 bytes)"
              : fmt == SF_BF16 ? "bf16 (single bf16 values: reduced precision, opt-in)" : "fp32 words";
        const bool wave16 = s16 && S == 16 && D == 64 && !split;
        const std::string st0 = wave16 ? "k_s16_state<0>" : "k_sp_state", out = wave16 ? "k_s16_out" : "k_sp_out";
        std::string mix0, mix1, dw;
        const std::string dwr = std::string("k_sp_dwr<") + (M <= 128 ? "2" : M <= 192 ? "3" : "4") + (fmt == SF_H16 ? ",h16>" : ">");   // whole-matrix dW kernel (incl. the <dn, z> term)
        if (fmt == SF_H16) {
            const bool m2 = sp_mixh2_applies(M, E, (long)B * H, S);
            mix0 = m2 ? "k_sp_mixh2<0>" : "k_sp_mixh<0>"; mix1 = M <= 128 ? "k_sp_mixh<1,dw>" : (m2 ? "k_sp_mixh2<1>" : "k_sp_mixh<1>"); dw = M <= 128 ? "" : dwr;
        }
        else if (s16 && M > 192 && mixr) { mix0 = "k_sp_mixr_dma<0>"; mix1 = "k_sp_mixr_dma<1>"; dw = dwr; }
        else if (mixr) {
            mix0 = "k_sp_mixr<0>";
            if (!s16 && M <= 128) { mix1 = "k_sp_mixr<1,dw>"; dw = ""; }
            else { mix1 = "k_sp_mixr<1>"; dw = s16 ? (M > 64 ? dwr : (M <= 16 ? "k_sp_dw<16>" : M <= 32 ? "k_sp_dw<32>" : "k_sp_dw")) : (fmt == SF_P24 ? "k_sp_dwt" : "k_sp_dw"); }
        } else { mix0 = "k_sp_mix<0>"; mix1 = "k_sp_mix<1>"; dw = s16 && M > 64 && M <= 256 && E % 64 == 0 ? dwr : (M <= 16 ? "k_sp_dw<16>" : M <= 32 ? "k_sp_dw<32>" : "k_sp_dw"); }
        const bool wzf = fmt == SF_BF16 ? (M > 192 && M <= 256 && S <= 16 && mixr) : (mixr && evenS);   // the normaliser's product rides in the mixing kernel
        fwd = st0 + " " + mix0 + (wzf ? "" : " k_wz<0>") + " " + out;
        bwd = std::string(wave16 ? "k_s16_state<1>" : "k_sp_state<1>") + " " + mix1 + (wzf ? "" : " k_wz<1>") + (dw.empty() ? "" : " " + dw) +
              ((dw.empty() && wzf) || dw == dwr ? "" : " k_dw") + " k_dw_reduce " + (wave16 ? "k_s16_bwd_dq k_s16_bwd_dkv" : "k_sp_bwd_dq k_sp_bwd_dkv");
    } else {
        fam = "generic (exact fp32 MFMA)";
        sum = "fp32 words (dense rows)";
        fwd = "k_bm_state<0> k_mix<0,0> k_wz<0> k_bm_out"; bwd = "k_bm_state<0> k_mix<0,0> k_wz<0> k_bm_state<1> k_wz<1> k_mix<1,0> k_dw k_dw_reduce k_bm_bwd_tok";   // (no state is kept: the backward recomputes KV, G, 1 / n)
    }
    const std::string txt = "family=" + fam + "; summaries=" + sum + "; fwd=" + fwd + "; bwd=" + bwd;
    if (buf && cap) {
        const size_t n = txt.size() < cap - 1 ? txt.size() : cap - 1;
        memcpy(buf, txt.data(), n);
        buf[n] = 0;
    }
    return (int)txt.size();
}

// Did the last mhla_blockmix_bwd on this workspace run into a hand-over flag that never arrived (fused.hpp, tile_wait)?
// Synchronises `stream`, reads the launch's error word.  MHLA_OK also for problems that do not take the in-launch hand-over.
int mhla_blockmix_bwd_status(const void* ws, size_t ws_bytes, int B, int H, int M, int S, int D, int dtype, int split, unsigned flags,
                             void* stream) {
    if (!ws) return fail(MHLA_EINVAL, "workspace null");
    if (!fast_shape_ok(M, D, dtype, split != 0, flags) || (flags & MHLA_FLAG_FORCE_GENERIC) ||
        (sn_shape_ok(M, S, D, dtype, split != 0) && !(flags & MHLA_FLAG_NO_SMALLN)))
        return MHLA_OK;
    const size_t need = mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dtype, split, flags);
    if (ws_bytes < need) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    int word = 0;
    hipError_t e = hipMemcpyAsync(&word, bwd_err_word(const_cast<void*>(ws), B, H, M, S, D, dtype, split, flags), sizeof(int),
                                  hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(MHLA_ELAUNCH, "mhla_blockmix_bwd_status: %s", hipGetErrorString(e));
    if (word != 0)
        return fail(MHLA_ELAUNCH, "k_t16_bwd: a dK/dV tile gave up waiting for its dQ tile's dksum rows (hand-over flag never raised); "
                                  "dk is invalid -- the library switches this process to two launches on its own at its next backward (mhla_set_option(\"bwd_two_launches\", 1) forces it)");
    return MHLA_OK;
}

}  // extern "C"
