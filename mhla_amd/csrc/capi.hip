// C ABI of libmhla_hip.so (see include/mhla_hip.h).  Validates arguments, carves the caller's
// workspace, picks template instantiations and enqueues kernels on the caller's stream.
#include "../../include/mhla_hip.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "blockmix.hpp"
#include "causal.hpp"
#include "causal_bf16.hpp"
#include "causal_mix.hpp"
#include "epilogue.hpp"
#include "fused.hpp"
#include "lepe.hpp"
#include "fused_tile16.hpp"
#include "smalln.hpp"
#include "split.hpp"

using namespace mhla;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline size_t al4(size_t n) { return (n + 3) & ~(size_t)3; }

// Optional per-launch timing (mhla_prof_*): hipEvents recorded on the launch stream around every
// kernel, so bench.py can report each kernel's average duration live (not only rocprof offline).
struct ProfRec { const char* name; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof;

template <typename K>
int launch(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t stream, const char* name, auto... args) {
    if (smem > 48 * 1024) {
        // opt in to > 48 KB of dynamic LDS once per (kernel, device); the driver call is kept off the steady-state launch path
        static std::mutex mu;
        static std::map<std::pair<const void*, int>, size_t> done;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const std::pair<const void*, int> key(reinterpret_cast<const void*>(kernel), dev);
        std::lock_guard<std::mutex> lk(mu);
        auto it = done.find(key);
        if (it == done.end() || it->second < smem) {
            hipError_t e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return fail(MHLA_ELAUNCH, "%s: hipFuncSetAttribute(%zu B LDS): %s", name, smem, hipGetErrorString(e));
            done[key] = smem;
        }
    }
    ProfRec rec{name, nullptr, nullptr};
    const bool prof = g_prof_on;
    if (prof) {
        (void)hipEventCreate(&rec.e0);
        (void)hipEventCreate(&rec.e1);
        (void)hipEventRecord(rec.e0, stream);
    }
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, args...);
    if (prof) {
        (void)hipEventRecord(rec.e1, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MHLA_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e));
    return MHLA_OK;
}

// debugging aid: per-workgroup phase timestamps of the tile kernels (mhla_debug_set_trace)
std::atomic<unsigned long long*> g_trace{nullptr};

View cv(const mhla_view& v) { return View{v.ptr, (long)v.sb, (long)v.sn, (long)v.sh}; }
MView cmv(const mhla_mview& v) { return MView{v.ptr, (long)v.sb, (long)v.sn, (long)v.sh}; }

int check_view(const char* name, const void* ptr, int64_t sb, int64_t sn, int64_t sh, int dtype) {
    if (!ptr) return fail(MHLA_EINVAL, "%s: null pointer", name);
    const int esz = dtype == MHLA_F32 ? 4 : 2;
    if (((uintptr_t)ptr) % (4 * esz) != 0) return fail(MHLA_EINVAL, "%s: pointer not %d-byte aligned", name, 4 * esz);
    if ((sb | sn | sh) & 3) return fail(MHLA_EINVAL, "%s: strides (%lld, %lld, %lld) must be multiples of 4 elements", name,
                                        (long long)sb, (long long)sn, (long long)sh);
    return MHLA_OK;
}
#define CHECK_VIEW(v) do { int rc_ = check_view(#v, (v).ptr, (v).sb, (v).sn, (v).sh, dtype); if (rc_) return rc_; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// E-slices of the dW GEMM so that the launch has ~1000+ workgroups (at most 16 slices, slices >= 256 columns)
constexpr int DW_MAX_SPLIT = 16;
int dw_splits(long wgs, long E) {
    long ns = (1024 + wgs - 1) / wgs;
    if (ns > DW_MAX_SPLIT) ns = DW_MAX_SPLIT;
    while (ns > 1 && E / ns < 256) --ns;
    return (int)(ns < 1 ? 1 : ns);
}

int dt_for(int D) { return D <= 32 ? 2 : D <= 64 ? 4 : D <= 80 ? 5 : D <= 96 ? 6 : D <= 128 ? 8 : 0; }

// dispatch on (dtype, DT)
#define DISPATCH_T(dtype, ...)                                                   \
    switch (dtype) {                                                             \
        case MHLA_F32: { using ET = float; __VA_ARGS__; break; }                  \
        case MHLA_BF16: { using ET = bf16_t; __VA_ARGS__; break; }                \
        case MHLA_F16: { using ET = f16_t; __VA_ARGS__; break; }                  \
        default: return fail(MHLA_EINVAL, "unknown dtype %d", dtype);            \
    }
#define DISPATCH_DT(dt, ...)                                                     \
    switch (dt) {                                                                \
        case 2: { constexpr int DT = 2; __VA_ARGS__; break; }                    \
        case 4: { constexpr int DT = 4; __VA_ARGS__; break; }                    \
        case 5: { constexpr int DT = 5; __VA_ARGS__; break; }                    \
        case 6: { constexpr int DT = 6; __VA_ARGS__; break; }                    \
        case 8: { constexpr int DT = 8; __VA_ARGS__; break; }                    \
        default: return fail(MHLA_ENOTSUP, "head dim tile %d not supported", dt);\
    }

struct BmWs {
    float *kv, *g, *z, *ksum, *ninv, *dg, *dkv, *dn, *dz, *dks, *dwp;
    size_t total_fwd, total_bwd;
};
// sum16: the D x D block summaries (KV, G, dG, dKV) are stored as bf16 (split-operand path on bf16 tensors): half the floats
BmWs bm_carve(void* ws, int B, int H, int M, int S, int D, bool sum16) {
    const size_t bh = (size_t)B * H, st = al4(sum16 ? (bh * M * D * D + 1) / 2 : bh * M * D * D), zs = al4(bh * M * S), ks = al4(bh * M * D);
    float* p = (float*)ws;
    BmWs w;
    w.kv = p; p += st;
    w.g = p; p += st;
    w.z = p; p += zs;
    w.ksum = p; p += ks;
    w.ninv = p; p += zs;
    w.total_fwd = (size_t)(p - (float*)ws) * 4;
    w.dg = p; p += st;
    w.dkv = p; p += st;
    w.dn = p; p += zs;
    w.dz = p; p += zs;
    w.dks = p; p += ks;
    w.dwp = p; p += al4(bh * M * M * DW_MAX_SPLIT);
    w.total_bwd = (size_t)(p - (float*)ws) * 4;
    return w;
}

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

struct FastWs {
    fast::u16 *state, *dstate;
    float *z, *ksum, *ninv, *dn, *dz, *dwp, *dksum;
    int* done;   // per 16-block tile: dQ done (k_tile_bwd)
    size_t total_fwd, total_bwd;
    int njg;
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
    w.dksum = (float*)p; p += al4(bh * M * 64) * 4;
    w.dwp = (float*)p; p += bh * fast::DW_SPLIT * 4096 * 4;
    w.done = (int*)p; p += al4(bh * ((w.njg + 1) / 2)) * 4;
    w.total_bwd = (size_t)(p - (char*)ws);
    return w;
}
bool view_ok16(const mhla_view& v) { return v.ptr && ((uintptr_t)v.ptr % 16) == 0 && ((v.sb | v.sn | v.sh) & 7) == 0; }
// split-bf16 MFMA kernels (split.hpp): head dims that are multiples of 8, any dtype
bool sp_shape_ok(int D, unsigned flags) { return (D & 7) == 0 && !(flags & MHLA_FLAG_FORCE_GENERIC); }
bool view_ok16m(const mhla_mview& v) { return v.ptr && ((uintptr_t)v.ptr % 16) == 0 && ((v.sb | v.sn | v.sh) & 7) == 0; }
// bf16-MFMA token kernels of the causal operator (causal_bf16.hpp)
// MHLA_CAUSAL_GENERIC is a testing aid that the parity tests flip inside one process (bf16 pipeline vs generic kernels on the
// same inputs), so it is looked up per call: one scan of the environment per operator call, beside five or more launches
bool cs_bf16_ok(int K, int V, int dtype) { return dtype == MHLA_BF16 && (K & 63) == 0 && (V & 63) == 0 && !getenv("MHLA_CAUSAL_GENERIC"); }
bool bm_sum16(int D, int dtype, unsigned flags) { return dtype == MHLA_BF16 && sp_shape_ok(D, flags); }
bool fast_shape_ok(int M, int D, int dtype, bool split) { return dtype == MHLA_BF16 && D == 64 && M <= 64 && !split; }
// small-sequence single-launch path (smalln.hpp): S = 16 tokens per block, at most 16 blocks, D <= 80
bool sn_shape_ok(int M, int S, int D, int dtype, bool split) {
    return dtype == MHLA_BF16 && S == 16 && M <= 16 && D <= 80 && (D & 7) == 0 && !split;
}

int bm_check(int B, int H, int M, int S, int D, int dtype, unsigned flags, bool normalize, bool split) {
    if (B <= 0 || H <= 0 || M <= 0 || S <= 0 || D <= 0) return fail(MHLA_EINVAL, "non-positive dimension B=%d H=%d M=%d S=%d D=%d", B, H, M, S, D);
    if (D % 4) return fail(MHLA_EINVAL, "D=%d must be a multiple of 4", D);
    if (!dt_for(D)) return fail(MHLA_ENOTSUP, "block-mix head dim D=%d > 128 not supported", D);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (flags & ~(MHLA_FLAG_RELU_EPS | MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) return fail(MHLA_EINVAL, "unknown flags 0x%x", flags);
    if ((flags & MHLA_FLAG_RELU_EPS) && split) return fail(MHLA_EINVAL, "MHLA_FLAG_RELU_EPS needs q_den/k_den to alias q_num/k_num");
    if ((size_t)B * H > 65535) return fail(MHLA_ENOTSUP, "B*H=%zu exceeds grid limit 65535", (size_t)B * H);
    (void)normalize;
    return MHLA_OK;
}

// KV/ksum/z, G for the forward and the recompute leg of the backward.
template <typename T, int DT>
int bm_state_and_mix(const mhla_view& q_num, const mhla_view& k_num, const mhla_view& v, const mhla_view& q_den,
                     const mhla_view& k_den, const float* W, int ldw, const int32_t* idx, const BmWs& w, int B, int H,
                     int M, int S, int D, float eps, unsigned flags, bool normalize, bool split, hipStream_t st,
                     const float* rcos = nullptr, const float* rsin = nullptr, long ldr = 0) {
    (void)q_num;
    StateArgs a{};
    a.rcos = rcos; a.rsin = rsin; a.ldr = ldr;
    a.x = cv(k_num); a.y = cv(v); a.kd = cv(k_den); a.qd = cv(q_den); a.idx = idx;
    a.out = w.kv; a.ksum = w.ksum; a.zo = w.z;
    a.H = H; a.M = M; a.S = S; a.D = D; a.eps = eps;
    a.relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0; a.normalize = normalize; a.split = split;
    MixArgs m{W, ldw, w.kv, w.g, M, (long)D * D};
    if (sp_shape_ok(D, flags)) {   // split-bf16 MFMA kernels (split.hpp)
        if (a.rcos) RC(launch(sp::k_sp_state<T, DT, 0, true>, dim3(M, B * H), dim3(NTHREADS), sp::sp_state_smem<DT>(), st, "k_sp_state<rope>", a));
        else        RC(launch(sp::k_sp_state<T, DT, 0>, dim3(M, B * H), dim3(NTHREADS), sp::sp_state_smem<DT>(), st, "k_sp_state", a));
        RC(launch(sp::k_sp_mix<0, sp::Sum16<T>::value>, dim3((unsigned)((m.E + sp::SPM_TE - 1) / sp::SPM_TE), (M + 63) / 64, B * H), dim3(NTHREADS), sp::sp_mix_smem<sp::Sum16<T>::value>(), st, "k_sp_mix<0>", m));
        if (normalize)
            RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
        return MHLA_OK;
    }
    RC(launch(k_bm_state<T, DT, 0>, dim3(M, B * H), dim3(NTHREADS), state_smem_floats<DT>() * 4, st, "k_bm_state<0>", a));
    dim3 grid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (M + MIX_TI - 1) / MIX_TI, B * H);
    RC(launch(k_mix<0, 0>, grid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,0>", m));
    if (normalize)
        RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
    return MHLA_OK;
}

struct CsWs {
    float *S, *P, *dP, *dS, *dwp, *diag;
    size_t total_fwd, total_bwd;
};
// esz: bytes per summary element (2 on the bf16 pipeline, 4 on the generic one)
// Launch plan of the resident-sequence mixing kernels (causal_mix.hpp): waves per workgroup (16 chunks each), workgroups and
// slices per workgroup.  One workgroup per CU at 8 waves (132 KB of LDS), two / four at 4 / 2 waves.
struct Mix2Plan { int nw, te, wgs, spw; long total; };
Mix2Plan mix2_plan(size_t bh, int n, long E, bool bwd) {
    static const char* const knob = getenv("MHLA_CAUSAL_MIX_TE");   // tuning knob, read once: forward slice width 128 (default) or 256
    Mix2Plan p{};
    p.nw = n <= 32 ? 2 : n <= 64 ? 4 : 8;
    p.te = (!bwd && knob && knob[0] == '2' && E % 256 == 0) ? 256 : 128;
    p.total = (long)bh * (E / p.te);
    // workgroups the chip holds at once: forward 2 tiles of 16 nw rows and 64 nw threads, backward 3 tiles and 128 nw threads
    const long slots = 256L * (8 / p.nw) * (bwd ? 1 : 256 / p.te);
    if (p.total <= 0) return p;   // (summaries smaller than a slice: the dispatcher does not take this path)
    const long wgs = std::min(p.total, slots);
    p.spw = (int)((p.total + wgs - 1) / wgs);
    p.wgs = (int)((p.total + p.spw - 1) / p.spw);
    return p;
}
bool cs_mix2_ok(int n, long E) {
    static const char* const knob = getenv("MHLA_CAUSAL_MIX");   // tuning knob, read once: "old" keeps k_csf_mix / k_csf_dw
    return n <= 128 && E % fast::MF_TE == 0 && !(knob && knob[0] == 'o');
}
CsWs cs_carve(void* ws, int B, int T, int H, int K, int V, int chunk, int esz) {
    const size_t bh = (size_t)B * H, n = (size_t)(T + chunk - 1) / chunk, st = al4(bh * n * K * V) * esz / 4;
    const size_t parts = std::max(bh * DW_MAX_SPLIT, (size_t)mix2_plan(bh, (int)n, (long)K * V, true).wgs);
    float* p = (float*)ws;
    CsWs w;
    w.S = p; p += st;
    w.P = p; p += st;
    w.total_fwd = (size_t)(p - (float*)ws) * 4;
    w.dP = p; p += st;
    w.dS = p; p += st;
    w.dwp = p; p += al4(parts * n * n);
    w.diag = p; p += al4(bh * n);
    w.total_bwd = (size_t)(p - (float*)ws) * 4;
    return w;
}
int cs_check(int B, int T, int H, int K, int V, int chunk, int dtype) {
    if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || V <= 0) return fail(MHLA_EINVAL, "non-positive dimension B=%d T=%d H=%d K=%d V=%d", B, T, H, K, V);
    if (chunk != 64) return fail(MHLA_ENOTSUP, "chunk=%d: only 64 is supported", chunk);
    if ((K | V) & 3) return fail(MHLA_EINVAL, "K=%d and V=%d must be multiples of 4", K, V);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if ((size_t)B * H > 65535) return fail(MHLA_ENOTSUP, "B*H=%zu exceeds grid limit 65535", (size_t)B * H);
    return MHLA_OK;
}

// S_j (or dP_i) = alpha X_j^T Y_j with 64x64 strips
template <typename T>
int cs_xty(const mhla_view& x, const mhla_view& y, float* out, float alpha, int B, int T_, int H, int n, int DX,
                  int DY, hipStream_t st) {
    StateArgs a{};
    a.x = cv(x); a.y = cv(y); a.out = out; a.H = H; a.M = n; a.S = CS; a.D = 64; a.DX = DX; a.DY = DY; a.T = T_;
    a.alpha = alpha;
    const int strips = ((DX + 63) / 64) * ((DY + 63) / 64);
    return launch(k_bm_state<T, 4, 2>, dim3(n, B * H, strips), dim3(NTHREADS), state_smem_floats<4>() * 4, st, "k_bm_state<2>", a);
}

int norm_check(const void* x, const void* y, int64_t rows, int D, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (rows <= 0 || D <= 0 || (D & 3) || D > 512) return fail(MHLA_EINVAL, "rows=%lld D=%d: need D %% 4 == 0 and D <= 512", (long long)rows, D);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    return MHLA_OK;
}
int norm_grid(int64_t rows) {   // backward: one dw partial row per workgroup, so the grid is capped
    int64_t g = (rows + 3) / 4;
    return (int)(g < 8192 ? g : 8192);
}
int norm_fwd_grid(int64_t rows, int rows_per_wave) {   // forward: a wave per row group, no grid-stride serialisation
    int64_t g = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
    return (int)(g < (1 << 20) ? g : (1 << 20));
}

}  // namespace

extern "C" {

int mhla_abi_version(void) { return MHLA_ABI_VERSION; }

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

void mhla_debug_set_trace(void* buf) { g_trace = (unsigned long long*)buf; }

// 1 when mhla_blockmix_fwd leaves reusable block summaries in its workspace for this problem (pass it as fwd_ws to the
// backward), 0 when the forward is stateless (generic recompute / small-sequence path).
int mhla_blockmix_fwd_keeps_state(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    (void)B; (void)H;
    if (flags & MHLA_FLAG_FORCE_GENERIC) return 0;
    if (sn_shape_ok(M, S, D, dtype, split != 0) && !(flags & MHLA_FLAG_NO_SMALLN)) return 0;
    if (fast_shape_ok(M, D, dtype, split != 0)) return 1;
    return sp_shape_ok(D, flags) ? 1 : 0;   // split-operand path: KV, G, z, ksum, 1/n (fp32)
}

// Upper bound over the paths the library may take for this problem (the fast path needs less).
// (the fast path needs less than the split-operand path, but which one runs also depends on the alignment of the views, which
// these queries do not see: the bound covers both)
size_t mhla_blockmix_fwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    const size_t gen = bm_carve(nullptr, B, H, M, S, D, bm_sum16(D, dtype, flags)).total_fwd;
    if (fast_shape_ok(M, D, dtype, split != 0) && !(flags & MHLA_FLAG_FORCE_GENERIC)) return std::max(gen, fast_carve(nullptr, B, H, M, S).total_fwd);
    return gen;
}
size_t mhla_blockmix_bwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags) {
    const size_t gen = bm_carve(nullptr, B, H, M, S, D, bm_sum16(D, dtype, flags)).total_bwd;
    if (fast_shape_ok(M, D, dtype, split != 0) && !(flags & MHLA_FLAG_FORCE_GENERIC)) return std::max(gen, fast_carve(nullptr, B, H, M, S).total_bwd);
    return gen;
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
        if (D <= 64) RC(launch(fast::k_sn_fwd<4>, dim3(B * H), dim3(fast::SN_T), fast::sn_fwd_smem<4>(), st, "k_sn_fwd<4>", sa));
        else         RC(launch(fast::k_sn_fwd<5>, dim3(B * H), dim3(fast::SN_T), fast::sn_fwd_smem<5>(), st, "k_sn_fwd<5>", sa));
        return MHLA_OK;
    }
    if (!rcos && !epi && fast_shape_ok(M, D, dtype, split) && !(flags & MHLA_FLAG_FORCE_GENERIC) && view_ok16(q_num) && view_ok16(k_num) &&
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
        RC(launch(fast::k_tile_out<16>, dim3(((f.njg + 1) / 2) * B * H), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_out", oa));
        return MHLA_OK;
    }
    const BmWs w = bm_carve(ws, B, H, M, S, D, bm_sum16(D, dtype, flags));
    if (ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    const int dt = dt_for(D);
    DISPATCH_T(dtype, DISPATCH_DT(dt, {
        RC((bm_state_and_mix<ET, DT>(q_num, k_num, v, q_den, k_den, W, ldw, block_index, w, B, H, M, S, D, eps, flags, normalize, split, st, rcos, rsin, ldr)));
        OutArgs o{};
        o.rcos = rcos; o.rsin = rsin; o.ldr = ldr;
        o.q = cv(q_num); o.o = cmv(out); o.idx = block_index; o.W = W; o.ldw = ldw; o.g = w.g; o.ninv = w.ninv;
        o.H = H; o.M = M; o.S = S; o.D = D; o.eps = eps;
        o.relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0; o.normalize = normalize;
        if (epi) {
            if constexpr (std::is_same<ET, float>::value) {
                o.nw = nw; o.neps = neps; o.gate = cv(gate);
                const dim3 g(M, B * H), blk(sp::SP_OUT_T);
                if (out_dtype == MHLA_BF16)     RC(launch(sp::k_sp_out<float, DT, bf16_t, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                else if (out_dtype == MHLA_F16) RC(launch(sp::k_sp_out<float, DT, f16_t, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                else                            RC(launch(sp::k_sp_out<float, DT, float, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
            }
        } else if (sp_shape_ok(D, flags))
            RC(launch(sp::k_sp_out<ET, DT>, dim3(M, B * H), dim3(sp::SP_OUT_T), sp::sp_out_smem<DT, sp::Sum16<ET>::value>(), st, "k_sp_out", o));
        else
            RC(launch(k_bm_out<ET, DT>, dim3(M, B * H), dim3(NTHREADS), out_smem_floats<DT>() * 4, st, "k_bm_out", o));
    }));
    return MHLA_OK;
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

int mhla_blockmix_bwd(mhla_view q_num, mhla_view k_num, mhla_view v, mhla_view q_den, mhla_view k_den, const float* W,
                      int ldw, mhla_view out, mhla_view dout, mhla_mview dq_num, mhla_mview dk_num, mhla_mview dv,
                      mhla_mview dq_den, mhla_mview dk_den, float* dW, const int32_t* block_index, void* ws,
                      size_t ws_bytes, const void* fwd_ws, int B, int H, int M, int S, int D, int dtype, float eps,
                      unsigned flags, void* stream) {
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
    const int dt = dt_for(D);
    const int relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0;
    {
        const mhla_view dqv{dq_num.ptr, dq_num.sb, dq_num.sn, dq_num.sh}, dkv_{dk_num.ptr, dk_num.sb, dk_num.sn, dk_num.sh},
            dvv{dv.ptr, dv.sb, dv.sn, dv.sh};
        if (sn_shape_ok(M, S, D, dtype, split) && !(flags & (MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN)) && view_ok16(q_num) &&
            view_ok16(k_num) && view_ok16(v) && view_ok16(dout) && (!normalize || view_ok16(out)) && view_ok16(dqv) &&
            view_ok16(dkv_) && view_ok16(dvv)) {
            const size_t need = (size_t)B * H * M * M * 4;
            if (ws_bytes < need) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, need);
            fast::SnArgs sa{};
            sa.q = cv(q_num); sa.k = cv(k_num); sa.v = cv(v); sa.o = normalize ? cv(out) : cv(q_num); sa.dout = cv(dout);
            sa.dq = cmv(dq_num); sa.dk = cmv(dk_num); sa.dv = cmv(dv); sa.idx = block_index; sa.W = W; sa.ldw = ldw;
            sa.dwp = (float*)ws; sa.H = H; sa.M = M; sa.D = D; sa.eps = eps; sa.relu = relu; sa.normalize = normalize;
            sa.trace = g_trace.load();
            if (D <= 64) RC(launch(fast::k_sn_bwd<4>, dim3(B * H), dim3(fast::SN_TB), fast::sn_bwd_smem<4>(), st, "k_sn_bwd<4>", sa));
            else         RC(launch(fast::k_sn_bwd<5>, dim3(B * H), dim3(fast::SN_TB), fast::sn_bwd_smem<5>(), st, "k_sn_bwd<5>", sa));
            RC(launch(fast::k_sn_dw_reduce, dim3(M * M), dim3(256), 0, st, "k_sn_dw_reduce", (const float*)ws, dW, M * M, B * H));
            return MHLA_OK;
        }
        if (fast_shape_ok(M, D, dtype, split) && !(flags & MHLA_FLAG_FORCE_GENERIC) && view_ok16(q_num) && view_ok16(k_num) &&
            view_ok16(v) && view_ok16(dout) && (!normalize || view_ok16(out)) && view_ok16(dqv) && view_ok16(dkv_) && view_ok16(dvv)) {
            const FastWs f = fast_carve(ws, B, H, M, S);
            if (ws_bytes < f.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, f.total_bwd);
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
            fast::FsDwArgs da{f.dstate, state, normalize ? f.dn : nullptr, z, f.dwp, M, S, f.njg, W, ldw, f.dz, f.done, (f.njg + 1) / 2};
            hipStream_t sd = st;
            const int nwz = normalize ? (S + fast::WZ_C - 1) / fast::WZ_C : 0;
            RC(launch(fast::k_fs_dw, dim3(fast::DW_SPLIT + nwz, B * H), dim3(fast::FT8), fast::FS_DW_SMEM, sd, "k_fs_dw", da));
            fast::FsTokArgs ta{};
            ta.q = cv(q_num); ta.k = cv(k_num); ta.v = cv(v); ta.dout = cv(dout); ta.dq = cmv(dq_num); ta.dk = cmv(dk_num);
            ta.dv = cmv(dv); ta.idx = block_index; ta.W = W; ta.ldw = ldw; ta.state = state; ta.dstate = f.dstate; ta.ninv = ninv;
            ta.dz = f.dz; ta.ksum = ksum; ta.H = H; ta.M = M; ta.S = S; ta.njg = f.njg; ta.eps = eps; ta.relu = relu;
            ta.normalize = normalize;
            ta.dksum = f.dksum;
            const long ntile_wgs = (long)((f.njg + 1) / 2) * B * H;
            unsigned long long* tr = g_trace.load();   // regions: [0] k_t16_out, [1] dQ role, [2] dK/dV role (record = ntiles + blockIdx.x)
            ta.trace = tr ? tr + ntile_wgs * fast::TRACE_SLOTS : nullptr;
            ta.dwp = f.dwp; ta.dW = dW; ta.nparts = B * H * fast::DW_SPLIT; ta.ntiles = (int)ntile_wgs; ta.done = f.done;
            RC(launch(fast::k_tile_bwd<16>, dim3((unsigned)(2 * ntile_wgs) + fast::DWR_WGS), dim3(fast::FT8), fast::tile_smem<16>(), st, "k_t16_bwd", ta));
            return MHLA_OK;
        }
    }
    BmWs w = bm_carve(ws, B, H, M, S, D, bm_sum16(D, dtype, flags));
    if (ws_bytes < w.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bwd);
    // the forward's KV, G, z, ksum, 1/n are still in its workspace (only when the shape cannot have taken the bf16 fast path,
    // whose workspace has another layout)
    const bool reuse = fwd_ws && sp_shape_ok(D, flags) && !fast_shape_ok(M, D, dtype, split);
    if (reuse) {
        const BmWs f = bm_carve(const_cast<void*>(fwd_ws), B, H, M, S, D, bm_sum16(D, dtype, flags));
        w.kv = f.kv; w.g = f.g; w.z = f.z; w.ksum = f.ksum; w.ninv = f.ninv;
    }
    DISPATCH_T(dtype, DISPATCH_DT(dt, {
        if (!reuse)
            RC((bm_state_and_mix<ET, DT>(q_num, k_num, v, q_den, k_den, W, ldw, block_index, w, B, H, M, S, D, eps, flags, normalize, split, st)));
        // dG_i = Q_i^T (dO_i / n_i), dn_i
        StateArgs a{};
        a.x = cv(q_num); a.y = cv(dout); a.o = cv(out); a.idx = block_index; a.W = W; a.ldw = ldw; a.ninv = w.ninv;
        a.out = w.dg; a.dn = w.dn; a.H = H; a.M = M; a.S = S; a.D = D; a.eps = eps;
        a.relu = relu; a.normalize = normalize; a.split = split;
        const int tiles = (M + 63) / 64;
        TokArgs t{};
        t.q = cv(q_num); t.k = cv(k_num); t.v = cv(v); t.qd = cv(q_den); t.kd = cv(k_den); t.dout = cv(dout);
        t.dq = cmv(dq_num); t.dk = cmv(dk_num); t.dv = cmv(dv); t.dqd = cmv(dq_den); t.dkd = cmv(dk_den);
        t.idx = block_index; t.W = W; t.ldw = ldw; t.g = w.g; t.dkv = w.dkv; t.ninv = w.ninv; t.dz = w.dz; t.ksum = w.ksum;
        t.dks = w.dks;
        t.H = H; t.M = M; t.S = S; t.D = D; t.eps = eps; t.relu = relu; t.normalize = normalize; t.split = split;
        if (sp_shape_ok(D, flags)) {   // split-bf16 MFMA kernels (split.hpp)
            const long E = (long)D * D;
            RC(launch(sp::k_sp_state<ET, DT, 1>, dim3(M, B * H), dim3(NTHREADS), sp::sp_state_smem<DT>(), st, "k_sp_state<1>", a));
            if (normalize)
                RC(launch(k_wz<1>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<1>", W, ldw, (const float*)w.dn, w.dz, M, S, 0.f));
            MixArgs m{W, ldw, w.dg, w.dkv, M, E};
            RC(launch(sp::k_sp_mix<1, sp::Sum16<ET>::value>, dim3((unsigned)((E + sp::SPM_TE - 1) / sp::SPM_TE), tiles, B * H), dim3(NTHREADS), sp::sp_mix_smem<sp::Sum16<ET>::value>(), st, "k_sp_mix<1>", m));
            int nsplit = dw_splits(tiles * tiles * B * H, E);
            if (nsplit > DW_MAX_SPLIT - 1) nsplit = DW_MAX_SPLIT - 1;   // one more part per (b, h) holds the <dn_i, z_j> term
            DwArgs d{w.dg, w.kv, E, nullptr, nullptr, 0, w.dwp, M, tiles, nsplit};
            if (M <= 16)      RC(launch(sp::k_sp_dw<sp::Sum16<ET>::value, 1>, dim3(1, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw<16>", d));
            else if (M <= 32) RC(launch(sp::k_sp_dw<sp::Sum16<ET>::value, 2>, dim3(1, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw<32>", d));
            else              RC(launch(sp::k_sp_dw<sp::Sum16<ET>::value>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw", d));
            int nparts = B * H * nsplit;
            if (normalize) {
                DwArgs dzz{w.dn, w.z, (long)S, nullptr, nullptr, 0, w.dwp + (size_t)nparts * M * M, M, tiles, 1};
                RC(launch(k_dw<0>, dim3(tiles * tiles, B * H, 1), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw", dzz));
                nparts += B * H;
            }
            if (M * M <= 1024) RC(launch(k_dw_reduce<0, 16>, dim3((M * M + 15) / 16), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                      (const float*)nullptr, dW, M, M, nparts, B * H));
            else RC(launch(k_dw_reduce<0>, dim3((M * M + 63) / 64), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                      (const float*)nullptr, dW, M, M, nparts, B * H));
            RC(launch(sp::k_sp_bwd_dq<ET, DT>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, sp::Sum16<ET>::value>(), st, "k_sp_bwd_dq", t));
            RC(launch(sp::k_sp_bwd_dkv<ET, DT>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, sp::Sum16<ET>::value>(), st, "k_sp_bwd_dkv", t));
            break;
        }
        RC(launch(k_bm_state<ET, DT, 1>, dim3(M, B * H), dim3(NTHREADS), state_smem_floats<DT>() * 4, st, "k_bm_state<1>", a));
        if (normalize)
            RC(launch(k_wz<1>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<1>", W, ldw, (const float*)w.dn, w.dz, M, S, 0.f));
        // dKV = W^T dG
        MixArgs m{W, ldw, w.dg, w.dkv, M, (long)D * D};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (M + MIX_TI - 1) / MIX_TI, B * H);
        RC(launch(k_mix<1, 0>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<1,0>", m));
        // dW = sum_bh (<dG_i, KV_j> + <dn_i, z_j>)
        const int nsplit = dw_splits(tiles * tiles * B * H, (long)D * D);
        DwArgs d{w.dg, w.kv, (long)D * D, normalize ? w.dn : nullptr, normalize ? w.z : nullptr, (long)S, w.dwp, M, tiles, nsplit};
        RC(launch(k_dw<0>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw", d));
        if (M * M <= 1024) RC(launch(k_dw_reduce<0, 16>, dim3((M * M + 15) / 16), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                  (const float*)nullptr, dW, M, M, B * H * nsplit, B * H));
        else RC(launch(k_dw_reduce<0>, dim3((M * M + 63) / 64), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                  (const float*)nullptr, dW, M, M, B * H * nsplit, B * H));
        // dQ, dK, dV
        RC(launch(k_bm_bwd_tok<ET, DT>, dim3(M, B * H), dim3(NTHREADS), tok_smem_floats<DT>() * 4, st, "k_bm_bwd_tok", t));
    }));
    return MHLA_OK;
}

// ---------------------------------------------------------------------------------------------
// causal
// ---------------------------------------------------------------------------------------------
size_t mhla_causal_fwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_bf16_ok(K, V, dtype) ? 2 : 4).total_fwd;
}
size_t mhla_causal_bwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_bf16_ok(K, V, dtype) ? 2 : 4).total_bwd;
}

// chunk summaries X^T Y of the bf16 pipeline (S = K^T V, dP = scale Q^T dO)
static int cs_state16(const mhla_view& x, const mhla_view& y, uint16_t* out, float mul, int B, int T, int H, int n, int K, int V, hipStream_t st) {
    static const char* const knob = getenv("MHLA_CAUSAL_STATE");   // tuning knob, read once: "old" keeps the per-K-slice kernel
    fast::CsfStateArgs s{cv(x), cv(y), out, H, n, K, V, (long)T, mul};
    if (knob && knob[0] == 'o')
        return launch(fast::k_csf_state, dim3(n, B * H, K / 64), dim3(NTHREADS), fast::CSF_STATE_SMEM, st, "k_csf_state", s);
    const int blocks = ((K + fast::ST2_KW - 1) / fast::ST2_KW) * ((V + fast::ST2_VW - 1) / fast::ST2_VW);
    return launch(fast::k_csf_state2, dim3(n, B * H, blocks), dim3(NTHREADS), fast::CSF_STATE2_SMEM, st, "k_csf_state", s);
}

// P = strictly-lower mix of S (bf16 pipeline)
static int cs_mix_fwd(const float* mix, int ldmix, const uint16_t* S, uint16_t* P, int BH, int n, long E, hipStream_t st) {
    if (cs_mix2_ok(n, E)) {
        const Mix2Plan pl = mix2_plan((size_t)BH, n, E, false);
        fast::CsfMix2Args mf{mix, ldmix, S, nullptr, P, nullptr, n, E, pl.total, pl.spw};
#define MIXF(NW, TE) launch(fast::k_csf_mixf<NW, TE>, dim3(pl.wgs), dim3(64 * NW), fast::mixf_smem<NW, TE>(), st, "k_csf_mixf", mf)
        if (pl.te == 256) return pl.nw == 2 ? MIXF(2, 256) : pl.nw == 4 ? MIXF(4, 256) : MIXF(8, 256);
        return pl.nw == 2 ? MIXF(2, 128) : pl.nw == 4 ? MIXF(4, 128) : MIXF(8, 128);
#undef MIXF
    }
    fast::CsfMixArgs m{mix, ldmix, S, P, n, E};
    return launch(fast::k_csf_mix<0>, dim3((unsigned)(E / fast::MX_TE), (n + 63) / 64, BH), dim3(NTHREADS), fast::CSF_MIX_SMEM, st, "k_csf_mix<0>", m);
}

static int cs_fwd_impl(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                       size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream,
                       bool epi, const float* nw, float neps, mhla_view gate, mhla_mview y) {
    RC(cs_check(B, T, H, K, V, chunk, dtype));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v);
    if (!epi || out.ptr) CHECK_VIEW(out);
    if (epi) {
        if (!cs_bf16_ok(K, V, dtype) || V > 64 * fast::CSF_OUT_VS)
            return fail(MHLA_ENOTSUP, "fused norm x gate epilogue needs bf16 tensors, K %% 64 == 0 and V %% 64 == 0, V <= %d (K=%d V=%d dtype=%d)",
                        64 * fast::CSF_OUT_VS, K, V, dtype);
        const mhla_view yv{y.ptr, y.sb, y.sn, y.sh};
        if (!view_ok16(yv) || (gate.ptr && !view_ok16(gate)) || (out.ptr && !view_ok16m(out)))
            return fail(MHLA_EINVAL, "fused norm x gate epilogue: y, gate and out must be 16-byte aligned views (strides multiples of 8)");
    }
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n) return fail(MHLA_EINVAL, "mix null or ldmix=%d < n=%d chunks (T=%d)", ldmix, n, T);
    const bool pipe16 = cs_bf16_ok(K, V, dtype);
    if (pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && (epi || view_ok16m(out))))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    const CsWs w = cs_carve(ws, B, T, H, K, V, chunk, pipe16 ? 2 : 4);
    if (!ws || ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    if (((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long E = (long)K * V;
    if (pipe16) {
        // bf16 pipeline (causal_bf16.hpp): bf16 chunk summaries, bf16 MFMA everywhere
        RC(cs_state16(k, v, (uint16_t*)w.S, 1.f, B, T, H, n, K, V, st));
        RC(cs_mix_fwd(mix, ldmix, (const uint16_t*)w.S, (uint16_t*)w.P, B * H, n, E, st));
        CsOutArgs o{cv(q), cv(k), cv(v), cmv(out), mix, ldmix, w.P, H, n, K, V, (long)T, scale, cmv(y), cv(gate), nw, neps};
        if (epi) RC(launch(fast::k_csf_out<uint16_t, true>, dim3(n, B * H, 1), dim3(NTHREADS), fast::CSF_OUT_SMEM, st, "k_csf_out<norm>", o));
        else     RC(launch(fast::k_csf_out<uint16_t>, dim3(n, B * H, (V / 64 + fast::CSF_OUT_VS - 1) / fast::CSF_OUT_VS), dim3(NTHREADS), fast::CSF_OUT_SMEM, st, "k_csf_out", o));
        return MHLA_OK;
    }
    DISPATCH_T(dtype, {
        RC(cs_xty<ET>(k, v, w.S, 1.f, B, T, H, n, K, V, st));
        MixArgs m{mix, ldmix, w.S, w.P, n, E};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (n + MIX_TI - 1) / MIX_TI, B * H);
        RC(launch(k_mix<0, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,1>", m));
        CsOutArgs o{cv(q), cv(k), cv(v), cmv(out), mix, ldmix, w.P, H, n, K, V, (long)T, scale, cmv(out), cv(mhla_view{nullptr, 0, 0, 0}), nullptr, 0.f};
        RC(launch(k_cs_out<ET>, dim3(n, B * H, (V + 63) / 64), dim3(NTHREADS), CS_OUT_SMEM_FLOATS * 4, st, "k_cs_out", o));
    });
    return MHLA_OK;
}

int mhla_causal_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                    size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream) {
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, stream, false, nullptr, 0.f,
                       mhla_view{nullptr, 0, 0, 0}, mhla_mview{nullptr, 0, 0, 0});
}

int mhla_causal_normgate_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, mhla_view gate,
                             const float* norm_w, float norm_eps, mhla_mview y, void* ws, size_t ws_bytes, int B, int T, int H,
                             int K, int V, int chunk, float scale, int dtype, void* stream) {
    if (!y.ptr) return fail(MHLA_EINVAL, "y null");
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, stream, true, norm_w, norm_eps, gate, y);
}

int mhla_causal_bwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_view dout, mhla_mview dq,
                    mhla_mview dk, mhla_mview dv, float* dmix, int lddmix, void* ws, size_t ws_bytes, const void* fwd_ws,
                    int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream) {
    RC(cs_check(B, T, H, K, V, chunk, dtype));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v); CHECK_VIEW(dout); CHECK_VIEW(dq); CHECK_VIEW(dk); CHECK_VIEW(dv);
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n || !dmix || lddmix < n) return fail(MHLA_EINVAL, "mix/dmix null or leading dim < n=%d chunks", n);
    const bool pipe16 = cs_bf16_ok(K, V, dtype);
    if (pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && view_ok16(dout) && view_ok16m(dq) && view_ok16m(dk) && view_ok16m(dv)))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    CsWs w = cs_carve(ws, B, T, H, K, V, chunk, pipe16 ? 2 : 4);
    if (!ws || ws_bytes < w.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bwd);
    if (((uintptr_t)ws) % 16 || ((uintptr_t)fwd_ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    if (fwd_ws) {   // chunk summaries S, P left by mhla_causal_fwd with the same arguments: skip their recomputation
        const CsWs f = cs_carve(const_cast<void*>(fwd_ws), B, T, H, K, V, chunk, pipe16 ? 2 : 4);
        w.S = f.S;
        w.P = f.P;
    }
    hipStream_t st = (hipStream_t)stream;
    const long E = (long)K * V;
    const int tiles = (n + 63) / 64;
    const int nsplit = dw_splits(tiles * tiles * B * H, E);
    if (pipe16) {
        uint16_t *S = (uint16_t*)w.S, *P = (uint16_t*)w.P, *dP = (uint16_t*)w.dP, *dS = (uint16_t*)w.dS;
        const dim3 mgrid((unsigned)(E / fast::MX_TE), tiles, B * H);
        if (!fwd_ws) {
            RC(cs_state16(k, v, S, 1.f, B, T, H, n, K, V, st));
            RC(cs_mix_fwd(mix, ldmix, S, P, B * H, n, E, st));
        }
        RC(cs_state16(q, dout, dP, scale, B, T, H, n, K, V, st));
        const bool mix2 = cs_mix2_ok(n, E);
        int nparts = B * H * nsplit;
        if (mix2) {   // dS and the dmix partials from one pass over dP and S
            const Mix2Plan pl = mix2_plan((size_t)B * H, n, E, true);
            fast::CsfMix2Args mb{mix, ldmix, dP, S, dS, w.dwp, n, E, pl.total, pl.spw};
#define MIXB(NW) launch(fast::k_csf_mixb<NW>, dim3(pl.wgs), dim3(128 * NW), fast::mixb_smem<NW>(), st, "k_csf_mixb", mb)
            RC(pl.nw == 2 ? MIXB(2) : pl.nw == 4 ? MIXB(4) : MIXB(8));
#undef MIXB
            nparts = pl.wgs;
        } else {
            fast::CsfMixArgs mt{mix, ldmix, dP, dS, n, E};
            RC(launch(fast::k_csf_mix<1>, mgrid, dim3(NTHREADS), fast::CSF_MIX_SMEM, st, "k_csf_mix<1>", mt));
        }
        CsTokArgs t{cv(q), cv(k), cv(v), cv(dout), cmv(dq), cmv(dk), cmv(dv), mix, ldmix, w.P, w.dS, w.diag, H, n, K, V, (long)T, scale};
        static const char* const tokv = getenv("MHLA_CAUSAL_TOK");   // tuning knob, read once: "2" forces the K-slice-outer kernels
        const bool tok3 = !(tokv && tokv[0] == '2');
        if (tok3 && K <= 128)      RC(launch(fast::k_csf_bwd_tok3<uint16_t, 2>, dim3(n, B * H), dim3(NTHREADS), fast::csf_tok3_smem<2>(), st, "k_csf_bwd_tok3", t));
        else if (tok3 && K <= 256) RC(launch(fast::k_csf_bwd_tok3<uint16_t, 4>, dim3(n, B * H), dim3(NTHREADS), fast::csf_tok3_smem<4>(), st, "k_csf_bwd_tok3", t));
        else if (V <= 128) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 2>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else if (V <= 256) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 4>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else if (V <= 512) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 8>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else               RC(launch(fast::k_csf_bwd_tok<uint16_t>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK_SMEM, st, "k_csf_bwd_tok", t));
        if (!mix2) {
            fast::CsfDwArgs d{dP, S, E, w.dwp, n, tiles, nsplit};
            if (n <= 16)      RC(launch(fast::k_csf_dw<1>, dim3(1, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw<16>", d));
            else if (n <= 32) RC(launch(fast::k_csf_dw<2>, dim3(1, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw<32>", d));
            else              RC(launch(fast::k_csf_dw<4>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw", d));
        }
        // few elements, many partials (short sequences): 16 part-lanes per element instead of 4
        if (n <= 64) RC(launch(k_dw_reduce<1, 16>, dim3((n * n + 15) / 16), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                               (const float*)w.diag, dmix, lddmix, n, nparts, B * H));
        else         RC(launch(k_dw_reduce<1>, dim3((n * n + 63) / 64), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                               (const float*)w.diag, dmix, lddmix, n, nparts, B * H));
        return MHLA_OK;
    }
    DISPATCH_T(dtype, {
        MixArgs m{mix, ldmix, w.S, w.P, n, E};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (n + MIX_TI - 1) / MIX_TI, B * H);
        if (!fwd_ws) {
            RC(cs_xty<ET>(k, v, w.S, 1.f, B, T, H, n, K, V, st));
            RC(launch(k_mix<0, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,1>", m));
        }
        RC(cs_xty<ET>(q, dout, w.dP, scale, B, T, H, n, K, V, st));
        MixArgs mt{mix, ldmix, w.dP, w.dS, n, E};
        RC(launch(k_mix<1, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<1,1>", mt));
        CsTokArgs t{cv(q), cv(k), cv(v), cv(dout), cmv(dq), cmv(dk), cmv(dv), mix, ldmix, w.P, w.dS, w.diag, H, n, K, V, (long)T, scale};
        RC(launch(k_cs_bwd_tok<ET>, dim3(n, B * H), dim3(NTHREADS), CS_TOK_SMEM_FLOATS * 4, st, "k_cs_bwd_tok", t));
        DwArgs d{w.dP, w.S, E, nullptr, nullptr, 0, w.dwp, n, tiles, nsplit};
        RC(launch(k_dw<1>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw<1>", d));
        RC(launch(k_dw_reduce<1>, dim3((n * n + 63) / 64), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                  (const float*)w.diag, dmix, lddmix, n, B * H * nsplit, B * H));
    });
    return MHLA_OK;
}

// ---------------------------------------------------------------------------------------------
// per-head RMSNorm x gate
// ---------------------------------------------------------------------------------------------
int64_t mhla_rmsnorm_gate_dw_rows(int64_t rows) { return norm_grid(rows); }

int mhla_rmsnorm_gate_fwd(const void* x, int64_t ldx, const void* g, int64_t ldg, const float* w, void* y, int64_t ldy,
                          float* rstd, int64_t rows, int D, float eps, int dtype, void* stream) {
    RC(norm_check(x, y, rows, D, dtype));
    if ((ldx | ldy | (g ? ldg : 0)) & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    NormArgs a{};
    a.x = x; a.ldx = ldx; a.g = g; a.ldg = ldg; a.w = w; a.y = y; a.ldy = ldy; a.rstd = rstd; a.rows = rows; a.D = D; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(norm_fwd_grid(rows, 1));
    DISPATCH_T(dtype, {
        if (D <= 64) {
            const dim3 g4(norm_fwd_grid(rows, 4));
            if (g) RC(launch(k_rmsnorm_gate_fwd_sub<ET, 16, true>, g4, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd_sub<ET, 16, false>, g4, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else if (D <= 128) {
            const dim3 g2(norm_fwd_grid(rows, 2));
            if (g) RC(launch(k_rmsnorm_gate_fwd_sub<ET, 32, true>, g2, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd_sub<ET, 32, false>, g2, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else if (D <= 256) {
            if (g) RC(launch(k_rmsnorm_gate_fwd<ET, 1, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd<ET, 1, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else {
            if (g) RC(launch(k_rmsnorm_gate_fwd<ET, 2, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd<ET, 2, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        }
    });
    return MHLA_OK;
}

// ---------------------------------------------------------------------------------------------
// LePE depthwise convolution on the block-major token layout
// ---------------------------------------------------------------------------------------------
static int lepe_check(const void* x, const void* y, int B, int pl, int bl, int C, int K, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (B <= 0 || pl <= 0 || bl <= 0 || C <= 0 || (C & 7)) return fail(MHLA_EINVAL, "B=%d pieces_len=%d block_len=%d C=%d: need positive sizes and C %% 8 == 0", B, pl, bl, C);
    if (K != 3 && K != 5) return fail(MHLA_ENOTSUP, "kernel size %d: 3 (DiT) and 5 (ViT) are supported", K);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (B > 65535) return fail(MHLA_ENOTSUP, "B=%d exceeds grid limit 65535", B);
    return MHLA_OK;
}
constexpr int LEPE_SLICES = 128;

int mhla_lepe2d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias, const void* add,
                int64_t add_sb, int64_t add_sn, void* y, int64_t y_sb, int64_t y_sn, int B, int pieces_len, int block_len,
                int C, int K, int flip, int dtype, void* stream) {
    RC(lepe_check(x, y, B, pieces_len, block_len, C, K, dtype));
    if (!w_taps) return fail(MHLA_EINVAL, "w_taps null");
    if ((x_sb | x_sn | y_sb | y_sn | (add ? (add_sb | add_sn) : 0)) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    LepeArgs a{x, (long)x_sb, (long)x_sn, w_taps, bias, add, (long)add_sb, (long)add_sn, y, (long)y_sb, (long)y_sn, B, pieces_len, block_len, C, K, flip ? 1 : 0};
    const long N = (long)pieces_len * pieces_len * block_len * block_len, work = N * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { RC(launch(k_lepe2d<ET>, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, "k_lepe2d", a)); });
    return MHLA_OK;
}

size_t mhla_lepe2d_wgrad_ws_bytes(int C, int K) { return (size_t)LEPE_SLICES * (K * K + 1) * C * 4; }

int mhla_lepe2d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn, float* dwb,
                      void* ws, size_t ws_bytes, int B, int pieces_len, int block_len, int C, int K, int dtype,
                      void* stream) {
    RC(lepe_check(x, dout, B, pieces_len, block_len, C, K, dtype));
    if (!dwb || !ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "dwb / workspace null or workspace not 16-byte aligned");
    if (ws_bytes < mhla_lepe2d_wgrad_ws_bytes(C, K)) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, mhla_lepe2d_wgrad_ws_bytes(C, K));
    if ((x_sb | x_sn | g_sb | g_sn) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    LepeWgradArgs a{x, (long)x_sb, (long)x_sn, dout, (long)g_sb, (long)g_sn, (float*)ws, B, pieces_len, block_len, C, K, LEPE_SLICES};
    hipStream_t st = (hipStream_t)stream;
    // a workgroup covers 4 waves x 8 channel groups of CH channels (CH = 8 for K = 3, 4 for K = 5)
    DISPATCH_T(dtype, {
        if (K == 3) RC(launch(k_lepe2d_wgrad<ET, 3, 8>, dim3((C + 255) / 256, LEPE_SLICES), dim3(256), 0, st, "k_lepe2d_wgrad", a));
        else        RC(launch(k_lepe2d_wgrad<ET, 5, 4>, dim3((C + 127) / 128, LEPE_SLICES), dim3(256), 0, st, "k_lepe2d_wgrad", a));
    });
    const int rows_c = (K * K + 1) * C;
    RC(launch(k_lepe2d_wgrad_reduce, dim3((rows_c + 63) / 64), dim3(256), 0, st, "k_lepe2d_wgrad_reduce", (const float*)ws, dwb, rows_c, LEPE_SLICES));
    return MHLA_OK;
}

static int lepe3d_check(const void* x, const void* y, int B, int F, int H, int W, int C, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (B <= 0 || F <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return fail(MHLA_EINVAL, "B=%d F=%d H=%d W=%d C=%d: need positive sizes and C %% 8 == 0", B, F, H, W, C);
    if ((long)F * H * W > (1L << 30)) return fail(MHLA_ENOTSUP, "F*H*W = %ld tokens exceed 2^30", (long)F * H * W);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (B > 65535) return fail(MHLA_ENOTSUP, "B=%d exceeds grid limit 65535", B);
    return MHLA_OK;
}

int mhla_lepe3d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias, const void* add,
                int64_t add_sb, int64_t add_sn, void* y, int64_t y_sb, int64_t y_sn, int B, int F, int H, int W, int C,
                int flip, int dtype, void* stream) {
    RC(lepe3d_check(x, y, B, F, H, W, C, dtype));
    if (!w_taps) return fail(MHLA_EINVAL, "w_taps null");
    if ((x_sb | x_sn | y_sb | y_sn | (add ? (add_sb | add_sn) : 0)) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    Lepe3dArgs a{x, (long)x_sb, (long)x_sn, w_taps, bias, add, (long)add_sb, (long)add_sn, y, (long)y_sb, (long)y_sn, B, F, H, W, C, flip ? 1 : 0};
    const long work = (long)F * H * W * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { RC(launch(k_lepe3d<ET>, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, "k_lepe3d", a)); });
    return MHLA_OK;
}

size_t mhla_lepe3d_wgrad_ws_bytes(int C) { return (size_t)LEPE_SLICES * 28 * C * 4; }

int mhla_lepe3d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn, float* dwb,
                      void* ws, size_t ws_bytes, int B, int F, int H, int W, int C, int dtype, void* stream) {
    RC(lepe3d_check(x, dout, B, F, H, W, C, dtype));
    if (!dwb || !ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "dwb / workspace null or workspace not 16-byte aligned");
    if (ws_bytes < mhla_lepe3d_wgrad_ws_bytes(C)) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, mhla_lepe3d_wgrad_ws_bytes(C));
    if ((x_sb | x_sn | g_sb | g_sn) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    Lepe3dWgradArgs a{x, (long)x_sb, (long)x_sn, dout, (long)g_sb, (long)g_sn, (float*)ws, B, F, H, W, C, LEPE_SLICES};
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { RC(launch(k_lepe3d_wgrad<ET>, dim3((C + 127) / 128, LEPE_SLICES), dim3(256), 0, st, "k_lepe3d_wgrad", a)); });
    const int rows_c = 28 * C;
    RC(launch(k_lepe2d_wgrad_reduce, dim3((rows_c + 63) / 64), dim3(256), 0, st, "k_lepe2d_wgrad_reduce", (const float*)ws, dwb, rows_c, LEPE_SLICES));
    return MHLA_OK;
}

static int prologue_check(const void* x, int64_t rows, int C, int dtype, int64_t ldx) {
    if (!x) return fail(MHLA_EINVAL, "null pointer");
    if (rows <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * 8) return fail(MHLA_EINVAL, "rows=%lld C=%d: need C %% 8 == 0 and C <= 4096", (long long)rows, C);
    if (ldx & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    return MHLA_OK;
}
static int prologue_rope_check(const float* cos, const float* sin, int64_t ld_tab, int ntok, int D, int C) {
    if (!cos || !sin) return fail(MHLA_EINVAL, "rope tables null");
    if (D <= 0 || (D & 7) || C % D) return fail(MHLA_EINVAL, "head dim D=%d must be a multiple of 8 dividing C=%d", D, C);
    if (ntok <= 0 || ld_tab < D / 2 || (ld_tab & 3) || ((uintptr_t)cos | (uintptr_t)sin) % 16)
        return fail(MHLA_EINVAL, "rope tables: ntok=%d, ld=%lld must be >= D/2 and a multiple of 4, tables 16-byte aligned", ntok, (long long)ld_tab);
    return MHLA_OK;
}

int mhla_qk_prologue(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy, int64_t rows, int C, int norm,
                     float norm_eps, float eps, int dtype, void* stream) {
    return mhla_qk_prologue_rope(x, ldx, w, y, ldy, nullptr, 0, nullptr, nullptr, 0, 0, 0, rows, C, norm, norm_eps, eps, dtype, stream);
}

int mhla_qk_prologue_rope(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy, float* y_rope, int64_t ldyr,
                          const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D, int64_t rows, int C,
                          int norm, float norm_eps, float eps, int dtype, void* stream) {
    RC(prologue_check(x, rows, C, dtype, ldx | ldy | (y_rope ? ldyr : 0)));
    if (!y) return fail(MHLA_EINVAL, "null pointer");
    if (y_rope) RC(prologue_rope_check(rope_cos, rope_sin, ld_tab, ntok, D, C));
    PrologueArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.y = y; a.ldy = ldy; a.rows = rows; a.C = C; a.norm_eps = norm_eps; a.eps = eps; a.norm = norm ? 1 : 0;
    a.yr = y_rope; a.ldyr = ldyr; a.rcos = rope_cos; a.rsin = rope_sin; a.ldr = ld_tab; a.ntok = ntok > 0 ? ntok : 1; a.D = D > 0 ? D : 8;
    hipStream_t st = (hipStream_t)stream;
    const int64_t gsz = (rows + 3) / 4;
    const dim3 grid((unsigned)(gsz < 16384 ? gsz : 16384));
    DISPATCH_T(dtype, {
        if (C <= 1024)      RC(launch(k_qk_prologue<ET, 2>, grid, dim3(256), 0, st, "k_qk_prologue", a));
        else if (C <= 2048) RC(launch(k_qk_prologue<ET, 4>, grid, dim3(256), 0, st, "k_qk_prologue", a));
        else                RC(launch(k_qk_prologue<ET, 8>, grid, dim3(256), 0, st, "k_qk_prologue", a));
    });
    return MHLA_OK;
}

static int prologue_bwd_grid(int64_t rows) {   // wide rows (C floats of dw partial each): fewer workgroups than the per-head norm
    const int64_t g = (rows + 3) / 4;
    return (int)(g < 2048 ? g : 2048);
}
int64_t mhla_qk_prologue_dw_rows(int64_t rows) { return prologue_bwd_grid(rows); }

int mhla_qk_prologue_bwd(const void* x, int64_t ldx, const float* w, const float* dy, int64_t lddy, const float* dy_rope,
                         int64_t lddyr, const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D,
                         void* dx, int64_t lddx, float* dw_partial, int64_t rows, int C, int norm, float norm_eps, int dtype,
                         void* stream) {
    RC(prologue_check(x, rows, C, dtype, ldx | lddx | (dy ? lddy : 0) | (dy_rope ? lddyr : 0)));
    if (!dx || (!dy && !dy_rope)) return fail(MHLA_EINVAL, "dx null or no upstream gradient");
    if (C > 2048) return fail(MHLA_ENOTSUP, "backward supports C <= 2048 (C=%d)", C);
    if (dy_rope) RC(prologue_rope_check(rope_cos, rope_sin, ld_tab, ntok, D, C));
    if (w && !dw_partial) return fail(MHLA_EINVAL, "dw_partial null");
    PrologueArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.rows = rows; a.C = C; a.norm_eps = norm_eps; a.norm = norm ? 1 : 0;
    a.rcos = rope_cos; a.rsin = rope_sin; a.ldr = ld_tab; a.ntok = ntok > 0 ? ntok : 1; a.D = D > 0 ? D : 8;
    a.dy = dy; a.lddy = lddy; a.dyr = dy_rope; a.lddyr = lddyr; a.dx = dx; a.lddx = lddx; a.dwp = w ? dw_partial : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(prologue_bwd_grid(rows));
    DISPATCH_T(dtype, {
        if (C <= 1024) RC(launch(k_qk_prologue_bwd<ET, 2>, grid, dim3(256), 0, st, "k_qk_prologue_bwd", a));
        else           RC(launch(k_qk_prologue_bwd<ET, 4>, grid, dim3(256), 0, st, "k_qk_prologue_bwd", a));
    });
    return MHLA_OK;
}

int mhla_featmap_rotary(mhla_view x, mhla_view x_saved, const void* cos, const void* sin, int64_t ld_tab, int64_t t_offset,
                        mhla_mview y, int B, int T, int H, int K, int feature_map, int backward, int dtype, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || (K & 7)) return fail(MHLA_EINVAL, "B=%d T=%d H=%d K=%d: need positive sizes and K %% 8 == 0", B, T, H, K);
    if (feature_map < 0 || feature_map > 2) return fail(MHLA_EINVAL, "feature_map %d: 0 identity, 1 relu, 2 elu+1", feature_map);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    CHECK_VIEW(x); CHECK_VIEW(y);
    if (!cos || !sin || ld_tab < K / 2 || (ld_tab & 3) || t_offset < 0) return fail(MHLA_EINVAL, "cos/sin tables null, ld < K/2, ld %% 4 != 0 or negative offset");
    if (backward && feature_map) CHECK_VIEW(x_saved);
    FmRotArgs a{cv(x), cv(x_saved), cmv(y), cos, sin, (long)ld_tab, B, T, H, K, feature_map, (long)t_offset};
    const long total = (long)B * T * H * (K / 8);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((total + 255) / 256));
    DISPATCH_T(dtype, {
        if (backward) RC(launch(k_fmap_rotary<ET, true>, grid, dim3(256), 0, st, "k_fmap_rotary<bwd>", a));
        else          RC(launch(k_fmap_rotary<ET, false>, grid, dim3(256), 0, st, "k_fmap_rotary", a));
    });
    return MHLA_OK;
}

int mhla_rmsnorm_gate_bwd(const void* x, int64_t ldx, const void* g, int64_t ldg, const float* w, const void* dy,
                          int64_t lddy, void* dx, int64_t lddx, void* dg, int64_t lddg, float* dw_partial, int64_t rows,
                          int D, float eps, int dtype, void* stream) {
    RC(norm_check(x, dx, rows, D, dtype));
    if (!dy || !dw_partial || (g && !dg)) return fail(MHLA_EINVAL, "null pointer");
    if ((ldx | lddy | lddx | (g ? (ldg | lddg) : 0)) & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    NormArgs a{};
    a.x = x; a.ldx = ldx; a.g = g; a.ldg = ldg; a.w = w; a.dy = dy; a.lddy = lddy; a.dx = dx; a.lddx = lddx;
    a.dg = dg; a.lddg = lddg; a.dwp = dw_partial; a.rows = rows; a.D = D; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(norm_grid(rows));
    DISPATCH_T(dtype, {
        if (D <= 64) {
            if (g) RC(launch(k_rmsnorm_gate_bwd_sub<ET, 16, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd_sub<ET, 16, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else if (D <= 128) {
            if (g) RC(launch(k_rmsnorm_gate_bwd_sub<ET, 32, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd_sub<ET, 32, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else if (D <= 256) {
            if (g) RC(launch(k_rmsnorm_gate_bwd<ET, 1, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd<ET, 1, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else {
            if (g) RC(launch(k_rmsnorm_gate_bwd<ET, 2, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd<ET, 2, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        }
    });
    return MHLA_OK;
}

}  // extern "C"
