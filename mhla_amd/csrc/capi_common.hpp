// Shared host-side helpers of the C ABI translation units (capi*.hip): error string, launch wrapper with the per-launch event
// hook, argument checks, workspace carving, dtype / head-dim dispatch.  The library is built from several translation units
// compiled in parallel (mhla_amd/build.py); state that must be one per library is an inline variable here.
#pragma once
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

#include "common.hpp"

namespace mhla {
namespace capi {


inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
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
inline std::mutex g_prof_mu;
inline std::atomic<bool> g_prof_on{false};
inline std::vector<ProfRec> g_prof;

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
inline std::atomic<unsigned long long*> g_trace{nullptr};

// h16 summaries of 129 .. 256 blocks: the re-cut resident mixing kernel (mixh2.hpp k_sp_mixh2: the rescaled weights kept per (b, h)) serves
// launches whose workgroups get at least this many 64-element slices each -- fewer do not pay for its rebuilds (capi_bm_typed.hpp sp_mixh;
// capi.hip mhla_describe_dispatch)
#ifndef SP_MIXH2_MIN_SLICES
#define SP_MIXH2_MIN_SLICES 8
#endif
// (mhla_set_option("recut_kernels", 0): the kernels they replaced -- k_sp_mixh at twelve / sixteen waves, the block-per-workgroup Wan output
// kernel -- for A/B timing and for the bit-equality tests; MHLA_WAN_FLAT=0 / MHLA_RECUT=0 in the environment set the start value)
inline std::atomic<int> g_recut{[] { const char* e = getenv("MHLA_RECUT"); const char* f = getenv("MHLA_WAN_FLAT"); return ((e && e[0] == '0') || (f && f[0] == '0')) ? 0 : 1; }()};
inline bool sp_mixh2_applies(int M, long E, long BH, long S = 0) {
    if (M <= 128 || M > 256 || !g_recut.load()) return false;
    // (its buffer descriptors address 32-bit byte offsets: the (b, h)'s summary rows -- E / 2 + 288 floats each -- and normaliser rows must fit)
    if (BH * M * (E / 2 + 288) * 4 >= (1L << 31) || BH * M * S * 4 >= (1L << 31)) return false;
    const long total = BH * ((E + 63) / 64), wgs = std::min<long>(total, 256);
    return wgs > 0 && (total + wgs - 1) / wgs >= SP_MIXH2_MIN_SLICES;
}
// mhla_set_option("fp32_summaries") / MHLA_FP32_SUMMARIES=1 (read once): the resident-mixing pipeline keeps its summaries as fp32 instead of 24-bit floats
inline std::atomic<int> g_no_p24{[] { const char* e = getenv("MHLA_FP32_SUMMARIES"); return (e && e[0] == '1') ? 1 : 0; }()};

inline View cv(const mhla_view& v) { return View{v.ptr, (long)v.sb, (long)v.sn, (long)v.sh}; }
inline MView cmv(const mhla_mview& v) { return MView{v.ptr, (long)v.sb, (long)v.sn, (long)v.sh}; }

inline int check_view(const char* name, const void* ptr, int64_t sb, int64_t sn, int64_t sh, int dtype) {
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
inline int dw_splits(long wgs, long E) {
    long ns = (1024 + wgs - 1) / wgs;
    if (ns > DW_MAX_SPLIT) ns = DW_MAX_SPLIT;
    while (ns > 1 && E / ns < 256) --ns;
    return (int)(ns < 1 ? 1 : ns);
}

inline int dt_for(int D) { return D <= 32 ? 2 : D <= 64 ? 4 : D <= 80 ? 5 : D <= 96 ? 6 : D <= 128 ? 8 : 0; }

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

// Storage format of the D x D block summaries KV, G, dG, dKV in the workspace (bm_sumfmt: a function of the call's shape, dtype and
// flags -- and of the process-wide "fp32_summaries" option -- so that a forward and the backward that reuses its state agree)
enum { SF_F32 = 0,    // fp32 words
       SF_P24 = 1,    // 24-bit floats in two planes per row (split.hpp p24): 16 significand bits, 3 bytes
       SF_H16 = 2,    // fp16 payload x one power-of-two multiplier per row (split.hpp h16): 11 significand bits, 2 bytes -- the default for 16-bit tensors
       SF_BF16 = 3 }; // single bf16 values (MHLA_FLAG_BF16_SUMMARIES: reduced precision, opt-in)
struct BmWs {
    float *kv, *g, *z, *ksum, *ninv, *dg, *dkv, *dn, *dz, *dks, *dwp;
    unsigned short* olo;   // forward region: O - fl(O) as bf16 [bh][M S][D] (16-bit tensors at the default arithmetic), or null
    size_t total_fwd, total_bwd;
    long es;   // from one block's D x D summary to the next, in floats (SF_BF16: in 16-bit elements): the format's row + padding
    int fmt;   // SF_*
};
// sum16: the D x D block summaries (KV, G, dG, dKV) are stored as bf16 (split-operand path on bf16 tensors): half the floats
// padded (split-operand path): every summary row carries 1152 bytes of padding.  The mixing and dW kernels read the same 128-byte
// piece of EVERY block's summary at once, i.e. at the row stride: at a power-of-two stride (8 KB at D = 64 bf16, 64 KB at D = 128
// fp32) those requests fall on a fraction of the HBM channels -- k_sp_dwr ran at 3.7 TB/s at D = 64 and at 4.9 / 5.4 TB/s at
// D = 56 / 72 (the causal pipeline met the same effect, causal_bf16.hpp).  The generic fp32-MFMA kernels keep dense rows.
inline long bm_row_elems(int D, int fmt, bool padded) {
    const long E = (long)D * D;
    switch (fmt) {
        case SF_BF16: return E + (padded ? 576 : 0);        // (16-bit elements)
        case SF_P24: return 3 * E / 4 + 288;                // two planes: E x u16, E x u8
        case SF_H16: return E / 2 + 288;                    // E x fp16; the row's multiplier is the first word of the padding.  (Rows must start on
                                                            // 128-byte lines: with 16 bytes more every 256-byte slice piece of the mixing kernels
                                                            // straddled three lines instead of two -- 1.67x the reads, 1.23x the writes in the counters)
        default: return E + (padded ? 288 : 0);
    }
}
inline BmWs bm_carve(void* ws, int B, int H, int M, int S, int D, int fmt, bool padded, bool olo = false) {
    const size_t es = (size_t)bm_row_elems(D, fmt, padded);
    // (every region starts on a 128-byte line -- 32 floats -- so that the summary rows, whose stride is a whole number of lines, do too)
    auto al32 = [](size_t n) { return (n + 31) & ~(size_t)31; };
    const size_t bh = (size_t)B * H, st = al32(fmt == SF_BF16 ? (bh * M * es + 1) / 2 : bh * M * es), zs = al32(bh * M * S), ks = al32(bh * M * D);
    float* p = (float*)ws;
    BmWs w;
    w.fmt = fmt;
    w.es = (long)es;
    w.kv = p; p += st;
    w.g = p; p += st;
    w.z = p; p += zs;
    w.ksum = p; p += ks;
    w.ninv = p; p += zs;
    w.olo = nullptr;
    if (olo) { w.olo = (unsigned short*)p; p += al32((bh * M * S * D + 1) / 2); }
    w.total_fwd = (size_t)(p - (float*)ws) * 4;
    w.dg = p; p += st;
    w.dkv = p; p += st;
    w.dn = p; p += zs;
    w.dz = p; p += zs;
    w.dks = p; p += ks;
    // dW partials: (b, h) x E-slices of k_sp_dw / k_dw, or one per workgroup of the fused mixing + dW kernel (<= 512; <= 1024 of up to 32 x 32 at two waves) + one per (b, h)
    w.dwp = p; p += al4(std::max(bh * DW_MAX_SPLIT, M <= 32 ? bh + 1024 : M <= 128 ? bh + 512 : (size_t)0) * M * M);
    w.total_bwd = (size_t)(p - (float*)ws) * 4;
    return w;
}

inline bool view_ok16(const mhla_view& v) { return v.ptr && ((uintptr_t)v.ptr % 16) == 0 && ((v.sb | v.sn | v.sh) & 7) == 0; }
// split-bf16 MFMA kernels (split.hpp): head dims that are multiples of 8, any dtype
inline bool sp_shape_ok(int D, unsigned flags) { return (D & 7) == 0 && !(flags & MHLA_FLAG_FORCE_GENERIC); }
inline bool view_ok16m(const mhla_mview& v) { return v.ptr && ((uintptr_t)v.ptr % 16) == 0 && ((v.sb | v.sn | v.sh) & 7) == 0; }
// bf16 block summaries (and single-bf16 intermediate operands): only when the caller asked for them (MHLA_FLAG_BF16_SUMMARIES)
inline bool bm_sum16(int D, int dtype, unsigned flags) { return dtype == MHLA_BF16 && sp_shape_ok(D, flags) && (flags & MHLA_FLAG_BF16_SUMMARIES); }
// The summary format of a block-mix call on the generic / split-operand path.  h16 and p24 live on the resident-mixing pipeline
// (split.hpp k_sp_mixr: up to 256 blocks, summaries of whole 64-element slices):
//   h16: 16-bit tensors, head dims 32 .. 96, 4 .. 256 blocks (up to 128: dW fused into the mixing; beyond: k_sp_dwr) of at least 16 tokens (11-bit summaries lean on averaging
//        over the block, the head dim and the blocks: in the model, tools/sim_h16.py, 2 blocks reach 5e-3 in dW -- a difference
//        of nearly equal terms there -- head dim 8 reaches 7e-4, a 2 x 1-token case 1.4e-3; inside the rule: <= 6e-4) -- unless
//        the caller asks for >= 16 significand bits
//        (MHLA_FLAG_FP32_GRADE_SUMMARIES) or the process keeps fp32 words ("fp32_summaries");
//   p24: the same 16-bit shapes when h16 is declined, and fp32 tensors at head dims 97 .. 128 with up to 192 blocks (the Wan shape).
inline int bm_sumfmt(int M, int S, int D, int dtype, unsigned flags) {
    if (!sp_shape_ok(D, flags)) return SF_F32;
    if (bm_sum16(D, dtype, flags)) return SF_BF16;
    const bool mixr = M <= 256 && ((long)D * D) % 64 == 0;
    if (!mixr || g_no_p24.load()) return SF_F32;
    if (dtype != MHLA_F32 && D <= 96 && S >= 16 && M >= 4 && D >= 32 && !(flags & MHLA_FLAG_FP32_GRADE_SUMMARIES)) return SF_H16;   // (M <= 256: mixr)
    if (dtype != MHLA_F32 && D <= 96 && M <= 128) return SF_P24;
    if (dtype == MHLA_F32 && D > 96 && M <= 192) return SF_P24;
    return SF_F32;
}
// 16-bit tensors on the h16 / p24 pipeline with head dims up to 64: the backward forms its row dots dO . O from G_i itself --
// dO' . (Q_i G_i), one more small product in k_sp_state<1> -- and reads neither the stored output nor a residual of it
inline bool bm_rowdots_from_g(int M, int S, int D, int dtype, unsigned flags) {
    const int f = bm_sumfmt(M, S, D, dtype, flags);
    return dtype != MHLA_F32 && (f == SF_H16 || f == SF_P24) && D <= 64;
}
// other 16-bit tensors at the default arithmetic: the forward keeps what its store of O rounded away (BmWs::olo) for the backward
inline bool bm_olo(int M, int S, int D, int dtype, unsigned flags) { return dtype != MHLA_F32 && !bm_sum16(D, dtype, flags) && !bm_rowdots_from_g(M, S, D, dtype, flags); }
// the bf16-summary fast path (fused.hpp): its summaries are single bf16 values, so it serves the opt-in arithmetic only
inline bool fast_shape_ok(int M, int D, int dtype, bool split, unsigned flags) {
    return dtype == MHLA_BF16 && D == 64 && M <= 64 && !split && (flags & MHLA_FLAG_BF16_SUMMARIES);
}
// small-sequence single-launch path (smalln.hpp): S = 16 tokens per block, at most 16 blocks, D <= 80
inline bool sn_shape_ok(int M, int S, int D, int dtype, bool split) {
    return dtype == MHLA_BF16 && S == 16 && M <= 16 && D <= 80 && (D & 7) == 0 && !split;
}

// the same regime with fp32 tensors (smalln_f32.hpp: hi + lo bf16 operands)
inline bool snf_shape_ok(int M, int S, int D, int dtype, bool split) {
    return dtype == MHLA_F32 && S == 16 && M <= 16 && D <= 80 && (D & 7) == 0 && !split;
}

inline int bm_check(int B, int H, int M, int S, int D, int dtype, unsigned flags, bool normalize, bool split) {
    if (B <= 0 || H <= 0 || M <= 0 || S <= 0 || D <= 0) return fail(MHLA_EINVAL, "non-positive dimension B=%d H=%d M=%d S=%d D=%d", B, H, M, S, D);
    if (D % 4) return fail(MHLA_EINVAL, "D=%d must be a multiple of 4", D);
    if (!dt_for(D)) return fail(MHLA_ENOTSUP, "block-mix head dim D=%d > 128 not supported", D);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (flags & ~(MHLA_FLAG_RELU_EPS | MHLA_FLAG_FORCE_GENERIC | MHLA_FLAG_NO_SMALLN | MHLA_FLAG_BF16_SUMMARIES | MHLA_FLAG_NO_BWD_STATE | MHLA_FLAG_FP32_GRADE_SUMMARIES)) return fail(MHLA_EINVAL, "unknown flags 0x%x", flags);
    if ((flags & MHLA_FLAG_BF16_SUMMARIES) && (flags & MHLA_FLAG_FP32_GRADE_SUMMARIES)) return fail(MHLA_EINVAL, "MHLA_FLAG_BF16_SUMMARIES and MHLA_FLAG_FP32_GRADE_SUMMARIES exclude each other");
    if ((flags & MHLA_FLAG_BF16_SUMMARIES) && dtype == MHLA_F16) return fail(MHLA_EINVAL, "MHLA_FLAG_BF16_SUMMARIES (single-bf16 summaries) serves bf16 tensors only");
    if ((flags & MHLA_FLAG_RELU_EPS) && split) return fail(MHLA_EINVAL, "MHLA_FLAG_RELU_EPS needs q_den/k_den to alias q_num/k_num");
    if ((size_t)B * H > 65535) return fail(MHLA_ENOTSUP, "B*H=%zu exceeds grid limit 65535", (size_t)B * H);
    (void)normalize;
    return MHLA_OK;
}

// One call of the generic / split-operand block-mix path, handed from capi.hip to the per-dtype translation units
// (capi_bm_f32.hip, capi_bm_bf16.hip, capi_bm_f16.hip instantiate bm_fwd_typed / bm_bwd_typed for their element type).
struct BmCall {
    mhla_view q_num, k_num, v, q_den, k_den, outv, dout, gate;
    mhla_mview out, dq_num, dk_num, dv, dq_den, dk_den;
    const float* W; int ldw; float* dW;
    const int32_t* block_index;
    BmWs w;
    int B, H, M, S, D; float eps; unsigned flags;
    bool normalize, split, reuse, epi;
    hipStream_t st;
    const float *rcos, *rsin; long ldr;
    const float* nw; float neps; int out_dtype;
    unsigned short* olo_own;   // backward: the O-residual region of the backward's OWN workspace (w.olo may point into the kept forward workspace)
};
template <typename ET, bool S16> int bm_fwd_typed(const BmCall& c);   // S16: bf16 block summaries (bf16 tensors + MHLA_FLAG_BF16_SUMMARIES)
template <typename ET, bool S16> int bm_bwd_typed(const BmCall& c);

}  // namespace capi
}  // namespace mhla
