// NOT PART OF THE LIBRARY -- the round-5 prototype of a fused forward (summaries, mixing and output of a (b,h) in ONE launch, the
// fp32 summaries handed from phase to phase through the XCD's L2), kept as the record of a measured negative (verdict r4 item 3;
// DESIGN.md 3b "round 5", profiles/r5_c2_xcd_ring.md).  It is correct (workspace and output equal to the three split-operand launches:
// tools/debug_xf.py at the time) and SLOWER: 385 us against 66 + 69 + 10 + 73 = 218 us at C2.  Phase trace (k_xf_fwd, 4096 workgroups,
// two per CU): phase A 7.8 us, barrier 7.9, phase B 5.0, barrier 9.3, phase C 8.8 -- a workgroup lives 41 us, 17 of them in the two
// group barriers, and the L2 (4 MB per XCD: two (b,h) of 2 MB fp32 summaries) caps the groups in flight at two per XCD, i.e. at the
// occupancy of two latency chains per CU.  To build it again it needs mhla_amd/csrc/fused.hpp on the include path.
// One lesson worth keeping: the `sc1` loads are inline asm, invisible to hipcc's wait bookkeeping -- the `s_waitcnt vmcnt(0)` behind
// them must name the loaded registers as in/out operands, or VALU instructions that read them are scheduled ABOVE the wait.
//
// Fused forward of the block-mixing operator at the DEFAULT arithmetic (fp32 block summaries, bf16 hi + lo operands) for bf16 tensors,
// head dim 64, 16 <= M <= 64 blocks of at most 64 tokens (round 5).
//
// The split-operand path runs this shape as three launches -- summaries KV_j (k_sp_state), mixing G = W KV (k_sp_mixr), output
// O_i = Q_i G_i / n_i (k_sp_out) -- and every summary byte makes an HBM round trip between them: at C2 the forward moves 870 MB for
// 268 MB of tokens.  Here the THREE PHASES RUN IN ONE LAUNCH and the summaries are read back from the XCD's L2, where the producing
// workgroups have just left them (tools/probes/xcd_ring.hip: same-XCD read-after-write through plain stores + L1-bypassing loads,
// verified word by word; the stores still go out to memory -- all bytes leave the L2 -- so the workspace ends up exactly as the
// three launches leave it, and the backward is unchanged):
//   * a (b,h) belongs to a GROUP of 32 workgroups that run on one XCD (blockIdx = 256 stripe + 8 member + xcd: blocks with equal
//     blockIdx % 8 land on one XCD, MI355X_MICROARCH.md "Workgroup dispatch"; every workgroup checks its XCC id against its group's
//     and the launch's error word says so if the placement ever differs -- the results are then NaN, never silently stale);
//   * phase A: workgroup w forms KV_j = K_j^T V_j (fp32), ksum_j, z_j for blocks j = w and w + 32 and stores them (plain stores);
//   * group barrier (a counter per group: relaxed agent atomics, bounded poll);
//   * phase B: workgroup w mixes the e-slice [128 w, 128 w + 128) of all blocks: G[i][e] = sum_j W[i][j] KV[j][e], operands as bf16
//     hi + lo (three MFMAs per product), stored to the G workspace;
//   * group barrier;
//   * phase C: O_i = (Q_i G_i) / n_i for blocks i = w and w + 32 (G_i as hi + lo in LDS, Q rows straight from memory as the MFMA's
//     B operand, transposed product: a lane owns four consecutive output features), 1 / n_i = 1 / (W z + eps) formed on the spot,
//     full 128-byte rows stored through an LDS staging tile; the bf16 residual of the store (BmWs::olo) beside it.
// Two workgroups per CU (55 KB of LDS, <= 128 VGPRs): one group's barrier waits are the other group's phases.
// Forward progress: a workgroup waits only for members of its own group, which are dispatched right after it (in-order dispatch per
// XCD) whatever else runs; the waits are bounded all the same (XF_WAIT_POLLS), and an expired wait raises the error word.
#pragma once
#include "fused.hpp"

namespace mhla {
namespace xf {

using fast::bf16x8;
using fast::mfma_bf16;
using fast::tr_read8;
using fast::u16;
using fast::s16x8;
using fast::gt_off;
using fast::TLD;

constexpr int XF_T = 512;
constexpr int XF_GS = 32;              // workgroups per (b,h) group
constexpr int XF_ES = 4096 / XF_GS;    // 128 summary elements per mixing slice
constexpr int XF_LDB = XF_ES + 8;      // LDS row stride (bf16) of the staged slice [64 blocks][128]
constexpr int XF_LDW = 72;             // LDS row stride (bf16) of the staged weights [64][64]
constexpr int XF_WAIT_POLLS = 1 << 20;

struct XfFwdArgs {
    View q, k, v;
    MView o;
    const int* idx;
    const float* W;
    int ldw;
    float* kv;      // [bh][M][es]  KV_j = K_j^T V_j, [d1][d2] (the split-operand path's workspace)
    float* g;       // [bh][M][es]  G_i, [d1][d2]
    float* z;       // [bh][M][S]
    float* ksum;    // [bh][M][64]
    float* ninv;    // [bh][M][S]
    u16* olo;       // [bh][M S][64] or null
    int* ctr;       // [groups][4]: barrier 1, barrier 2, XCC mask, -
    int* err;       // error word of the launch
    int H, M, S, BH;
    long es;
    float eps;
    int relu, normalize;
    unsigned long long* trace;   // debugging aid (mhla_debug_set_trace): per-workgroup phase timestamps (16 slots each), or null
};

// LDS map (bytes)
constexpr int XF_OFF_WH = 0;                                  // [64][XF_LDW] bf16 hi parts of W
constexpr int XF_OFF_WL = XF_OFF_WH + 64 * XF_LDW * 2;        // lo parts
constexpr int XF_OFF_SCR = XF_OFF_WL + 64 * XF_LDW * 2;       // scratch: A: Xs, Ys | B: Th, Tl | C: Gh, Gl, Os, Ls
constexpr int XF_SCR_BYTES = 2 * 64 * XF_LDB * 2;             // 34 816 (phase B is the largest)
constexpr int XF_OFF_SIDE = XF_OFF_SCR + XF_SCR_BYTES;        // ksum_s[64], ninv_s[64], red[8][64], flag words
constexpr int XF_FWD_SMEM = XF_OFF_SIDE + (64 + 64 + 8 * 64 + 16) * 4;

__device__ __forceinline__ f32x4 ld_l2_f4(const float* p) {   // 16 bytes past the L1 (sc1: served by the XCD's L2)
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the same wait as a DEPENDENCE of the registers the asm loads above were given: hipcc knows nothing of the loads in flight, and
// without it VALU instructions that read those registers (selects, conversions) may be scheduled above the wait
__device__ __forceinline__ void wait_vm0(f32x4& a, f32x4& b) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void wait_vm0(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

// arrive at / wait for a group counter; returns false when the wait expired (error word raised)
__device__ __forceinline__ bool group_barrier(int* ctr, int target, int* err, int* lds_word, int tid) {
    wait_vm0();                      // this workgroup's stores have reached the L2
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0, ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > XF_WAIT_POLLS) {
                __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *lds_word = ok;
    }
    __syncthreads();
    return *lds_word != 0;
}

// 4 fp32 -> hi / lo bf16 pairs
__device__ __forceinline__ void split4(const f32x4& x, uint2& hi, uint2& lo) {
    hi.x = pack_bf16x2(x[0], x[1]);
    hi.y = pack_bf16x2(x[2], x[3]);
    lo.x = pack_bf16x2(x[0] - __uint_as_float(hi.x << 16), x[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack_bf16x2(x[2] - __uint_as_float(hi.y << 16), x[3] - __uint_as_float(hi.y & 0xffff0000u));
}
__device__ __forceinline__ bf16x8 row8(const u16* tile, int ld, int r0, int k0, int lane) {   // T[r0 + (lane & 15)][k0 + 8 (lane >> 4) ..]
    return *reinterpret_cast<const bf16x8*>(tile + (r0 + (lane & 15)) * ld + k0 + (lane >> 4) * 8);
}

template <bool IDX>
__global__ __launch_bounds__(XF_T, 4) void k_xf_fwd(const XfFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u16* Wh = reinterpret_cast<u16*>(smem + XF_OFF_WH);
    u16* Wl = reinterpret_cast<u16*>(smem + XF_OFF_WL);
    unsigned char* scr = smem + XF_OFF_SCR;
    float* ksum_s = reinterpret_cast<float*>(smem + XF_OFF_SIDE);
    float* ninv_s = ksum_s + 64;
    float* red = ninv_s + 64;                  // [8][64]
    int* word = reinterpret_cast<int*>(red + 8 * 64);
    const int* const idx = IDX ? a.idx : nullptr;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    // group = (b,h); its 32 members share blockIdx % 8, i.e. one XCD
    const int stripe = blockIdx.x >> 8, xcd = blockIdx.x & 7, w = (blockIdx.x & 255) >> 3;
    const int bh = stripe * 8 + xcd;
    if (bh >= a.BH) return;
    const int b = bh / a.H, h = bh - b * a.H, M = a.M, S = a.S;
    int* ctr = a.ctr + bh * 4;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(ctr + 2, 1 << (xcc & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    fast::trace_mark(a.trace, 0);
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int lrow = min(srow, S - 1);
    const bool valid = srow < S;

    // ---- the mixing weights as bf16 hi + lo in LDS (rows / columns beyond M: zero); requested first, committed behind phase A's loads
    float wreg[8];
    {
        const int i = tid >> 3, j0 = (tid & 7) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) wreg[t] = gld<float>(a.W + (long)min(i, M - 1) * a.ldw + min(j0 + t, M - 1));
    }

    // ---- phase A: KV_j^T, ksum_j, z_j of blocks w and w + 32
    u16* Xs = reinterpret_cast<u16*>(scr);     // K tile [64][TLD]
    u16* Ys = Xs + 64 * TLD;                   // V tile
    s16x8 ones_;
#pragma unroll
    for (int t = 0; t < 8; ++t) ones_[t] = (short)0x3F80;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_);
    const int d2t = wave & 3, d1h = wave >> 2, rfill = (S + 31) & ~31;
    uint4 xr[2], yr[2], tr_[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {   // both blocks' rows requested at once (rows of a missing second block: the first block's, never used)
        const int j = min(w + 32 * x, M - 1);
        const long tr = IDX ? (long)idx[(long)j * S + lrow] : (long)j * S + lrow;
        xr[x] = gld_stream16(kb + tr * a.k.sn + scol);
        yr[x] = gld_stream16(vb + tr * a.v.sn + scol);
        tr_[x] = a.normalize ? gld<uint4>(qb + tr * a.q.sn + scol) : make_uint4(0, 0, 0, 0);
    }
    {   // commit W
        const int i = tid >> 3, j0 = (tid & 7) * 8;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float w0 = (i < M && j0 + 2 * t < M) ? wreg[2 * t] : 0.f, w1 = (i < M && j0 + 2 * t + 1 < M) ? wreg[2 * t + 1] : 0.f;
            hw[t] = pack_bf16x2(w0, w1);
            lw[t] = pack_bf16x2(w0 - __uint_as_float(hw[t] << 16), w1 - __uint_as_float(hw[t] & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(Wh + i * XF_LDW + j0) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        *reinterpret_cast<uint4*>(Wl + i * XF_LDW + j0) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int j = w + 32 * x;
        if (j >= M) break;   // (uniform)
        uint4 kx = xr[x], vx = yr[x], qx = tr_[x];
        if (a.relu) {
            kx = fast::relu_eps8(kx, a.eps);
            if (a.normalize) qx = fast::relu_eps8(qx, a.eps);
        }
        if (x) __syncthreads();   // the first block's tiles and ksum_s are consumed
        *reinterpret_cast<uint4*>(Xs + srow * TLD + scol) = make_uint4(valid ? kx.x : 0u, valid ? kx.y : 0u, valid ? kx.z : 0u, valid ? kx.w : 0u);
        *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = make_uint4(valid ? vx.x : 0u, valid ? vx.y : 0u, valid ? vx.z : 0u, valid ? vx.w : 0u);
        __syncthreads();
        f32x4 acc[2], ks[2];
        acc[0] = acc[1] = ks[0] = ks[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < rfill; k0 += 32) {
            const bf16x8 av = tr_read8(Ys, TLD, k0, d2t * 16, lane);                 // A[m = d2][k = s] = V[s][d2]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 bv = tr_read8(Xs, TLD, k0, (2 * d1h + t) * 16, lane);   // B[k = s][n = d1] = K[s][d1]
                acc[t] = mfma_bf16(av, bv, acc[t]);
                if (a.normalize && d2t == 0) ks[t] = mfma_bf16(ones, bv, ks[t]);     // every row: sum_s K[s][d1]
            }
        }
        // C[m = d2 = 16 d2t + 4 kg + r][n = d1]: the lane's four consecutive d2 of row d1 of KV_j = K_j^T V_j (the split-operand
        // path's layout [d1][d2]: the backward reads this workspace)
        float* kvb = a.kv + ((long)bh * M + j) * a.es;
#pragma unroll
        for (int t = 0; t < 2; ++t) gst<f32x4>(kvb + ((2 * d1h + t) * 16 + n) * 64 + d2t * 16 + kg * 4, acc[t]);
        if (a.normalize) {
            if (d2t == 0 && lane < 16) {
#pragma unroll
                for (int t = 0; t < 2; ++t) ksum_s[(2 * d1h + t) * 16 + lane] = ks[t][0];
            }
            __syncthreads();
            if (tid < 64) gst<float>(a.ksum + ((long)bh * M + j) * 64 + tid, ksum_s[tid]);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(ksum_s + scol), hi = *reinterpret_cast<const f32x4*>(ksum_s + scol + 4);
            const unsigned qw[4] = {qx.x, qx.y, qx.z, qx.w};
            float d = __uint_as_float(qw[0] << 16) * lo[0] + __uint_as_float(qw[0] & 0xffff0000u) * lo[1] +
                      __uint_as_float(qw[1] << 16) * lo[2] + __uint_as_float(qw[1] & 0xffff0000u) * lo[3] +
                      __uint_as_float(qw[2] << 16) * hi[0] + __uint_as_float(qw[2] & 0xffff0000u) * hi[1] +
                      __uint_as_float(qw[3] << 16) * hi[2] + __uint_as_float(qw[3] & 0xffff0000u) * hi[3];
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (valid && (tid & 7) == 0) gst<float>(a.z + ((long)bh * M + j) * S + srow, d);
        }
    }
    fast::trace_mark(a.trace, 1);
    bool ok = group_barrier(ctr + 0, XF_GS, a.err, word, tid);
    fast::trace_mark(a.trace, 2);
    if (tid == 0) {   // every member has published its XCC by now: one XCD, or the L2 hand-over below is not valid
        const int mask = __hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mask & (mask - 1)) {
            __hip_atomic_store(a.err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *word = 0;
        }
    }
    __syncthreads();
    ok = ok && *word != 0;
    __syncthreads();

    // ---- phase B: G[i][e] = sum_j W[i][j] KV[j][e] for e in [128 w, 128 w + 128)
    {
        u16* Th = reinterpret_cast<u16*>(scr);   // [64 j][XF_LDB]
        u16* Tl = Th + 64 * XF_LDB;
        f32x4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int vv = tid + XF_T * u, j = vv >> 5, p = vv & 31;
            raw[u] = ld_l2_f4(a.kv + ((long)bh * M + min(j, M - 1)) * a.es + XF_ES * w + p * 4);
        }
        wait_vm0(raw[0], raw[1], raw[2], raw[3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int vv = tid + XF_T * u, j = vv >> 5, p = vv & 31;
            uint2 hi, lo;
            split4(j < M ? raw[u] : f32x4{0.f, 0.f, 0.f, 0.f}, hi, lo);
            *reinterpret_cast<uint2*>(Th + j * XF_LDB + p * 4) = hi;
            *reinterpret_cast<uint2*>(Tl + j * XF_LDB + p * 4) = lo;
        }
        __syncthreads();
        const int et = wave;   // this wave's 16 elements of the slice
        const int nks = (M + 31) >> 5, nit = (M + 15) >> 4;
        for (int it = 0; it < nit; ++it) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < nks; ++ks) {
                const bf16x8 ah = tr_read8(Th, XF_LDB, ks * 32, et * 16, lane), al = tr_read8(Tl, XF_LDB, ks * 32, et * 16, lane);   // A[m = e][k = j]
                const bf16x8 bh_ = row8(Wh, XF_LDW, it * 16, ks * 32, lane), bl_ = row8(Wl, XF_LDW, it * 16, ks * 32, lane);     // B[k = j][n = i] = W[i][j]
                c = mfma_bf16(ah, bh_, c);
                c = mfma_bf16(ah, bl_, c);
                c = mfma_bf16(al, bh_, c);
            }
            const int i = it * 16 + n;
            if (i < M) gst<f32x4>(a.g + ((long)bh * M + i) * a.es + XF_ES * w + et * 16 + kg * 4, c);   // rows e = 4 kg + r of column i
        }
    }
    fast::trace_mark(a.trace, 3);
    ok = group_barrier(ctr + 1, XF_GS, a.err, word, tid) && ok;
    fast::trace_mark(a.trace, 4);

    // ---- phase C: O_i = (Q_i G_i) / n_i of blocks w and w + 32
    u16* Gh = reinterpret_cast<u16*>(scr);   // [64 d1][64 d2], 16-byte pieces swizzled (gt_off)
    u16* Gl = Gh + 64 * 64;
    u16* Os = Gl + 64 * 64;                  // output staging
    u16* Ls = Os + 64 * 64;                  // residual staging
    const int st = wave & 3, tp = wave >> 2;
    for (int x = 0; x < 2; ++x) {
        const int i = w + 32 * x;
        if (i >= M) break;
        const long p0 = (long)i * S;
        // this wave's 16 token rows of Q as the MFMA's B operand, straight from memory (L1 / L2: phase A read them)
        bf16x8 qa[2];
        {
            const int row = st * 16 + n;
            const u16* src = qb + tok_row(idx, p0 + (row < S ? row : 0)) * a.q.sn + kg * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 qv = gld<uint4>(src + ks * 32);
                if (a.relu) qv = fast::relu_eps8(qv, a.eps);
                qa[ks] = __builtin_bit_cast(bf16x8, qv);
            }
        }
        f32x4 graw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) graw[u] = ld_l2_f4(a.g + ((long)bh * M + i) * a.es + (tid + XF_T * u) * 4);
        float zpart = 0.f;
        if (a.normalize) {   // n_i[s] = eps + sum_j W[i][j] z_j[s]: 8 parts of 8 blocks each
            const int s = tid & 63, part = tid >> 6, sc = min(s, S - 1);
            float zr[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) zr[t] = __hip_atomic_load(a.z + ((long)bh * M + min(part * 8 + t, M - 1)) * S + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int t = 0; t < 8; ++t) zpart += (part * 8 + t < M) ? gld<float>(a.W + (long)i * a.ldw + min(part * 8 + t, M - 1)) * zr[t] : 0.f;
        }
        wait_vm0(graw[0], graw[1]);
        if (x) __syncthreads();   // the previous block's staging tiles are stored
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int vv = tid + XF_T * u, d1 = vv >> 4, d2 = (vv & 15) * 4;
            uint2 hi, lo;
            split4(graw[u], hi, lo);
            *reinterpret_cast<uint2*>(Gh + gt_off(d1, d2)) = hi;
            *reinterpret_cast<uint2*>(Gl + gt_off(d1, d2)) = lo;
        }
        if (a.normalize) red[(tid >> 6) * 64 + (tid & 63)] = zpart;
        __syncthreads();
        if (a.normalize && tid < 64) {
            float nn = a.eps;
#pragma unroll
            for (int p = 0; p < 8; ++p) nn += red[p * 64 + tid];
            const float ni = 1.f / nn;
            ninv_s[tid] = ni;
            if (tid < S) gst<float>(a.ninv + ((long)bh * M + i) * S + tid, ni);
        }
        f32x4 acc[2];
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tn = 2 * tp + t;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // A[m = d2][k = d1] = G[d1][d2]: hardware transpose reads of the [d1][d2] tile; B = Q rows
                acc[t] = mfma_bf16(fast::tr_read8_gt(Gh, ks * 32, tn * 16, lane), qa[ks], acc[t]);
                acc[t] = mfma_bf16(fast::tr_read8_gt(Gl, ks * 32, tn * 16, lane), qa[ks], acc[t]);
            }
        }
        __syncthreads();   // 1 / n visible
        const float ni = a.normalize ? ninv_s[st * 16 + n] : 1.f;
        const float poison = ok ? 1.f : __builtin_nanf("");
#pragma unroll
        for (int t = 0; t < 2; ++t) {   // the lane's token row 16 st + n, columns 16 tn + 4 kg .. + 3
            const f32x4 o4 = acc[t] * (ni * poison);
            const int off = gt_off(st * 16 + n, (2 * tp + t) * 16 + kg * 4);
            *reinterpret_cast<uint2*>(Os + off) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
            if (a.olo) *reinterpret_cast<uint2*>(Ls + off) = store_residual4<bf16_t>(o4);
        }
        __syncthreads();
        {   // one 16-byte piece per thread: full 128-byte rows
            const int row = tid >> 3, c = (tid & 7) * 8;
            if (row < S) {
                gst<uint4>(ob + tok_row(idx, p0 + row) * a.o.sn + c, *reinterpret_cast<const uint4*>(Os + gt_off(row, c)));
                if (a.olo) gst<uint4>(a.olo + (((long)bh * M + i) * S + row) * 64 + c, *reinterpret_cast<const uint4*>(Ls + gt_off(row, c)));
            }
        }
    }
    fast::trace_mark(a.trace, 5);
    if (a.trace) { wait_vm0(); fast::trace_mark(a.trace, 6); }
}

}  // namespace xf
}  // namespace mhla
