// Block-mixing forward kernels for head dims 64 and 128 in any dtype (the Wan2.1 shape: fp32, D = 128, M = 150 blocks of
// 210 tokens; wan/mhla_utils.py:331-341), computing on the bf16 MFMA with split operands.
//
// An fp32 value x is carried as two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi), which keep 16 mantissa bits;
// a product a b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi with fp32 accumulation (the dropped lo*lo term is
// below 2^-17 |a b|).  Three v_mfma_f32_16x16x32_bf16 do the work of eight v_mfma_f32_16x16x4_f32 at a sixteenth of the
// issue cycles each, and a bf16 operand fetch from LDS moves 8 reduction steps per lane instead of 1 -- the generic
// fp32-MFMA kernels are bound by exactly those two.  bf16 tensors have lo = 0 and skip the extra products.
// Workspace formats are the generic path's (blockmix.cuh): KV, G fp32 [bh][M][D][D]; ksum [bh][M][D]; z, 1/n [bh][M][S].
//   k_sp_state : KV_j = K_j^T V_j, ksum_j, z_j                     grid (M, bh)
//   k_sp_mix   : G = W . KV  (or W^T .)                            grid (D^2 / 128, ceil(M / 64), bh)
//   k_sp_out   : O_i = (Q_i G_i) / n_i, computed transposed so that a lane owns 4 consecutive output features
#pragma once
#include <type_traits>

#include "blockmix.cuh"
#include "fused.cuh"

namespace mhla {
namespace sp {

using fast::bf16x8;
using fast::mfma_bf16;
using fast::tr_read8;
using fast::u16;

template <typename T>
__device__ __forceinline__ void ld8(const T* p, f32x4& a, f32x4& b) {
    a = Io<T>::ld4(p);
    b = Io<T>::ld4(p + 4);
}
__device__ __forceinline__ void relu8(f32x4& a, f32x4& b, float eps) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = fmaxf(a[i], 0.f) + eps;
        b[i] = fmaxf(b[i], 0.f) + eps;
    }
}
// 8 fp32 -> 8 bf16 hi + 8 bf16 lo (packed, element 0 in the low half of word 0)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, uint4& hi, uint4& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
        l[i] = pack_bf16x2(x[2 * i] - __uint_as_float(h[i] << 16), x[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int DT>
__host__ __device__ constexpr int sp_state_smem() {
    constexpr int D = DT * 16;
    return 4 * 32 * (D + 8) * 2 + (256 / (D / 8)) * D * 4 + D * 4;
}

template <typename T, int DT>
__global__ __launch_bounds__(NTHREADS) void k_sp_state(const StateArgs a) {
    constexpr int D = DT * 16, LD = D + 8, CGS = D / 8, RPP = NTHREADS / CGS, IT = 32 / RPP, RT = DT / 4, TILE = 32 * LD;
    constexpr bool LO = !std::is_same<T, bf16_t>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Kh = reinterpret_cast<u16*>(smem_raw);
    u16* Kl = Kh + TILE;
    u16* Vh = Kl + TILE;
    u16* Vl = Vh + TILE;
    float* cs = reinterpret_cast<float*>(Vl + TILE);   // [RPP][D] column-sum partials
    float* vecd = cs + RPP * D;                        // [D] ksum
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, S = a.S;
    const long p0 = (long)blk * S;
    const T* kb = (const T*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const T* vb = (const T*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const T* kdb = (const T*)a.kd.ptr + b * a.kd.sb + h * a.kd.sh;
    const bool den = a.normalize && a.split;   // the normaliser's keys are a separate tensor
    const int r0 = tid / CGS, cg = (tid % CGS) * 8;

    f32x4 kx[IT][2], vx[IT][2], dx[IT][2];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = c0 + r0 + RPP * it;
            kx[it][0] = kx[it][1] = vx[it][0] = vx[it][1] = dx[it][0] = dx[it][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < S) {
                const long row = tok_row(a.idx, p0 + r);
                ld8(kb + row * a.x.sn + cg, kx[it][0], kx[it][1]);
                ld8(vb + row * a.y.sn + cg, vx[it][0], vx[it][1]);
                if (a.relu) relu8(kx[it][0], kx[it][1], a.eps);
                if (den) ld8(kdb + row * a.kd.sn + cg, dx[it][0], dx[it][1]);
            }
        }
    };
    float ksp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int off = (r0 + RPP * it) * LD + cg;
            uint4 hi, lo;
            split8(kx[it][0], kx[it][1], hi, lo);
            *reinterpret_cast<uint4*>(Kh + off) = hi;
            if (LO) *reinterpret_cast<uint4*>(Kl + off) = lo;
            split8(vx[it][0], vx[it][1], hi, lo);
            *reinterpret_cast<uint4*>(Vh + off) = hi;
            if (LO) *reinterpret_cast<uint4*>(Vl + off) = lo;
            const f32x4* s = den ? dx[it] : kx[it];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ksp[i] += s[0][i];
                ksp[4 + i] += s[1][i];
            }
        }
    };

    f32x4 acc[RT][DT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int c0 = 0; c0 < S; c0 += 32) {
        commit();
        __syncthreads();
        if (c0 + 32 < S) fetch(c0 + 32);
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ah[rt] = tr_read8(Kh, LD, 0, (wave * RT + rt) * 16, lane);
            if (LO) al[rt] = tr_read8(Kl, LD, 0, (wave * RT + rt) * 16, lane);
        }
#pragma unroll
        for (int ct = 0; ct < DT; ++ct) {
            const bf16x8 bh_ = tr_read8(Vh, LD, 0, ct * 16, lane);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(ah[rt], bh_, acc[rt][ct]);
            if (LO) {
                const bf16x8 bl_ = tr_read8(Vl, LD, 0, ct * 16, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(ah[rt], bl_, acc[rt][ct]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(al[rt], bh_, acc[rt][ct]);
            }
        }
        __syncthreads();
    }

    // KV_j -> ws  (C layout: row d1 = 16 tile + 4 kg + r, column d2 = 16 ct + nl)
    float* ob = a.out + ((long)bh * a.M + blk) * D * D;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long)((wave * RT + rt) * 16 + kg * 4 + r) * D + ct * 16 + nl] = acc[rt][ct][r];

    if (a.normalize) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[r0 * D + cg + i] = ksp[i];
        __syncthreads();
        if (tid < D) {
            float s = 0.f;
#pragma unroll 4
            for (int r = 0; r < RPP; ++r) s += cs[r * D + tid];
            vecd[tid] = s;
            a.ksum[((long)bh * a.M + blk) * D + tid] = s;
        }
        __syncthreads();
        // z_j[s] = Qden_j[s] . ksum_j : CGS lanes per token row
        const T* qb = (const T*)a.qd.ptr + b * a.qd.sb + h * a.qd.sh;
        float kv8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv8[i] = vecd[cg + i];
        for (int rb = 0; rb < S; rb += RPP) {
            const int r = rb + r0;
            float d = 0.f;
            if (r < S) {
                f32x4 x0, x1;
                ld8(qb + tok_row(a.idx, p0 + r) * a.qd.sn + cg, x0, x1);
                if (a.relu) relu8(x0, x1, a.eps);
#pragma unroll
                for (int i = 0; i < 4; ++i) d += x0[i] * kv8[i] + x1[i] * kv8[4 + i];
            }
#pragma unroll
            for (int o = CGS / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (r < S && (tid % CGS) == 0) a.zo[((long)bh * a.M + blk) * S + r] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------------
constexpr int SPM_TE = 128, SPM_LD = SPM_TE + 8, SPM_TILE = 32 * SPM_LD;
constexpr int SP_MIX_SMEM = 4 * SPM_TILE * 2;   // two buffers of (hi, lo) [32][SPM_LD] bf16; reused as [64][SPM_TE + 4] fp32 staging
static_assert(64 * (SPM_TE + 4) * 4 <= SP_MIX_SMEM, "output staging must fit in the input tiles");

template <int TRANS>
__global__ __launch_bounds__(NTHREADS, 2) void k_sp_mix(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const long e0 = (long)blockIdx.x * SPM_TE;
    const int i0 = blockIdx.y * 64, bh = blockIdx.z, M = a.M;
    const float* inb = a.in + (long)bh * M * a.E + e0;
    float* outb = a.out + (long)bh * M * a.E + e0;
    const int steps = (M + 31) / 32;
    const int sr = tid >> 3, sc = (tid & 7) * 8;   // staging: row sr, 8 floats at columns sc and sc + 64

    f32x4 pre[2][2];
    auto fetch = [&](int step) {
        const int row = step * 32 + sr;
#pragma unroll
        for (int u = 0; u < 2; ++u) pre[u][0] = pre[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < M) {
            const float* src = inb + (long)row * a.E + sc;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pre[u][0] = *reinterpret_cast<const f32x4*>(src + 64 * u);
                pre[u][1] = *reinterpret_cast<const f32x4*>(src + 64 * u + 4);
            }
        }
    };
    auto commit = [&](int buf) {
        u16* th = Xs + buf * 2 * SPM_TILE + sr * SPM_LD + sc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 hi, lo;
            split8(pre[u][0], pre[u][1], hi, lo);
            *reinterpret_cast<uint4*>(th + 64 * u) = hi;
            *reinterpret_cast<uint4*>(th + SPM_TILE + 64 * u) = lo;
        }
    };

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int orow = i0 + wave * 16 + nl;

    fetch(0);
    commit(0);
    for (int step = 0; step < steps; ++step) {
        const int buf = step & 1;
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
        const int k0 = step * 32 + kg * 8;
        bf16x8 ah, al;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int kk = k0 + t;
            float w = 0.f;
            if (orow < M && kk < M) w = TRANS ? a.W[(long)kk * a.ldw + orow] : a.W[(long)orow * a.ldw + kk];
            const __bf16 hi = (__bf16)w;
            ah[t] = hi;
            al[t] = (__bf16)(w - (float)hi);
        }
        const u16* th = Xs + buf * 2 * SPM_TILE;
#pragma unroll
        for (int t4 = 0; t4 < 8; t4 += 4) {
            bf16x8 bh_[4], bl_[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bh_[t] = tr_read8(th, SPM_LD, 0, (t4 + t) * 16, lane);
                bl_[t] = tr_read8(th + SPM_TILE, SPM_LD, 0, (t4 + t) * 16, lane);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(ah, bh_[t], acc[t4 + t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(ah, bl_[t], acc[t4 + t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(al, bh_[t], acc[t4 + t]);
        }
        if (step + 1 < steps) commit(buf ^ 1);
    }
    __syncthreads();
    float* Os = reinterpret_cast<float*>(smem_raw);   // [64][SPM_TE + 4]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(wave * 16 + kg * 4 + r) * (SPM_TE + 4) + t * 16 + nl] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int v0 = 0; v0 < 8; ++v0) {
        const int v = tid + v0 * NTHREADS, r = v >> 5, c = (v & 31) * 4;
        if (i0 + r < M) *reinterpret_cast<f32x4*>(outb + (long)(i0 + r) * a.E + c) = *reinterpret_cast<const f32x4*>(Os + r * (SPM_TE + 4) + c);
    }
}

// -------------------------------------------------------------------------------------------------
template <int DT>
__host__ __device__ constexpr int sp_out_smem() { return 2 * DT * 16 * (DT * 16 + 8) * 2; }

template <typename T, int DT>
__global__ __launch_bounds__(NTHREADS, 2) void k_sp_out(const OutArgs a) {
    constexpr int D = DT * 16, LD = D + 8, CGS = D / 8, RPP = NTHREADS / CGS, KST = D / 32, TILE = D * LD;
    constexpr bool LO = !std::is_same<T, bf16_t>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gh = reinterpret_cast<u16*>(smem_raw);   // [d1][d2]
    u16* Gl = Gh + TILE;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, S = a.S;
    const long p0 = (long)blk * S;
    const T* qb = (const T*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    T* ob = (T*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    {   // G_i -> LDS as hi / lo
        const float* g = a.g + ((long)bh * a.M + blk) * D * D;
        const int r0 = tid / CGS, cg = (tid % CGS) * 8;
        constexpr int PASSES = D / RPP, UB = PASSES < 4 ? PASSES : 4;
        for (int pb = 0; pb < PASSES; pb += UB) {
            f32x4 x[UB][2];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const float* src = g + (long)(r0 + RPP * (pb + u)) * D + cg;
                x[u][0] = *reinterpret_cast<const f32x4*>(src);
                x[u][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                uint4 hi, lo;
                split8(x[u][0], x[u][1], hi, lo);
                const int off = (r0 + RPP * (pb + u)) * LD + cg;
                *reinterpret_cast<uint4*>(Gh + off) = hi;
                *reinterpret_cast<uint4*>(Gl + off) = lo;
            }
        }
    }
    __syncthreads();
    const float* ninvb = a.ninv + ((long)bh * a.M + blk) * S;
    for (int tt = wave; tt * 16 < S; tt += 4) {
        const int s = tt * 16 + nl, sv = min(s, S - 1);
        const long row = tok_row(a.idx, p0 + sv);
        const T* qrow = qb + row * a.q.sn + kg * 8;
        bf16x8 qh[KST], ql[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            f32x4 x0, x1;
            ld8(qrow + ks * 32, x0, x1);
            if (a.relu) relu8(x0, x1, a.eps);
            uint4 hi, lo;
            split8(x0, x1, hi, lo);
            qh[ks] = as_bf16x8(hi);
            ql[ks] = as_bf16x8(lo);
        }
        const float ninv = a.normalize ? ninvb[sv] : 1.f;
        T* orow = ob + row * a.o.sn + kg * 4;
#pragma unroll
        for (int ct = 0; ct < DT; ct += 2) {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const bf16x8 a0h = tr_read8(Gh, LD, ks * 32, ct * 16, lane), a1h = tr_read8(Gh, LD, ks * 32, ct * 16 + 16, lane);
                const bf16x8 a0l = tr_read8(Gl, LD, ks * 32, ct * 16, lane), a1l = tr_read8(Gl, LD, ks * 32, ct * 16 + 16, lane);
                c0 = mfma_bf16(a0h, qh[ks], c0);
                c1 = mfma_bf16(a1h, qh[ks], c1);
                c0 = mfma_bf16(a0l, qh[ks], c0);
                c1 = mfma_bf16(a1l, qh[ks], c1);
                if (LO) {
                    c0 = mfma_bf16(a0h, ql[ks], c0);
                    c1 = mfma_bf16(a1h, ql[ks], c1);
                }
            }
            if (s < S) {
                Io<T>::st4(orow + ct * 16, c0 * ninv);
                Io<T>::st4(orow + ct * 16 + 16, c1 * ninv);
            }
        }
    }
}

}  // namespace sp
}  // namespace mhla
