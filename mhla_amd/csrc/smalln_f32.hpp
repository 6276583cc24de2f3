// Small-sequence path for fp32 tensors: block size S = 16, N = 16 M <= 256 tokens, D <= 80 -- the DiT-XL/2 operator as the
// reference trains it (mhla_dit/train.py:12-13: fp32 tensors, no autocast).  Same attention-form evaluation as smalln.hpp
//      O_i = sum_j W[i][j] (Q_i K_j^T) V_j / n_i            (no block summaries: they would be 4.5x the token bytes at this shape)
// with fp32 accuracy on the bf16 matrix pipe: every operand is carried as bf16 hi + lo (16 significand bits) and every product
// is three MFMAs (hi hi + hi lo + lo hi, fp32 accumulation), including the score tile, which is split again on its way from the
// accumulators into the second contraction.  One workgroup (8 waves) per (b, h); wave w owns blocks w and w + 8, one after the
// other (the joint two-block form of the bf16 kernel does not fit the register file with hi / lo operands).  The hi / lo planes of
// 256 rows do not fit in LDS (4 planes x 256 x 88 x 2 B = 180 KB): keys (forward, pass A) resp. queries (pass B) are staged in two
// halves of 128 rows, accumulators live across the halves.
#pragma once
#include "smalln.hpp"

namespace mhla {
namespace fast {

constexpr int SNF_T = 512, SNF_W = SNF_T / 64, SNF_HR = 128;

__device__ __forceinline__ void snf_split8(const f32x4& a, const f32x4& b, uint4& hi, uint4& lo) {
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
// two C-layout tiles -> A operand of the next contraction, as hi + lo parts
__device__ __forceinline__ void snf_pack_hl(const f32x4& c0, const f32x4& c1, bf16x8& hi, bf16x8& lo) {
    uint4 h, l;
    snf_split8(c0, c1, h, l);
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ f32x4 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
    c = mfma_bf16(ah, bh, c);
    c = mfma_bf16(ah, bl, c);
    return mfma_bf16(al, bh, c);
}

// rows [row0, row0 + nrows) (nrows <= 128) of an fp32 view -> hi / lo planes [128][LDR] (rows past nrows, columns past D: zeros);
// optional relu + eps, optional scaling of row r by rowscale[row0 + r].  All loads first, from clamped addresses.
template <int DT, bool SCALE>
__device__ __forceinline__ void snf_stage(u16* __restrict__ Hh, u16* __restrict__ Hl, const float* __restrict__ base, long sn,
                                          const int* __restrict__ idx, int row0, int nrows, int D, float eps, int tid, bool relu,
                                          const float* __restrict__ rowscale) {
    constexpr int LDR = sn_ldr<DT>(), PV = DT * 2, MAXIT = (SNF_HR * PV + SNF_T - 1) / SNF_T;
    const int dv = D >> 3;
    f32x4 reg[MAXIT][2];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SNF_T, r = v / PV, p = v - r * PV;
        const float* src = base + tok_row(idx, row0 + min(r, nrows - 1)) * sn + min(p, dv - 1) * 8;
        reg[t][0] = gld<f32x4>(src);
        reg[t][1] = gld<f32x4>(src + 4);
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SNF_T, r = v / PV, p = v - r * PV;
        if (r < SNF_HR) {
            f32x4 x0 = reg[t][0], x1 = reg[t][1];
            if (relu) {   // (uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i) { x0[i] = fmaxf(x0[i], 0.f) + eps; x1[i] = fmaxf(x1[i], 0.f) + eps; }
            }
            if (SCALE) {
                const float sc = rowscale[row0 + min(r, nrows - 1)];
                x0 *= sc;
                x1 *= sc;
            }
            uint4 hi, lo;
            snf_split8(x0, x1, hi, lo);
            const bool ok = r < nrows && p < dv;
            *reinterpret_cast<uint4*>(Hh + r * LDR + p * 8) = sel4(ok, hi);
            *reinterpret_cast<uint4*>(Hl + r * LDR + p * 8) = sel4(ok, lo);
        }
    }
}
// two tensors at once: both tensors' loads are issued before either is split and written (one memory round trip instead of two)
template <int DT, bool SCALE1>
__device__ __forceinline__ void snf_stage2(u16* __restrict__ H0h, u16* __restrict__ H0l, const float* __restrict__ base0, long sn0, float eps0, bool relu0,
                                           u16* __restrict__ H1h, u16* __restrict__ H1l, const float* __restrict__ base1, long sn1,
                                           const int* __restrict__ idx, int row0, int nrows, int D, int tid, const float* __restrict__ rowscale1) {
    constexpr int LDR = sn_ldr<DT>(), PV = DT * 2, MAXIT = (SNF_HR * PV + SNF_T - 1) / SNF_T;
    const int dv = D >> 3;
    f32x4 ra[MAXIT][2], rb[MAXIT][2];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SNF_T, r = v / PV, p = v - r * PV;
        const long row = tok_row(idx, row0 + min(r, nrows - 1));
        const float* s0 = base0 + row * sn0 + min(p, dv - 1) * 8;
        const float* s1 = base1 + row * sn1 + min(p, dv - 1) * 8;
        ra[t][0] = gld<f32x4>(s0);
        ra[t][1] = gld<f32x4>(s0 + 4);
        rb[t][0] = gld<f32x4>(s1);
        rb[t][1] = gld<f32x4>(s1 + 4);
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SNF_T, r = v / PV, p = v - r * PV;
        if (r < SNF_HR) {
            f32x4 x0 = ra[t][0], x1 = ra[t][1], y0 = rb[t][0], y1 = rb[t][1];
            if (relu0) {   // (uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i) { x0[i] = fmaxf(x0[i], 0.f) + eps0; x1[i] = fmaxf(x1[i], 0.f) + eps0; }
            }
            if (SCALE1) {
                const float sc = rowscale1[row0 + min(r, nrows - 1)];
                y0 *= sc;
                y1 *= sc;
            }
            const bool ok = r < nrows && p < dv;
            uint4 hi, lo;
            snf_split8(x0, x1, hi, lo);
            *reinterpret_cast<uint4*>(H0h + r * LDR + p * 8) = sel4(ok, hi);
            *reinterpret_cast<uint4*>(H0l + r * LDR + p * 8) = sel4(ok, lo);
            snf_split8(y0, y1, hi, lo);
            *reinterpret_cast<uint4*>(H1h + r * LDR + p * 8) = sel4(ok, hi);
            *reinterpret_cast<uint4*>(H1l + r * LDR + p * 8) = sel4(ok, lo);
        }
    }
}
// 16 rows of an fp32 view as an MFMA operand (lane: row lane & 15, columns 32 ks + 8 kg ..), hi + lo; optional relu + eps and a
// per-row scale (lane's row).  Columns past D: a clamped address, zeroed on arrival.
template <int KS>
__device__ __forceinline__ void snf_issue_rows(f32x4 (&v)[KS][2], const float* __restrict__ base, long sn, const int* __restrict__ idx,
                                               int row0, int D, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    const float* src = base + tok_row(idx, row0 + m) * sn;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = min(ks * 32 + kg * 8, D - 8);
        v[ks][0] = gld<f32x4>(src + c);
        v[ks][1] = gld<f32x4>(src + c + 4);
    }
}
template <int KS>
__device__ __forceinline__ void snf_finish_rows(bf16x8 (&h)[KS], bf16x8 (&l)[KS], const f32x4 (&v)[KS][2], int D, float eps, int lane,
                                                bool relu, float scale = 1.f) {
    const int kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        f32x4 x0 = v[ks][0], x1 = v[ks][1];
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { x0[i] = fmaxf(x0[i], 0.f) + eps; x1[i] = fmaxf(x1[i], 0.f) + eps; }
        }
        x0 *= scale;
        x1 *= scale;
        uint4 hi, lo;
        snf_split8(x0, x1, hi, lo);
        const bool ok = ks * 32 + kg * 8 < D;
        h[ks] = __builtin_bit_cast(bf16x8, sel4(ok, hi));
        l[ks] = __builtin_bit_cast(bf16x8, sel4(ok, lo));
    }
}
// column sums of 16-row blocks straight from the fp32 view: ksum[j][d] = sum_r relu?(x[16 j + r][d])  (M x DP values, all threads)
template <int DT>
__device__ __forceinline__ void snf_ksum(float* __restrict__ ksum_s, const float* __restrict__ base, long sn, const int* __restrict__ idx,
                                         int M, int D, float eps, int tid, bool relu) {
    constexpr int DP = DT * 16;
    for (int v = tid; v < M * DP; v += SNF_T) {
        const int j = v / DP, d = v - j * DP;
        float s = 0.f;
        if (d < D) {
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = gld<float>(base + tok_row(idx, j * 16 + r) * sn + d);
#pragma unroll
            for (int r = 0; r < 16; ++r) s += relu ? fmaxf(x[r], 0.f) + eps : x[r];
        }
        ksum_s[v] = s;
    }
}
// C-layout accumulators of one 16-row block (lane: rows 4 kg + r, column 16 tn + n) -> fp32 token rows, optional relu mask from `mbase`
template <int DT, bool MASK>
__device__ __forceinline__ void snf_store16(float* __restrict__ base, long sn, const int* __restrict__ idx, int row0, int D,
                                            const f32x4 (&acc)[DT], const float* __restrict__ mbase, long msn, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long tr = tok_row(idx, row0 + kg * 4 + r);
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            const int d = tn * 16 + n;
            if (d < D) {
                float x = acc[tn][r];
                if (MASK) x = mbase[tr * msn + d] > 0.f ? x : 0.f;
                base[tr * sn + d] = x;
            }
        }
    }
}

template <int DT>
__host__ __device__ constexpr int snf_smem() {
    return 4 * SNF_HR * sn_ldr<DT>() * 2 + (2 * 16 * DT * 16 + 5 * 256) * 4;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int DT, bool GATHER, bool M16 = false>
__global__ __launch_bounds__(SNF_T, 2) void k_snf_fwd(const SnArgs a) {
    // GATHER: the launch has a block_index map.  As a template parameter the row lookups carry no branch: with `idx ? idx[p] : p`
    // decided at run time hipcc branched around every map load and waited for ALL loads in flight at each join (s_waitcnt
    // vmcnt(0) after every group of row loads: the staging became a chain of dependent round trips).
    const int* const idx = GATHER ? a.idx : nullptr;
    if constexpr (GATHER) __builtin_assume(idx != nullptr);
    constexpr int DP = DT * 16, LDR = sn_ldr<DT>(), KS = (DP + 31) / 32, PL = SNF_HR * LDR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Kh = reinterpret_cast<u16*>(smem_raw);
    u16* Kl = Kh + PL;
    u16* Vh = Kl + PL;
    u16* Vl = Vh + PL;
    float* ksum_s = reinterpret_cast<float*>(Vl + PL);   // [M][DP]
    float* zs = ksum_s + 2 * 16 * DP;                     // [M][16]   (the layout of snf_smem is shared with the backward)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int bh = xcd_swizzle(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
    const int M = M16 ? 16 : a.M, D = a.D, N = M * 16;
    const float* qb = (const float*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const float* kb = (const float*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const float* vb = (const float*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    float* ob = (float*)a.out.ptr + b * a.out.sb + h * a.out.sh;
    const bool relu = a.relu != 0;

    __shared__ float Wsh[16 * 17];
    if (tid < 256) Wsh[(tid >> 4) * 17 + (tid & 15)] = ((tid >> 4) < M && (tid & 15) < M) ? a.W[(long)(tid >> 4) * a.ldw + (tid & 15)] : 0.f;
    if (a.normalize) snf_ksum<DT>(ksum_s, kb, a.k.sn, idx, M, D, a.eps, tid, relu);

    f32x4 acc[2][DT];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) acc[x][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    float zmine[2] = {0.f, 0.f};

    for (int half = 0; half * SNF_HR < N; ++half) {
        const int r0 = half * SNF_HR, nr = min(SNF_HR, N - r0), jb = half * 8;
        __syncthreads();   // the previous half's readers are done (first round: Wsh / ksum_s written)
        snf_stage2<DT, false>(Kh, Kl, kb, a.k.sn, a.eps, relu, Vh, Vl, vb, a.v.sn, idx, r0, nr, D, tid, nullptr);
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = wave + SNF_W * x;
            if (i >= M) continue;
            bf16x8 qh[KS], ql[KS];
            {
                f32x4 qraw[KS][2];
                snf_issue_rows<KS>(qraw, qb, a.q.sn, idx, i * 16, D, lane);
                snf_finish_rows<KS>(qh, ql, qraw, D, a.eps, lane, relu);
            }
            if (half == 0 && a.normalize) {   // z_i[s] = q_i[s] . ksum_i (lane: row n, 8-column pieces kg)
                float z = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks * 32 + kg * 8 < D) {
                        const s16x8 hs = __builtin_bit_cast(s16x8, qh[ks]), ls = __builtin_bit_cast(s16x8, ql[ks]);
#pragma unroll
                        for (int t = 0; t < 8; ++t) z += (bf((u16)hs[t]) + bf((u16)ls[t])) * ksum_s[i * DP + ks * 32 + kg * 8 + t];
                    }
                }
                z += __shfl_xor(z, 16, 64);
                z += __shfl_xor(z, 32, 64);
                zmine[x] = z;
            }
            for (int jp = 0; jp < 4; ++jp) {
                const int j0 = jb + 2 * jp;
                if (j0 >= M) break;
                const int j1 = j0 + 1 < M ? j0 + 1 : j0, l0 = (j0 - jb) * 16, l1 = (j1 - jb) * 16;
                const float w0 = Wsh[i * 17 + j0], w1 = j0 + 1 < M ? Wsh[i * 17 + j0 + 1] : 0.f;
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
                {
                    bf16x8 ah[KS], al[KS];
                    sn_lds_rows<KS>(ah, Kh, LDR, l0, D, lane);
                    sn_lds_rows<KS>(al, Kl, LDR, l0, D, lane);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) c0 = mfma3(ah[ks], al[ks], qh[ks], ql[ks], c0);   // S^T tile (j0, i): rows t, cols s
                    sn_lds_rows<KS>(ah, Kh, LDR, l1, D, lane);
                    sn_lds_rows<KS>(al, Kl, LDR, l1, D, lane);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) c1 = mfma3(ah[ks], al[ks], qh[ks], ql[ks], c1);
                }
                bf16x8 ph, pl;
                snf_pack_hl(c0 * w0, c1 * w1, ph, pl);
#pragma unroll
                for (int tn = 0; tn < DT; ++tn)
                    acc[x][tn] = mfma3(ph, pl, sn_tr_pair(Vh, LDR, l0, l1, tn * 16, lane), sn_tr_pair(Vl, LDR, l0, l1, tn * 16, lane), acc[x][tn]);
            }
        }
    }
    if (a.normalize) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = wave + SNF_W * x;
            if (i < M && kg == 0) zs[i * 16 + n] = zmine[x];
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int i = wave + SNF_W * x;
        if (i >= M) continue;
        float ninv = 1.f;   // lane n = row s of the block
        if (a.normalize) {
            float nn = a.eps;
            for (int j = 0; j < M; ++j) nn += Wsh[i * 17 + j] * zs[j * 16 + n];
            ninv = 1.f / nn;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ni = __shfl(ninv, kg * 4 + r, 64);
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) acc[x][tn][r] *= ni;
        }
        snf_store16<DT, false>(ob, a.out.sn, idx, i * 16, D, acc[x], nullptr, 0, lane);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward: dQ, dK, dV and the per-(b,h) partial of dW, one launch (the structure of k_sn_bwd, one block at a time, operands hi + lo)
//   pass A (wave owns query block i, K / V halves in LDS):  S^T, dP^T tiles -> dW[i][:], dS -> dQ_i
//   pass B (wave owns key block j, Q / dO' halves in LDS):  S, dP tiles -> P^T dO' = dV_j ; dS^T Q = dK_j ; dksum_j
// ------------------------------------------------------------------------------------------------------------------
template <int DT, bool GATHER, bool M16 = false>
__global__ __launch_bounds__(SNF_T, 2) void k_snf_bwd(const SnArgs a) {
    // GATHER: the launch has a block_index map.  As a template parameter the row lookups carry no branch: with `idx ? idx[p] : p`
    // decided at run time hipcc branched around every map load and waited for ALL loads in flight at each join (s_waitcnt
    // vmcnt(0) after every group of row loads: the staging became a chain of dependent round trips).
    const int* const idx = GATHER ? a.idx : nullptr;
    if constexpr (GATHER) __builtin_assume(idx != nullptr);
    constexpr int DP = DT * 16, LDR = sn_ldr<DT>(), KS = (DP + 31) / 32, PL = SNF_HR * LDR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ah = reinterpret_cast<u16*>(smem_raw);   // K (pass A) / Q (pass B), hi
    u16* Al = Ah + PL;
    u16* Bh = Al + PL;                            // V (pass A) / dO' (pass B), hi
    u16* Bl = Bh + PL;
    float* ksum_s = reinterpret_cast<float*>(Bl + PL);   // [M][DP]
    float* zs = ksum_s + 2 * 16 * DP;                    // [M][16]  (a second [M][DP] array is reserved after ksum_s)
    float* rds = zs + 256;                               // row dots dO . O
    float* nis = rds + 256;                              // 1 / n
    float* dns = nis + 256;                              // dn
    float* dzs = dns + 256;                              // dz
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int bh = xcd_swizzle(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
    const int M = M16 ? 16 : a.M, D = a.D, N = M * 16;
    auto base = [&](const View& w) { return (const float*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (float*)w.ptr + b * w.sb + h * w.sh; };
    const float *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *ob = base(a.o), *gb = base(a.dout);
    float *dqb = mbase(a.dq), *dkb = mbase(a.dk), *dvb = mbase(a.dv);
    float* dwp = a.dwp + (long)bh * M * M;
    const bool relu = a.relu != 0;

    __shared__ float Wsh[16 * 17];
    if (tid < 256) Wsh[(tid >> 4) * 17 + (tid & 15)] = ((tid >> 4) < M && (tid & 15) < M) ? a.W[(long)(tid >> 4) * a.ldw + (tid & 15)] : 0.f;
    for (int v = tid; v < 256; v += SNF_T) { nis[v] = 1.f; dns[v] = 0.f; dzs[v] = 0.f; zs[v] = 0.f; }
    if (a.normalize) {
        snf_ksum<DT>(ksum_s, kb, a.k.sn, idx, M, D, a.eps, tid, relu);
        __syncthreads();
        // z_i, row dots dO . O of the wave's blocks (lane: row n, 8-column pieces kg)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = wave + SNF_W * x;
            if (i >= M) continue;
            f32x4 qr[KS][2], gr[KS][2], orr[KS][2];
            snf_issue_rows<KS>(qr, qb, a.q.sn, idx, i * 16, D, lane);
            snf_issue_rows<KS>(gr, gb, a.dout.sn, idx, i * 16, D, lane);
            snf_issue_rows<KS>(orr, ob, a.o.sn, idx, i * 16, D, lane);
            float z = 0.f, rd = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks * 32 + kg * 8 < D) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        float qv = qr[ks][t >> 2][t & 3];
                        if (relu) qv = fmaxf(qv, 0.f) + a.eps;
                        z += qv * ksum_s[i * DP + ks * 32 + kg * 8 + t];
                        rd += gr[ks][t >> 2][t & 3] * orr[ks][t >> 2][t & 3];
                    }
                }
            }
            z += __shfl_xor(z, 16, 64); z += __shfl_xor(z, 32, 64);
            rd += __shfl_xor(rd, 16, 64); rd += __shfl_xor(rd, 32, 64);
            if (kg == 0) { zs[i * 16 + n] = z; rds[i * 16 + n] = rd; }
        }
        __syncthreads();
        for (int v = tid; v < N; v += SNF_T) {   // 1 / n, dn
            const int i = v >> 4, sx = v & 15;
            float nn = a.eps;
            for (int j = 0; j < M; ++j) nn += Wsh[i * 17 + j] * zs[j * 16 + sx];
            const float ni = 1.f / nn;
            nis[v] = ni;
            dns[v] = -rds[v] * ni;
        }
        __syncthreads();
        for (int v = tid; v < N; v += SNF_T) {   // dz = W^T dn
            const int j = v >> 4, sx = v & 15;
            float dz = 0.f;
            for (int i = 0; i < M; ++i) dz += Wsh[i * 17 + j] * dns[i * 16 + sx];
            dzs[v] = dz;
        }
    }
    __syncthreads();

    // ---- pass A ----
#pragma unroll 1
    for (int x = 0; x < 2; ++x) {
        const int i = wave + SNF_W * x;
        const bool live = i < M;
        const int ic = live ? i : 0;
        bf16x8 qh[KS], ql[KS], gh[KS], gl[KS];
        {
            f32x4 raw[KS][2];
            snf_issue_rows<KS>(raw, qb, a.q.sn, idx, ic * 16, D, lane);
            snf_finish_rows<KS>(qh, ql, raw, D, a.eps, lane, relu);
            snf_issue_rows<KS>(raw, gb, a.dout.sn, idx, ic * 16, D, lane);
            snf_finish_rows<KS>(gh, gl, raw, D, 0.f, lane, false, nis[ic * 16 + n]);   // dO' = dO / n
        }
        f32x4 acc[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ew[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) ew[j] = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half * SNF_HR >= N) break;   // (uniform)
            const int r0 = half * SNF_HR, nr = min(SNF_HR, N - r0), jb = half * 8;
            __syncthreads();
            snf_stage2<DT, false>(Ah, Al, kb, a.k.sn, a.eps, relu, Bh, Bl, vb, a.v.sn, idx, r0, nr, D, tid, nullptr);
            __syncthreads();
            if (live) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const int j0 = jb + 2 * jp;
                    if (j0 < M) {
                        const bool has1 = j0 + 1 < M;
                        const int j1 = has1 ? j0 + 1 : j0, l0 = (j0 - jb) * 16, l1 = (j1 - jb) * 16;
                        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                        {
                            bf16x8 th[KS], tl[KS];
                            sn_lds_rows<KS>(th, Ah, LDR, l0, D, lane);
                            sn_lds_rows<KS>(tl, Al, LDR, l0, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) s0 = mfma3(th[ks], tl[ks], qh[ks], ql[ks], s0);   // S^T (j0, i)
                            sn_lds_rows<KS>(th, Ah, LDR, l1, D, lane);
                            sn_lds_rows<KS>(tl, Al, LDR, l1, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) s1 = mfma3(th[ks], tl[ks], qh[ks], ql[ks], s1);
                            sn_lds_rows<KS>(th, Bh, LDR, l0, D, lane);
                            sn_lds_rows<KS>(tl, Bl, LDR, l0, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) p0 = mfma3(th[ks], tl[ks], gh[ks], gl[ks], p0);   // dP^T (j0, i)
                            sn_lds_rows<KS>(th, Bh, LDR, l1, D, lane);
                            sn_lds_rows<KS>(tl, Bl, LDR, l1, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) p1 = mfma3(th[ks], tl[ks], gh[ks], gl[ks], p1);
                        }
                        // dW[i][j] = sum(dP . S) + sum_s dn_i[s] z_j[s]: lane partials, reduced once per query block below
                        float e0 = s0[0] * p0[0] + s0[1] * p0[1] + s0[2] * p0[2] + s0[3] * p0[3];
                        float e1 = s1[0] * p1[0] + s1[1] * p1[1] + s1[2] * p1[2] + s1[3] * p1[3];
                        if (a.normalize && kg == 0) {
                            e0 += dns[i * 16 + n] * zs[j0 * 16 + n];
                            e1 += dns[i * 16 + n] * zs[j1 * 16 + n];
                        }
                        ew[2 * jp + 8 * half] = e0;
                        ew[2 * jp + 1 + 8 * half] = has1 ? e1 : 0.f;
                        const float w0 = Wsh[i * 17 + j0], w1 = has1 ? Wsh[i * 17 + j1] : 0.f;
                        bf16x8 dh, dl;
                        snf_pack_hl(p0 * w0, p1 * w1, dh, dl);   // dS^T pair -> A operand (m = s, k-slots = t)
#pragma unroll
                        for (int tn = 0; tn < DT; ++tn)
                            acc[tn] = mfma3(dh, dl, sn_tr_pair(Ah, LDR, l0, l1, tn * 16, lane), sn_tr_pair(Al, LDR, l0, l1, tn * 16, lane), acc[tn]);
                    }
                }
            }
        }
        if (live) {
            const float tot = wave_reduce16(ew, lane);
            const int jw = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            if ((lane & 3) == 0 && jw < M) dwp[i * M + jw] = tot;
            if (a.normalize) {   // + dz (x) ksum
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dz = dzs[i * 16 + kg * 4 + r];
#pragma unroll
                    for (int tn = 0; tn < DT; ++tn) acc[tn][r] += dz * ksum_s[i * DP + tn * 16 + n];
                }
            }
            if (relu) snf_store16<DT, true>(dqb, a.dq.sn, idx, i * 16, D, acc, qb, a.q.sn, lane);
            else      snf_store16<DT, false>(dqb, a.dq.sn, idx, i * 16, D, acc, nullptr, 0, lane);
        }
    }

    // ---- pass B ----
#pragma unroll 1
    for (int x = 0; x < 2; ++x) {
        const int j = wave + SNF_W * x;
        const bool live = j < M;
        const int jc = live ? j : 0;
        bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
        {
            f32x4 raw[KS][2];
            snf_issue_rows<KS>(raw, kb, a.k.sn, idx, jc * 16, D, lane);
            snf_finish_rows<KS>(kh, kl, raw, D, a.eps, lane, relu);
            snf_issue_rows<KS>(raw, vb, a.v.sn, idx, jc * 16, D, lane);
            snf_finish_rows<KS>(vh, vl, raw, D, 0.f, lane, false);
        }
        f32x4 accK[DT], accV[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) accK[tn] = accV[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half * SNF_HR >= N) break;   // (uniform)
            const int r0 = half * SNF_HR, nr = min(SNF_HR, N - r0), ib = half * 8;
            __syncthreads();
            snf_stage2<DT, true>(Ah, Al, qb, a.q.sn, a.eps, relu, Bh, Bl, gb, a.dout.sn, idx, r0, nr, D, tid, nis);
            __syncthreads();
            if (live) {
#pragma unroll
                for (int ip = 0; ip < 4; ++ip) {
                    const int i0 = ib + 2 * ip;
                    if (i0 < M) {
                        const bool has1 = i0 + 1 < M;
                        const int i1 = has1 ? i0 + 1 : i0, l0 = (i0 - ib) * 16, l1 = (i1 - ib) * 16;
                        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
                        {
                            bf16x8 th[KS], tl[KS];
                            sn_lds_rows<KS>(th, Ah, LDR, l0, D, lane);
                            sn_lds_rows<KS>(tl, Al, LDR, l0, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) s0 = mfma3(th[ks], tl[ks], kh[ks], kl[ks], s0);   // S (i0, j): rows s, cols t
                            sn_lds_rows<KS>(th, Ah, LDR, l1, D, lane);
                            sn_lds_rows<KS>(tl, Al, LDR, l1, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) s1 = mfma3(th[ks], tl[ks], kh[ks], kl[ks], s1);
                            sn_lds_rows<KS>(th, Bh, LDR, l0, D, lane);
                            sn_lds_rows<KS>(tl, Bl, LDR, l0, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) p0 = mfma3(th[ks], tl[ks], vh[ks], vl[ks], p0);   // dP (i0, j)
                            sn_lds_rows<KS>(th, Bh, LDR, l1, D, lane);
                            sn_lds_rows<KS>(tl, Bl, LDR, l1, D, lane);
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) p1 = mfma3(th[ks], tl[ks], vh[ks], vl[ks], p1);
                        }
                        const float w0 = Wsh[i0 * 17 + j], w1 = has1 ? Wsh[i1 * 17 + j] : 0.f;
                        bf16x8 ph, pl, dh, dl;
                        snf_pack_hl(s0 * w0, s1 * w1, ph, pl);   // P pair  -> A operand (m = t, k-slots = s)
                        snf_pack_hl(p0 * w0, p1 * w1, dh, dl);   // dS pair
#pragma unroll
                        for (int tn = 0; tn < DT; ++tn) {
                            accV[tn] = mfma3(ph, pl, sn_tr_pair(Bh, LDR, l0, l1, tn * 16, lane), sn_tr_pair(Bl, LDR, l0, l1, tn * 16, lane), accV[tn]);   // dV += P^T dO'
                            accK[tn] = mfma3(dh, dl, sn_tr_pair(Ah, LDR, l0, l1, tn * 16, lane), sn_tr_pair(Al, LDR, l0, l1, tn * 16, lane), accK[tn]);   // dK += dS^T Q
                        }
                    }
                }
                if (a.normalize && (j >> 3) == half) {
                    // dksum_j[d] = sum_s dz_j[s] q_j[s][d]: every row of the A operand is dz_j (k-slots of the first tile of the pair),
                    // the B operand the transposed Q_j tile of the staged half; all rows of the result are equal
                    f32x4 dz4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz4[r] = dzs[j * 16 + kg * 4 + r];
                    bf16x8 zh, zl;
                    snf_pack_hl(dz4, f32x4{0.f, 0.f, 0.f, 0.f}, zh, zl);
                    const int lj = (j - ib) * 16;
#pragma unroll
                    for (int tn = 0; tn < DT; ++tn) {
                        f32x4 dk = {0.f, 0.f, 0.f, 0.f};
                        dk = mfma3(zh, zl, sn_tr_pair(Ah, LDR, lj, lj, tn * 16, lane), sn_tr_pair(Al, LDR, lj, lj, tn * 16, lane), dk);
#pragma unroll
                        for (int r = 0; r < 4; ++r) accK[tn][r] += dk[0];
                    }
                }
            }
        }
        if (live) {
            if (relu) snf_store16<DT, true>(dkb, a.dk.sn, idx, j * 16, D, accK, kb, a.k.sn, lane);
            else      snf_store16<DT, false>(dkb, a.dk.sn, idx, j * 16, D, accK, nullptr, 0, lane);
            snf_store16<DT, false>(dvb, a.dv.sn, idx, j * 16, D, accV, nullptr, 0, lane);
        }
    }
}

}  // namespace fast
}  // namespace mhla
