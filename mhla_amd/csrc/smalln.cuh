// Small-sequence path of the block-mixing MHLA operator: bf16, block size S = 16, N = 16 M <= 256 tokens, D <= 80
// (the DiT / ViT regime: N = 256, M = 16, D = 64 or 72).  One workgroup per (b, h), ONE launch per direction.
//
// With S = 16 every (query block i, key block j) pair is exactly one 16 x 16 MFMA tile, so the operator is evaluated in
// its attention form      O_i = sum_j W[i][j] (Q_i K_j^T) V_j / n_i
// where the mixing weight is a scalar multiply of the score tile.  At N = 256 this costs ~2.4x the FLOPs of the
// summary form but needs no block summaries at all: Q, K, V are read once, O written once (the summary form moves
// 6.75x the token bytes through the KV / G workspaces at this shape).
//
// Data flow of one block row i (owned by one wave), flash-attention style without softmax:
//   S^T tile (j, i) = K_j Q_i^T   (A = rows of K from LDS, B = Q_i held in registers; C layout: lane = (s, 4 t's))
//   scale by W[i][j], pack two tiles (j0, j1) to bf16  ->  this IS the A operand (m = s, 32 k-slots = the 2 x 16 t's) of
//   O_i += P V     with B = V rows fetched by two hardware transpose reads that follow the same k-slot order.
// No shuffle or LDS round trip sits between the two contractions.
#pragma once
#include "fused.cuh"

namespace mhla {
namespace fast {

constexpr int SN_T = 256;   // threads

struct SnArgs {
    View q, k, v, o, dout;
    MView out, dq, dk, dv;
    const int* idx;
    const float* W;
    int ldw;
    float* dwp;     // [bh][M][M] partial dW (backward)
    int H, M, D;
    float eps;
    int relu, normalize;
};

template <int DT>
__host__ __device__ constexpr int sn_ldr() { return DT * 16 + 8; }   // LDS row stride (bf16): 72 (144 B) / 88 (176 B)

// stage `nrows` token rows (D valid columns, zero up to DP) into an LDS tile [nrows][LDR]
template <int DT, bool RELU>
__device__ __forceinline__ void sn_stage(u16* __restrict__ dst, const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                         int nrows, int D, float eps, int tid) {
    constexpr int LDR = sn_ldr<DT>(), PV = DT * 2;   // 16-byte pieces per padded row
    const int dv = D >> 3;
    constexpr int MAXIT = (256 * PV + SN_T - 1) / SN_T;
    uint4 reg[MAXIT];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SN_T, r = v / PV, p = v - r * PV;
        reg[t] = make_uint4(0, 0, 0, 0);
        if (r < nrows && p < dv) {
            reg[t] = *reinterpret_cast<const uint4*>(base + tok_row(idx, r) * sn + p * 8);
            if (RELU) reg[t] = relu_eps8(reg[t], eps);
        }
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * SN_T, r = v / PV, p = v - r * PV;
        if (r < nrows) *reinterpret_cast<uint4*>(dst + r * LDR + p * 8) = reg[t];
    }
}

// 16 rows x KS k-steps of an MFMA operand straight from global: lane (m = lane & 15, kg) -> row0 + m, cols 32 ks + 8 kg ..
template <int KS, bool RELU>
__device__ __forceinline__ void sn_load_rows(bf16x8 (&a)[KS], const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                             int row0, int D, float eps, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    const u16* src = base + tok_row(idx, row0 + m) * sn + kg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ks * 32 + kg * 8 < D) {
            v = *reinterpret_cast<const uint4*>(src + ks * 32);
            if (RELU) v = relu_eps8(v, eps);
        }
        a[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// the same from an LDS tile [rows][LDR]
template <int KS>
__device__ __forceinline__ void sn_lds_rows(bf16x8 (&a)[KS], const u16* __restrict__ tile, int ldr, int row0, int D, int lane) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ks * 32 + kg * 8 < D) v = *reinterpret_cast<const uint4*>(tile + (row0 + m) * ldr + ks * 32 + kg * 8);
        a[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// B operand whose 32 k-slots are (tile t0 rows 4 kg .. +3, tile t1 rows 4 kg .. +3): two transpose reads
__device__ __forceinline__ bf16x8 sn_tr_pair(const u16* __restrict__ tile, int ldr, int row_t0, int row_t1, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const u16* p0 = tile + (row_t0 + g * 4 + (li >> 2)) * ldr + c0 + (li & 3) * 4;
    const u16* p1 = tile + (row_t1 + g * 4 + (li >> 2)) * ldr + c0 + (li & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 sn_pack_pair(f32x4 c0, f32x4 c1) {
    uint4 v;
    v.x = pack_bf16x2(c0[0], c0[1]); v.y = pack_bf16x2(c0[2], c0[3]);
    v.z = pack_bf16x2(c1[0], c1[1]); v.w = pack_bf16x2(c1[2], c1[3]);
    return __builtin_bit_cast(bf16x8, v);
}

template <int DT>
__host__ __device__ constexpr int sn_fwd_smem() {
    return 2 * 256 * sn_ldr<DT>() * 2 + (16 * DT * 16 + 256 + 4 * 16 * (DT * 16 + 8) / 2) * 4;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(SN_T) void k_sn_fwd(const SnArgs a) {
    constexpr int DP = DT * 16, LDR = sn_ldr<DT>(), KS = (DP + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ks = reinterpret_cast<u16*>(smem_raw);          // [N][LDR]
    u16* Vs = Ks + 256 * LDR;                            // [N][LDR]
    float* ksum_s = reinterpret_cast<float*>(Vs + 256 * LDR);   // [M][DP]
    float* zs = ksum_s + 16 * DP;                        // [M][16]
    u16* Ost = reinterpret_cast<u16*>(zs + 256);         // [4 waves][16][LDR] output staging
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int M = a.M, D = a.D, N = M * 16;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    u16* ob = (u16*)a.out.ptr + b * a.out.sb + h * a.out.sh;

    if (a.relu) sn_stage<DT, true>(Ks, kb, a.k.sn, a.idx, N, D, a.eps, tid);
    else        sn_stage<DT, false>(Ks, kb, a.k.sn, a.idx, N, D, a.eps, tid);
    sn_stage<DT, false>(Vs, vb, a.v.sn, a.idx, N, D, 0.f, tid);
    __syncthreads();
    if (a.normalize) {
        for (int v = tid; v < M * DP; v += SN_T) {
            const int j = v / DP, d = v - j * DP;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += bf(Ks[(j * 16 + r) * LDR + d]);
            ksum_s[v] = s;
        }
        __syncthreads();
    }
    // own blocks: i = wave, wave + 4, ...   Q_i rows as MFMA operands; z_i
    bf16x8 qa[4][KS];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int i = wave + 4 * x;
        if (i < M) {
            if (a.relu) sn_load_rows<KS, true>(qa[x], qb, a.q.sn, a.idx, i * 16, D, a.eps, lane);
            else        sn_load_rows<KS, false>(qa[x], qb, a.q.sn, a.idx, i * 16, D, a.eps, lane);
            if (a.normalize) {
                float z = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const s16x8 qs = __builtin_bit_cast(s16x8, qa[x][ks]);
                    if (ks * 32 + kg * 8 < D) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) z += bf((u16)qs[t]) * ksum_s[i * DP + ks * 32 + kg * 8 + t];
                    }
                }
                z += __shfl_xor(z, 16, 64);
                z += __shfl_xor(z, 32, 64);
                if (kg == 0) zs[i * 16 + n] = z;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int i = wave + 4 * x;
        if (i >= M) continue;
        float ninv = 1.f;   // lane n = row s of the block
        if (a.normalize) {
            float nn = a.eps;
            for (int j = 0; j < M; ++j) nn += a.W[(long)i * a.ldw + j] * zs[j * 16 + n];
            ninv = 1.f / nn;
        }
        f32x4 acc[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < M; j0 += 2) {
            const int j1 = j0 + 1 < M ? j0 + 1 : j0;
            const float w0 = a.W[(long)i * a.ldw + j0], w1 = j0 + 1 < M ? a.W[(long)i * a.ldw + j0 + 1] : 0.f;
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
            bf16x8 ka[KS], kb2[KS];
            sn_lds_rows<KS>(ka, Ks, LDR, j0 * 16, D, lane);
            sn_lds_rows<KS>(kb2, Ks, LDR, j1 * 16, D, lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                c0 = mfma_bf16(ka[ks], qa[x][ks], c0);     // S^T tile (j0, i): rows t, cols s
                c1 = mfma_bf16(kb2[ks], qa[x][ks], c1);
            }
            const bf16x8 pa = sn_pack_pair(c0 * w0, c1 * w1);
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) acc[tn] = mfma_bf16(pa, sn_tr_pair(Vs, LDR, j0 * 16, j1 * 16, tn * 16, lane), acc[tn]);
        }
        // O rows: C layout lane (col d2 = 16 tn + n, rows s = 4 kg + r); scale by 1/n[s]; stage; coalesced store
        u16* Os = Ost + wave * 16 * LDR;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ni = __shfl(ninv, kg * 4 + r, 64);
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) Os[(kg * 4 + r) * LDR + tn * 16 + n] = cvt_bf16(acc[tn][r] * ni);
        }
        wave_lds_fence();
        const int dv = D >> 3;
        for (int v = lane; v < 16 * dv; v += 64) {
            const int r = v / dv, p = v - r * dv;
            *reinterpret_cast<uint4*>(ob + tok_row(a.idx, i * 16 + r) * a.out.sn + p * 8) = *reinterpret_cast<const uint4*>(Os + r * LDR + p * 8);
        }
        wave_lds_fence();
    }
}

}  // namespace fast
}  // namespace mhla
