// Block-mixing token kernels for blocks of exactly 16 tokens: bf16 tensors, D = 64, more than 64 blocks (the 256 x 16 variant of
// the micro-bench; DiT-style block_size = 16 on long sequences).  Same workspace formats and the same arithmetic
// as split.hpp's kernels for bf16 tensors (16-bit summaries [bh][M][64][64], fp32 ksum / z / 1/n / dn / dz), so each kernel is a
// drop-in for its split.hpp counterpart; what changes is the shape of the work:
//
//   split.hpp gives a block to a workgroup of four waves.  With 16 tokens a block is ONE 16-row MFMA tile: three waves only help
//   to stage the 8 KB summary and then wait, and a workgroup is a chain of three dependent memory round trips with 4-12 KB in
//   flight (k_sp_state 2.8 TB/s, k_sp_bwd_dq 1.9 TB/s at M = 256).
//
//   Here a block belongs to ONE WAVE: its tiles are private to the wave (no block barrier anywhere, `wave_lds_fence` orders the
//   wave's own LDS writes and transposed reads), every global access is a 16-byte piece, and whatever is already an MFMA operand
//   in memory order is loaded straight into the operand registers:
//     k_s16_state<0> : KV_j = K_j^T V_j (v_mfma 16x16x16: the 16 tokens are the whole reduction), ksum_j, z_j
//     k_s16_state<1> : dG_i = Q_i^T (dO_i / n_i), dn_i
//     k_s16_out      : O_i = Q_i G_i / n_i              G_i through LDS (transposed operand), Q rows direct
//     k_s16_bwd_dq   : dQ_i = (dO_i / n_i) G_i^T + dz_i ksum_i^T       no LDS at all: G rows and dO rows are operands as stored
//     k_s16_bwd_dkv  : dK_j = V_j dKV_j^T + 1 dksum_j^T, dV_j = K_j dKV_j; dKV rows direct (dK) and through LDS (dV, transposed);
//                      dksum_j = Qden_j^T dz_j is two more MFMAs per feature tile (dz as bf16 hi + lo in every column of the B
//                      operand) accumulated straight into dK -- no dksum round trip through HBM, no shuffles
//   k_sp_dwr: dW partials for 64 < M <= 256 with the whole M x M matrix in one workgroup (one 64 x 64 tile per wave): every
//   summary row is read ONCE per (b, h) (k_sp_dw: once per 64-column tile of dW, i.e. four times through L2 at M = 256), rows
//   staged through LDS in whole 128-byte lines, the <dn_i, z_j> term is one more stage of the same loop.
#pragma once
#include <type_traits>
#include "split.hpp"

namespace mhla {
namespace s16 {

using fast::bf16x8;
using fast::dot8;
using fast::mask_pos8;
using fast::mfma_bf16;
using fast::pair_pieces;
using fast::relu_eps8;
using fast::s16x4;
using fast::tr_read8;
using fast::u16;
using fast::wave_lds_fence;
using sp::as_bf16x8;

constexpr int WPB = 4;    // waves (= blocks) per workgroup
constexpr int LD = 72;    // LDS row stride (bf16) of the token tiles [16][64] and summary matrices [64][64]

__host__ __device__ constexpr int state_smem() { return WPB * 2 * 16 * LD * 2; }
__host__ __device__ constexpr int out_smem() { return WPB * 64 * LD * 2; }
__host__ __device__ constexpr int dkv_smem() { return WPB * 80 * LD * 2; }

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
// operand of the 16-step MFMA from a row-major 16-row tile: lane (c = lane & 15, g = lane >> 4) receives T[4 g + 0..3][c0 + c]
__device__ __forceinline__ s16x4 tr_read4(const u16* tile, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(tile + (g * 4 + (li >> 2)) * LD + c0 + (li & 3) * 4));
}
__device__ __forceinline__ uint4 scale8(uint4 v, float s) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(__uint_as_float(w[i] << 16) * s, __uint_as_float(w[i] & 0xffff0000u) * s);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void unpack8(uint4 v, float (&x)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[2 * i] = __uint_as_float(w[i] << 16);
        x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint2 pack4(f32x4 c) { return make_uint2(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3])); }
// feature offset of the 16-byte piece a lane owns after pair_pieces of the feature tiles 2 p and 2 p + 1
__device__ __forceinline__ int pair_col(int p, int kg) { return (2 * p + (kg & 1)) * 16 + 8 * (kg >> 1); }

// MODE 0: out = KV_j = K_j^T V_j; ksum_j; z_j          x = k_num, y = v, qd = q_den (= q_num: no separate normaliser pair here)
// MODE 1: out = dG_i = Q_i^T (dO_i / n_i); dn_i        x = q_num, y = dout, o = forward output
template <int MODE>
__global__ __launch_bounds__(64 * WPB) void k_s16_state(const StateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x * WPB + wave, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, M = a.M;
    if (blk >= M) return;   // (no block barriers below)
    u16* Xs = reinterpret_cast<u16*>(smem_raw) + wave * (2 * 16 * LD);
    u16* Ys = Xs + 16 * LD;
    const long p0 = (long)blk * 16, sb = (long)bh * M + blk;
    const int r = lane >> 3, c8 = (lane & 7) * 8;   // staging: the lane's 16-byte piece of token rows r and r + 8
    const long row0 = tok_row(a.idx, p0 + r), row1 = tok_row(a.idx, p0 + r + 8);
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh + c8;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh + c8;
    const View& third = MODE == 0 ? a.qd : a.o;
    const u16* tb = (const u16*)third.ptr + b * third.sb + h * third.sh + c8;
    uint4 x0 = gld<uint4>(xb + row0 * a.x.sn), x1 = gld<uint4>(xb + row1 * a.x.sn);
    uint4 y0 = gld<uint4>(yb + row0 * a.y.sn), y1 = gld<uint4>(yb + row1 * a.y.sn);
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    float n0 = 1.f, n1 = 1.f;
    if (a.normalize) {
        t0 = gld<uint4>(tb + row0 * third.sn);
        t1 = gld<uint4>(tb + row1 * third.sn);
        if (MODE == 1) {
            n0 = gld<float>(a.ninv + sb * 16 + r);
            n1 = gld<float>(a.ninv + sb * 16 + r + 8);
        }
    }
    if (a.relu) {
        x0 = relu_eps8(x0, a.eps);
        x1 = relu_eps8(x1, a.eps);
    }
    if (MODE == 1 && a.normalize) {   // dn[s] = -(dO[s] . O[s]) / n[s]; dO / n feeds the product
        float d0 = dot8(y0, t0), d1 = dot8(y1, t1);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            d0 += __shfl_xor(d0, o, 64);
            d1 += __shfl_xor(d1, o, 64);
        }
        if ((lane & 7) == 0) {
            a.dn[sb * 16 + r] = -d0 * n0;
            a.dn[sb * 16 + r + 8] = -d1 * n1;
        }
        y0 = scale8(y0, n0);
        y1 = scale8(y1, n1);
    }
    *reinterpret_cast<uint4*>(Xs + r * LD + c8) = x0;
    *reinterpret_cast<uint4*>(Xs + (r + 8) * LD + c8) = x1;
    *reinterpret_cast<uint4*>(Ys + r * LD + c8) = y0;
    *reinterpret_cast<uint4*>(Ys + (r + 8) * LD + c8) = y1;
    wave_lds_fence();
    // C[m = d2][n = d1] = sum_s Y[s][d2] X[s][d1]: a lane ends up with 4 consecutive d2 of row d1 = 16 tn + nl
    s16x4 bx[4], ay[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bx[t] = tr_read4(Xs, t * 16, lane);
        ay[t] = tr_read4(Ys, t * 16, lane);
    }
    u16* ob = reinterpret_cast<u16*>(a.out) + sb * a.es + nl * 64;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 c0 = mfma16(ay[2 * p], bx[tn], zero), c1 = mfma16(ay[2 * p + 1], bx[tn], zero);
            gst<uint4>(ob + tn * 16 * 64 + pair_col(p, kg), pair_pieces(pack4(c0), pack4(c1)));
        }
    if (MODE == 0 && a.normalize) {
        // ksum_j[d] = sum_s k[s][d]: the lane's two rows, then the 8 lanes that hold the same columns
        float ka[8], kb[8], ks[8];
        unpack8(x0, ka);
        unpack8(x1, kb);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ks[i] = ka[i] + kb[i];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) ks[i] += __shfl_xor(ks[i], o, 64);
        }
        if (lane < 8) {
            float* kd = a.ksum + sb * 64 + c8;
            gst<f32x4>(kd, f32x4{ks[0], ks[1], ks[2], ks[3]});
            gst<f32x4>(kd + 4, f32x4{ks[4], ks[5], ks[6], ks[7]});
        }
        // z_j[s] = Qden_j[s] . ksum_j
        if (a.relu) {
            t0 = relu_eps8(t0, a.eps);
            t1 = relu_eps8(t1, a.eps);
        }
        float qa[8], qb[8], z0 = 0.f, z1 = 0.f;
        unpack8(t0, qa);
        unpack8(t1, qb);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            z0 += qa[i] * ks[i];
            z1 += qb[i] * ks[i];
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            z0 += __shfl_xor(z0, o, 64);
            z1 += __shfl_xor(z1, o, 64);
        }
        if ((lane & 7) == 0) {
            a.zo[sb * 16 + r] = z0;
            a.zo[sb * 16 + r + 8] = z1;
        }
    }
}

// O_i^T[d2][s] = sum_d1 G_i[d1][d2] q[s][d1]: a lane ends up with 4 consecutive features of token s = nl per feature tile
template <int UNIT = 0>   // (a template so that the header can be part of several translation units)
__global__ __launch_bounds__(64 * WPB) void k_s16_out(const OutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x * WPB + wave, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, M = a.M;
    if (blk >= M) return;
    u16* Gs = reinterpret_cast<u16*>(smem_raw) + wave * (64 * LD);
    const long p0 = (long)blk * 16, sb = (long)bh * M + blk;
    const int r = lane >> 3, c8 = (lane & 7) * 8;
    const u16* g = reinterpret_cast<const u16*>(a.g) + sb * a.es + r * 64 + c8;
    uint4 gv[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) gv[p] = gld<uint4>(g + p * 8 * 64);
    const long row = tok_row(a.idx, p0 + nl);
    const u16* qrow = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh + row * a.q.sn + kg * 8;
    uint4 q0 = gld<uint4>(qrow), q1 = gld<uint4>(qrow + 32);
    const float ninv = a.normalize ? gld<float>(a.ninv + sb * 16 + nl) : 1.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<uint4*>(Gs + (p * 8 + r) * LD + c8) = gv[p];
    if (a.relu) {
        q0 = relu_eps8(q0, a.eps);
        q1 = relu_eps8(q1, a.eps);
    }
    wave_lds_fence();
    f32x4 c[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        c[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        c[ct] = mfma_bf16(tr_read8(Gs, LD, 0, ct * 16, lane), as_bf16x8(q0), c[ct]);
        c[ct] = mfma_bf16(tr_read8(Gs, LD, 32, ct * 16, lane), as_bf16x8(q1), c[ct]);
    }
    u16* orow = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh + row * a.o.sn;
#pragma unroll
    for (int p = 0; p < 2; ++p) gst<uint4>(orow + pair_col(p, kg), pair_pieces(pack4(c[2 * p] * ninv), pack4(c[2 * p + 1] * ninv)));
}

// dQ_i^T[d1][s] = sum_d2 G_i[d1][d2] (dO[s][d2] / n[s]) + ksum_i[d1] dz[s]: both operands are contiguous along d2 in memory
template <int UNIT = 0>   // (a template so that the header can be part of several translation units)
__global__ __launch_bounds__(64 * WPB) void k_s16_bwd_dq(const TokArgs a) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x * WPB + wave, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, M = a.M;
    if (blk >= M) return;
    const long p0 = (long)blk * 16, sb = (long)bh * M + blk;
    const u16* g = reinterpret_cast<const u16*>(a.g) + sb * a.es + nl * 64 + kg * 8;
    uint4 A[4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        A[ct][0] = gld<uint4>(g + ct * 16 * 64);
        A[ct][1] = gld<uint4>(g + ct * 16 * 64 + 32);
    }
    const long row = tok_row(a.idx, p0 + nl);
    const u16* grow = (const u16*)a.dout.ptr + b * a.dout.sb + h * a.dout.sh + row * a.dout.sn + kg * 8;
    uint4 d0 = gld<uint4>(grow), d1 = gld<uint4>(grow + 32);
    float dz = 0.f;
    f32x4 ks4[4];
    if (a.normalize) {
        const float ninv = gld<float>(a.ninv + sb * 16 + nl);
        dz = gld<float>(a.dz + sb * 16 + nl);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) ks4[ct] = gld<f32x4>(a.ksum + sb * 64 + ct * 16 + kg * 4);
        d0 = scale8(d0, ninv);
        d1 = scale8(d1, ninv);
    }
    const u16* qrow = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh + row * a.q.sn;
    uint4 qm[2];
    if (a.relu) {
        qm[0] = gld<uint4>(qrow + pair_col(0, kg));
        qm[1] = gld<uint4>(qrow + pair_col(1, kg));
    }
    f32x4 c[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        c[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        c[ct] = mfma_bf16(as_bf16x8(A[ct][0]), as_bf16x8(d0), c[ct]);
        c[ct] = mfma_bf16(as_bf16x8(A[ct][1]), as_bf16x8(d1), c[ct]);
        if (a.normalize) c[ct] += dz * ks4[ct];
    }
    u16* orow = (u16*)a.dq.ptr + b * a.dq.sb + h * a.dq.sh + row * a.dq.sn;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint4 v = pair_pieces(pack4(c[2 * p]), pack4(c[2 * p + 1]));
        if (a.relu) v = mask_pos8(v, qm[p]);
        gst<uint4>(orow + pair_col(p, kg), v);
    }
}

template <int UNIT = 0>   // (a template so that the header can be part of several translation units)
__global__ __launch_bounds__(64 * WPB) void k_s16_bwd_dkv(const TokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x * WPB + wave, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, M = a.M;
    if (blk >= M) return;
    u16* Gs = reinterpret_cast<u16*>(smem_raw) + wave * (80 * LD);   // dKV_j [d1][d2]
    u16* Qs = Gs + 64 * LD;                                           // Qden_j [s][d] (normalised operator)
    const long p0 = (long)blk * 16, sb = (long)bh * M + blk;
    // dKV rows in the layout of the dK product's A operand (m = d1 = 16 ct + nl, k = d2 = 32 ks + 8 kg ..): used as loaded,
    // and written to LDS for the transposed reads of the dV product
    const u16* g = reinterpret_cast<const u16*>(a.dkv) + sb * a.es + nl * 64 + kg * 8;
    uint4 A[4][2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        A[ct][0] = gld<uint4>(g + ct * 16 * 64);
        A[ct][1] = gld<uint4>(g + ct * 16 * 64 + 32);
    }
    const long row = tok_row(a.idx, p0 + nl);
    const u16* krow = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh + row * a.k.sn;
    const u16* vrow = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh + row * a.v.sn;
    uint4 k0 = gld<uint4>(krow + kg * 8), k1 = gld<uint4>(krow + kg * 8 + 32);
    const uint4 v0 = gld<uint4>(vrow + kg * 8), v1 = gld<uint4>(vrow + kg * 8 + 32);
    s16x4 dzh = {0, 0, 0, 0}, dzl = dzh;
    if (a.normalize) {
        const int r = lane >> 3, c8 = (lane & 7) * 8;
        const u16* qb = (const u16*)a.qd.ptr + b * a.qd.sb + h * a.qd.sh + c8;
        uint4 qa = gld<uint4>(qb + tok_row(a.idx, p0 + r) * a.qd.sn), qc = gld<uint4>(qb + tok_row(a.idx, p0 + r + 8) * a.qd.sn);
        const f32x4 dz4 = gld<f32x4>(a.dz + sb * 16 + kg * 4);
        if (a.relu) {
            qa = relu_eps8(qa, a.eps);
            qc = relu_eps8(qc, a.eps);
        }
        *reinterpret_cast<uint4*>(Qs + r * LD + c8) = qa;
        *reinterpret_cast<uint4*>(Qs + (r + 8) * LD + c8) = qc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned short hi = cvt_bf16(dz4[i]);
            dzh[i] = (short)hi;
            dzl[i] = (short)cvt_bf16(dz4[i] - bf16_to_f32(hi));
        }
    }
    uint4 km[2];
    if (a.relu) {
        km[0] = gld<uint4>(krow + pair_col(0, kg));
        km[1] = gld<uint4>(krow + pair_col(1, kg));
        k0 = relu_eps8(k0, a.eps);
        k1 = relu_eps8(k1, a.eps);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        *reinterpret_cast<uint4*>(Gs + (ct * 16 + nl) * LD + kg * 8) = A[ct][0];
        *reinterpret_cast<uint4*>(Gs + (ct * 16 + nl) * LD + kg * 8 + 32) = A[ct][1];
    }
    wave_lds_fence();
    f32x4 ck[4], cv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        ck[ct] = cv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // dK^T[d1][s] = sum_d2 dKV[d1][d2] V[s][d2];  dV^T[d2][s] = sum_d1 dKV[d1][d2] K[s][d1]
        ck[ct] = mfma_bf16(as_bf16x8(A[ct][0]), as_bf16x8(v0), ck[ct]);
        ck[ct] = mfma_bf16(as_bf16x8(A[ct][1]), as_bf16x8(v1), ck[ct]);
        cv[ct] = mfma_bf16(tr_read8(Gs, LD, 0, ct * 16, lane), as_bf16x8(k0), cv[ct]);
        cv[ct] = mfma_bf16(tr_read8(Gs, LD, 32, ct * 16, lane), as_bf16x8(k1), cv[ct]);
        if (a.normalize) {   // + dksum_j[d1] = sum_s Qden[s][d1] dz[s] in every column s
            const s16x4 qt = tr_read4(Qs, ct * 16, lane);
            ck[ct] = mfma16(qt, dzh, ck[ct]);
            ck[ct] = mfma16(qt, dzl, ck[ct]);
        }
    }
    u16* dkrow = (u16*)a.dk.ptr + b * a.dk.sb + h * a.dk.sh + row * a.dk.sn;
    u16* dvrow = (u16*)a.dv.ptr + b * a.dv.sb + h * a.dv.sh + row * a.dv.sn;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint4 dk = pair_pieces(pack4(ck[2 * p]), pack4(ck[2 * p + 1]));
        if (a.relu) dk = mask_pos8(dk, km[p]);
        gst<uint4>(dkrow + pair_col(p, kg), dk);
        gst<uint4>(dvrow + pair_col(p, kg), pair_pieces(pack4(cv[2 * p]), pack4(cv[2 * p + 1])));
    }
}

// -------------------------------------------------------------------------------------------------
// k_sp_dwr: dwp[bh][split][i][j] = sum_{e in slice} x_i[e] y_j[e] (+ sum_s x2_i[s] y2_j[s] in split 0) for 64 < M <= 64 TT,
// 16-bit summaries.  One workgroup of TT x TT waves holds the whole M x M partial (wave (ti, tj) owns a 64 x 64 tile).
//
// Staging is LDS-DMA (global_load_lds_dwordx4): at 1024 threads the 64 accumulator registers leave no room for a register
// pipeline deeper than one stage (two register sets: 24 spilled VGPRs), and a register stage is requested only after the previous
// one has been written to LDS.  A stage is 64 elements (one 128-byte line) of every row of x and y = 64 KB at TT = 4, two LDS
// buffers: the DMA of stage st + 1 is issued right behind the barrier that releases its buffer and lands while stage st is
// multiplied -- no staging registers, no ds_write pass.  (Stages of 32 elements in four buffers -- three in flight -- were
// measured first: the two halves of a line are then requested a stage apart, by which time the line has left the L2 --
// FETCH_SIZE 1 024 MB for 541 MB of operands, 195 us.  Touching the lines of the stage after next -- one dword per half line into a
// kept register, to have it in the L2 when its copy is issued -- was slower as well: 164 -> 205 us.)
// The DMA writes lane-linear images (wave-uniform base + 16 lane): a wave instruction fills 8 rows x 8 pieces; the bank
// swizzle of the tile kernels (fast::gt_off) is applied on the SOURCE side -- LDS position q of row r holds the row's piece
// q ^ ((r ^ (r >> 1)) & 7) -- and again by the operand reads.
// The fp32 pair (dn, z) goes through the same product as bf16 hi + lo in one more stage per 16 values (ordinary loads, after
// the DMA pipeline has drained):   x: [hi(16) | lo(16) | hi(16) | 0]     y: [hi(16) | hi(16) | lo(16) | 0]   ->  hi hi + lo hi + hi lo
// -------------------------------------------------------------------------------------------------
struct DwrArgs {
    const u16* x;
    const u16* y;
    long E;
    long es;           // row stride of x / y in elements (E + padding)
    const float* x2;   // [bh][M][S2] or null
    const float* y2;
    int S2;
    float* out;        // [bh][nsplit][M][M]
    int M, nsplit;
};
constexpr int DWR_SE = 64;     // elements per stage
constexpr int DWR_NBUF = 2;    // LDS buffers
template <int TT> __host__ __device__ constexpr int dwr_smem() { return DWR_NBUF * 2 * 64 * TT * DWR_SE * 2; }

using sp::wait_vmcnt;

// H16: x and y are h16 payload rows (fp16, the row's fp32 multiplier right behind its E elements, split.hpp): the products run on the fp16
// MFMA (exact in the fp32 accumulators) and the partial is scaled by mult_x[i] mult_y[j] once, before the fp32 stages of (x2, y2) join.
template <int TT, bool H16 = false>
__global__ __launch_bounds__(64 * TT * TT) void k_sp_dwr(const DwrArgs a) {
    constexpr int NW = TT * TT, R = 64 * TT, UNITS = 2 * R / 8, UPW = (UNITS + NW - 1) / NW;   // unit: 8 rows of x or of y = one DMA instruction
    constexpr int IMG = R * DWR_SE;   // elements of one matrix image
    static_assert(DWR_SE == fast::GLD, "the images use the tile kernels' swizzle");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* lds = reinterpret_cast<u16*>(smem_raw);   // [NBUF][x | y][R][64]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int ti = wave / TT, tj = wave - ti * TT;
    const int split = blockIdx.x, bh = blockIdx.y, M = a.M;
    const long per = ((a.E / DWR_SE + a.nsplit - 1) / a.nsplit) * DWR_SE;
    const long ebeg = (long)split * per, eend = min(a.E, ebeg + per);
    const int nst = eend > ebeg ? (int)((eend - ebeg) / DWR_SE) : 0;
    // the fp32 stages of (x2, y2) are dealt over the E-slices' workgroups like the payload stages (all in slice 0 they were a chain of S2 / 16
    // load -> LDS -> barrier -> product round trips on ONE workgroup per (b, h): 16 of them at 256 tokens per block, 101 us of a 385 us step
    // at the DiT-S/2 4096^2 shape)
    const int tot2 = a.x2 ? (a.S2 + 15) / 16 : 0, per2 = (tot2 + a.nsplit - 1) / a.nsplit;
    const int st2b = min(tot2, split * per2), st2e = min(tot2, st2b + per2);

    // the wave's DMA units: unit u -> matrix u / (R / 8), rows 8 (u % (R / 8)) ..; lane -> row + lane / 8, LDS position lane % 8.
    // (units past the last wrap around: a duplicate copy of identical bytes keeps the instruction count per stage the same for every wave)
    const u16* src[UPW];
    int dst[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = (wave + i * NW) % UNITS, mat = u / (R / 8), r0 = (u - mat * (R / 8)) * 8, row = r0 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row ^ (row >> 1)) & 7);
        src[i] = (mat ? a.y : a.x) + ((long)bh * M + min(row, M - 1)) * a.es + ebeg + piece * 8;
        dst[i] = mat * IMG + r0 * DWR_SE;   // (wave-uniform)
    }
    auto issue = [&](int st) {
        u16* buf = lds + (st % DWR_NBUF) * (2 * IMG);
#pragma unroll
        for (int i = 0; i < UPW; ++i)
            __builtin_amdgcn_global_load_lds((const MHLA_GLOBAL_AS void*)(src[i] + (long)st * DWR_SE),
                                             (__attribute__((address_space(3))) void*)(buf + dst[i]), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // operand read of row 16 t + nl (+ 64 ti), pieces 4 ks + kg: the swizzle term depends on nl only
    const int sw = (nl ^ (nl >> 1)) & 7, rd0 = nl * DWR_SE + ((kg ^ sw) << 3), rd1 = nl * DWR_SE + (((4 + kg) ^ sw) << 3);
    auto compute = [&]<bool F16>(std::bool_constant<F16>, const u16* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u16* xs = buf + (ti * 64) * DWR_SE + (ks ? rd1 : rd0);
            const u16* ys = buf + IMG + (tj * 64) * DWR_SE + (ks ? rd1 : rd0);
            bf16x8 xa[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xa[t] = *reinterpret_cast<const bf16x8*>(xs + t * 16 * DWR_SE);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 ya = *reinterpret_cast<const bf16x8*>(ys + j * 16 * DWR_SE);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (F16) acc[i][j] = fast::mfma_f16(fast::as_f16x8(xa[i]), fast::as_f16x8(ya), acc[i][j]);
                    else acc[i][j] = mfma_bf16(xa[i], ya, acc[i][j]);
                }
            }
        }
    };

    // a stage is complete for this wave at vmcnt(0) (nothing younger is in flight); the barrier then (a) makes every wave's part of
    // it visible and (b) says that everyone has finished reading the other buffer, which the next issue refills
    if (nst > 0) issue(0);
    for (int st = 0; st < nst; ++st) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (st + 1 < nst) issue(st + 1);
        compute(std::bool_constant<H16>{}, lds + (st % DWR_NBUF) * (2 * IMG));
    }
    if constexpr (H16) {   // payload products -> values: rows i = 64 ti + 16 i + 4 kg + r of x, columns j = 64 tj + 16 j + nl of y
        float my[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) my[j] = gld<float>(a.y + ((long)bh * M + min(tj * 64 + j * 16 + nl, M - 1)) * a.es + a.E);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mx = gld<float>(a.x + ((long)bh * M + min(ti * 64 + i * 16 + kg * 4 + r, M - 1)) * a.es + a.E);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] *= mx * my[j];
            }
    }
    // fp32 stages: 16 values of (x2, y2) each; thread -> (row, 4 values)
    const bool vec2 = (a.S2 & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.x2) | reinterpret_cast<uintptr_t>(a.y2)) & 15) == 0;   // (uniform)
    for (int st = st2b; st < st2e; ++st) {
        __syncthreads();
        const int s0 = st * 16;
        for (int v = tid; v < R * 4; v += 64 * NW) {
            const int row = v >> 2, q4 = (v & 3) * 4, rr = min(row, M - 1);
            float xv[4], yv[4];
            if (vec2) {   // (S2 % 4 == 0: the four values are one aligned piece, inside the row or wholly past it)
                const long o = ((long)bh * M + rr) * a.S2 + min(s0 + q4, a.S2 - 4);
                const f32x4 xq = gld<f32x4>(a.x2 + o), yq = gld<f32x4>(a.y2 + o);
                const bool in = s0 + q4 < a.S2;
#pragma unroll
                for (int i = 0; i < 4; ++i) { xv[i] = in ? xq[i] : 0.f; yv[i] = in ? yq[i] : 0.f; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = s0 + q4 + i;
                    xv[i] = s < a.S2 ? gld<float>(a.x2 + ((long)bh * M + rr) * a.S2 + s) : 0.f;
                    yv[i] = s < a.S2 ? gld<float>(a.y2 + ((long)bh * M + rr) * a.S2 + s) : 0.f;
                }
            }
            unsigned xh[2], xl[2], yh[2], yl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xh[i] = pack_bf16x2(xv[2 * i], xv[2 * i + 1]);
                xl[i] = pack_bf16x2(xv[2 * i] - __uint_as_float(xh[i] << 16), xv[2 * i + 1] - __uint_as_float(xh[i] & 0xffff0000u));
                yh[i] = pack_bf16x2(yv[2 * i], yv[2 * i + 1]);
                yl[i] = pack_bf16x2(yv[2 * i] - __uint_as_float(yh[i] << 16), yv[2 * i + 1] - __uint_as_float(yh[i] & 0xffff0000u));
            }
            const uint2 XH = make_uint2(xh[0], xh[1]), XL = make_uint2(xl[0], xl[1]), YH = make_uint2(yh[0], yh[1]), YL = make_uint2(yl[0], yl[1]),
                        Z = make_uint2(0, 0);
            u16* xd = lds;
            u16* yd = lds + IMG;
            *reinterpret_cast<uint2*>(xd + fast::gt_off(row, q4)) = XH;      *reinterpret_cast<uint2*>(yd + fast::gt_off(row, q4)) = YH;
            *reinterpret_cast<uint2*>(xd + fast::gt_off(row, 16 + q4)) = XL; *reinterpret_cast<uint2*>(yd + fast::gt_off(row, 16 + q4)) = YH;
            *reinterpret_cast<uint2*>(xd + fast::gt_off(row, 32 + q4)) = XH; *reinterpret_cast<uint2*>(yd + fast::gt_off(row, 32 + q4)) = YL;
            *reinterpret_cast<uint2*>(xd + fast::gt_off(row, 48 + q4)) = Z;  *reinterpret_cast<uint2*>(yd + fast::gt_off(row, 48 + q4)) = Z;
        }
        __syncthreads();
        compute(std::false_type{}, lds);
    }
    float* out = a.out + ((long)bh * a.nsplit + split) * M * M;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 64 + i * 16 + kg * 4 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = tj * 64 + j * 16 + nl;
                if (row < M && col < M) out[(long)row * M + col] = acc[i][j][r];
            }
        }
}

}  // namespace s16
}  // namespace mhla
