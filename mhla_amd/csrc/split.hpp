// Block-mixing kernels for any head dim D <= 128 with D % 8 == 0 in any dtype (the Wan2.1 shape: fp32, D = 128, M = 150
// blocks of 210 tokens, wan/mhla_utils.py:331-341; DiT-XL/2: D = 72), computing on the bf16 MFMA with split operands.
// DT = ceil(D / 16) output tiles per side; LDS tiles are 64 or 128 columns wide with the columns >= D zero.
//
// An fp32 value x is carried as two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi), which keep 16 mantissa bits;
// a product a b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi with fp32 accumulation (the dropped lo*lo term is
// below 2^-17 |a b|).  Three v_mfma_f32_16x16x32_bf16 do the work of eight v_mfma_f32_16x16x4_f32 at a sixteenth of the
// issue cycles each, and a bf16 operand fetch from LDS moves 8 reduction steps per lane instead of 1 -- the generic
// fp32-MFMA kernels are bound by exactly those two.  bf16 tensors have lo = 0 and skip the extra products.
// Workspace formats are the generic path's (blockmix.hpp): KV, G fp32 [bh][M][D][D]; ksum [bh][M][D]; z, 1/n [bh][M][S].
//   k_sp_state : KV_j = K_j^T V_j, ksum_j, z_j                     grid (M, bh)
//   k_sp_mix   : G = W . KV  (or W^T .)                            grid (D^2 / 128, ceil(M / 64), bh)
//   k_sp_out   : O_i = (Q_i G_i) / n_i, computed transposed so that a lane owns 4 consecutive output features
#pragma once
#include <type_traits>

#include "blockmix.hpp"
#include "fused.hpp"

namespace mhla {
namespace sp {

using fast::bf16x8;
using fast::mfma_bf16;
using fast::tr_read8;
using fast::s16x4;
using fast::s16x8;
using fast::u16;

__device__ __forceinline__ bf16x8 row_read8(const u16* tile, int ld, int c0, int k0, int lane);   // (defined with the token-gradient kernels)

template <typename T>
__device__ __forceinline__ void ld8(const T* p, f32x4& a, f32x4& b) {
    a = Io<T>::ld4(p);
    b = Io<T>::ld4(p + 4);
}
// 4 elements of TO as loaded (packed for 16-bit types) and their conversion to fp32
template <typename TO> struct Raw4 { typedef uint2 type; };
template <> struct Raw4<float> { typedef f32x4 type; };
__device__ __forceinline__ f32x4 raw4_to_f32(bf16_t, uint2 r) {
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 raw4_to_f32(f16_t, uint2 r) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 h = __builtin_bit_cast(h4, r);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ f32x4 raw4_to_f32(float, f32x4 r) { return r; }
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
// rotate the 4 consecutive channel pairs (x[2i], x[2i+1]) held in (a, b) by the angles (c[i], s[i])
// (rope_apply, wan/mhla_utils.py:127-156: complex multiply of consecutive pairs)
__device__ __forceinline__ void rope8(f32x4& a, f32x4& b, const f32x4& c, const f32x4& s) {
    const f32x4 a0 = a, b0 = b;
    a[0] = a0[0] * c[0] - a0[1] * s[0]; a[1] = a0[0] * s[0] + a0[1] * c[0];
    a[2] = a0[2] * c[1] - a0[3] * s[1]; a[3] = a0[2] * s[1] + a0[3] * c[1];
    b[0] = b0[0] * c[2] - b0[1] * s[2]; b[1] = b0[0] * s[2] + b0[1] * c[2];
    b[2] = b0[2] * c[3] - b0[3] * s[3]; b[3] = b0[2] * s[3] + b0[3] * c[3];
}

// transposed rotation of a gradient in output layout: the lane's 4 consecutive features are 2 pairs with angles (c[0], s[0]), (c[1], s[1])
__device__ __forceinline__ void unrope4(f32x4& g, const f32x2& c, const f32x2& s) {
    const f32x4 g0 = g;
    g[0] = g0[0] * c[0] + g0[1] * s[0]; g[1] = g0[1] * c[0] - g0[0] * s[0];
    g[2] = g0[2] * c[1] + g0[3] * s[1]; g[3] = g0[3] * c[1] - g0[2] * s[1];
}

// -------------------------------------------------------------------------------------------------
// p24: block summaries as 24-bit floats -- sign, 8 exponent bits, 15 stored mantissa bits: the top three bytes of the fp32, i.e. 16
// significand bits, which is what the hi + lo bf16 operand pair of the products carries anyway -- in 3 / 4 of the bytes.  A summary row
// of E elements is two planes: [E x u16: the top two bytes = the bf16 hi operand as it is][E x u8: the third byte].  Whole-block readers
// and writers see two contiguous runs; a 64-element slice of the resident mixing is one 128-byte line of the hi plane and half a line
// of the lo plane (its neighbour slice, taken next by the same workgroup, owns the other half).  (192-byte
// groups of 64 elements -- hi and lo side by side -- put every second hi piece across two lines, and the streaming loads dropped the
// shared line between slices: the mixing kernels ran 10-25 % slower than on fp32 summaries.)
// The lo operand, value - hi, has at most 8 significant bits: exact in bf16.  Used by the resident-mixing pipeline on 16-bit tensors
// (capi_bm_typed.hpp bm_p24); rows keep their stride in float units (BmWs::es), 3 E / 4 + padding.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned p24_round(float x) { return (__float_as_uint(x) + 0x80u) & 0xffffff00u; }
// 8 values -> their hi piece (16 bytes) and lo piece (8 bytes)
__device__ __forceinline__ void p24_pack8(const f32x4& a, const f32x4& b, uint4& hi, uint2& lo) {
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = p24_round(a[i]); r[4 + i] = p24_round(b[i]); }
    hi = make_uint4(__builtin_amdgcn_perm(r[1], r[0], 0x07060302u), __builtin_amdgcn_perm(r[3], r[2], 0x07060302u),
                    __builtin_amdgcn_perm(r[5], r[4], 0x07060302u), __builtin_amdgcn_perm(r[7], r[6], 0x07060302u));
    // byte 1 of each value: (r1.b1, r0.b1) -> low half, (r3.b1, r2.b1) -> high half
    lo = make_uint2(__builtin_amdgcn_perm(__builtin_amdgcn_perm(r[3], r[2], 0x0c0c0501u), __builtin_amdgcn_perm(r[1], r[0], 0x0c0c0501u), 0x05040100u),
                    __builtin_amdgcn_perm(__builtin_amdgcn_perm(r[7], r[6], 0x0c0c0501u), __builtin_amdgcn_perm(r[5], r[4], 0x0c0c0501u), 0x05040100u));
}
// the bf16 lo operands of 8 stored values: (hi | third byte) - hi, exact
__device__ __forceinline__ uint4 p24_lo8(const uint4& hi, const uint2& lo) {
    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w};
    const unsigned lw[2] = {lo.x, lo.y};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // elements 2 j (low half of the word) and 2 j + 1
        const unsigned l = lw[j >> 1];
        // v_perm_b32 (S0, S1): selector bytes 0-3 pick from S1, 4-7 from S0, 0x0c is a zero byte
        const unsigned h0 = hw[j] << 16, h1 = hw[j] & 0xffff0000u;
        const unsigned f0 = __builtin_amdgcn_perm(hw[j], l, (j & 1) ? 0x05040200u | 0x0cu : 0x05040000u | 0x0cu);
        const unsigned f1 = __builtin_amdgcn_perm(hw[j], l, (j & 1) ? 0x07060300u | 0x0cu : 0x07060100u | 0x0cu);
        const float d0 = __uint_as_float(f0) - __uint_as_float(h0), d1 = __uint_as_float(f1) - __uint_as_float(h1);
        o[j] = __builtin_amdgcn_perm(__float_as_uint(d1), __float_as_uint(d0), 0x07060302u);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// the 8 stored values as floats
__device__ __forceinline__ void p24_unpack8(const uint4& hi, const uint2& lo, f32x4& a, f32x4& b) {
    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w};
    const unsigned lw[2] = {lo.x, lo.y};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned l = lw[j >> 1];
        f[2 * j] = __uint_as_float(__builtin_amdgcn_perm(hw[j], l, (j & 1) ? 0x05040200u | 0x0cu : 0x05040000u | 0x0cu));
        f[2 * j + 1] = __uint_as_float(__builtin_amdgcn_perm(hw[j], l, (j & 1) ? 0x07060300u | 0x0cu : 0x07060100u | 0x0cu));
    }
    a = f32x4{f[0], f[1], f[2], f[3]};
    b = f32x4{f[4], f[5], f[6], f[7]};
}

// -------------------------------------------------------------------------------------------------
// h16 (summary format 2, round 6): block summaries as an fp16 payload x ONE power-of-two multiplier per block row -- 11 significand
// bits, the precision the reference's own products run at (TF32: mhla_dit/train.py:12-13 turns allow_tf32 on for the matmul / 1x1-conv
// of mhla_dit/mhla/mhla.py:262-263) -- in 2 bytes per element where p24 takes 3.  A summary row of E elements is [E x fp16][fp32
// multiplier m at byte 2 E]: value = payload * m.  The kernels that own a whole row (k_sp_state) take m from the row's measured
// maximum (payload maximum in [2^14, 2^15)); the mixing kernels, whose output rows are spread over workgroups, take it from the bound
// |out_o[e]| <= sum_r |W(o, r)| 2^15 m_r, which every workgroup computes alike from the input rows' multipliers -- a loose bound costs
// nothing: fp16 keeps 11 bits over 29 binades.  Consumers decode to the same bf16 hi + lo operand pair as every other fp32-grade format
// (hi = top 16 bits of payload * m, lo = the remaining 3 significand bits: exact).  16-bit tensors with blocks of >= 16 tokens
// (capi_common.hpp bm_sumfmt); tools/sim_h16.py is the CPU model of its error (<= 4e-4 of a result's maximum on every BASELINE shape).
// -------------------------------------------------------------------------------------------------
// (the format's helpers -- h16_mult_from_max / _from_bound, h16_inv, h16_pack2 / _pack8, h16_split8 -- are in common.hpp: the causal pipeline uses them too)

// Block summaries (KV, G, dG, dKV) are fp32 in the workspace, except for bf16 tensors: there they are stored as bf16 (as on
// the bf16 fast path), which halves the summary traffic -- the larger share of the bytes when S is small -- and needs no lo part.
template <typename T> struct Sum16 { static constexpr bool value = std::is_same<T, bf16_t>::value; };

template <int DT> struct Geo {
    static constexpr int CGS = DT > 4 ? 16 : 8;        // column groups of 8 per tile row
    static constexpr int DW = CGS * 8;                 // tile width (columns), >= 16 DT
    static constexpr int LD = DW + 16;                 // LDS row stride (bf16) of k_sp_state's token tiles (row r at mat_row(r), below) ...
    static constexpr int LDR = DW + 8;                 // ... except in its row-dots-from-G variant, whose fourth workgroup per CU needs the 2 KB
    static constexpr int RPP = NTHREADS / CGS;         // tile rows covered per pass of the 256 threads
    static constexpr int RT = (DT + 3) / 4;            // 16-row output tiles per wave
    static constexpr int KST = (DT + 1) / 2;           // reduction steps of 32 over a head dim
};

// (used by k_sp_state's row-dots-from-G variant as well as by the output and token-gradient kernels further down)
// LDS row stride (bf16) of a staged D x D summary matrix [KST * 32 rows][KST * 32 columns + 8]: the reads cover KST * 32
// columns, not the DW of the token tiles (D = 72: 104 instead of 136 -> a third workgroup per CU for the fp32 kernels)
// Bank layout (round 6).  gfx950's LDS has 64 banks and serves the 16-byte row reads in groups of 16 lanes, the transpose reads in groups
// of 32 (MI355X_MICROARCH.md, LDS): with the round-1 padding of 8 elements (derived for 32 banks) every operand read of these kernels is
// 2-way conflicted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.28 - 0.44 at C2; `tools/lds_conflicts.py split` reproduces it).  In the
// new layout rows are padded by 16 elements (consecutive rows 8 banks apart) and matrix row r is kept at LDS row mat_row(r) = r with bits
// 2 and 3 swapped: the 32 lanes of a transpose read (rows 8 g + j, g = 0, 1) then touch 8 consecutive LDS rows = all 64 banks once, and a
// 16-byte row read's lane groups keep their sets of rows.  Counters after: 0.00 (k_sp_out, k_sp_bwd_dq), 0.16 (k_sp_bwd_dkv: its strip
// stores).  What it buys is small, because these kernels wait for HBM, not for the LDS: C2 step 0.382 -> 0.373 ms (k_sp_bwd_dkv 81.8 ->
// 76.6 us; at 256 blocks of 16 tokens 214 -> 191 us), nothing at D = 72 .. 96, and at D = 128 the Wan inference output kernel LOSES 11 %
// (5.69 -> 6.3 ms per forward, twice) -- so the staged matrices take it for D <= 64 only (MAT_NEW_MAX_DT; `tools/ab_configs.sh` is the
// A/B), k_sp_state's token tiles always except in the row-dots-from-G variant (whose fourth workgroup per CU needs the 2 KB).
// Everything that touches a staged matrix goes through mat_row / mat_tr_read8 / mat_row_read8.
#ifndef MAT_NEW_MAX_DT
#define MAT_NEW_MAX_DT 4
#endif
template <int DT> __host__ __device__ constexpr bool mat_new() { return DT <= MAT_NEW_MAX_DT; }
template <int DT>
__host__ __device__ constexpr int mat_ld() { return Geo<DT>::KST * 32 + (mat_new<DT>() ? 16 : 8); }
template <bool NEW = true>
__host__ __device__ constexpr int mat_row(int r) { return NEW ? ((r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)) : r; }
// tr_read8 of a staged matrix: lane (c = lane & 15, g = lane >> 4) receives T[k0 + 8 g + 0..7][c0 + c]   (k0 a multiple of 32)
template <bool NEW = true>
__device__ __forceinline__ bf16x8 mat_tr_read8(const u16* tile, int ld, int k0, int c0, int lane) {
    if constexpr (!NEW) return tr_read8(tile, ld, k0, c0, lane);
    const int g = lane >> 4, li = lane & 15;
    const u16* p = tile + (k0 + (g >> 1) * 16 + (g & 1) * 4 + (li >> 2)) * ld + c0 + (li & 3) * 4;   // = mat_row(k0 + 8 g + (li >> 2))
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p + 8 * ld));                  // = mat_row(.. + 4)
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
// row_read8 of a staged matrix: A[m][k] = T[c0 + m][k0 + 8 kg .. + 7]   (c0 a multiple of 16)
template <bool NEW = true>
__device__ __forceinline__ bf16x8 mat_row_read8(const u16* tile, int ld, int c0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (c0 + mat_row<NEW>(lane & 15)) * ld + k0 + (lane >> 4) * 8);
}
template <int DT, bool S16 = false>   // bf16 summaries: no lo tile
__host__ __device__ constexpr int sp_out_smem() { return (S16 ? 1 : 2) * Geo<DT>::KST * 32 * mat_ld<DT>() * 2; }

// EPI: the per-head RMSNorm (x SiLU gate) that follows the operator in the Wan host (wan/mhla_utils.py:356-362) is applied
// to the token's D outputs before they are stored, in the dtype TO of the host's activations: O is rounded to TO (the
// `.to(dtype)` at :356), normalised over the head dim in fp32, scaled by the norm weight and the gate, stored once.
// D x D summary matrix (fp32, or bf16 when S16) -> LDS [KP][LD] hi (/ lo) tiles; rows and columns >= D zero.  NT threads.
template <int DT, bool S16, int NT = NTHREADS, int P24 = 0>   // P24: 0 fp32 words (or bf16 when S16), 1 24-bit floats, 2 h16
__device__ __forceinline__ void stage_mat_split(u16* __restrict__ Gh, u16* __restrict__ Gl, const float* __restrict__ base, long elem_off, int D, int tid) {
    constexpr int LD = mat_ld<DT>(), CGS = Geo<DT>::CGS, RPP = NT / CGS, KP = Geo<DT>::KST * 32;
    const int r0 = tid / CGS, cg = (tid % CGS) * 8;
    constexpr int PASSES = (KP + RPP - 1) / RPP, UB = PASSES < 4 ? PASSES : (PASSES % 4 == 0 ? 4 : (PASSES % 3 == 0 ? 3 : 2));
    static_assert(PASSES % UB == 0, "staging batches must tile the passes");
    const float* g = base + elem_off;
    const u16* g16 = reinterpret_cast<const u16*>(base) + elem_off;
    float hm = 0.f;   // h16: the row's multiplier
    if constexpr (P24 == 2) hm = gld<float>(reinterpret_cast<const char*>(base + elem_off) + 2 * D * D);
    for (int pb = 0; pb < PASSES; pb += UB) {
        f32x4 x[UB][2];
        uint4 x16[UB];
        uint2 xl[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            // (every load unconditional, from a clamped position: rows / columns past D are zeroed below.  Behind `if (r < D && cg < D)` the
            // pieces of a batch were round trips of their own -- hipcc waits for everything in flight where a branch with a load in it joins)
            const int r = min(r0 + RPP * (pb + u), D - 1), c = min(cg, D - 8);
            x[u][0] = x[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            x16[u] = make_uint4(0, 0, 0, 0);
            xl[u] = make_uint2(0, 0);
            if (P24) {   // 8 elements: a piece of the hi plane and one of the lo plane (elem_off counts floats: the row's start)
                const int e = r * D + c;
                const char* rowp = reinterpret_cast<const char*>(base + elem_off);
                x16[u] = gld<uint4>(rowp + 2 * e);
                if constexpr (P24 == 1) xl[u] = gld<uint2>(rowp + 2 * D * D + e);
            } else if (S16) {
                x16[u] = gld<uint4>(g16 + (long)r * D + c);
            } else {
                const float* src = g + (long)r * D + c;
                x[u][0] = gld<f32x4>(src);
                x[u][1] = gld<f32x4>(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = r0 + RPP * (pb + u), off = mat_row<mat_new<DT>()>(r) * LD + cg;
            if (!(r < D && cg < D)) {   // (no memory operation in here)
                x[u][0] = x[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                x16[u] = make_uint4(0, 0, 0, 0);
                xl[u] = make_uint2(0, 0);
            }
            if (r < KP && cg < KP) {
                if (P24 == 2) {
                    uint4 hi, lo;
                    h16_split8(x16[u], hm, hi, lo);
                    *reinterpret_cast<uint4*>(Gh + off) = hi;
                    *reinterpret_cast<uint4*>(Gl + off) = lo;
                } else if (P24) {
                    *reinterpret_cast<uint4*>(Gh + off) = x16[u];
                    *reinterpret_cast<uint4*>(Gl + off) = p24_lo8(x16[u], xl[u]);
                } else if (S16) {
                    *reinterpret_cast<uint4*>(Gh + off) = x16[u];
                } else {
                    uint4 hi, lo;
                    split8(x[u][0], x[u][1], hi, lo);
                    *reinterpret_cast<uint4*>(Gh + off) = hi;
                    *reinterpret_cast<uint4*>(Gl + off) = lo;
                }
            }
        }
    }
}

// stage_mat_split in two halves for 24-bit summaries staged in ONE batch (D = 128 by 512 threads: four passes): the loads are requested,
// the workgroup multiplies a round of token tiles, then the pieces are committed (k_sp_out<.., FLAT>)
template <int DT, int NT>
struct MatStage {
    static constexpr int PASSES = (Geo<DT>::KST * 32 + NT / Geo<DT>::CGS - 1) / (NT / Geo<DT>::CGS);
    uint4 x16[PASSES];
    uint2 xl[PASSES];
};
// (real = false: the loads are issued all the same -- no branch around them -- but every lane reads the matrix's first piece)
template <int DT, int NT>
__device__ __forceinline__ void stage_mat_issue(MatStage<DT, NT>& g, const float* __restrict__ base, long elem_off, int D, int tid, bool real = true) {
    constexpr int CGS = Geo<DT>::CGS, RPP = NT / CGS;
    const int r0 = tid / CGS, cg = (tid % CGS) * 8;
    const char* rowp = reinterpret_cast<const char*>(base + elem_off);
#pragma unroll
    for (int u = 0; u < MatStage<DT, NT>::PASSES; ++u) {   // (unconditional, from clamped addresses: rows / columns past D are zeroed at the commit)
        const int r = min(r0 + RPP * u, D - 1), c = min(cg, D - 8), e = real ? r * D + c : 0;
        g.x16[u] = gld<uint4>(rowp + 2 * e);
        g.xl[u] = gld<uint2>(rowp + 2 * D * D + e);
    }
}
template <int DT, int NT>
__device__ __forceinline__ void stage_mat_commit(u16* __restrict__ Gh, u16* __restrict__ Gl, const MatStage<DT, NT>& g, int D, int tid) {
    constexpr int LD = mat_ld<DT>(), CGS = Geo<DT>::CGS, RPP = NT / CGS, KP = Geo<DT>::KST * 32;
    const int r0 = tid / CGS, cg = (tid % CGS) * 8;
#pragma unroll
    for (int u = 0; u < MatStage<DT, NT>::PASSES; ++u) {
        const int r = r0 + RPP * u, off = mat_row<mat_new<DT>()>(r) * LD + cg;
        const bool in = r < D && cg < D;
        const uint4 hi = in ? g.x16[u] : make_uint4(0, 0, 0, 0);
        const uint2 lo = in ? g.xl[u] : make_uint2(0, 0);
        if (r < KP && cg < KP) {
            *reinterpret_cast<uint4*>(Gh + off) = hi;
            *reinterpret_cast<uint4*>(Gl + off) = p24_lo8(hi, lo);
        }
    }
}

template <int DT>
__host__ __device__ constexpr int sp_state_smem() {
    // the column-sum partials [RPP][DW] of the epilogue reuse the tiles
    static_assert(Geo<DT>::RPP * Geo<DT>::DW * 4 <= 4 * 32 * Geo<DT>::LD * 2, "column-sum partials must fit in the tiles");
    return 4 * 32 * Geo<DT>::LD * 2 + Geo<DT>::DW * 4;
}

// MODE 1 on 16-bit tensors with 24-bit summaries and D <= 64 (RD): the row dots dO . O come from G_i -- two more tiles (G_i as hi / lo,
// [64][72] bf16 each) and the partial dots of a 32-token chunk
template <int DT> __host__ __device__ constexpr int sp_state_rd_smem() {   // (its token tiles keep the rows of DW + 8: four workgroups per CU)
    return 4 * 32 * Geo<DT>::LDR * 2 + Geo<DT>::DW * 4 + 2 * Geo<DT>::KST * 32 * mat_ld<DT>() * 2 + 2 * 32 * 4;
}

// MODE 0 (forward):  out = KV_j = K_j^T V_j; ksum_j; z_j                      x = k_num, y = v, kd = k_den, qd = q_den
// MODE 1 (backward): out = dG_i = Q_i^T (dO_i / n_i); dn_i[s] = -(dO_i[s] . O_i[s]) / n_i[s]     x = q_num, y = dout, o = out
// ROPE: x (the keys of the KV product in MODE 0, the queries of dG = Q^T dP in MODE 1) is rotated on load (a.rcos / a.rsin); a
// template flag so that the plain variants do not carry the angle registers (3 waves per SIMD need <= 168 VGPRs)
// NT: threads.  512 (D = 128 only): a 32-row chunk is one staging pass and every wave owns ONE 16-row tile of the summary -- half
// the staging and accumulator registers per thread (<= 128 VGPRs: two workgroups = 16 waves per CU) and half the time per block,
// which matters at the Wan shape for a second reason: 1 800 blocks on 768 four-wave slots are 2.3 rounds that cost 3, on 512
// eight-wave slots of half the duration 3.5 rounds that cost 4 (of half the length).
// S16: the summary is stored as bf16 (the opt-in MHLA_FLAG_BF16_SUMMARIES arithmetic on bf16 tensors); otherwise fp32, and the
// one operand that is an INTERMEDIATE (dP = dO / n, MODE 1) is split into hi + lo parts whatever the tensor type
// PRO (MODE 0, Wan inference): x and qd are 16-bit projection outputs; relu(x * rstd[token] * w[channel]) + eps is applied on load
// (StateArgs::pro_*), the values then carry a lo part like fp32 tensors
template <typename T, int DT, int MODE, bool ROPE = false, int NT = NTHREADS, bool S16 = Sum16<T>::value, int P24 = 0, bool PRO = false>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : (ROPE ? 2 : 3)) void k_sp_state(const StateArgs a) {
    static_assert(!PRO || (MODE == 0 && !S16), "the prologue on load serves the forward's summary kernel");
    static_assert(!P24 || !S16, "p24 is a format of the fp32-grade summaries");
    // RD: the row dots dO . O = dO' . (Q G_i) are formed from the mixed summary G_i (staged as hi / lo tiles beside the operand tiles): the
    // stored output and its residual are not read, and the forward does not write the residual (capi_common.hpp bm_rowdots_from_g)
    constexpr bool RD = MODE == 1 && P24 && sizeof(T) == 2 && DT <= 4 && !ROPE;
    constexpr bool TN = !RD;   // the token tiles in the conflict-free layout (Geo::LD, mat_row)
    constexpr int DW = Geo<DT>::DW, LD = TN ? Geo<DT>::LD : Geo<DT>::LDR, CGS = Geo<DT>::CGS, RPP = NT / CGS, IT = 32 / RPP, NWV = NT / 64,
                  RT = (DT + NWV - 1) / NWV, TILE = 32 * LD;
    static_assert(IT >= 1 && RPP * IT == 32, "a 32-row chunk must be whole staging passes");
    static_assert(RPP * DW * 4 <= 4 * 32 * LD * 2, "column-sum partials must fit in the tiles");
    constexpr bool LO = !std::is_same<T, bf16_t>::value || PRO;   // the token operands carry a lo part
    constexpr bool LOY = !std::is_same<T, bf16_t>::value || (MODE == 1 && !S16);   // ... and so does y = dO / n, unless the reduced-precision form was asked for
    constexpr int GLD = mat_ld<DT>(), GT = Geo<DT>::KST * 32 * GLD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Kh = reinterpret_cast<u16*>(smem_raw);
    u16* Kl = Kh + TILE;
    u16* Vh = Kl + TILE;
    u16* Vl = Vh + TILE;
    float* cs = reinterpret_cast<float*>(smem_raw);    // [RPP][DW] column-sum partials: over the tiles, after the last product
    float* vecd = reinterpret_cast<float*>(Vl + TILE); // [DW] ksum
    u16* Gh = reinterpret_cast<u16*>(vecd + DW);       // RD: G_i [d1][d2] hi, lo ([KST * 32][GLD] each), then the chunk's partial row dots [2][32]
    u16* Gl = Gh + GT;
    float* rdp = reinterpret_cast<float*>(Gl + GT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, S = a.S, D = a.D;
    const long p0 = (long)blk * S;
    const T* kb = (const T*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const T* vb = (const T*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const View& third = MODE == 0 ? a.kd : a.o;   // MODE 0: the normaliser's keys; MODE 1: the forward output
    const T* kdb = (const T*)third.ptr + b * third.sb + h * third.sh;
    const bool den = a.normalize && ((MODE == 1 && !RD) || (MODE == 0 && a.split));   // a third tensor is read
    const int r0 = tid / CGS, cg = (tid % CGS) * 8;
    const float* ninvb = a.ninv + ((long)bh * a.M + blk) * S;   // MODE 1

    // the chunk's rows AS LOADED (kr, vr, dr): converted in `settle`, at the commit -- `ld8` converted them as they arrived, i.e. the wait for
    // the next chunk sat right behind its request, in front of this chunk's products (tools/isa_waits.py: `L L L L W3 W2` at the loop top)
    typename Raw4<T>::type kr[IT][2], vr[IT][2], dr[IT][2];
    f32x4 kx[IT][2], vx[IT][2], dx[IT][2], rc[ROPE ? IT : 1], rs[ROPE ? IT : 1];
    float nv[IT];
    float prr[PRO ? IT : 1];   // PRO: the rows' rstd
    f32x4 pw[PRO ? 2 : 1];     // ... and the norm weights of the thread's 8 channels
    constexpr bool rope = ROPE;
    int crow = 0;   // first token of the chunk held in registers
    // Gather map: the rows of a chunk are looked up ONE FETCH EARLIER than they are used (clamped, unconditional), so that a fetch
    // is one memory round trip, not two (map entry, then the row it names) -- with the map the chunk-ahead prefetch was a chain
    // of two latencies against one 32-row chunk of products.
    int nraw[IT], npos[IT];   // the map entry as loaded, and the position it was loaded for (used when there is no map)
    auto lookup = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long p = p0 + min(c0 + r0 + RPP * it, S - 1);
            // (no branch around the load -- hipcc waits for everything in flight where a branch with a load in it joins: without
            // a map the load reads the first word of x and its result is dropped -- and no use of the loaded value before the next
            // fetch: a select right here would wait for it, and for the data loads issued before it)
            nraw[it] = gld<int>(a.idx ? (const void*)(a.idx + p) : a.x.ptr);
            npos[it] = (int)p;
        }
    };
    lookup(0);
    // Every load of a fetch is unconditional (rows past the block's end and column groups past D are clamped onto valid ones and
    // zeroed in `commit`; a tensor the call does not have is replaced by x and its values dropped): a load behind a branch makes
    // hipcc wait for everything in flight where the branch joins, i.e. before the products the prefetch is meant to overlap.
    const int cgc = min(cg, D - 8);
    if constexpr (PRO) {
        pw[0] = pw[1] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (a.pro_wk) {
            pw[0] = *reinterpret_cast<const f32x4*>(a.pro_wk + h * D + cgc);
            pw[1] = *reinterpret_cast<const f32x4*>(a.pro_wk + h * D + cgc + 4);
        }
    }
    const T* kdq = den ? kdb : kb;                  // (uniform selects)
    const long kdsn = den ? third.sn : a.x.sn;
    const float* nvp = (MODE == 1 && a.normalize) ? ninvb : reinterpret_cast<const float*>(a.x.ptr);
    // MODE 1, 16-bit tensors at the default arithmetic: the residual of the forward's store of O (StateArgs::olo), fetched with
    // the rows (unconditionally: without it, the first bytes of x, dropped in `settle`)
    constexpr bool OLO = MODE == 1 && !S16 && sizeof(T) == 2 && !RD;
    const bool olo_on = OLO && a.normalize && a.olo != nullptr;
    const u16* olop = olo_on ? a.olo + ((long)bh * a.M * S + p0) * D + cgc : reinterpret_cast<const u16*>(a.x.ptr);
    uint4 ox[OLO ? IT : 1];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
        crow = c0;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int rcl = min(c0 + r0 + RPP * it, S - 1);
            const long row = a.idx ? nraw[it] : npos[it];
            kr[it][0] = gld<typename Raw4<T>::type>(kb + row * a.x.sn + cgc);
            kr[it][1] = gld<typename Raw4<T>::type>(kb + row * a.x.sn + cgc + 4);
            vr[it][0] = gld<typename Raw4<T>::type>(vb + row * a.y.sn + cgc);
            vr[it][1] = gld<typename Raw4<T>::type>(vb + row * a.y.sn + cgc + 4);
            dr[it][0] = gld<typename Raw4<T>::type>(kdq + row * kdsn + cgc);
            dr[it][1] = gld<typename Raw4<T>::type>(kdq + row * kdsn + cgc + 4);
            nv[it] = gld<float>(nvp + ((MODE == 1 && a.normalize) ? rcl : 0));
            if constexpr (PRO) prr[it] = gld<float>(a.pro_rk ? a.pro_rk + b * a.pro_n + row : reinterpret_cast<const float*>(a.x.ptr));
            if constexpr (OLO) ox[it] = gld<uint4>(olop + (olo_on ? (long)rcl * D : 0));
            if constexpr (rope) {
                rc[it] = *reinterpret_cast<const f32x4*>(a.rcos + row * a.ldr + cgc / 2);
                rs[it] = *reinterpret_cast<const f32x4*>(a.rsin + row * a.ldr + cgc / 2);
            }
        }
        lookup(c0 + 32);   // (the rows of the next fetch)
    };
    // what `fetch` used to do behind its branch: zeros for padded rows / columns, relu + eps on the keys, 1 for an absent 1 / n
    auto settle = [&]() __attribute__((always_inline)) {   // (two call sites since the loop's last chunk was peeled: left to the inliner's size
                                                            //  heuristics, the register arrays these lambdas share went to scratch)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool valid = crow + r0 + RPP * it < S && cg < D;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                kx[it][hf] = raw4_to_f32(T{}, kr[it][hf]);
                vx[it][hf] = raw4_to_f32(T{}, vr[it][hf]);
                dx[it][hf] = raw4_to_f32(T{}, dr[it][hf]);
            }
            if constexpr (PRO) {
                const float r = a.pro_rk ? prr[it] : 1.f;   // ((x rstd) w: the order of k_qk_prologue, so that the two paths agree bit for bit)
                kx[it][0] = (kx[it][0] * r) * pw[0];
                kx[it][1] = (kx[it][1] * r) * pw[1];
            }
            if (a.relu) relu8(kx[it][0], kx[it][1], a.eps);
            kx[it][0] = valid ? kx[it][0] : z4; kx[it][1] = valid ? kx[it][1] : z4;
            vx[it][0] = valid ? vx[it][0] : z4; vx[it][1] = valid ? vx[it][1] : z4;
            dx[it][0] = (valid && den) ? dx[it][0] : z4; dx[it][1] = (valid && den) ? dx[it][1] : z4;
            if constexpr (OLO) {   // O = stored value + what the store rounded away
                const unsigned w[4] = {ox[it].x, ox[it].y, ox[it].z, ox[it].w};
                const bool on = valid && olo_on;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    dx[it][0][2 * i] += on ? __uint_as_float(w[i] << 16) : 0.f;
                    dx[it][0][2 * i + 1] += on ? __uint_as_float(w[i] & 0xffff0000u) : 0.f;
                    dx[it][1][2 * i] += on ? __uint_as_float(w[2 + i] << 16) : 0.f;
                    dx[it][1][2 * i + 1] += on ? __uint_as_float(w[2 + i] & 0xffff0000u) : 0.f;
                }
            }
            nv[it] = (valid && MODE == 1 && a.normalize) ? nv[it] : 1.f;
            if constexpr (rope) {   // padded rows: 0 * (uninitialised angle) could be NaN
                rc[it] = valid ? rc[it] : z4;
                rs[it] = valid ? rs[it] : z4;
            }
        }
    };
    float ksp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float nvc[IT];   // RD: 1 / n and the first row of the chunk whose row dots are in flight
    int crowc = 0;
    auto commit = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);   // (the fetched values are not touched before this point: hipcc would hoist `settle` above the products and wait there)
        settle();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int off = mat_row<TN>(r0 + RPP * it) * LD + cg;
            uint4 hi, lo;
            if (MODE == 1 && a.normalize) {   // dn[s] and the 1/n scaling of dO
                if constexpr (!RD) {
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d += vx[it][0][i] * dx[it][0][i] + vx[it][1][i] * dx[it][1][i];
#pragma unroll
                    for (int o = CGS / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
                    const int r = crow + r0 + RPP * it;
                    if (r < S && (tid % CGS) == 0) a.dn[((long)bh * a.M + blk) * S + r] = -d * nv[it];
                } else {
                    nvc[it] = nv[it];   // (the next fetch overwrites nv and crow before this chunk's dots are complete)
                    crowc = crow;
                }
                vx[it][0] *= nv[it];
                vx[it][1] *= nv[it];
            }
            if constexpr (rope) {   // KV takes the rotated keys, ksum the plain ones
                f32x4 r0 = kx[it][0], r1 = kx[it][1];
                rope8(r0, r1, rc[it], rs[it]);
                split8(r0, r1, hi, lo);
            } else {
                split8(kx[it][0], kx[it][1], hi, lo);
            }
            *reinterpret_cast<uint4*>(Kh + off) = hi;
            if (LO) *reinterpret_cast<uint4*>(Kl + off) = lo;
            split8(vx[it][0], vx[it][1], hi, lo);
            *reinterpret_cast<uint4*>(Vh + off) = hi;
            if (LOY) *reinterpret_cast<uint4*>(Vl + off) = lo;
            if (MODE == 0) {
                const f32x4* s = den ? dx[it] : kx[it];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ksp[i] += s[0][i];
                    ksp[4 + i] += s[1][i];
                }
            }
        }
    };

    f32x4 acc[RT][DT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    if constexpr (RD) {   // (its loads travel with the first chunk's; the loop's first barrier covers the tiles)
        if (a.normalize) stage_mat_split<DT, false, NT, P24>(Gh, Gl, a.g, ((long)bh * a.M + blk) * a.es, D, tid);
    }
    // one 32-token chunk: commit, request the next one (PF: there is one -- a compile-time fact of the call site, so that no branch sits
    // around the request: where such a branch joins hipcc waits for the loads in it), products
    auto chunk = [&]<bool PF>(std::bool_constant<PF>, int c0) __attribute__((always_inline)) {
        commit();
        __syncthreads();
        if constexpr (PF) fetch(c0 + 32);
        if constexpr (RD) {
            // dots[s] = sum_d1 q[s][d1] T[d1][s],  T = G_i dO'^T (transposed product: a lane gets T[16 ct + 4 kg + r][s = nl]); wave w takes
            // token tile w & 1 of the chunk and the feature tiles of half w >> 1, the two halves meet in LDS after the loop's barrier
            if (a.normalize) {
                constexpr int KSTG = Geo<DT>::KST, CTH = (DT + 1) / 2;
                const int tt = wave & 1, hf = wave >> 1;
                bf16x8 bh_[KSTG], bl_[KSTG];
#pragma unroll
                for (int ks = 0; ks < KSTG; ++ks) {
                    bh_[ks] = mat_row_read8<TN>(Vh, LD, tt * 16, ks * 32, lane);
                    bl_[ks] = mat_row_read8<TN>(Vl, LD, tt * 16, ks * 32, lane);
                }
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < CTH; ++c) {
                    const int ct = hf * CTH + c;
                    if (ct < DT) {   // (uniform)
                        f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KSTG; ++ks) {
                            const bf16x8 gh = mat_row_read8<mat_new<DT>()>(Gh, GLD, ct * 16, ks * 32, lane), gl = mat_row_read8<mat_new<DT>()>(Gl, GLD, ct * 16, ks * 32, lane);
                            t4 = mfma_bf16(gh, bh_[ks], t4);
                            t4 = mfma_bf16(gl, bh_[ks], t4);
                            t4 = mfma_bf16(gh, bl_[ks], t4);
                        }
                        const uint2 qr = *reinterpret_cast<const uint2*>(Kh + (tt * 16 + mat_row<TN>(nl)) * LD + ct * 16 + kg * 4);   // q[s][16 ct + 4 kg ..]: hi part
                        f32x4 q4 = {__uint_as_float(qr.x << 16), __uint_as_float(qr.x & 0xffff0000u), __uint_as_float(qr.y << 16), __uint_as_float(qr.y & 0xffff0000u)};
                        if (LO) {   // (fp16 tensors: + lo part; bf16 values are their hi part)
                            const uint2 ql = *reinterpret_cast<const uint2*>(Kl + (tt * 16 + mat_row<TN>(nl)) * LD + ct * 16 + kg * 4);
                            q4 += f32x4{__uint_as_float(ql.x << 16), __uint_as_float(ql.x & 0xffff0000u), __uint_as_float(ql.y << 16), __uint_as_float(ql.y & 0xffff0000u)};
                        }
                        dot += t4[0] * q4[0] + t4[1] * q4[1] + t4[2] * q4[2] + t4[3] * q4[3];
                    }
                }
                dot += __shfl_xor(dot, 16, 64);
                dot += __shfl_xor(dot, 32, 64);
                if (kg == 0) rdp[hf * 32 + tt * 16 + nl] = dot;
            }
        }
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {   // row tiles past DT read zero / unused columns of the tile: harmless, not stored
            const int c0 = min(wave * RT + rt, DW / 16 - 1) * 16;
            ah[rt] = mat_tr_read8<TN>(Kh, LD, 0, c0, lane);
            if (LO) al[rt] = mat_tr_read8<TN>(Kl, LD, 0, c0, lane);
        }
#pragma unroll
        for (int ct = 0; ct < DT; ++ct) {
            const bf16x8 bh_ = mat_tr_read8<TN>(Vh, LD, 0, ct * 16, lane);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(ah[rt], bh_, acc[rt][ct]);
            if (LOY) {
                const bf16x8 bl_ = mat_tr_read8<TN>(Vl, LD, 0, ct * 16, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(ah[rt], bl_, acc[rt][ct]);
            }
            if (LO) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = mfma_bf16(al[rt], bh_, acc[rt][ct]);
            }
        }
        __syncthreads();
        if constexpr (RD) {   // dn[s] = -(dO . O)[s] / n[s] from the two halves of the chunk's row dots
            if (a.normalize && (tid % CGS) == 0) {
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int lr = r0 + RPP * it, r = crowc + lr;
                    if (r < S) a.dn[((long)bh * a.M + blk) * S + r] = -(rdp[lr] + rdp[32 + lr]) * nvc[it];
                }
            }
        }
    };
    int c0 = 0;
    for (; c0 + 32 < S; c0 += 32) chunk(std::true_type{}, c0);
    chunk(std::false_type{}, c0);   // (the block's last chunk)

    // KV_j -> ws  (C layout: row d1 = 16 tile + 4 kg + r, column d2 = 16 ct + nl)
    float* ob = a.out + ((long)bh * a.M + blk) * a.es;
    // fp32 summaries: through a wave-private LDS tile [16][CW + 4] (the operand tiles are free after the last product), so that a lane
    // stores 16-byte pieces and 16 lanes cover a whole row of the summary -- in the C layout a store instruction wrote four 64-byte
    // segments (k_sp_state ran at 3.9 TB/s of its bytes at C2)
    constexpr int CW = (DW > 64 && NWV == 8) ? 64 : DW, LDO = CW + 4, PPRO = CW / 4;
    static_assert(NWV * 16 * LDO * 4 <= 4 * 32 * LD * 2, "the staging tiles of all waves must fit in the operand tiles");
    const bool staged = !S16 && (P24 || ((D & 3) == 0 && (a.es & 3) == 0));   // (uniform; p24: D % 8 == 0 by the path's shape test)
    float hinv = 0.f;   // h16: 1 / the row's multiplier
    if constexpr (P24 == 2) {
        // the row's multiplier from the summary's largest magnitude: lane -> wave (shuffles) -> workgroup (NWV floats in `vecd`, free until
        // the column sums); padded rows / columns are zero products and do not matter
        float mx = 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < DT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(acc[rt][ct][r]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) vecd[wave] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWV; ++w) mx = fmaxf(mx, vecd[w]);
        const float hm = h16_mult_from_max(mx);
        hinv = h16_inv(hm);
        if (tid == 0) gst<float>(reinterpret_cast<char*>(ob) + 2 * D * D, hm);
    }
    if (staged) {
        float* Os = reinterpret_cast<float*>(smem_raw) + wave * 16 * LDO;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int tile = wave * RT + rt;
            if (tile * 16 < D) {   // (uniform)
#pragma unroll
                for (int c0 = 0; c0 < DW; c0 += CW) {
#pragma unroll
                    for (int ct = c0 / 16; ct < (c0 + CW) / 16; ++ct)
                        if (ct < DT)
#pragma unroll
                            for (int r = 0; r < 4; ++r) Os[(kg * 4 + r) * LDO + ct * 16 - c0 + nl] = acc[rt][ct][r];
                    __builtin_amdgcn_wave_barrier();
                    if constexpr (P24) {   // units of 8 elements: a 16-byte piece of the hi plane and an 8-byte piece of the lo plane
#pragma unroll
                        for (int p = 0; p < CW / 32; ++p) {
                            const int v = lane + 64 * p, row = v / (CW / 8), c8 = v % (CW / 8);
                            const int grow = tile * 16 + row, gcol = c0 + c8 * 8;
                            if (grow < D && gcol < D) {
                                const float* src = Os + row * LDO + c8 * 8;
                                const int e = grow * D + gcol;
                                if constexpr (P24 == 2) {
                                    gst<uint4>(reinterpret_cast<char*>(ob) + 2 * e, h16_pack8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), hinv));
                                } else {
                                uint4 hi;
                                uint2 lo;
                                p24_pack8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), hi, lo);
                                gst<uint4>(reinterpret_cast<char*>(ob) + 2 * e, hi);
                                gst<uint2>(reinterpret_cast<char*>(ob) + 2 * D * D + e, lo);
                                }
                            }
                        }
                    } else {
#pragma unroll
                    for (int p = 0; p < CW / 16; ++p) {
                        const int v = lane + 64 * p, row = v / PPRO, c4 = v % PPRO;
                        const int grow = tile * 16 + row, gcol = c0 + c4 * 4;
                        if (grow < D && gcol < D) gst<f32x4>(ob + (long)grow * D + gcol, *reinterpret_cast<const f32x4*>(Os + row * LDO + c4 * 4));
                    }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (MODE == 0 && a.normalize) __syncthreads();   // (the column sums below reuse the tiles)
    } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (wave * RT + rt) * 16 + kg * 4 + r, col = ct * 16 + nl;
                if (row < D && col < D) {
                    if (S16) reinterpret_cast<u16*>(a.out)[((long)bh * a.M + blk) * a.es + (long)row * D + col] = cvt_bf16(acc[rt][ct][r]);
                    else                 ob[(long)row * D + col] = acc[rt][ct][r];
                }
            }
    }

    if (MODE == 0 && a.normalize) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[r0 * DW + cg + i] = ksp[i];
        __syncthreads();
        if (tid < DW) {
            float s = 0.f;
#pragma unroll 4
            for (int r = 0; r < RPP; ++r) s += cs[r * DW + tid];
            vecd[tid] = s;
            if (tid < D) a.ksum[((long)bh * a.M + blk) * D + tid] = s;
        }
        __syncthreads();
        // z_j[s] = Qden_j[s] . ksum_j : CGS lanes per token row
        const T* qb = (const T*)a.qd.ptr + b * a.qd.sb + h * a.qd.sh;
        float kv8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv8[i] = vecd[cg + i];
        f32x4 qw[PRO ? 2 : 1];   // PRO: the q norm weights of the thread's channels
        if constexpr (PRO) {
            qw[0] = qw[1] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (a.pro_wq && cg < D) {
                qw[0] = *reinterpret_cast<const f32x4*>(a.pro_wq + h * D + cg);
                qw[1] = *reinterpret_cast<const f32x4*>(a.pro_wq + h * D + cg + 4);
            }
        }
        constexpr int ZB = 4;   // token rows in flight per thread: the loads of a batch are issued before any is used
        for (int rb = 0; rb < S; rb += RPP * ZB) {
            f32x4 x0[ZB], x1[ZB];
            int rw[ZB];   // the batch's rows first (the map lookups travel together), then the batch's data loads
            if (a.idx) {
#pragma unroll
                for (int u = 0; u < ZB; ++u) rw[u] = gld<int>(a.idx + p0 + min(rb + u * RPP + r0, S - 1));
            } else {
#pragma unroll
                for (int u = 0; u < ZB; ++u) rw[u] = (int)p0 + min(rb + u * RPP + r0, S - 1);
            }
#pragma unroll
            for (int u = 0; u < ZB; ++u) {
                const int r = rb + u * RPP + r0;
                x0[u] = x1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < S && cg < D) ld8(qb + (long)rw[u] * a.qd.sn + cg, x0[u], x1[u]);
            }
#pragma unroll
            for (int u = 0; u < ZB; ++u) {
                const int r = rb + u * RPP + r0;
                float d = 0.f;
                if (r < S && cg < D) {
                    if constexpr (PRO) {
                        const float rq = a.pro_rq ? a.pro_rq[b * a.pro_n + rw[u]] : 1.f;
                        x0[u] = (x0[u] * rq) * qw[0];
                        x1[u] = (x1[u] * rq) * qw[1];
                    }
                    if (a.relu) relu8(x0[u], x1[u], a.eps);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d += x0[u][i] * kv8[i] + x1[u][i] * kv8[4 + i];
                }
#pragma unroll
                for (int o = CGS / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
                if (r < S && (tid % CGS) == 0) a.zo[((long)bh * a.M + blk) * S + r] = d;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
constexpr int SPM_TE = 128, SPM_LD = SPM_TE + 8, SPM_TILE = 32 * SPM_LD;
constexpr int SP_MIX_SMEM = 4 * SPM_TILE * 2;   // two buffers of (hi, lo) [32][SPM_LD] bf16; reused as [64][SPM_TE + 4] fp32 staging
static_assert(64 * (SPM_TE + 4) * 4 <= SP_MIX_SMEM, "output staging must fit in the input tiles");
// bf16 summaries: no lo tiles and a bf16 output staging [64][SPM_LD]: half the LDS, twice the resident workgroups
constexpr int SP_MIX_SMEM16 = 2 * SPM_TILE * 2;
static_assert(64 * SPM_LD * 2 <= SP_MIX_SMEM16, "bf16 output staging must fit in the input tiles");
template <bool S16> constexpr int sp_mix_smem() { return S16 ? SP_MIX_SMEM16 : SP_MIX_SMEM; }

template <int TRANS, bool S16 = false>   // S16: summaries stored as bf16 (no lo part)
__global__ __launch_bounds__(NTHREADS, 2) void k_sp_mix(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const long e0 = (long)blockIdx.x * SPM_TE;
    const int i0 = blockIdx.y * 64, bh = blockIdx.z, M = a.M;
    const float* inb = a.in + (long)bh * M * a.es + e0;
    float* outb = a.out + (long)bh * M * a.es + e0;
    const u16* inb16 = reinterpret_cast<const u16*>(a.in) + (long)bh * M * a.es + e0;
    u16* outb16 = reinterpret_cast<u16*>(a.out) + (long)bh * M * a.es + e0;
    const int steps = (M + 31) / 32;
    const int sr = tid >> 3, sc = (tid & 7) * 8;   // staging: row sr, 8 floats at columns sc and sc + 64

    f32x4 pre[2][2];
    uint4 pre16[2];
    auto fetch = [&](int step) {
        const int row = step * 32 + sr;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pre[u][0] = pre[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            pre16[u] = make_uint4(0, 0, 0, 0);
        }
        if (row < M) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (e0 + sc + 64 * u < a.E) {   // E is a multiple of 8
                    if (S16) {
                        pre16[u] = *reinterpret_cast<const uint4*>(inb16 + (long)row * a.es + sc + 64 * u);
                    } else {
                        const float* src = inb + (long)row * a.es + sc;
                        pre[u][0] = *reinterpret_cast<const f32x4*>(src + 64 * u);
                        pre[u][1] = *reinterpret_cast<const f32x4*>(src + 64 * u + 4);
                    }
                }
        }
    };
    auto commit = [&](int buf) {
        u16* th = Xs + buf * (S16 ? 1 : 2) * SPM_TILE + sr * SPM_LD + sc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (S16) {
                *reinterpret_cast<uint4*>(th + 64 * u) = pre16[u];
            } else {
                uint4 hi, lo;
                split8(pre[u][0], pre[u][1], hi, lo);
                *reinterpret_cast<uint4*>(th + 64 * u) = hi;
                *reinterpret_cast<uint4*>(th + SPM_TILE + 64 * u) = lo;
            }
        }
    };

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int orow = i0 + wave * 16 + nl;

    // this lane's 8 mixing weights of a step (row orow, columns step * 32 + kg * 8 ..): fetched one step ahead like the tiles,
    // so their L2 latency is not on the MFMA chain
    float wc[8], wn[8];
    const bool wvec = !TRANS && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 && (a.ldw & 3) == 0;
    const bool wvec2 = !TRANS && (reinterpret_cast<uintptr_t>(a.W) & 7) == 0 && (a.ldw & 1) == 0;
    auto fetch_w = [&](int step, float (&w)[8]) {
        const int k0 = step * 32 + kg * 8;
        if (wvec && orow < M && k0 + 8 <= M) {   // 8 consecutive weights of one row: two 16-byte loads
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.W + (long)orow * a.ldw + k0);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.W + (long)orow * a.ldw + k0 + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { w[t] = w0[t]; w[4 + t] = w1[t]; }
            return;
        }
        if (wvec2 && orow < M && k0 + 8 <= M) {  // even row stride (M = 150): four 8-byte loads
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 wp = *reinterpret_cast<const float2*>(a.W + (long)orow * a.ldw + k0 + 2 * t);
                w[2 * t] = wp.x; w[2 * t + 1] = wp.y;
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int kk = k0 + t;
            w[t] = (orow < M && kk < M) ? (TRANS ? a.W[(long)kk * a.ldw + orow] : a.W[(long)orow * a.ldw + kk]) : 0.f;
        }
    };
    fetch(0);
    fetch_w(0, wc);
    commit(0);
    for (int step = 0; step < steps; ++step) {
        const int buf = step & 1;
        __syncthreads();
        if (step + 1 < steps) {
            fetch(step + 1);
            fetch_w(step + 1, wn);
        }
        bf16x8 ah, al;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const __bf16 hi = (__bf16)wc[t];
            ah[t] = hi;
            al[t] = (__bf16)(wc[t] - (float)hi);
        }
        const u16* th = Xs + buf * (S16 ? 1 : 2) * SPM_TILE;
#pragma unroll
        for (int t4 = 0; t4 < 8; t4 += 4) {
            bf16x8 bh_[4], bl_[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bh_[t] = tr_read8(th, SPM_LD, 0, (t4 + t) * 16, lane);
                if (!S16) bl_[t] = tr_read8(th + SPM_TILE, SPM_LD, 0, (t4 + t) * 16, lane);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(ah, bh_[t], acc[t4 + t]);
            if (!S16)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(ah, bl_[t], acc[t4 + t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(al, bh_[t], acc[t4 + t]);
        }
        if (step + 1 < steps) {
            commit(buf ^ 1);
#pragma unroll
            for (int t = 0; t < 8; ++t) wc[t] = wn[t];
        }
    }
    __syncthreads();
    if constexpr (S16) {
        u16* Os = Xs;   // [64][SPM_LD] bf16
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(wave * 16 + kg * 4 + r) * SPM_LD + t * 16 + nl] = cvt_bf16(acc[t][r]);
        __syncthreads();
#pragma unroll
        for (int v0 = 0; v0 < 4; ++v0) {
            const int v = tid + v0 * NTHREADS, r = v >> 4, c = (v & 15) * 8;
            if (i0 + r < M && e0 + c < a.E)   // E is a multiple of 8
                *reinterpret_cast<uint4*>(outb16 + (long)(i0 + r) * a.es + c) = *reinterpret_cast<const uint4*>(Os + r * SPM_LD + c);
        }
    } else {
        float* Os = reinterpret_cast<float*>(smem_raw);   // [64][SPM_TE + 4]
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(wave * 16 + kg * 4 + r) * (SPM_TE + 4) + t * 16 + nl] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int v0 = 0; v0 < 8; ++v0) {
            const int v = tid + v0 * NTHREADS, r = v >> 5, c = (v & 31) * 4;
            if (i0 + r < M && e0 + c < a.E)
                *reinterpret_cast<f32x4*>(outb + (long)(i0 + r) * a.es + c) = *reinterpret_cast<const f32x4*>(Os + r * (SPM_TE + 4) + c);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_sp_mixr<NW, TRANS, S16>: the same mixing with EVERY block of the (b,h) resident in the workgroup (M <= 16 NW <= 256), the
// causal path's k_csf_mixf for a full matrix.  A workgroup owns slices of 256-byte row pieces (TE = 64 fp32 or 128 bf16 elements):
// the slice's rows of all blocks sit in LDS (fp32 summaries as bf16 hi + lo tiles), so every summary byte is read from HBM once
// (k_sp_mix re-reads a slice per 64-row output tile: 1.5x at M = 150, 4x at M = 256) and the mixing weights of a wave's 16 output
// blocks live in its registers as bf16 hi + lo for the whole launch (k_sp_mix rebuilt them per step and was bound by its LDS
// operand reads: 3.1 TB/s at the Wan shape, 2.3 TB/s at M = 256).  One wave per 16 output blocks; the summaries enter the MFMA
// through the transpose read as the A operand, so a lane ends up with four consecutive elements of one output block (16-byte /
// 8-byte staging writes, full 256-byte rows out); a workgroup walks `spw` consecutive slices with the next slice's rows in
// flight in registers.  grid: wgs <= 256 persistent workgroups of 64 NW threads.
//   TRANS 0: out[i][e] = sum_j W[i][j] in[j][e]      TRANS 1: out[j][e] = sum_i W[i][j] in[i][e]
// -------------------------------------------------------------------------------------------------
// Mixing weights of a wave's NB tiles of 16 output blocks as MFMA B operands, bf16 hi + lo:
//   B[k = r][n = o] = Wm(o, r), o = obase + 16 n + (lane & 15), r = 32 ks + 8 (lane >> 4) + t   (Wm = W, or W^T with TRANS)
// Every workgroup needs the whole matrix: it is fetched in chunks of 64 input blocks with coalesced 16-byte loads into LDS (`Tf`,
// any region of ROWS x 68 / 64 x (ROWS + 4) floats that is free before the first slice) and picked from there.  Per-lane dword
// loads straight from global memory touched 16 lines per instruction for 16 bytes of each and re-fetched them for every t: 35 us
// of a 190 us launch at M = 256.  Blocks past M get zero weights.  Ends with a barrier (Tf may be reused).
template <int TRANS, int NTH, int ROWS, int NK, int NB>
__device__ __forceinline__ void mixr_weights(bf16x8 (&wh)[NK][NB], bf16x8 (&wl)[NK][NB], float* __restrict__ Tf, const float* __restrict__ W,
                                             int ldw, int M, int obase, int tid) {
    constexpr int TR = TRANS ? 64 : ROWS, TC = TRANS ? ROWS : 64, LDT = TC + 4, PPRW = TC / 4, NCH = (NK + 1) / 2;
    const int lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (ldw & 3) == 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        for (int v = tid; v < TR * PPRW; v += NTH) {
            const int row = v / PPRW, c4 = (v - row * PPRW) * 4;
            const int gr = TRANS ? c * 64 + row : row, gc = TRANS ? c4 : c * 64 + c4;   // row / first column in W
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (gr < M) {
                const float* src = W + (long)gr * ldw + gc;
                if (vec && gc + 4 <= M) {
                    x = gld<f32x4>(src);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (gc + i < M) x[i] = gld<float>(src + i);
                }
            }
            *reinterpret_cast<f32x4*>(Tf + row * LDT + c4) = x;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int o = obase + n * 16 + nl;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                if (2 * c + k2 < NK) {
                    float w[8];
                    if (TRANS) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) w[t] = Tf[(k2 * 32 + kg * 8 + t) * LDT + o];
                    } else {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8);
                        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8 + 4);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { w[t] = lo4[t]; w[4 + t] = hi4[t]; }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const __bf16 h = (__bf16)w[t];
                        wh[2 * c + k2][n][t] = h;
                        wl[2 * c + k2][n][t] = (__bf16)(w[t] - (float)h);
                    }
                }
            }
        }
    }
    __syncthreads();
}

struct MixrArgs {
    const float* W;
    int ldw;
    const void* in;
    void* out;
    int M;
    long E;        // elements per block summary (multiple of the slice width)
    long es;       // row stride of `in` / `out` in elements (E + padding)
    long total;    // slices = bh * E / TE
    int spw;       // slices per workgroup
    // k_sp_mixr_dma only: the normaliser's product with the same weights, out = f(sum_r Wm(o, r) zin[bh][r][s]) for S <= 16 fp32
    // values per block (k_wz<0>: f = 1 / (eps + .), Wm = W with TRANS 0; k_wz<1>: f = identity, Wm = W^T with TRANS 1); null: none
    const float* zin;
    float* zout;
    int S;
    float eps;
    unsigned long long* trace;   // debugging aid (mhla_debug_set_trace): s_memtime stamps of the first workgroups' slice loop, or null
    // k_sp_mixr<.., DW> (backward, fp32 summaries): the OTHER summary set of the dW product (KV, laid out like `in` = dG) and the
    // per-workgroup partials dwp[workgroup][M][M] of dW[i][j] = sum_bh sum_e dG[i][e] KV[j][e] -- formed from the slice rows this
    // kernel stages anyway, so that dG is read from HBM once for dKV and dW together
    const void* in2;
    float* dwp;
    // k_sp_mixr, fp32 summaries: the normaliser's product rides along as EXTRA SLICES after the summaries' -- block r's S values zin[bh][r][:]
    // are one more row set mixed with the same weights (slices of TE values, ztotal = bh * ceil(S / TE) of them, dealt round-robin over the workgroups; S even), stored
    // through f into zout.  DW: zin = dn, zin2 = z, and the <dn_i, z_j> term of dW falls out of the same products (k_dw is not launched).
    long ztotal;
    const float* zin2;
};
// slice width: 256-byte row pieces; 128-byte ones for 16 waves (1024 threads on 128 VGPRs: half the accumulators and staging registers)
template <int NW, bool S16> __host__ __device__ constexpr int mixr_te() { return (S16 ? 128 : 64) / (NW > 12 ? 2 : 1); }
template <int NW, bool S16, bool DW = false>
__host__ __device__ constexpr int sp_mixr_smem() {
    constexpr int TE = mixr_te<NW, S16>(), ROWS = 16 * NW;
    return (S16 ? 1 : (DW ? 4 : 2)) * ROWS * (TE + 8) * 2 + (S16 ? ROWS * (TE + 8) * 2 : ROWS * (TE + 4) * 4);
}

// DW (TRANS 1, fp32 summaries, M <= 128): the kernel also stages the KV rows of every slice (hi + lo, two more tiles) and accumulates
// dW[i][j] += sum_{e in slice} dG[i][e] KV[j][e] over ALL its slices -- every (b,h) it meets: dW is their sum anyway -- in
// registers (wave w owns rows i = 16 w .. 16 w + 15, all columns: NW tiles); one [M][M] partial per workgroup at the end, summed in a
// fixed order by k_dw_reduce.  Replaces k_sp_dw, which read dG and KV a second time (C2 at the default arithmetic: 81 us).
// P24: the summaries (in, in2, out) are stored as 24-bit floats in two planes per row (p24_pack8).  A thread's unit is then 8 elements
// of a row -- one 16-byte piece of the hi plane and one 8-byte piece of the lo plane -- instead of a 16-byte piece of 4 floats.
// (h16 summaries are mixed by k_sp_mixh, mixh.hpp: on the payload, with the fp16 MFMA.  A decode-at-the-commit variant of THIS kernel was
// built first in round 6 and was no faster than p24: 2-byte rows in 64-element slices put fewer bytes in flight per workgroup.)
template <int NW, int TRANS, bool S16, bool DW = false, int P24 = 0>
__global__ __launch_bounds__(64 * NW, (NW == 8 && !DW && P24 == 1) ? 4 : (NW + 3) / 4) void k_sp_mixr(const MixrArgs a) {   // (eight waves on 24-bit summaries without dW: 128 VGPRs, two workgroups per CU instead of one at 136)
    static_assert(!DW || (TRANS == 1 && !S16 && NW <= 8), "dW rides in the backward's fp32 mixing kernel, M <= 128 (twelve waves: 77 spilled registers)");
    static_assert(P24 == 0 || (P24 == 1 && !S16 && NW <= 12), "p24: fp32-grade summaries, slices of 64 elements");
    constexpr int TE = mixr_te<NW, S16>(), ROWS = 16 * NW, LD = TE + 8, LDO = TE + 4, NK = (NW + 1) / 2, NT = TE / 16;
    constexpr int PPR = TE * (S16 ? 2 : 4) / 16;          // 16-byte pieces per row of the slice (16, or 8 with 16 waves)
    constexpr int NTH = 64 * NW, NP = P24 ? ROWS * 8 / NTH : ROWS * PPR / NTH;   // pieces (P24: units of 8 elements) per thread and slice
    constexpr int UPR = P24 ? 8 : PPR;                    // ... per row
    static_assert(NP * NTH == ROWS * UPR, "pieces must tile the slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Th = reinterpret_cast<u16*>(smem_raw);
    u16* Tl = Th + ROWS * LD;                                        // (fp32 summaries only)
    u16* Kh = Tl + ROWS * LD;                                        // (DW only: the KV rows of the slice)
    u16* Kl = Kh + ROWS * LD;
    unsigned char* Os = smem_raw + (S16 ? 1 : (DW ? 4 : 2)) * ROWS * LD * 2;     // bf16 [ROWS][LD] or fp32 [ROWS][LDO]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const long nsl = a.E / TE;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt_s = (int)max(0L, min((long)a.spw, a.total - s0));   // this workgroup's summary slices: a consecutive range
    const int nzs = (a.S + TE - 1) / TE;                              // normaliser slices per (b, h): dealt round-robin, see the loops
    if (cnt_s <= 0 && (S16 || (long)blockIdx.x >= a.ztotal)) return;
    // B operand: B[k = r][n = o] = weight of input block r in output block o = 16 wave + nl, r = 32 ks + 8 kg + t
    bf16x8 wh[NK][1], wl[NK][1];
    static_assert((TRANS ? 64 * (ROWS + 4) : ROWS * 68) * 4 <= sp_mixr_smem<NW, S16, DW>(), "weight chunk must fit in the tiles");
    mixr_weights<TRANS, 64 * NW, ROWS, NK, 1>(wh, wl, reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, wave * 16, tid);
    constexpr int ESZ = S16 ? 2 : 4;
    constexpr int SLB = P24 ? 128 : TE * ESZ;             // bytes of a slice in a summary row (P24: of its hi plane)
    // P24: the lo piece of a unit relative to its hi piece (goff + 16 c of slice es): lo plane at 2 E, slice at 64 es, piece at 8 c
    auto lo_rel = [&](int es, int c) { return (long)2 * a.E - 64 * es - 8 * c; };
    // byte offset of slice (bh, es); a workgroup's slices are consecutive, so the pair is advanced rather than divided out per
    // slice (the 64-bit division was 150 instructions with branches between the barrier and the next slice's loads)
    auto slice_off = [&](int bh, int es) { return (long)bh * M * a.es * ESZ + (long)es * SLB; };
    auto zslice_off = [&](int bh, int es) { return ((long)bh * M * a.S + (long)es * TE) * 4; };   // (normaliser rows: S plain floats)
    auto advance = [&](int& bh, int& es) { if (++es == (int)nsl) { es = 0; ++bh; } };
    // the thread's pieces: piece v = tid + p NTH -> row v / PPR, 16 bytes at column piece v % PPR (rows past M: the last row, zeroed)
    unsigned goff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int v = tid + p * NTH, row = v / UPR, c = v % UPR;
        goff[p] = (unsigned)((long)(row < M ? row : M - 1) * a.es * ESZ + c * 16);   // (P24: the unit's hi piece; its lo piece: lo_rel)
    }
    // staging registers.  P24: pre = hi piece, prl = lo piece; a normaliser slice (plain floats) takes the unit's 8 floats in pre + prz
    struct Stage {
        uint4 pre[NP], pre2[DW ? NP : 1], prz[P24 ? NP : 1], prz2[(P24 && DW) ? NP : 1];
        u32x2_t prl[P24 ? NP : 1], prl2[(P24 && DW) ? NP : 1];   // (native vectors: an array of uint2 in here stays in scratch)
    };
    // (A second slice in flight per workgroup -- two Stage objects, the loop unrolled by two -- gained nothing: hipcc's wait in front of
    // the commit is vmcnt(0), i.e. for both.  Nor did starting the workgroups of a CU a fraction of an iteration apart, or fetching and
    // storing the lo pieces of an even / odd slice pair together, whole 128-byte lines at a time.)
    constexpr int NBUF = 1;
    Stage sga;
    // a normaliser slice: rows of S floats, pieces past the row's end are not touched (zeros)
    constexpr int ZPB = P24 ? 32 : 16, ZPF = ZPB / 4;   // bytes / floats of a thread's piece of a normaliser row
    auto zoff = [&](int p) { const int v = tid + p * NTH, row = v / UPR, c = v % UPR; return (unsigned)((row < M ? row : M - 1) * a.S * 4 + c * ZPB); };
    auto zlive = [&](int p, int es, int half = 0) { return ((tid + p * NTH) % UPR) * ZPF + half * 4 + es * TE < a.S; };
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // four floats of a normaliser row (piece `half` of unit p): one 16-byte load when rows are multiples of 4 floats, two 8-byte loads
    // when they are only even (S = 210: rows start on 8-byte boundaries), each half on its own side of the row's end
    // Unconditional, from clamped addresses -- a load behind a lane-dependent branch makes hipcc wait for everything in flight where the
    // branch joins (eight serial round trips per normaliser slice, and those workgroups were the kernel's tail); floats past the row's end
    // are zeroed at the commit (zmask).
    const bool zwide = (a.S & 3) == 0;   // (uniform)
    auto zf0 = [&](int p, int es, int half) { return ((tid + p * NTH) % UPR) * ZPF + half * 4 + es * TE; };
    auto zld = [&](const char* base, int p, int es, int half) __attribute__((always_inline)) {
        const int f0 = zf0(p, es, half);
        const char* src = base + zoff(p) + half * 16;
        if (zwide) return gld_stream16(f0 < a.S ? src : base);
        const uint2 lo = gld<uint2>(f0 < a.S ? src : base), hi = gld<uint2>(f0 + 2 < a.S ? src + 8 : base);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto zmask = [&](const uint4& x, bool ok, int p, int es, int half) {
        const int f0 = zf0(p, es, half);
        return make_uint4((ok && f0 < a.S) ? x.x : 0u, (ok && f0 + 1 < a.S) ? x.y : 0u, (ok && f0 + 2 < a.S) ? x.z : 0u, (ok && f0 + 3 < a.S) ? x.w : 0u);
    };
    // ZS: a normaliser slice.  A compile-time flag, and the summary slices and the normaliser slices of a workgroup are two loops:
    // with both kinds of loads behind one runtime branch, writing the same staging registers, hipcc waited for ALL loads right after
    // issuing them (`L L L L W0 M M`: k_sp_mixr<1,dw> 78 -> 121 us at C2).
    auto issue = [&]<bool ZS>(std::bool_constant<ZS>, Stage& g, int bh, int es) __attribute__((always_inline)) {
        const long boff = ZS ? zslice_off(bh, es) : slice_off(bh, es);
        if constexpr (ZS) {
            const char* base = reinterpret_cast<const char*>(a.zin) + boff;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                g.pre[p] = zld(base, p, es, 0);
                if constexpr (P24) g.prz[p] = zld(base, p, es, 1);
            }
            if constexpr (DW) {
                const char* base2 = reinterpret_cast<const char*>(a.zin2) + boff;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    g.pre2[p] = zld(base2, p, es, 0);
                    if constexpr (P24) g.prz2[p] = zld(base2, p, es, 1);
                }
            }
            return;
        }
        const char* base = reinterpret_cast<const char*>(a.in) + boff;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            g.pre[p] = gld_stream16(base + goff[p]);
            if constexpr (P24) g.prl[p] = *(const MHLA_GLOBAL_AS u32x2_t*)(base + goff[p] + lo_rel(es, (tid + p * NTH) % UPR));
        }
        if constexpr (DW) {
            const char* base2 = reinterpret_cast<const char*>(a.in2) + boff;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                g.pre2[p] = gld_stream16(base2 + goff[p]);
                if constexpr (P24) g.prl2[p] = *(const MHLA_GLOBAL_AS u32x2_t*)(base2 + goff[p] + lo_rel(es, (tid + p * NTH) % UPR));
            }
        }
    };
    f32x4 dwacc[DW ? NW : 1];
#pragma unroll
    for (int t = 0; t < (DW ? NW : 1); ++t) dwacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four floats -> four bf16 hi + four bf16 lo at tile position (row, 4 c)
    auto commit_hl = [&](u16* th, u16* tl, const uint4& x, int row, int c) {
        const float f[4] = {__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)};
        float l[4];
        unsigned short hs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hs[i] = cvt_bf16(f[i]);
            l[i] = f[i] - __uint_as_float((unsigned)hs[i] << 16);
        }
        *reinterpret_cast<uint2*>(th + row * LD + c * 4) = make_uint2(hs[0] | ((unsigned)hs[1] << 16), hs[2] | ((unsigned)hs[3] << 16));
        *reinterpret_cast<uint2*>(tl + row * LD + c * 4) = make_uint2(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]));
    };
    // The stores of a slice are issued at the top of the NEXT iteration, after that slice's loads have been committed to LDS: hipcc
    // waits with vmcnt(0) in front of the commit -- with the stores at the end of the iteration that wait included their
    // acknowledgement (loads and stores retire in order), a round trip per slice on top of the load's.  The staging tile keeps the
    // results across the iteration boundary: it is not written again before the barrier that follows the stores.
    long poff = 0;
    bool pz = false;
    int pzes = 0;
    auto store_slice = [&](long off, bool zslice, int zes) __attribute__((always_inline)) {
        if constexpr (!S16) {
            if (zslice) {   // the normaliser's rows: 1 / (eps + .) in the forward, as they are in the backward
                char* zb = reinterpret_cast<char*>(a.zout) + off;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int v = tid + p * NTH, row = v / UPR, c = v % UPR;
#pragma unroll
                    for (int hf = 0; hf < ZPF / 4; ++hf) {
                        if (row < M && zlive(p, zes, hf)) {
                            f32x4 x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Os) + row * LDO + c * ZPF + hf * 4);
                            if (TRANS == 0)
#pragma unroll
                                for (int i = 0; i < 4; ++i) x[i] = 1.f / (a.eps + x[i]);
                            if (zwide) {
                                *reinterpret_cast<f32x4*>(zb + zoff(p) + hf * 16) = x;
                            } else {   // (even rows: 8-byte pieces, the second one only inside the row)
                                *reinterpret_cast<f32x2*>(zb + zoff(p) + hf * 16) = f32x2{x[0], x[1]};
                                if (((tid + p * NTH) % UPR) * ZPF + hf * 4 + zes * TE + 2 < a.S) *reinterpret_cast<f32x2*>(zb + zoff(p) + hf * 16 + 8) = f32x2{x[2], x[3]};
                            }
                        }
                    }
                }
                return;
            }
        }
        char* ob = reinterpret_cast<char*>(a.out) + off;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int v = tid + p * NTH, row = v / UPR, c = v % UPR;
            if constexpr (P24) {
                if (row < M) {
                    const float* src = reinterpret_cast<const float*>(Os) + row * LDO + c * 8;
                    uint4 hi;
                    uint2 lo;
                    p24_pack8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), hi, lo);
                    gst<uint4>(ob + goff[p], hi);
                    gst<uint2>(ob + goff[p] + lo_rel(zes, c), lo);
                }
                continue;
            }
            if (row < M) {
                const uint4 x = S16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(Os) + row * LD + c * 8)
                                    : *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(Os) + row * LDO + c * 4);
                gst<uint4>(ob + goff[p], x);
            }
        }
    };
    // it: the slice's position in this phase; first: nothing to store yet; more: another slice of the SAME kind follows (prefetch it)
    auto body = [&]<bool ZS>(std::bool_constant<ZS> zs, Stage& g, bool first, bool more, int bh, int es, int nbh, int nes) __attribute__((always_inline)) {
        const long off = ZS ? zslice_off(bh, es) : slice_off(bh, es);
        constexpr bool zslice = ZS;
        const int zes = es;
#ifdef MIXR_X_NOCOMMIT
        if (first)
#endif
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int v = tid + p * NTH, row = v / UPR, c = v % UPR;
            const bool ok = row < M;   // rows past the last block: zeros (their weights are zero too, but 0 x NaN is not)
            const uint4 x = make_uint4(ok ? g.pre[p].x : 0u, ok ? g.pre[p].y : 0u, ok ? g.pre[p].z : 0u, ok ? g.pre[p].w : 0u);
            if constexpr (P24) {
                if (zslice) {   // 8 plain floats
                    commit_hl(Th, Tl, zmask(g.pre[p], ok, p, zes, 0), row, 2 * c);
                    commit_hl(Th, Tl, zmask(g.prz[p], ok, p, zes, 1), row, 2 * c + 1);
                    if constexpr (DW) {
                        commit_hl(Kh, Kl, zmask(g.pre2[p], ok, p, zes, 0), row, 2 * c);
                        commit_hl(Kh, Kl, zmask(g.prz2[p], ok, p, zes, 1), row, 2 * c + 1);
                    }
                } else {   // the hi piece is the operand; the lo operand is rebuilt from the third bytes
                    const uint2 l = make_uint2(ok ? g.prl[p][0] : 0u, ok ? g.prl[p][1] : 0u);
                    *reinterpret_cast<uint4*>(Th + row * LD + c * 8) = x;
                    *reinterpret_cast<uint4*>(Tl + row * LD + c * 8) = p24_lo8(x, l);
                    if constexpr (DW) {
                        const uint4 y = make_uint4(ok ? g.pre2[p].x : 0u, ok ? g.pre2[p].y : 0u, ok ? g.pre2[p].z : 0u, ok ? g.pre2[p].w : 0u);
                        const uint2 yl = make_uint2(ok ? g.prl2[p][0] : 0u, ok ? g.prl2[p][1] : 0u);
                        *reinterpret_cast<uint4*>(Kh + row * LD + c * 8) = y;
                        *reinterpret_cast<uint4*>(Kl + row * LD + c * 8) = p24_lo8(y, yl);
                    }
                }
            } else if constexpr (S16) {
                *reinterpret_cast<uint4*>(Th + row * LD + c * 8) = x;
            } else if (zslice) {   // (fp32 summaries: the unit is the 16-byte piece)
                commit_hl(Th, Tl, zmask(g.pre[p], ok, p, zes, 0), row, c);
                if constexpr (DW) commit_hl(Kh, Kl, zmask(g.pre2[p], ok, p, zes, 0), row, c);
            } else {   // four floats -> four bf16 hi + four bf16 lo
                commit_hl(Th, Tl, x, row, c);
                if constexpr (DW) {
                    const uint4 y = make_uint4(ok ? g.pre2[p].x : 0u, ok ? g.pre2[p].y : 0u, ok ? g.pre2[p].z : 0u, ok ? g.pre2[p].w : 0u);
                    commit_hl(Kh, Kl, y, row, c);
                }
            }
        }
#ifndef MIXR_X_NOSTORE   // (experiment builds only, tools/build_variant.sh: MIXR_X_* drop one phase of the slice loop -- results are wrong)
        if (!first) store_slice(poff, pz, pzes);
#endif
        __syncthreads();
        if (more) {
#ifndef MIXR_X_NOLOAD
            issue(zs, g, nbh, nes);
#endif
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kend = (M + 31) / 32;   // (uniform) reduction steps that hold a block
#ifndef MIXR_X_NOMFMA
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks < kend) {
                constexpr int TB = NT < 4 ? NT : 4;   // operand tiles per batch
#pragma unroll
                for (int t4 = 0; t4 < NT; t4 += TB) {
                    bf16x8 sv[TB], sl[S16 ? 1 : TB];
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        sv[t] = tr_read8(Th, LD, ks * 32, (t4 + t) * 16, lane);
                        if constexpr (!S16) sl[t] = tr_read8(Tl, LD, ks * 32, (t4 + t) * 16, lane);
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wh[ks][0], acc[t4 + t]);
                    if constexpr (!S16) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sl[t], wh[ks][0], acc[t4 + t]);
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wl[ks][0], acc[t4 + t]);
                }
            }
        }
#endif
        if constexpr (DW) {   // dW[i][j] += sum_e dG[i][e] KV[j][e]: rows i of this wave, every column tile that holds a block
            if (wave * 16 < M) {   // (uniform)
#pragma unroll
                for (int k2 = 0; k2 < TE / 32; ++k2) {
                    const bf16x8 ah = row_read8(Th, LD, wave * 16, k2 * 32, lane), al = row_read8(Tl, LD, wave * 16, k2 * 32, lane);   // A[m = i][k = e]
#pragma unroll
                    for (int jt = 0; jt < NW; ++jt) {
                        if (jt * 16 < M) {   // (uniform)
                            const bf16x8 bh_ = row_read8(Kh, LD, jt * 16, k2 * 32, lane), bl_ = row_read8(Kl, LD, jt * 16, k2 * 32, lane);   // B[k = e][n = j]
                            dwacc[jt] = mfma_bf16(ah, bh_, dwacc[jt]);
                            dwacc[jt] = mfma_bf16(ah, bl_, dwacc[jt]);
                            dwacc[jt] = mfma_bf16(al, bh_, dwacc[jt]);
                        }
                    }
                }
            }
        }
        // lane: elements 16 t + 4 kg .. + 3 of output block 16 wave + nl -> staging tile [block][element]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (S16)
                *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(Os) + (wave * 16 + nl) * LD + t * 16 + kg * 4) =
                    make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));
            else
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Os) + (wave * 16 + nl) * LDO + t * 16 + kg * 4) = acc[t];
        }
        __syncthreads();
        poff = off;
        pz = zslice;
        pzes = zes;
    };
    // The workgroup's summary slices (a consecutive range), then its share of the normaliser slices: slice zi goes to workgroup
    // zi % gridDim.x -- at C2 one of them for a quarter of the workgroups -- so that no workgroup is left with a tail of them (as a
    // range behind the summaries they were 17 latency-bound iterations for the last eight workgroups).
    if (cnt_s > 0) {
        int bh = (int)(s0 / nsl), es = (int)(s0 - (long)bh * nsl);
        issue(std::false_type{}, sga, bh, es);
        for (int it = 0; it < cnt_s; ++it) {
            int nb = bh, ne = es;
            advance(nb, ne);
            body(std::false_type{}, sga, it == 0, it + 1 < cnt_s, bh, es, nb, ne);
            bh = nb;
            es = ne;
        }
    }
    if constexpr (!S16) {
        bool firstz = cnt_s <= 0;
        for (long zi = blockIdx.x; zi < a.ztotal; zi += gridDim.x) {   // (the first one is not prefetched under the last summary slice)
            const long nzi = zi + gridDim.x;
            if (zi == (long)blockIdx.x) issue(std::true_type{}, sga, (int)(zi / nzs), (int)(zi % nzs));
            body(std::true_type{}, sga, firstz, nzi < a.ztotal, (int)(zi / nzs), (int)(zi % nzs), (int)(nzi / nzs), (int)(nzi % nzs));
            firstz = false;
        }
    }
    store_slice(poff, pz, pzes);
    if constexpr (DW) {   // C layout: rows i = 16 wave + 4 kg + r, column j = 16 jt + nl
        float* dp = a.dwp + (long)blockIdx.x * M * M;
#pragma unroll
        for (int jt = 0; jt < NW; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = wave * 16 + kg * 4 + r, j = jt * 16 + nl;
                if (i < M && j < M) dp[(long)i * M + j] = dwacc[jt][r];
            }
    }
}

// -------------------------------------------------------------------------------------------------
// k_sp_mixr_dma<TRANS>: resident-sequence mixing for 16-bit summaries and 192 < M <= 256 blocks.
// k_sp_mixr<16> (sixteen waves, sixteen output blocks each, 64-element slices, register staging) ran at 3 TB/s.  This kernel
//   * runs EIGHT waves with 32 output blocks each (256-register budget): every operand read feeds four MFMAs (two block tiles x
//     weight hi / lo), the next reduction step's operands are requested before the current step's products, half the LDS reads;
//   * stages by LDS-DMA (global_load_lds_dwordx4) into THREE images: two slices (64 KB) in flight while one is multiplied, no
//     staging registers, no ds_write pass.  The DMA writes lane-linear images (8 rows x 8 pieces per wave instruction); the tile
//     kernels' bank swizzle (fast::gt_off) is applied to the SOURCE piece index and again by the transposed operand reads;
//   * stages its results in TWO tiles, so that the stores of slice k - 1 are issued at the top of iteration k, beside the request
//     for slice k + 2, and both travel under the products of slice k: one barrier per slice;
//   * fetches the weights once per workgroup through LDS (mixr_weights) and computes the normaliser's small product (k_wz) with them.
// What it gained is the weight prologue (35 -> 8 us) and the two k_wz launches; the slice loop itself takes the same 170 us in
// every arrangement tried -- DESIGN.md 3d' has the measurements (tools/trace_mixr.py): the launch runs at the chip's power limit.
// vmcnt counts loads and stores in order: slice k's copy is complete when at most the instructions issued after it are
// outstanding (counted per iteration, see the loop).
// Rows past M repeat the last row (their weights are zero), so every reduction step runs unguarded.
// -------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
constexpr int MIXR_DMA_NBUF = 3;   // input images; two staging tiles follow them
constexpr int MIXR_DMA_T = 512;
__host__ __device__ constexpr int sp_mixr_dma_smem() { return (MIXR_DMA_NBUF + 2) * 256 * 64 * 2; }

template <int TRANS>
__global__ __launch_bounds__(MIXR_DMA_T) void k_sp_mixr_dma(const MixrArgs a) {
    constexpr int NW = 8, NB = 2, TE = 64, ROWS = 256, NK = 8, NT = 4, NBUF = MIXR_DMA_NBUF, IMG = ROWS * TE, NTH = MIXR_DMA_T, NP = 4;
    static_assert(NW * NB * 16 == ROWS && NTH == 64 * NW, "eight waves of 32 output blocks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* lds = reinterpret_cast<u16*>(smem_raw);   // [NBUF] input images, then two output staging tiles
    u16* Os = lds + NBUF * IMG;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const long nsl = a.E / TE;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt = (int)min((long)a.spw, a.total - s0);
    if (cnt <= 0) return;
    // B operands of the wave's 32 output blocks (hi + lo), through LDS (mixr_weights; the not yet used images)
    bf16x8 wh[NK][NB], wl[NK][NB];
    static_assert((TRANS ? 64 * (ROWS + 4) : ROWS * 68) * 4 <= NBUF * IMG * 2, "weight chunk must fit in the images");
    mixr_weights<TRANS, NTH, ROWS, NK, NB>(wh, wl, reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, wave * 32, tid);
    // the weights are in, and converted before the first copy is issued (the compiler's own waits for them would otherwise sit
    // behind the prologue's copies and drain them): from here on the outstanding-instruction count is ours
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int n = 0; n < NB; ++n) asm volatile("" : "+v"(wh[ks][n]), "+v"(wl[ks][n]));
    wait_vmcnt<0>();
    auto slice_off = [&](int bh, int es) { return ((long)bh * M * a.es + (long)es * TE) * 2; };   // bytes
    auto advance = [&](int& bh, int& es) { if (++es == (int)nsl) { es = 0; ++bh; } };
    int cbh = (int)(s0 / nsl), ces = (int)(s0 - (long)cbh * nsl), nbh = cbh, nes = ces;
    // The normaliser's small product for every (b, h) whose first slice is this workgroup's: z as bf16 hi + lo in a swizzled tile
    // [256 r][hi s 0..15 | lo s 16..31], the weights already in registers, three products (hi hi + hi lo + lo hi) per step.
    // (Two launches of k_wz -- 30 + 37 us at the 256 x 16 shape for 2 MB of z -- become 3 us here.)
    if (a.zin) {
        u16* Zt = lds;
        const int S = a.S;
        int zbh = cbh + (ces ? 1 : 0);
        for (long first = (long)zbh * nsl; first < s0 + cnt; first += nsl, ++zbh) {
            __syncthreads();
            {
                const int row = tid >> 1, h8 = (tid & 1) * 8;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (row < M && h8 + i < S) ? gld<float>(a.zin + ((long)zbh * M + row) * S + h8 + i) : 0.f;
                uint4 hi, lo;
                split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
                *reinterpret_cast<uint4*>(Zt + fast::gt_off(row, h8)) = hi;
                *reinterpret_cast<uint4*>(Zt + fast::gt_off(row, 16 + h8)) = lo;
            }
            __syncthreads();
            f32x4 za[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) za[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const bf16x8 zh = fast::tr_read8_gt(Zt, ks * 32, 0, lane), zl = fast::tr_read8_gt(Zt, ks * 32, 16, lane);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    za[n] = mfma_bf16(zh, wh[ks][n], za[n]);
                    za[n] = mfma_bf16(zh, wl[ks][n], za[n]);
                    za[n] = mfma_bf16(zl, wh[ks][n], za[n]);
                }
            }
            // lane: s = 4 kg .. + 3 of output block o = 32 wave + 16 n + nl
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int o = wave * 32 + n * 16 + nl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sc = kg * 4 + i;
                    if (o < M && sc < S) a.zout[((long)zbh * M + o) * S + sc] = TRANS ? za[n][i] : 1.f / (a.eps + za[n][i]);
                }
            }
        }
        __syncthreads();
        wait_vmcnt<0>();
    }
    // DMA units of a slice: 32 x (8 rows x 128 bytes); wave w copies rows 8 w + 64 p ..; lane -> row + lane / 8,
    // LDS position lane % 8 <- source piece (lane % 8) ^ swizzle(row)
    unsigned soff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = p * 64 + wave * 8 + (lane >> 3), piece = (lane & 7) ^ ((row ^ (row >> 1)) & 7);
        soff[p] = (unsigned)((long)min(row, M - 1) * a.es * 2 + piece * 16);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int k, long boff) {
        const char* base = reinterpret_cast<const char*>(a.in) + boff;
        const unsigned buf = lds0 + (unsigned)((k % NBUF) * IMG * 2) + (unsigned)(wave * 8 * TE * 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) glds16(base + soff[p], buf + (unsigned)(p * 64 * TE * 2));
    };
    // this thread's four 16-byte store pieces: piece v = tid + p NTH -> row v / 8, column piece v % 8
    unsigned goff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int v = tid + p * NTH, row = v >> 3, c = v & 7;
        goff[p] = (unsigned)((long)min(row, M - 1) * a.es * 2 + c * 16);
    }
    // transposed operand reads: the piece permutation depends on the row's low four bits only, so the addresses of reduction step 0
    // serve every step with a constant offset, and element tile t is tile 0 with bit t of the piece index flipped (offset ^ 16 t)
    const int g = lane >> 4, li = lane & 15;
    const int tr0 = fast::gt_off(g * 8 + (li >> 2), (li & 3) * 4), tr1 = fast::gt_off(g * 8 + (li >> 2) + 4, (li & 3) * 4);
    auto operands = [&](const u16* Th, int ks, bf16x8 (&sv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(Th + ks * 32 * TE + (tr0 ^ (t << 4))));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(Th + ks * 32 * TE + (tr1 ^ (t << 4))));
            fast::s16x8 r8;
            r8[0] = lo4[0]; r8[1] = lo4[1]; r8[2] = lo4[2]; r8[3] = lo4[3];
            r8[4] = hi4[0]; r8[5] = hi4[1]; r8[6] = hi4[2]; r8[7] = hi4[3];
            sv[t] = __builtin_bit_cast(bf16x8, r8);
        }
    };

    // One barrier per slice.  Iteration k: wait for slice k's copy, barrier, THEN store slice k - 1 from its staging tile and request
    // slice k + 2 -- both travel while slice k is multiplied -- and stage slice k's result in the other staging tile.
    //   staging tile (k & 1): written at the end of iteration k, read at the top of iteration k + 1, rewritten in k + 2 (barrier k + 2 between)
    //   image k % 3: refilled by the copy of slice k + 3, issued after barrier k + 1, i.e. after everyone's products of slice k
    // vmcnt (in order): younger than slice k's copy are the stores of slice k - 2 (k >= 2) and the copy of slice k + 1.
    for (int k = 0; k < NBUF - 1 && k < cnt; ++k) {
        issue(k, slice_off(nbh, nes));
        advance(nbh, nes);
    }
    auto store_slice = [&](const u16* St, long off) {
        char* ob = reinterpret_cast<char*>(a.out) + off;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int v = tid + p * NTH, row = v >> 3, c = v & 7;
            const uint4 x = *reinterpret_cast<const uint4*>(St + fast::gt_off(row, c * 8));
            if (row < M) gst<uint4>(ob + goff[p], x);
        }
    };
    long off_prev = 0;
    for (int k = 0; k < cnt; ++k) {
        const long off = slice_off(cbh, ces);
        advance(cbh, ces);
        const int younger = (k + 1 < cnt ? 1 : 0) + (k >= 2 ? 1 : 0);   // in units of NP instructions
        auto stamp = [&](int i) { if (a.trace && tid == 0 && blockIdx.x < 8 && k < 32) a.trace[(blockIdx.x * 32 + k) * 8 + i] = __builtin_amdgcn_s_memtime(); };
        stamp(0);
        if (younger == 2)      wait_vmcnt<2 * NP>();
        else if (younger == 1) wait_vmcnt<NP>();
        else                   wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's staging writes of iteration k - 1)
        stamp(4);
        __builtin_amdgcn_s_barrier();
        stamp(1);
        if (k >= 1) store_slice(Os + ((k - 1) & 1) * IMG, off_prev);
        if (k + NBUF - 1 < cnt) {
            issue(k + NBUF - 1, slice_off(nbh, nes));
            advance(nbh, nes);
        }
        stamp(2);
        off_prev = off;
        const u16* Th = lds + (k % NBUF) * IMG;
        u16* St = Os + (k & 1) * IMG;
        f32x4 acc[NB][NT];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 svA[NT], svB[NT];
        operands(Th, 0, svA);
#pragma unroll
        for (int ks = 0; ks < NK; ks += 2) {
            operands(Th, ks + 1, svB);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[n][t] = mfma_bf16(svA[t], wh[ks][n], acc[n][t]);
#ifndef MIXR_NOLO   // (experiment builds only, tools/build_variant.sh -DMIXR_NOLO=1: without the weights' lo products -- results are wrong)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[n][t] = mfma_bf16(svA[t], wl[ks][n], acc[n][t]);
#endif
            }
            if (ks + 2 < NK) operands(Th, ks + 2, svA);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[n][t] = mfma_bf16(svB[t], wh[ks + 1][n], acc[n][t]);
#ifndef MIXR_NOLO
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[n][t] = mfma_bf16(svB[t], wl[ks + 1][n], acc[n][t]);
#endif
            }
        }
        // lane: elements 16 t + 4 kg .. + 3 of output block 32 wave + 16 n + nl; neighbouring element tiles paired into 16-byte pieces
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                const uint4 pc = fast::pair_pieces(make_uint2(pack_bf16x2(acc[n][t][0], acc[n][t][1]), pack_bf16x2(acc[n][t][2], acc[n][t][3])),
                                                   make_uint2(pack_bf16x2(acc[n][t + 1][0], acc[n][t + 1][1]), pack_bf16x2(acc[n][t + 1][2], acc[n][t + 1][3])));
                *reinterpret_cast<uint4*>(St + fast::gt_off(wave * 32 + n * 16 + nl, (t + (kg & 1)) * 16 + 8 * (kg >> 1))) = pc;
            }
        stamp(3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    store_slice(Os + ((cnt - 1) & 1) * IMG, off_prev);
}

// -------------------------------------------------------------------------------------------------
// 8 fp32 values -> one 16-byte piece of a 16-bit type
__device__ __forceinline__ uint4 pack8_16(bf16_t, f32x4 a, f32x4 b) {
    return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
}
__device__ __forceinline__ uint4 pack8_16(f16_t, f32x4 a, f32x4 b) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    return __builtin_bit_cast(uint4, h);
}
__device__ __forceinline__ uint4 pack8_16(float, f32x4, f32x4) { return make_uint4(0, 0, 0, 0); }   // (never called: fp32 stores 16-byte pieces already)

// Two neighbouring feature tiles of a token in the products' output layout (a lane owns features 16 ct + 4 kg .. + 3 of each) -> the
// token's row.  16-bit tensors on 16-byte aligned rows (`wide`): lane pairs (kg, kg ^ 1) swap halves, a lane then owns 8 consecutive
// features -- one 16-byte piece -- and the four lanes of a token cover 64 contiguous bytes per instruction (the 8-byte pieces of
// the plain layout cost k_sp_out 1.47x its bytes in HBM writes at C2).  EVERY lane must call this (shuffles); `live`: the lane's
// token exists.
template <typename T>
__device__ __forceinline__ void store_tile_pair(T* tok, int ct, int ntiles, int D, int kg, const f32x4& v0, const f32x4& v1, bool live, bool wide) {
    if constexpr (sizeof(T) == 2) {
        if (wide && ct + 1 < ntiles) {   // (uniform)
            const int podd = kg & 1;
            const f32x4 send = podd ? v0 : v1, keep = podd ? v1 : v0;
            f32x4 recv;
#pragma unroll
            for (int i = 0; i < 4; ++i) recv[i] = __shfl_xor(send[i], 16, 64);
            const int f0 = (ct + podd) * 16 + (kg >> 1) * 8;
            if (live && f0 < D) gst<uint4>(tok + f0, pack8_16(T{}, podd ? recv : keep, podd ? keep : recv));
            return;
        }
    }
    if (live) {
        const int d0 = ct * 16 + kg * 4;
        if (d0 < D) Io<T>::st4(tok + d0, v0);
        if (ct + 1 < ntiles && d0 + 16 < D) Io<T>::st4(tok + d0 + 16, v1);
    }
}
template <typename V>
__device__ __forceinline__ bool view16(const V& w) { return (reinterpret_cast<uintptr_t>(w.ptr) & 15) == 0 && ((w.sb | w.sn | w.sh) & 7) == 0; }

constexpr int SP_OUT_T = 512;   // 8 waves share the staged G_i: twice the loads in flight per LDS byte
// FLAT (24-bit summaries, 16-bit tensors): the launch is one persistent workgroup per CU and the token tiles of ALL blocks are one flat list
// cut into gridDim.x equal ranges; a workgroup walks its range in rounds of eight consecutive tiles (one per wave) with the summaries of
// the (at most two) blocks a round touches in two LDS buffers, the next block's pieces requested a round ahead.  A block per workgroup
// quantises twice at the Wan shape -- 14 tiles on 8 waves are two rounds (87.5 %), 1 800 workgroups on 256 CUs are eight waves of
// workgroups for 7.03 (88 %) -- 77 % together; the flat list leaves the last round of a range (98.4 tiles in 13 rounds: 94.6 %).
// OutArgs::nbh = B H; needs at least 8 tiles per block (a round then spans at most two blocks, the next round at most one more).
template <typename T, int DT, typename TO = T, bool EPI = false, bool S16 = Sum16<T>::value, int P24 = 0, bool PRO = false, bool FLAT = false>   // PRO: the q prologue on load (OutArgs::pro_*)
#ifndef SP_OUT_EPI_WAVES
#define SP_OUT_EPI_WAVES 2   // the fused-epilogue variant takes 142 VGPRs: one workgroup per CU without spills (157 us at C4) beats two with 28 spilled registers (163 us)
#endif
__global__ __launch_bounds__(SP_OUT_T, EPI ? SP_OUT_EPI_WAVES : 2) void k_sp_out(const OutArgs a) {
    constexpr int LD = mat_ld<DT>(), KST = Geo<DT>::KST, KP = KST * 32, TILE = KP * LD;
    constexpr bool LO = !std::is_same<T, bf16_t>::value || PRO;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(!FLAT || (P24 == 1 && sizeof(T) == 2 && !S16), "the flat tile list serves 16-bit tensors with 24-bit summaries");
    u16* Gh = reinterpret_cast<u16*>(smem_raw);   // [d1][d2], rows >= D and columns >= D zero   (FLAT: two (hi, lo) pairs, block parity)
    u16* Gl = Gh + TILE;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int S = a.S, D = a.D;
    // the block in hand (FLAT: of the wave's tile of this round)
    int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    long p0 = (long)blk * S;
    const T* qb = (const T*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    TO* ob = (TO*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    // The lane's token row of a 16-token tile as loaded: 8 q features per reduction step, 1 / n, (PRO) the token's rstd.  The first tile's
    // rows are requested BEFORE G_i is staged (one memory round trip per workgroup instead of two: a workgroup of the C2 step has one
    // tile per wave); 16-bit tensors also keep the wave's next tile in flight under the products (8 registers per reduction step).
    // Every load is unconditional, from clamped addresses (rows past the block: its last row; columns past D: the row's first piece).
    constexpr bool DBL = sizeof(T) == 2;
    // (Fetching the token's rope angles with its rows too -- 64 more registers in the fused-epilogue variants, which have them -- was
    // measured and lost: k_sp_out<norm,pro> 189 -> 218 us per Wan layer, the fp32 variant 176 -> 184.)
    struct QRows {
        typename Raw4<T>::type x[KST][2];
        float ninv, rq;
        long row;
    };
    QRows cur, nxt;
    // FLAT: tile g of the flat list = tile g % TPI of block g / TPI (blocks in the order of the summaries: (b, h) major)
    const int TPI = (S + 15) / 16;
    // the gather map's entry for the lane's row of a tile: looked up ONE FETCH AHEAD of the rows it names (unconditional: without a map the
    // load reads the summaries' first word and is dropped) -- looked up inside the fetch it was a dependent round trip in front of every
    // tile's loads, and behind `idx ? idx[p] : p` a branch with a load in it
    auto look = [&](int tt) __attribute__((always_inline)) {
        int fblk = blk;
        if constexpr (FLAT) {
            const int item = tt / TPI;
            tt -= item * TPI;
            fblk = item % a.M;
        }
        const long p = (long)fblk * S + min(tt * 16 + nl, S - 1);
        return gld<int>(a.idx ? a.idx + p : reinterpret_cast<const int*>(a.g));
    };
    auto fetch = [&](int tt, QRows& R, int looked) __attribute__((always_inline)) {   // (FLAT: tt is the tile's place in the flat list)
        int fb = b, fh = h, fbh = bh, fblk = blk;
        if constexpr (FLAT) {
            const int item = tt / TPI;
            tt -= item * TPI;
            fbh = item / a.M; fblk = item - fbh * a.M; fb = fbh / a.H; fh = fbh - fb * a.H;
        }
        const int sv = min(tt * 16 + nl, S - 1);
        R.row = a.idx ? (long)looked : (long)fblk * S + sv;
        const T* qrow = (const T*)a.q.ptr + fb * a.q.sb + fh * a.q.sh + R.row * a.q.sn;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int c = ks * 32 + kg * 8 < D ? ks * 32 + kg * 8 : 0;
            R.x[ks][0] = *reinterpret_cast<const typename Raw4<T>::type*>(qrow + c);
            R.x[ks][1] = *reinterpret_cast<const typename Raw4<T>::type*>(qrow + c + 4);
        }
        R.ninv = gld<float>(a.normalize ? a.ninv + ((long)fbh * a.M + fblk) * S + sv : a.W);
        R.rq = 1.f;
        if constexpr (PRO) R.rq = gld<float>(a.pro_rq ? a.pro_rq + fb * a.pro_n + R.row : a.W);
    };
    // EARLY (the fused-epilogue variants: Wan): the prologue's channel weights and the rope angles of ALL reduction steps are requested
        // first, unconditionally (a table the call does not have is replaced by the summaries, its values dropped).  Fetched where they are
        // used -- behind `if (column < D)`, `if (a.pro_wq)`, `if (a.rcos)` -- every step's pieces were a round trip of their own (hipcc waits
        // for everything in flight where a branch with a load in it joins): ~17 serial L2 round trips per 16-token tile with the norm
        // weights of the epilogue, 12 us of a 13 us tile (tools/isa_waits.py: `L L W1 W0` x 8).
    // (`early` is called for the tile in hand BEFORE the next tile's rows and the next block's summary are requested: the wait for its values
    // then leaves those in flight -- the counter retires in order)
    constexpr bool EARLY = EPI;
    f32x4 pw[(EARLY && PRO) ? KST : 1][2], rcv[EARLY ? KST : 1], rsv[EARLY ? KST : 1];
    auto early = [&]() __attribute__((always_inline)) {
        if constexpr (EARLY) {
            const long row = cur.row;
            const float* standin = a.g;   // (always there, readable well past any offset used below)
            const float* pwb = (PRO && a.pro_wq) ? a.pro_wq + h * D : standin;
            const float* rcb = a.rcos ? a.rcos + row * a.ldr : standin;
            const float* rsb = a.rcos ? a.rsin + row * a.ldr : standin;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const int c = ks * 32 + kg * 8 < D ? ks * 32 + kg * 8 : 0;
                if constexpr (PRO) {
                    pw[ks][0] = gld<f32x4>(pwb + c);
                    pw[ks][1] = gld<f32x4>(pwb + c + 4);
                }
                rcv[ks] = gld<f32x4>(rcb + c / 2);
                rsv[ks] = gld<f32x4>(rsb + c / 2);
            }
        }
    };
    // the products, epilogue and stores of one tile of the block in hand, from `cur`; `live`: the tile exists (every lane runs this: shuffles)
    auto tile = [&](int tt, bool live) __attribute__((always_inline)) {
        const int s = live ? tt * 16 + nl : S, sv = min(s, S - 1);
        const long row = cur.row;
        bf16x8 qh[KST], ql[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if constexpr (EARLY) {   // (no branch: selects)
                const bool in = ks * 32 + kg * 8 < D;
                x0 = raw4_to_f32(T{}, cur.x[ks][0]);
                x1 = raw4_to_f32(T{}, cur.x[ks][1]);
                if constexpr (PRO) {   // relu(q rstd[token] w[channel]) + eps: the values k_qk_prologue used to materialise
                    const float rq = a.pro_rq ? cur.rq : 1.f;
                    x0 *= rq;
                    x1 *= rq;
                    const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                    x0 *= a.pro_wq ? pw[ks][0] : one;
                    x1 *= a.pro_wq ? pw[ks][1] : one;
                }
                if (a.relu) relu8(x0, x1, a.eps);
                if (a.rcos) rope8(x0, x1, rcv[ks], rsv[ks]);   // (uniform; no memory operation inside)
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                x0 = in ? x0 : z4;
                x1 = in ? x1 : z4;
            } else if (ks * 32 + kg * 8 < D) {
                x0 = raw4_to_f32(T{}, cur.x[ks][0]);
                x1 = raw4_to_f32(T{}, cur.x[ks][1]);
                if constexpr (PRO) {   // relu(q rstd[token] w[channel]) + eps: the values k_qk_prologue used to materialise
                    const float rq = a.pro_rq ? cur.rq : 1.f;
                    x0 *= rq;
                    x1 *= rq;
                    if (a.pro_wq) {
                        x0 *= *reinterpret_cast<const f32x4*>(a.pro_wq + h * D + ks * 32 + kg * 8);
                        x1 *= *reinterpret_cast<const f32x4*>(a.pro_wq + h * D + ks * 32 + kg * 8 + 4);
                    }
                }
                if (a.relu) relu8(x0, x1, a.eps);
                if (a.rcos) {
                    const long ro = row * a.ldr + ks * 16 + kg * 4;
                    rope8(x0, x1, *reinterpret_cast<const f32x4*>(a.rcos + ro), *reinterpret_cast<const f32x4*>(a.rsin + ro));
                }
            }
            uint4 hi, lo;
            split8(x0, x1, hi, lo);
            qh[ks] = as_bf16x8(hi);
            ql[ks] = as_bf16x8(lo);
        }
        (void)sv;
        const float ninv = a.normalize ? cur.ninv : 1.f;
        TO* orow = ob + row * a.o.sn + kg * 4;
        f32x4 res[EPI ? DT + 1 : 1];
        // gate values of this lane's features, fetched (packed) before the products: latency off the epilogue.
        // 16-bit activations with 16-byte aligned rows (WIDE): the epilogue runs in a STORE layout -- lane pairs (kg, kg ^ 1)
        // swap halves of two neighbouring feature tiles, so that a lane owns 8 consecutive features (one 16-byte piece) of the
        // token and the four lanes of a token cover 64 contiguous bytes per instruction; gate pieces are read the same way.
        // (In the product layout a lane owns 4 features: 8-byte pieces, 32 bytes per token and instruction -- the gate read of
        // 97 MB cost 70 us and every output line was written by four instructions.)
        constexpr bool WIDE_T = EPI && sizeof(TO) == 2;
        constexpr int NPAIR = DT / 2;
        bool wide = false;
        if constexpr (WIDE_T)
            wide = (((reinterpret_cast<uintptr_t>(a.o.ptr) | reinterpret_cast<uintptr_t>(a.gate.ptr)) & 15) == 0) &&
                   (((a.o.sb | a.o.sn | a.o.sh | (a.gate.ptr ? (a.gate.sb | a.gate.sn | a.gate.sh) : 0)) & 7) == 0);
        const int podd = kg & 1, phalf = (kg >> 1) * 8;   // store layout: tile 2 j + podd, features phalf .. phalf + 7 of it
        bool wide2 = false;   // the plain (no-epilogue) store of 16-bit results in the same layout
        if constexpr (!EPI && sizeof(TO) == 2)
            wide2 = a.skip_out || ((reinterpret_cast<uintptr_t>(a.o.ptr) & 15) == 0 && ((a.o.sb | a.o.sn | a.o.sh) & 7) == 0);
        typename Raw4<TO>::type gv[EPI ? DT : 1];
        uint4 gv8[WIDE_T ? (NPAIR ? NPAIR : 1) : 1];
        f32x4 nwv[WIDE_T ? (NPAIR ? NPAIR : 1) : 1][2];   // the norm weights of the lane's pieces (wide layout), requested with the gate
        if constexpr (WIDE_T) {
            if (wide) {   // (uniform; every load inside is unconditional)
                const TO* gtok = a.gate.ptr ? (const TO*)a.gate.ptr + b * a.gate.sb + row * a.gate.sn + h * a.gate.sh : reinterpret_cast<const TO*>(a.g);
                const float* nwb = a.nw ? a.nw : a.g;
#pragma unroll
                for (int j = 0; j < NPAIR; ++j) {
                    const int f0 = (2 * j + podd) * 16 + phalf, fc = f0 < D ? f0 : 0;
                    gv8[j] = gld<uint4>(gtok + fc);
                    nwv[j][0] = gld<f32x4>(nwb + fc);
                    nwv[j][1] = gld<f32x4>(nwb + fc + 4);
                }
            }
        }
        if constexpr (EPI) {
            if (a.gate.ptr) {
                const TO* gtok = (const TO*)a.gate.ptr + b * a.gate.sb + row * a.gate.sn + h * a.gate.sh;
                if (WIDE_T && wide) {
                    if (DT & 1) {
                        if ((DT - 1) * 16 + kg * 4 < D) gv[DT - 1] = *reinterpret_cast<const typename Raw4<TO>::type*>(gtok + kg * 4 + (DT - 1) * 16);
                    }
                } else {
#pragma unroll
                    for (int ct = 0; ct < DT; ++ct)
                        if (ct * 16 + kg * 4 < D) gv[ct] = *reinterpret_cast<const typename Raw4<TO>::type*>(gtok + kg * 4 + ct * 16);
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < DT; ct += 2) {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const bf16x8 a0h = mat_tr_read8<mat_new<DT>()>(Gh, LD, ks * 32, ct * 16, lane), a1h = mat_tr_read8<mat_new<DT>()>(Gh, LD, ks * 32, ct * 16 + 16, lane);
                c0 = mfma_bf16(a0h, qh[ks], c0);
                c1 = mfma_bf16(a1h, qh[ks], c1);
                if (!S16) {
                    const bf16x8 a0l = mat_tr_read8<mat_new<DT>()>(Gl, LD, ks * 32, ct * 16, lane), a1l = mat_tr_read8<mat_new<DT>()>(Gl, LD, ks * 32, ct * 16 + 16, lane);
                    c0 = mfma_bf16(a0l, qh[ks], c0);
                    c1 = mfma_bf16(a1l, qh[ks], c1);
                }
                if (LO) {
                    c0 = mfma_bf16(a0h, ql[ks], c0);
                    c1 = mfma_bf16(a1h, ql[ks], c1);
                }
            }
            if constexpr (EPI) {
                res[ct] = c0 * ninv;
                res[ct + 1] = c1 * ninv;
            } else {
                const f32x4 o0 = c0 * ninv, o1 = c1 * ninv;
                bool stored = false;
                if constexpr (sizeof(TO) == 2) {
                    if (wide2 && ct + 1 < DT) {   // (uniform) the store layout of the fused epilogue: 16-byte pieces, 64 contiguous bytes per token and instruction
                        const f32x4 send = podd ? o0 : o1, keep = podd ? o1 : o0;
                        f32x4 recv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) recv[i] = __shfl_xor(send[i], 16, 64);   // (every lane takes part)
                        const f32x4 lo4 = podd ? recv : keep, hi4 = podd ? keep : recv;
                        const int f0 = (ct + podd) * 16 + phalf;
                        if (s < S && f0 < D) {
                            if (!a.skip_out) gst<uint4>(ob + row * a.o.sn + f0, pack8_16(TO{}, lo4, hi4));
                            if (a.olo) {   // what the 16-bit store loses, for the backward's row dots (OutArgs::olo)
                                const uint2 r0 = store_residual4<TO>(lo4), r1 = store_residual4<TO>(hi4);
                                gst<uint4>(a.olo + ((long)bh * a.M * S + p0 + s) * D + f0, make_uint4(r0.x, r0.y, r1.x, r1.y));
                            }
                        }
                        stored = true;
                    }
                }
                if (!stored && s < S) {
                    if (!a.skip_out) {
                        if (ct * 16 + kg * 4 < D) Io<TO>::st4(orow + ct * 16, o0);
                        if (ct * 16 + 16 + kg * 4 < D) Io<TO>::st4(orow + ct * 16 + 16, o1);
                    }
                    if constexpr (sizeof(TO) == 2) {
                        if (a.olo) {
                            u16* lo = a.olo + ((long)bh * a.M * S + p0 + s) * D + kg * 4;
                            if (ct * 16 + kg * 4 < D) *reinterpret_cast<uint2*>(lo + ct * 16) = store_residual4<TO>(o0);
                            if (ct * 16 + 16 + kg * 4 < D) *reinterpret_cast<uint2*>(lo + ct * 16 + 16) = store_residual4<TO>(o1);
                        }
                    }
                }
            }
        }
        if constexpr (EPI) {
            float ss = 0.f;
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) {
                if (!std::is_same<TO, float>::value)   // out.to(dtype) before the norm
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        res[ct][i] = std::is_same<TO, bf16_t>::value ? bf16_to_f32(cvt_bf16(res[ct][i])) : (float)(_Float16)res[ct][i];
                if (ct * 16 + kg * 4 < D) ss += res[ct][0] * res[ct][0] + res[ct][1] * res[ct][1] + res[ct][2] * res[ct][2] + res[ct][3] * res[ct][3];
            }
            ss += __shfl_xor(ss, 16, 64);   // the token's features live in the 4 lanes kg = 0..3
            ss += __shfl_xor(ss, 32, 64);
            const float rstd = 1.f / sqrtf(ss / (float)D + a.neps);
            int ct_first = 0;   // tiles below it were stored in the wide layout
            if constexpr (WIDE_T) {
                if (wide) {
                    TO* otok = ob + row * a.o.sn;
#pragma unroll
                    for (int j = 0; j < NPAIR; ++j) {
                        const f32x4 send = podd ? res[2 * j] : res[2 * j + 1], keep = podd ? res[2 * j + 1] : res[2 * j];
                        f32x4 recv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) recv[i] = __shfl_xor(send[i], 16, 64);   // (every lane takes part)
                        const f32x4 lo4 = podd ? recv : keep, hi4 = podd ? keep : recv;
                        const int f0 = (2 * j + podd) * 16 + phalf;
                        if (s < S && f0 < D) {
                            f32x4 y0 = lo4 * rstd, y1 = hi4 * rstd;
                            if (a.nw) {   // (uniform; the values were requested before the products)
                                y0 *= nwv[j][0];
                                y1 *= nwv[j][1];
                            }
                            if (a.gate.ptr) {
                                const f32x4 g0 = raw4_to_f32(TO{}, make_uint2(gv8[j].x, gv8[j].y)), g1 = raw4_to_f32(TO{}, make_uint2(gv8[j].z, gv8[j].w));
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    y0[i] *= g0[i] / (1.f + __expf(-g0[i]));
                                    y1[i] *= g1[i] / (1.f + __expf(-g1[i]));
                                }
                            }
                            gst<uint4>(otok + f0, pack8_16(TO{}, y0, y1));
                        }
                    }
                    ct_first = 2 * NPAIR;
                }
            }
            if (s < S) {
#pragma unroll
                for (int ct = 0; ct < DT; ++ct) {
                    if (ct < ct_first) continue;
                    const int d0 = ct * 16 + kg * 4;
                    if (d0 < D) {
                        f32x4 y = res[ct] * rstd;
                        if (a.nw) y *= *reinterpret_cast<const f32x4*>(a.nw + d0);
                        if (a.gate.ptr) {
                            const f32x4 gf = raw4_to_f32(TO{}, gv[ct]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) y[i] *= gf[i] / (1.f + __expf(-gf[i]));
                        }
                        Io<TO>::st4(orow + ct * 16, y);
                    }
                }
            }
        }
    };
    if constexpr (!FLAT) {
        constexpr int NWV = SP_OUT_T / 64;
        int lk = 0;
        if (a.idx) lk = gld<int>(a.idx + p0 + min(wave * 16 + nl, S - 1));   // (uniform branch; nothing is in flight yet)
        fetch(wave, cur, lk);
        lk = look(wave + NWV);   // (for the next fetch)
        // (S16: G_i stored as bf16: no lo tile)
        stage_mat_split<DT, S16, SP_OUT_T, P24>(Gh, Gl, a.g, ((long)bh * a.M + blk) * a.es, D, tid);
        __syncthreads();
        for (int tt = wave; tt * 16 < S; tt += NWV) {
            if constexpr (DBL) {
                early();
                fetch(tt + NWV, nxt, lk);
                lk = look(tt + 2 * NWV);
            } else {
                if (tt != wave) {   // (uniform)
                    fetch(tt, cur, lk);
                    lk = look(tt + NWV);
                }
                early();
            }
            tile(tt, true);
            if constexpr (DBL) cur = nxt;
        }
    } else {
        constexpr int NWV = SP_OUT_T / 64;
        u16* const G0 = Gh;
        const long total = (long)a.nbh * a.M * TPI;
        const int g0 = (int)(total * blockIdx.x / gridDim.x), g1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
        if (g1 <= g0) return;
        const int nrounds = (g1 - g0 + NWV - 1) / NWV;
        MatStage<DT, SP_OUT_T> stg;
        int staged = g0 / TPI;   // the last block whose summary is in LDS (or on its way)
        int lk = 0;
        if (a.idx) lk = look(min(g0 + wave, g1 - 1));   // (uniform branch; nothing is in flight yet)
        fetch(min(g0 + wave, g1 - 1), cur, lk);
        lk = look(min(g0 + wave + NWV, g1 - 1));   // (for the next fetch)
        {   // the first round's blocks: one, or -- when it already ends in the next block -- two, requested together (one round trip)
            const bool two = min(g0 + NWV - 1, g1 - 1) / TPI > staged;   // (uniform)
            MatStage<DT, SP_OUT_T> stg2;
            stage_mat_issue<DT, SP_OUT_T>(stg, a.g, (long)staged * a.es, D, tid);
            stage_mat_issue<DT, SP_OUT_T>(stg2, a.g, (long)(staged + (two ? 1 : 0)) * a.es, D, tid);
            stage_mat_commit<DT, SP_OUT_T>(G0 + (staged & 1) * 2 * TILE, G0 + (staged & 1) * 2 * TILE + TILE, stg, D, tid);
            if (two) {
                ++staged;
                stage_mat_commit<DT, SP_OUT_T>(G0 + (staged & 1) * 2 * TILE, G0 + (staged & 1) * 2 * TILE + TILE, stg2, D, tid);
            }
        }
        __syncthreads();
        for (int r = 0; r < nrounds; ++r) {
            const int g = g0 + r * NWV + wave, gc = min(g, g1 - 1), item = gc / TPI;
            // the block a later round needs first: requested now, committed behind this round's products (uniform)
            const bool ahead = r + 1 < nrounds && min(g0 + (r + 1) * NWV + NWV - 1, g1 - 1) / TPI > staged;
            bh = item / a.M; blk = item - bh * a.M; b = bh / a.H; h = bh - b * a.H;
            p0 = (long)blk * S;
            qb = (const T*)a.q.ptr + b * a.q.sb + h * a.q.sh;
            ob = (TO*)a.o.ptr + b * a.o.sb + h * a.o.sh;
            Gh = G0 + (item & 1) * 2 * TILE;
            Gl = Gh + TILE;
            early();   // this tile's channel weights and rope angles first, then what is needed later
            fetch(min(g + NWV, g1 - 1), nxt, lk);
            lk = look(min(g + 2 * NWV, g1 - 1));
            stage_mat_issue<DT, SP_OUT_T>(stg, a.g, (long)(staged + (ahead ? 1 : 0)) * a.es, D, tid, ahead);   // (no branch around the loads)
            tile(gc - item * TPI, g < g1);
            cur = nxt;
            if (ahead) {   // (the buffer it goes to held the block before the previous one: this round may still have read it)
                __syncthreads();
                ++staged;
                stage_mat_commit<DT, SP_OUT_T>(G0 + (staged & 1) * 2 * TILE, G0 + (staged & 1) * 2 * TILE + TILE, stg, D, tid);
                __syncthreads();
            }
        }
    }
}


// -------------------------------------------------------------------------------------------------
// k_sp_dw: dwp[bh][split][i][j] = sum_{e in slice} dG_i[e] KV_j[e]  (fp32 operands straight from HBM, split in
// registers; the reduction index is contiguous for both, lane kg takes e = 8 kg .. 8 kg + 7 of every 32).
// grid (tile pairs, bh, E-slices) and output layout as k_dw; each wave reduces a quarter of the slice for the whole
// 64 x 64 tile and the four partial tiles are summed through LDS in wave order.
// -------------------------------------------------------------------------------------------------
constexpr int SP_DW_LD = 68;
constexpr int SP_DW_SMEM = 64 * SP_DW_LD * 4;

// NT: 16-row tiles per side of the output tile (4: 64 x 64; 1 / 2 when M <= 16 / 32, where the full tile would spend 15/16 or 3/4 of
// its loads and MFMAs on clamped rows); UE = 4 / NT reduction steps per loop iteration keep the same number of loads in flight.
// P24: both summary sets are 24-bit floats in two planes per row (the hi plane is the hi operand; the lo operand is rebuilt, p24_lo8)
template <bool S16, int NT = 4, bool P24 = false>
__global__ __launch_bounds__(NTHREADS) void k_sp_dw(const DwArgs a) {
    constexpr int UE = 4 / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Rs = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    // the tile pairs of one (b, h, slice) share their operand rows: consecutive logical indices, which xcd_swizzle keeps on one
    // XCD (one L2) and next to each other in dispatch order
    const int np = gridDim.x, nbh = gridDim.y;
    const int L = fast::xcd_swizzle(blockIdx.x + np * (blockIdx.y + nbh * blockIdx.z), np * nbh * gridDim.z);
    const int unit = L / np, pair = L - unit * np, split = unit / nbh, bh = unit - split * nbh, M = a.M;
    const int it = pair / a.tiles, jt = pair - it * a.tiles;
    const int i0 = it * 64, j0 = jt * 64;
    const long E = a.E;
    const long per = ((E + a.nsplit - 1) / a.nsplit + 127) & ~127L;   // slice: multiple of 4 waves x 32 elements
    const long ebeg = (long)split * per, eend = min(E, ebeg + per);
    const long span = eend > ebeg ? eend - ebeg : 0;
    const long wper = ((span + 3) / 4 + 31) & ~31L;                    // E is a multiple of 32
    const long wbeg = ebeg + wave * wper, wend = min(eend, wbeg + wper);
    const float* xp[NT];
    const float* yp[NT];
    const u16* xp16[NT];
    const u16* yp16[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ri = min(i0 + t * 16 + nl, M - 1), rj = min(j0 + t * 16 + nl, M - 1);
        xp[t] = a.x + ((long)bh * M + ri) * a.es + kg * 8;
        yp[t] = a.y + ((long)bh * M + rj) * a.es + kg * 8;
        xp16[t] = reinterpret_cast<const u16*>(a.x) + ((long)bh * M + ri) * a.es + kg * 8;
        yp16[t] = reinterpret_cast<const u16*>(a.y) + ((long)bh * M + rj) * a.es + kg * 8;
    }
    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long e0 = wbeg; e0 < wend; e0 += 32 * UE) {
        if constexpr (S16) {   // bf16 summaries: operands as stored
            bf16x8 xa[UE][NT], ya[UE][NT];
#pragma unroll
            for (int u = 0; u < UE; ++u) {
                const long e = e0 + 32 * u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    xa[u][t] = ya[u][t] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
                    if (e < wend) {
                        xa[u][t] = *reinterpret_cast<const bf16x8*>(xp16[t] + e);
                        ya[u][t] = *reinterpret_cast<const bf16x8*>(yp16[t] + e);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UE; ++u)
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(xa[u][i], ya[u][j], acc[i][j]);
            continue;
        }
        uint4 xh[UE][NT], xl[UE][NT], yh[UE][NT], yl[UE][NT];
        if constexpr (P24) {   // 8 elements: 16 bytes of the hi plane at 2 e, 8 bytes of the lo plane at 2 E + e
            uint2 xb[UE][NT], yb[UE][NT];
#pragma unroll
            for (int u = 0; u < UE; ++u) {
                const long e = e0 + 32 * u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    xh[u][t] = yh[u][t] = make_uint4(0, 0, 0, 0);
                    xb[u][t] = yb[u][t] = make_uint2(0, 0);
                    if (e < wend) {
                        const char* xr = reinterpret_cast<const char*>(xp[t] - kg * 8);   // (the row's first byte)
                        const char* yr = reinterpret_cast<const char*>(yp[t] - kg * 8);
                        const long ee = e + kg * 8;
                        xh[u][t] = gld<uint4>(xr + 2 * ee);
                        xb[u][t] = gld<uint2>(xr + 2 * E + ee);
                        yh[u][t] = gld<uint4>(yr + 2 * ee);
                        yb[u][t] = gld<uint2>(yr + 2 * E + ee);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UE; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    xl[u][t] = p24_lo8(xh[u][t], xb[u][t]);
                    yl[u][t] = p24_lo8(yh[u][t], yb[u][t]);
                }
        } else {
        f32x4 raw[UE][2 * NT][2];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const long e = e0 + 32 * u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                raw[u][t][0] = raw[u][t][1] = raw[u][NT + t][0] = raw[u][NT + t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < wend) {
                    raw[u][t][0] = *reinterpret_cast<const f32x4*>(xp[t] + e);
                    raw[u][t][1] = *reinterpret_cast<const f32x4*>(xp[t] + e + 4);
                    raw[u][NT + t][0] = *reinterpret_cast<const f32x4*>(yp[t] + e);
                    raw[u][NT + t][1] = *reinterpret_cast<const f32x4*>(yp[t] + e + 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UE; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                split8(raw[u][t][0], raw[u][t][1], xh[u][t], xl[u][t]);
                split8(raw[u][NT + t][0], raw[u][NT + t][1], yh[u][t], yl[u][t]);
            }
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(as_bf16x8(xh[u][i]), as_bf16x8(yh[u][j]), acc[i][j]);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(as_bf16x8(xh[u][i]), as_bf16x8(yl[u][j]), acc[i][j]);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(as_bf16x8(xl[u][i]), as_bf16x8(yh[u][j]), acc[i][j]);
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = Rs + (i * 16 + kg * 4 + r) * SP_DW_LD + j * 16 + nl;
                        *d = (w == 0 ? 0.f : *d) + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float* out = a.out + ((long)bh * a.nsplit + split) * M * M;
    for (int v = tid; v < 64 * 64; v += NTHREADS) {
        const int r = v >> 6, c = v & 63;
        if (i0 + r < M && j0 + c < M) out[(long)(i0 + r) * M + j0 + c] = Rs[r * SP_DW_LD + c];
    }
}

// -------------------------------------------------------------------------------------------------
// k_sp_dwt<P24>: the same partials as k_sp_dw for fp32-grade summaries (fp32 words or 24-bit floats) and more than 128 blocks, with the
// operand rows STAGED THROUGH LDS: k_sp_dw's lanes fetch their MFMA operands straight from memory -- 16 rows x 64 bytes (32 for the lo
// plane) per load instruction -- and ran at 2.1-2.6 TB/s of its bytes (C4: 127 us, the 256 x 16 variant: 490 us).  Here a workgroup walks
// its E-slice in chunks of 64 elements: 64 rows of dG and 64 rows of KV as 256-byte (128 + 64) row pieces, 16 or 8 lanes per row, the
// next chunk in flight in registers while the current one is multiplied from hi / lo tiles ([64][72] bf16 x 4: 37 KB); wave w owns the
// 32 x 32 quadrant (w >> 1, w & 1) of the 64 x 64 output tile, so there is no reduction across waves.  Grid and output as k_sp_dw.
// Used on the 24-bit summaries of the Wan shape (E = 16 384, 9 tile pairs: 124 -> 92 us); on fp32 summaries with E = 4 096 and 16 tile
// pairs (the 256 x 16 variant) its 64 chunks of two barriers and 24 products per wave each ran 773 us against k_sp_dw's 490 -- that
// shape keeps k_sp_dw, whose waves multiply whole 64 x 64 tiles from registers.
// -------------------------------------------------------------------------------------------------
constexpr int SP_DWT_LD = 72, SP_DWT_SMEM = 4 * 64 * SP_DWT_LD * 2;
template <bool P24>
__global__ __launch_bounds__(NTHREADS, 2) void k_sp_dwt(const DwArgs a) {
    constexpr int LD = SP_DWT_LD, CE = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xh = reinterpret_cast<u16*>(smem_raw);
    u16* Xl = Xh + 64 * LD;
    u16* Yh = Xl + 64 * LD;
    u16* Yl = Yh + 64 * LD;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int np = gridDim.x, nbh = gridDim.y;
    const int L = fast::xcd_swizzle(blockIdx.x + np * (blockIdx.y + nbh * blockIdx.z), np * nbh * gridDim.z);
    const int unit = L / np, pair = L - unit * np, split = unit / nbh, bh = unit - split * nbh, M = a.M;
    const int it = pair / a.tiles, jt = pair - it * a.tiles;
    const int i0 = it * 64, j0 = jt * 64;
    const long E = a.E;
    const long per = ((E + a.nsplit - 1) / a.nsplit + CE - 1) / CE * CE;   // slice: whole chunks (E is a multiple of 64)
    const long ebeg = (long)split * per, eend = min(E, ebeg + per);
    // the thread's two units per operand: unit v = tid + 256 p -> row v / 8, elements 8 (v % 8) .. + 7 of the chunk
    const char* xr[2];
    const char* yr[2];
    int urow[2], ucol[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int v = tid + p * NTHREADS;
        urow[p] = v >> 3;
        ucol[p] = (v & 7) * 8;
        xr[p] = reinterpret_cast<const char*>(a.x + ((long)bh * M + min(i0 + urow[p], M - 1)) * a.es);
        yr[p] = reinterpret_cast<const char*>(a.y + ((long)bh * M + min(j0 + urow[p], M - 1)) * a.es);
    }
    uint4 xa[2], xb[2], ya[2], yb[2];   // fp32: the unit's 8 floats; P24: xa = hi piece, xb.xy = lo piece
    auto issue = [&](long e0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long e = e0 + ucol[p];
            if constexpr (P24) {
                xa[p] = gld_stream16(xr[p] + 2 * e);
                ya[p] = gld_stream16(yr[p] + 2 * e);
                const uint2 lx = gld<uint2>(xr[p] + 2 * E + e), ly = gld<uint2>(yr[p] + 2 * E + e);
                xb[p] = make_uint4(lx.x, lx.y, 0u, 0u);
                yb[p] = make_uint4(ly.x, ly.y, 0u, 0u);
            } else {
                xa[p] = gld_stream16(xr[p] + 4 * e);
                xb[p] = gld_stream16(xr[p] + 4 * e + 16);
                ya[p] = gld_stream16(yr[p] + 4 * e);
                yb[p] = gld_stream16(yr[p] + 4 * e + 16);
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool okx = i0 + urow[p] < M, oky = j0 + urow[p] < M;   // rows past the last block: zeros
            const int off = urow[p] * LD + ucol[p];
            uint4 hx, lx, hy, ly;
            if constexpr (P24) {
                hx = xa[p];
                lx = p24_lo8(xa[p], make_uint2(xb[p].x, xb[p].y));
                hy = ya[p];
                ly = p24_lo8(ya[p], make_uint2(yb[p].x, yb[p].y));
            } else {
                split8(__builtin_bit_cast(f32x4, xa[p]), __builtin_bit_cast(f32x4, xb[p]), hx, lx);
                split8(__builtin_bit_cast(f32x4, ya[p]), __builtin_bit_cast(f32x4, yb[p]), hy, ly);
            }
            // (component-wise: a select of whole uint4 structs goes through the stack)
            auto sel = [](bool ok, const uint4& v) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); };
            *reinterpret_cast<uint4*>(Xh + off) = sel(okx, hx);
            *reinterpret_cast<uint4*>(Xl + off) = sel(okx, lx);
            *reinterpret_cast<uint4*>(Yh + off) = sel(oky, hy);
            *reinterpret_cast<uint4*>(Yl + off) = sel(oky, ly);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wi = wave >> 1, wj = wave & 1;
    if (ebeg < eend) issue(ebeg);
    for (long e0 = ebeg; e0 < eend; e0 += CE) {
        commit();
        __syncthreads();
        if (e0 + CE < eend) issue(e0 + CE);
#pragma unroll
        for (int ks = 0; ks < CE / 32; ++ks) {
            bf16x8 ah[2], al[2], bh_[2], bl_[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = row_read8(Xh, LD, (2 * wi + t) * 16, ks * 32, lane);
                al[t] = row_read8(Xl, LD, (2 * wi + t) * 16, ks * 32, lane);
                bh_[t] = row_read8(Yh, LD, (2 * wj + t) * 16, ks * 32, lane);
                bl_[t] = row_read8(Yl, LD, (2 * wj + t) * 16, ks * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mfma_bf16(ah[i], bh_[j], acc[i][j]);
                    acc[i][j] = mfma_bf16(ah[i], bl_[j], acc[i][j]);
                    acc[i][j] = mfma_bf16(al[i], bh_[j], acc[i][j]);
                }
        }
        __syncthreads();
    }
    // C layout: rows i = 32 wi + 16 ti + 4 kg + r, column j = 32 wj + 16 tj + nl
    float* out = a.out + ((long)bh * a.nsplit + split) * M * M;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 32 * wi + 16 * ti + 4 * kg + r, j = j0 + 32 * wj + 16 * tj + nl;
                if (i < M && j < M) out[(long)i * M + j] = acc[ti][tj][r];
            }
}

// -------------------------------------------------------------------------------------------------
// Token gradients.  Both kernels keep one D x D matrix of the block in LDS as bf16 hi / lo and compute transposed
// products, so that the token tensors are MFMA B operands read straight from HBM (8 consecutive features per lane) and a
// lane ends up with 4 consecutive features of one token.
//   k_sp_bwd_dq : dQ_i = (dO_i / n_i) G_i^T (+ dz_i ksum_i^T) ; dksum_i = Qden_i^T dz_i ; dQden_i = dz_i ksum_i^T (split)
//   k_sp_bwd_dkv: dK_j = V_j dKV_j^T (+ dksum_j) ; dV_j = K_j dKV_j ; dKden_j = 1 dksum_j^T (split)
// -------------------------------------------------------------------------------------------------
template <int DT, bool S16 = false>
__host__ __device__ constexpr int sp_tok_smem() {
    // (+ k_sp_bwd_dkv's row staging for 16-bit tensors with rows of up to 128 bytes: [4 waves][dK, dV][16 tokens][DW + 8] 16-bit values)
    return sp_out_smem<DT, S16>() + Geo<DT>::DW * 4 * 4 + Geo<DT>::DW * 4 + (DT <= 4 ? 4 * 2 * 16 * (Geo<DT>::DW + 8) * 2 : 0);
}

// A operand with the reduction index along the rows' columns: A[m][k] = T[c0 + m][k0 + 8 kg .. + 7]
__device__ __forceinline__ bf16x8 row_read8(const u16* tile, int ld, int c0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (c0 + (lane & 15)) * ld + k0 + (lane >> 4) * 8);
}

// ROPE: the numerator pair was rotated inside the forward (q_den aliases the un-rotated q): dQ_rot is turned back by the
// transposed rotation before dz ksum^T is added, so the one stored tensor is the gradient of the un-rotated q
// WQ: q_den enters only dksum (normaliser on, no relu prologue) and is fetched in 16-byte pieces in the operand layout; otherwise in
// the output layout, where the relu gradient mask needs it (a template flag: both register sets at once cost the occupancy)
template <typename T, int DT, bool ROPE = false, bool S16 = Sum16<T>::value, bool WQ = false, int P24 = 0>
__global__ __launch_bounds__(NTHREADS, (DT <= 4 && WQ) ? 4 : 2) void k_sp_bwd_dq(const TokArgs a) {
    constexpr int LD = mat_ld<DT>(), DW = Geo<DT>::DW, KST = Geo<DT>::KST, TILE = KST * 32 * LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gh = reinterpret_cast<u16*>(smem_raw);
    u16* Gl = Gh + TILE;                                  // not allocated for bf16 summaries
    float* dksw = reinterpret_cast<float*>(Gh + (S16 ? 1 : 2) * TILE);   // [4 waves][DW]
    float* ksum = dksw + 4 * DW;                          // [DW]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, S = a.S, D = a.D, M = a.M;
    const long p0 = (long)blk * S;
    auto base = [&](const View& w) { return (const T*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (T*)w.ptr + b * w.sb + h * w.sh; };
    const T *gb = base(a.dout), *qdb = base(a.qd);
    T *dqb = mbase(a.dq), *dqdb = a.split ? mbase(a.dqd) : nullptr;
    const float* ninvb = a.ninv + ((long)bh * M + blk) * S;
    const float* dzb = a.dz + ((long)bh * M + blk) * S;
    constexpr bool wide_q = WQ;   // (the launch picks WQ = normalize && !relu)
    const bool wide_dq = view16(a.dq), wide_dqd = a.split && view16(a.dqd);   // (uniform) 16-byte store layout

    // one 16-token tile of this wave: the lane's token row, its dO features as the MFMA B operand (8 per reduction step) and
    // its q_den features in output layout (4 per feature tile); fetched one tile ahead of the tile being computed
    // (the rows AS LOADED, every load unconditional from a clamped address, nothing touched before the tile that consumes it, the gather
    // map looked up one fetch ahead: see k_sp_bwd_dkv)
    struct Rows {
        typename Raw4<T>::type g[KST][2];
        typename Raw4<T>::type qd[WQ ? 1 : DT];         // q_den in output layout (relu prologue: the gradient mask needs it there)
        typename Raw4<T>::type qw[WQ ? KST : 1][2];     // ... or in the operand layout of g (8 features per reduction step: 16-byte loads), for dksum only
        float ninv, dz;
        long row;
        bool live;
    } cur, nxt;
    const float* standin = a.g;
    auto look = [&](int tt) { return gld<int>(a.idx ? a.idx + p0 + min(tt * 16 + nl, S - 1) : reinterpret_cast<const int*>(a.g)); };
    auto fetch = [&](int tt, Rows& R, int looked) __attribute__((always_inline)) {
        const int s = tt * 16 + nl, sv = min(s, S - 1);
        R.live = s < S;
        R.row = a.idx ? (long)looked : p0 + sv;
        R.ninv = gld<float>(a.normalize ? ninvb + sv : standin);   // (1 / 0 are selected where the values are used)
        R.dz = gld<float>(a.normalize ? dzb + sv : standin);
        const T* grow = gb + R.row * a.dout.sn;
        const T* qrow = qdb + R.row * a.qd.sn;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int c = ks * 32 + kg * 8 < D ? ks * 32 + kg * 8 : 0;
            R.g[ks][0] = gld<typename Raw4<T>::type>(grow + c);
            R.g[ks][1] = gld<typename Raw4<T>::type>(grow + c + 4);
        }
        if constexpr (!WQ) {
#pragma unroll
            for (int ct = 0; ct < DT; ++ct) R.qd[ct] = gld<typename Raw4<T>::type>(qrow + (ct * 16 + kg * 4 < D ? ct * 16 + kg * 4 : 0));
        }
        // no relu prologue: q_den is needed for dksum = sum_s dz[s] q[s][:] alone -- 8 features per load, as the dO rows (the 4-feature
        // pieces of the output layout are 8-byte loads for 16-bit tensors: 0.54-0.70 of the 16-byte rate)
        if constexpr (WQ) {
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const int c = ks * 32 + kg * 8 < D ? ks * 32 + kg * 8 : 0;
                R.qw[ks][0] = gld<typename Raw4<T>::type>(qrow + c);
                R.qw[ks][1] = gld<typename Raw4<T>::type>(qrow + c + 4);
            }
        }
    };
    int lk = 0;
    if (a.idx) lk = gld<int>(a.idx + p0 + min(wave * 16 + nl, S - 1));   // (uniform branch; nothing is in flight yet)
    fetch(wave, cur, lk);   // in flight while G_i is staged
    lk = look(wave + 4);
    const float ksum_v = gld<float>(a.normalize ? a.ksum + ((long)bh * M + blk) * D + min(tid, D - 1) : standin);
    stage_mat_split<DT, S16, NTHREADS, P24>(Gh, Gl, a.g, ((long)bh * M + blk) * a.es, D, tid);
    if (tid < DW) ksum[tid] = (a.normalize && tid < D) ? ksum_v : 0.f;
    __syncthreads();
    f32x4 dksp[WQ ? 1 : DT], dksw8[WQ ? KST : 1][2];   // per-lane dksum partials: output layout / operand layout (WQ)
#pragma unroll
    for (int ct = 0; ct < (WQ ? 1 : DT); ++ct) dksp[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < (WQ ? KST : 1); ++ks) dksw8[ks][0] = dksw8[ks][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto tile = [&]() __attribute__((always_inline)) {
        const float cninv = a.normalize ? cur.ninv : 1.f, cdz = (a.normalize && cur.live) ? cur.dz : 0.f;
        bf16x8 gh[KST], gl[KST];   // dO / n : B[k = d2][n = s]
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            uint4 hi, lo;
            const bool in = ks * 32 + kg * 8 < D;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            split8((in ? raw4_to_f32(T{}, cur.g[ks][0]) : z4) * cninv, (in ? raw4_to_f32(T{}, cur.g[ks][1]) : z4) * cninv, hi, lo);
            gh[ks] = as_bf16x8(hi);
            gl[ks] = as_bf16x8(lo);
        }
        // gradient of one feature tile in output layout (4 features of the lane's token) and its q_den companion
        auto epilogue = [&](int ct, f32x4& c, f32x4& cd, const f32x2& rc, const f32x2& rs) {
            const int d0 = ct * 16 + kg * 4;
            if constexpr (ROPE) unrope4(c, rc, rs);
            f32x4 qd = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!WQ) {
                if ((a.normalize || a.relu) && d0 < D) qd = raw4_to_f32(T{}, cur.qd[ct]);   // (uniform / lane select; no memory operation)
            }
            if (a.relu)
#pragma unroll
                for (int i = 0; i < 4; ++i) qd[i] = fmaxf(qd[i], 0.f) + a.eps;
            const f32x4 ks4 = *reinterpret_cast<const f32x4*>(ksum + d0);
            cd = cdz * ks4;
            if (a.normalize) {
                if constexpr (!WQ) dksp[ct] += cdz * qd;
                if (!a.split) c += cd;
            }
            if (a.relu)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(qd[i] > a.eps)) c[i] = 0.f;
        };
#pragma unroll
        for (int ct = 0; ct < DT; ct += 2) {   // two feature tiles at a time: independent MFMA chains, adjacent stores
            constexpr int LAST = DT - 1;
            const int c1t = ct + 1 <= LAST ? ct + 1 : LAST;
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
            f32x2 rc0 = {1.f, 1.f}, rs0 = {0.f, 0.f}, rc1 = rc0, rs1 = rs0;   // the angles of the two tiles' features: in flight during the products
            if constexpr (ROPE) {
                const long ro = cur.row * a.ldr + kg * 2;
                if (ct * 16 + kg * 4 < D) { rc0 = *reinterpret_cast<const f32x2*>(a.rcos + ro + ct * 8); rs0 = *reinterpret_cast<const f32x2*>(a.rsin + ro + ct * 8); }
                if (c1t * 16 + kg * 4 < D) { rc1 = *reinterpret_cast<const f32x2*>(a.rcos + ro + c1t * 8); rs1 = *reinterpret_cast<const f32x2*>(a.rsin + ro + c1t * 8); }
            }
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const bf16x8 a0h = mat_row_read8<mat_new<DT>()>(Gh, LD, ct * 16, ks * 32, lane), a1h = mat_row_read8<mat_new<DT>()>(Gh, LD, c1t * 16, ks * 32, lane);
                c0 = mfma_bf16(a0h, gh[ks], c0);
                c1 = mfma_bf16(a1h, gh[ks], c1);
                if (!S16) {
                    const bf16x8 a0l = mat_row_read8<mat_new<DT>()>(Gl, LD, ct * 16, ks * 32, lane), a1l = mat_row_read8<mat_new<DT>()>(Gl, LD, c1t * 16, ks * 32, lane);
                    c0 = mfma_bf16(a0l, gh[ks], c0);
                    c1 = mfma_bf16(a1l, gh[ks], c1);
                }
                c0 = mfma_bf16(a0h, gl[ks], c0);
                c1 = mfma_bf16(a1h, gl[ks], c1);
            }
            f32x4 e0, e1 = {0.f, 0.f, 0.f, 0.f};
            epilogue(ct, c0, e0, rc0, rs0);
            if (ct + 1 < DT) epilogue(ct + 1, c1, e1, rc1, rs1);
            // the two 64-byte halves of a 128-byte line go out in consecutive store instructions (write combining):
            // half-line stores separated in time cost a fill read and a second write per line
            store_tile_pair<T>(dqb + cur.row * a.dq.sn, ct, DT, D, kg, c0, c1, cur.live, wide_dq);
            if (a.normalize && a.split) store_tile_pair<T>(dqdb + cur.row * a.dqd.sn, ct, DT, D, kg, e0, e1, cur.live, wide_dqd);
            __builtin_amdgcn_sched_barrier(0);   // keep the LDS operand reads of later tiles from being hoisted (register pressure)
        }
        if constexpr (WQ) {
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const bool in = ks * 32 + kg * 8 < D;
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                dksw8[ks][0] += cdz * (in ? raw4_to_f32(T{}, cur.qw[ks][0]) : z4);
                dksw8[ks][1] += cdz * (in ? raw4_to_f32(T{}, cur.qw[ks][1]) : z4);
            }
        }
    };
    for (int tt = wave; tt * 16 < S; tt += 4) {
        fetch((tt + 4) * 16 < S ? tt + 4 : tt, nxt, lk);   // (unconditional; the wave's last tile: itself again -- lines it has just read)
        lk = look(tt + 8);
        tile();
        cur = nxt;
    }
    if (a.normalize) {   // dksum[d] = sum_s dz[s] qden[s][d]: over the 16 token lanes, then over the waves
        if constexpr (WQ) {   // the lane's features 32 ks + 8 kg + 0..7
#pragma unroll
            for (int ks = 0; ks < KST; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = row16_sum(dksw8[ks][i >> 2][i & 3]);
                    if (nl == 0 && ks * 32 + kg * 8 + i < DW) dksw[wave * DW + ks * 32 + kg * 8 + i] = v;
                }
        } else {
#pragma unroll
            for (int ct = 0; ct < (WQ ? 1 : DT); ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = row16_sum(dksp[ct][i]);
                    if (nl == 0) dksw[wave * DW + ct * 16 + kg * 4 + i] = v;
                }
        }
        __syncthreads();
        if (tid < D) a.dks[((long)bh * M + blk) * D + tid] = dksw[tid] + dksw[DW + tid] + dksw[2 * DW + tid] + dksw[3 * DW + tid];
    }
}

// ROPE: k is the un-rotated tensor: it is rotated on its way into the dV product (KV was formed from the rotated keys), and
// dK_rot is turned back before dksum is added
template <typename T, int DT, bool ROPE = false, bool S16 = Sum16<T>::value, int P24 = 0>
__global__ __launch_bounds__(NTHREADS, 2) void k_sp_bwd_dkv(const TokArgs a) {   // (a bound of 4 -- 128 VGPRs, four workgroups per CU at D <= 64 -- measured: C2 +-0, blocks of 256 tokens 85 -> 94 us)
    constexpr int LD = mat_ld<DT>(), DW = Geo<DT>::DW, KST = Geo<DT>::KST, TILE = KST * 32 * LD;
    constexpr bool LO = !std::is_same<T, bf16_t>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gh = reinterpret_cast<u16*>(smem_raw);   // dKV_j [d1][d2]
    u16* Gl = Gh + TILE;                          // not allocated for bf16 summaries
    float* dks = reinterpret_cast<float*>(Gh + (S16 ? 1 : 2) * TILE);   // [DW]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, S = a.S, D = a.D, M = a.M;
    const long p0 = (long)blk * S;
    auto base = [&](const View& w) { return (const T*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (T*)w.ptr + b * w.sb + h * w.sh; };
    const T *kb = base(a.k), *vb = base(a.v);
    T *dkb = mbase(a.dk), *dvb = mbase(a.dv), *dkdb = a.split ? mbase(a.dkd) : nullptr;

    // A tile's rows AS LOADED (converted, masked and relu'd where they are used): every load of a fetch is unconditional, from clamped
    // addresses, and nothing touches a loaded value before the tile that consumes it.  The fetch used to convert on arrival, sit behind
    // `if (column < D)` / `if (a.relu)` and look its gather-map entry up itself: in the listing every piece was `load, s_waitcnt vmcnt(0)`
    // -- ten serial round trips in front of the first tile and seven per prefetch (tools/isa_waits.py).
    struct Rows {
        typename Raw4<T>::type v[KST][2], k[KST][2];   // B operands: 8 features per reduction step
        typename Raw4<T>::type km[ROPE ? 1 : DT];      // k in output layout (gradient mask of the relu prologue; never combined with the rotary one)
        f32x4 rc[ROPE ? KST : 1], rs[ROPE ? KST : 1];   // angles of the B-operand features (4 pairs per reduction step)
        long row;
        bool live;
    } cur, nxt;
    // the gather map's entry of the lane's row, looked up one fetch ahead (without a map: the summaries' first word, dropped)
    // (a wave whose tile is the block's last prefetches ITS OWN tile again -- lines it has just read; re-reading the block's last tile cost
    // the L2 a second pass over K and V, and a stand-in address selected by `has a next tile` became a branch around the loads.  A loop-free
    // second copy of the tile body for blocks of one tile per wave cost 30 registers and the fourth workgroup per CU)
    auto look = [&](int tt) { return gld<int>(a.idx ? a.idx + p0 + min(tt * 16 + nl, S - 1) : reinterpret_cast<const int*>(a.dkv)); };
    auto fetch = [&](int tt, Rows& R, int looked) __attribute__((always_inline)) {
        const int s = tt * 16 + nl, sv = min(s, S - 1);
        R.live = s < S;
        R.row = a.idx ? (long)looked : p0 + sv;
        const T* vrow = vb + R.row * a.v.sn;
        const T* krow = kb + R.row * a.k.sn;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const int c = ks * 32 + kg * 8 < D ? ks * 32 + kg * 8 : 0;
            R.v[ks][0] = gld<typename Raw4<T>::type>(vrow + c);
            R.v[ks][1] = gld<typename Raw4<T>::type>(vrow + c + 4);
            R.k[ks][0] = gld<typename Raw4<T>::type>(krow + c);
            R.k[ks][1] = gld<typename Raw4<T>::type>(krow + c + 4);
            if constexpr (ROPE) {   // unconditional, clamped: features past D are zeros whatever their angle (a branch or a select
                const int co = min(ks * 16 + kg * 4, D / 2 - 4);   // of whole vectors here puts the arrays on the stack)
                R.rc[ks] = *reinterpret_cast<const f32x4*>(a.rcos + R.row * a.ldr + co);
                R.rs[ks] = *reinterpret_cast<const f32x4*>(a.rsin + R.row * a.ldr + co);
            }
        }
        if constexpr (!ROPE) {   // (pieces of lines the operand loads fetch anyway; used under the relu prologue only)
            if (DT <= 4 || a.relu) {   // (head dims above 64: 32 registers of fp32 pieces -- only when they are used)
#pragma unroll
                for (int ct = 0; ct < DT; ++ct) R.km[ct] = gld<typename Raw4<T>::type>(krow + (ct * 16 + kg * 4 < D ? ct * 16 + kg * 4 : 0));
            }
        }
    };
    const bool wide_dk = view16(a.dk), wide_dv = view16(a.dv);   // (uniform) 16-byte store layout
    int lk = 0;
    if (a.idx) lk = gld<int>(a.idx + p0 + min(wave * 16 + nl, S - 1));   // (uniform branch; nothing is in flight yet)
    fetch(wave, cur, lk);
    lk = look(wave + 4);
    const float dks_v = gld<float>(a.normalize ? a.dks + ((long)bh * M + blk) * D + min(tid, D - 1) : reinterpret_cast<const float*>(a.dkv));
    stage_mat_split<DT, S16, NTHREADS, P24>(Gh, Gl, a.dkv, ((long)bh * M + blk) * a.es, D, tid);
    if (tid < DW) dks[tid] = (a.normalize && tid < D) ? dks_v : 0.f;
    __syncthreads();

    auto tile = [&]() __attribute__((always_inline)) {
        bf16x8 vh[KST], vl[KST], kh[KST], kl[KST];
        f32x4 dkst[2], dvst[2];
        // ROWST (16-bit tensors, rows of up to 128 bytes): the wave's dK and dV tiles go through a wave-private LDS strip and leave as WHOLE
        // rows -- 8 lanes x 16 bytes per token, eight rows per store instruction.  In the product layout a store covered 16 half rows (64
        // bytes each) of two alternating output streams, and the L2 wrote a quarter of the lines twice (WRITE_SIZE 165 MB for 134 MB of
        // results at C2, with the fill reads on top).  (Keeping the tiles in registers and storing a row's pieces back to back behind the
        // loop -- no LDS -- was measured first: 79.6 -> 85.3 us, the stores start later.)
        constexpr bool ROWST = sizeof(T) == 2 && DT <= 4;
        constexpr int SLD = DW + 8;
        u16* stk = reinterpret_cast<u16*>(dks + DW) + wave * 2 * 16 * SLD;   // [dK, dV][16][SLD]
        constexpr bool LATE = false;
        f32x4 dkall[LATE ? DT : 1], dvall[LATE ? DT : 1];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            uint4 hi, lo;
            const bool in = ks * 32 + kg * 8 < D;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            split8(in ? raw4_to_f32(T{}, cur.v[ks][0]) : z4, in ? raw4_to_f32(T{}, cur.v[ks][1]) : z4, hi, lo);
            vh[ks] = as_bf16x8(hi);
            vl[ks] = as_bf16x8(lo);
            f32x4 y0 = in ? raw4_to_f32(T{}, cur.k[ks][0]) : z4, y1 = in ? raw4_to_f32(T{}, cur.k[ks][1]) : z4;
            if constexpr (ROPE) rope8(y0, y1, cur.rc[ks], cur.rs[ks]);
            else if (a.relu && in) relu8(y0, y1, a.eps);
            split8(y0, y1, hi, lo);
            kh[ks] = as_bf16x8(hi);
            kl[ks] = as_bf16x8(lo);
        }
#pragma unroll
        for (int ct = 0; ct < DT; ++ct) {
            f32x4 ck = {0.f, 0.f, 0.f, 0.f}, cv = ck;
            f32x2 urc = {1.f, 1.f}, urs = {0.f, 0.f};
            if constexpr (ROPE) {
                if (ct * 16 + kg * 4 < D) {
                    urc = *reinterpret_cast<const f32x2*>(a.rcos + cur.row * a.ldr + ct * 8 + kg * 2);
                    urs = *reinterpret_cast<const f32x2*>(a.rsin + cur.row * a.ldr + ct * 8 + kg * 2);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                // dK^T[d1][s] = sum_d2 dKV[d1][d2] V[s][d2]
                const bf16x8 ah = mat_row_read8<mat_new<DT>()>(Gh, LD, ct * 16, ks * 32, lane);
                // dV^T[d2][s] = sum_d1 dKV[d1][d2] K[s][d1]
                const bf16x8 th = mat_tr_read8<mat_new<DT>()>(Gh, LD, ks * 32, ct * 16, lane);
                ck = mfma_bf16(ah, vh[ks], ck);
                cv = mfma_bf16(th, kh[ks], cv);
                if (!S16) {
                    const bf16x8 al = mat_row_read8<mat_new<DT>()>(Gl, LD, ct * 16, ks * 32, lane), tl = mat_tr_read8<mat_new<DT>()>(Gl, LD, ks * 32, ct * 16, lane);
                    ck = mfma_bf16(al, vh[ks], ck);
                    cv = mfma_bf16(tl, kh[ks], cv);
                }
                if (LO) {
                    ck = mfma_bf16(ah, vl[ks], ck);
                    cv = mfma_bf16(th, kl[ks], cv);
                }
            }
            const int d0 = ct * 16 + kg * 4;
            if constexpr (ROPE) unrope4(ck, urc, urs);
            if (a.normalize && !a.split) ck += *reinterpret_cast<const f32x4*>(dks + d0);
            if constexpr (!ROPE) {
                if (a.relu && ct * 16 + kg * 4 < D) {
                    const f32x4 kmf = raw4_to_f32(T{}, cur.km[ct]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (!(fmaxf(kmf[i], 0.f) + a.eps > a.eps)) ck[i] = 0.f;
                }
            }
            if constexpr (ROWST) {
                const uint2 pk = std::is_same<T, bf16_t>::value ? make_uint2(pack_bf16x2(ck[0], ck[1]), pack_bf16x2(ck[2], ck[3])) : make_uint2(h16_pack2(ck[0], ck[1]), h16_pack2(ck[2], ck[3]));
                const uint2 pv = std::is_same<T, bf16_t>::value ? make_uint2(pack_bf16x2(cv[0], cv[1]), pack_bf16x2(cv[2], cv[3])) : make_uint2(h16_pack2(cv[0], cv[1]), h16_pack2(cv[2], cv[3]));
                *reinterpret_cast<uint2*>(stk + nl * SLD + ct * 16 + kg * 4) = pk;
                *reinterpret_cast<uint2*>(stk + 16 * SLD + nl * SLD + ct * 16 + kg * 4) = pv;
            } else if constexpr (LATE) {
                dkall[ct] = ck;
                dvall[ct] = cv;
            } else {
            dkst[ct & 1] = ck;
            dvst[ct & 1] = cv;
            if ((ct & 1) || ct == DT - 1) {   // store pairs of feature tiles: whole 128-byte lines (fp32) / 16-byte pieces (16-bit tensors)
                const int cta = ct & ~1, nt2 = (ct & 1) ? DT : cta + 1;   // (a lone last tile: no partner)
                store_tile_pair<T>(dkb + cur.row * a.dk.sn, cta, nt2, D, kg, dkst[0], dkst[1], cur.live, wide_dk);
                store_tile_pair<T>(dvb + cur.row * a.dv.sn, cta, nt2, D, kg, dvst[0], dvst[1], cur.live, wide_dv);
                if (a.normalize && a.split && cur.live) {
                    const int da = cta * 16 + kg * 4, db = da + 16;
                    const bool sa = da < D, sb = (ct & 1) && db < D;
                    if (sa) Io<T>::st4(dkdb + cur.row * a.dkd.sn + da, *reinterpret_cast<const f32x4*>(dks + da));
                    if (sb) Io<T>::st4(dkdb + cur.row * a.dkd.sn + db, *reinterpret_cast<const f32x4*>(dks + db));
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ROWST) {
            __builtin_amdgcn_wave_barrier();
            const bool wide = wide_dk && wide_dv;   // (uniform) 16-byte aligned rows: whole-row stores; otherwise 8-byte pieces from the strip
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int r = (lane >> 3) + 8 * ps, c = (lane & 7) * 8;   // token r of the tile, features c .. c + 7
                const long grow = __shfl(cur.row, r, 64);                 // (lanes 0 .. 15 hold the rows of tokens 0 .. 15)
                const bool glive = __shfl((int)cur.live, r, 64) != 0;
                if (wide) {
                    if (glive && c < D) {
                        gst<uint4>(dkb + grow * a.dk.sn + c, *reinterpret_cast<const uint4*>(stk + r * SLD + c));
                        gst<uint4>(dvb + grow * a.dv.sn + c, *reinterpret_cast<const uint4*>(stk + 16 * SLD + r * SLD + c));
                    }
                } else if (glive) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        if (c + 4 * hf < D) {
                            *reinterpret_cast<uint2*>(dkb + grow * a.dk.sn + c + 4 * hf) = *reinterpret_cast<const uint2*>(stk + r * SLD + c + 4 * hf);
                            *reinterpret_cast<uint2*>(dvb + grow * a.dv.sn + c + 4 * hf) = *reinterpret_cast<const uint2*>(stk + 16 * SLD + r * SLD + c + 4 * hf);
                        }
                }
            }
            if (a.normalize && a.split && cur.live) {
#pragma unroll
                for (int ct = 0; ct < DT; ++ct) {
                    const int da = ct * 16 + kg * 4;
                    if (da < D) Io<T>::st4(dkdb + cur.row * a.dkd.sn + da, *reinterpret_cast<const f32x4*>(dks + da));
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (LATE) {
            // 16-bit rows of up to 128 bytes: all of a token's dK pieces, then all of its dV pieces, in consecutive store instructions -- the
            // two 64-byte halves of a line written with the other tensor's store and two tiles of products between them left the L2 as
            // partial lines (WRITE_SIZE 154 MB for 134 MB of results, and the fill reads on top)
#pragma unroll
            for (int cta = 0; cta < DT; cta += 2) store_tile_pair<T>(dkb + cur.row * a.dk.sn, cta, DT, D, kg, dkall[cta], dkall[cta + 1 < DT ? cta + 1 : cta], cur.live, wide_dk);
#pragma unroll
            for (int cta = 0; cta < DT; cta += 2) store_tile_pair<T>(dvb + cur.row * a.dv.sn, cta, DT, D, kg, dvall[cta], dvall[cta + 1 < DT ? cta + 1 : cta], cur.live, wide_dv);
            if (a.normalize && a.split && cur.live) {
#pragma unroll
                for (int ct = 0; ct < DT; ++ct) {
                    const int da = ct * 16 + kg * 4;
                    if (da < D) Io<T>::st4(dkdb + cur.row * a.dkd.sn + da, *reinterpret_cast<const f32x4*>(dks + da));
                }
            }
        }
    };
    for (int tt = wave; tt * 16 < S; tt += 4) {
#ifdef DKV_UNCOND_PREFETCH
        fetch((tt + 4) * 16 < S ? tt + 4 : tt, nxt, lk);   // (unconditional; the wave's last tile: itself again -- lines it has just read)
        lk = look(tt + 8);
#else
        // (a branch around the prefetch: where it joins hipcc waits for the prefetch itself, so blocks of more than 64 tokens do not overlap it
        // with the products -- but the unconditional form, which re-reads the wave's own tile when there is no next one, measured slower here
        // at every block length but 16: twelve row pieces and the mask pieces per lane are a lot of requests to issue twice)
        if ((tt + 4) * 16 < S) {
            fetch(tt + 4, nxt, lk);
            lk = look(tt + 8);
        }
#endif
        tile();
        cur = nxt;
    }
}

}  // namespace sp
}  // namespace mhla
