// bf16-MFMA kernels of the causal chunk-mixing operator (bf16 tensors, K and V multiples of 64, K <= 256, n <= 256 chunks).
// Same algorithm as k_cs_out / k_cs_bwd_tok (causal.hpp): every contraction is a 64 x 64 x 64 tile product, but the tiles
// live in LDS as bf16 ([64][72], row-major) and run on v_mfma_f32_16x16x32_bf16; operands whose reduction index is the row
// index of the staged tile come through the hardware transpose read.
//
// Arithmetic (template flag HL, the default): the reference computes the whole operator in fp32 (naive.py:39, :60-78) and
// rounds once at the end (:82).  bf16 inputs are exact on the matrix pipe and every product accumulates in fp32; what would
// lose bits are the intermediates that feed a SECOND contraction -- the chunk summaries S, P, dP, dS and the score tiles
// tril(QK^T), tril(dO V^T).  With HL each of them is kept as a bf16 hi + lo pair (x = hi + lo, hi = bf16(x), lo = bf16(x - hi):
// 16 significand bits, 2^-17 relative) and every product that consumes one runs twice (hi and lo operand, fp32 accumulate).
// HL = false is the reduced-precision variant (one bf16 rounding per intermediate: half the summary traffic, 2-3e-3 of the
// result's maximum), selected only by MHLA_CAUSAL_BF16_SUMMARIES.
//
// HL = 2 (round 6, the default): the SUMMARIES S, P, dP, dS are stored in the h16 format (common.hpp) -- one plane of fp16 payload per
// 64 x 64 tile and chunk, one power-of-two multiplier per 16-row strip of the tile (four floats right behind the plane, in the chunk
// tile's padding): 11 significand bits, the precision of the reference's TF32 matmuls, in 2 bytes per element instead of 4.  Producers
// (k_csf_state2) take a strip's multiplier from its measured maximum, the mixing kernels (causal_mix.hpp) from the bound of their inputs';
// the token kernels decode a tile into the same bf16 hi + lo LDS planes as HL = 1 while committing it, so their products and the
// score tiles (hi + lo pairs in LDS only) are unchanged.  tools/sim_h16_causal.py: <= 4.4e-4 of a result's maximum.
//
// Summary layout in the workspace: tile-major [bh][n][K / 64][V / 64][planes][64][64] bf16, planes = (hi, lo) -- every 64 x 64
// tile a kernel produces or stages is one contiguous 8 KB block per plane.  The mixing kernels (causal_mix.hpp) are
// elementwise across chunks and only need to agree on it.
#pragma once
#include "causal.hpp"
#include "fused.hpp"

namespace mhla {
namespace fast {

// LDS row stride (bf16) of the token kernels' 64 x 64 tiles: 80 elements (160 bytes) where the tiles fit -- in gfx950's 64-bank
// lane-group model (tools/lds_conflicts.py) the 16-byte row reads of 16 consecutive rows, the most frequent operand fetch of these
// kernels, are conflict-free at 160 bytes and two-way conflicted at the 144 bytes of rounds 1-3 (k_csf_bwd_tok4 at C5: 211 ->
// 197 us) -- and 72 where 17 tiles have to share the 160 KB (K = 256 with hi + lo pairs).  Template parameter LD of the helpers.
template <int LD> __host__ __device__ constexpr int tile_elems() { return CS * LD; }
constexpr int CLD = 72;                 // strips of k_csf_state2
constexpr int CTE = CS * CS;            // elements per tile plane in the workspace

// Summary layout of the 16-bit pipeline: [bh][K / 64][V / 64][chunk][plane][64][64] bf16 -- the chunk index INSIDE the tile index.
// The mixing kernels read the same 128-byte row piece of EVERY chunk of a sequence at once: with the chunk index outermost
// (rounds 3-4a: [bh][chunk][tile][plane]) those pieces were a whole chunk summary apart (128 KB at K = 128, V = 256; 512 KB at
// K = 256, V = 512, i.e. every row of a slice in another 2 MB page), and the kernels' rate fell with the head size -- k_csf_mixf
// 5.2 / 5.1 / 4.9 / 4.1 TB/s and k_csf_mixb 6.0 / 5.7 / 5.0 / 4.2 TB/s at K V = 16 K / 32 K / 64 K / 128 K.  Now a slice's rows are
// one tile (16 KB with hi + lo planes) apart whatever the head size, and the token kernels, which move whole tiles, find the
// tiles of a chunk n tiles apart instead of side by side.  CS_CHUNK_PAD (2 176 bytes per chunk tile, 13 % of a hi + lo tile) keeps
// consecutive chunks from falling on the same HBM channels at the power-of-two tile size: with one 128-byte line of padding the
// mixing kernels ran 30-40 % SLOWER than in the old layout, from 640 bytes on every padding measured the same (C5 0.72-0.75 ms,
// the K = 256, V = 512 shape 1.49 -> 1.27-1.29 ms).
constexpr int CS_CHUNK_PAD = 1088;
// memory planes / LDS planes of a summary tile per format HL (0: single bf16, 1: bf16 hi + lo, 2: h16 payload)
__host__ __device__ constexpr int cs_mplanes(int HL) { return HL == 1 ? 2 : 1; }
__host__ __device__ constexpr int cs_lplanes(int HL) { return HL ? 2 : 1; }
// h16: the thread's piece of a tile (cs8_issue_state: 16 bytes of row tid >> 3) and the multiplier of that row's 16-row strip
__device__ __forceinline__ float cs8_issue_mult(const u16* __restrict__ tile, int tid) { return gld<float>(reinterpret_cast<const float*>(tile + CS * CS) + (tid >> 7)); }
struct CsLayout {
    long bhs, ts, cst;   // bf16 elements from one (b,h) / one tile / one chunk to the next
};
__host__ __device__ __forceinline__ CsLayout cs_layout(int n, long E, int planes) {   // E = K V elements per chunk summary
    CsLayout L;
    L.cst = (long)planes * CTE + CS_CHUNK_PAD;
    L.ts = (long)n * L.cst;
    L.bhs = (E / CTE) * L.ts;
    return L;
}
// offset of tile (kk0 / 64, v0 / 64) inside a (b,h)'s summaries (the chunk's cst and the plane's CTE are added by the caller)
__host__ __device__ __forceinline__ long cs_tile_off(int kk0, int v0, int V, long ts) { return ((long)(kk0 >> 6) * (V >> 6) + (v0 >> 6)) * ts; }

__device__ __forceinline__ float bf_lo16(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi16(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// (x0, x1) -> packed bf16 hi parts and packed bf16 residuals
__device__ __forceinline__ void split_pack2(float x0, float x1, unsigned& h, unsigned& l) {
    h = pack_bf16x2(x0, x1);
    l = pack_bf16x2(x0 - bf_lo16(h), x1 - bf_hi16(h));
}

__device__ __forceinline__ void zero4(f32x4 (&x)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------------------------------------
// Eight-wave token kernels: every wave owns 16 rows x 32 columns of each 64 x 64 product (row tile rt = wave & 3, column half
// ch = wave >> 2).  Every thread moves 16 bytes of a tile (row = tid >> 3), so a ring slot is one uint4 and every load / store
// instruction of a wave covers eight FULL 128-byte rows.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int NT4 = 512;
__device__ __forceinline__ void cs8_issue_tok(uint4& t, const u16* __restrict__ base, long sn, long p0, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    t = gld<uint4>(base + (p0 + (r < rv ? r : 0)) * sn + c);
}
template <int LD>
__device__ __forceinline__ void cs8_commit_tok(u16* __restrict__ dst, const uint4& t, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const bool ok = r < rv;
    *reinterpret_cast<uint4*>(dst + r * LD + c) = make_uint4(ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u);
}
__device__ __forceinline__ void cs8_issue_state(uint4& t, const u16* __restrict__ tile, int tid) { t = gld<uint4>(tile + tid * 8); }   // [64][64] contiguous
template <int LD>
__device__ __forceinline__ void cs8_commit_state(u16* __restrict__ dst, const uint4& t, int tid) {
    *reinterpret_cast<uint4*>(dst + (tid >> 3) * LD + (tid & 7) * 8) = t;
}
template <int LD>
__device__ __forceinline__ void cs8_store_tok(u16* __restrict__ base, long sn, long p0, int rv, const u16* __restrict__ Os, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    if (r < rv) *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = *reinterpret_cast<const uint4*>(Os + r * LD + c);
}
// the same with the swish gate applied on the way out: y = staged * g * sigmoid(g)   (gate rows in the output's token layout)
template <int LD>
__device__ __forceinline__ void cs8_store_tok_gate(u16* __restrict__ base, long sn, const u16* __restrict__ gbase, long gsn, long p0,
                                                   int rv, const u16* __restrict__ Os, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    if (r < rv) {
        uint4 x = *reinterpret_cast<const uint4*>(Os + r * LD + c);
        if (gbase) {
            const uint4 g = gld<uint4>(gbase + (p0 + r) * gsn + c);
            unsigned xw[4] = {x.x, x.y, x.z, x.w};
            const unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g0 = bf_lo16(gw[i]), g1 = bf_hi16(gw[i]);
                const float y0 = bf_lo16(xw[i]) * g0 / (1.f + __expf(-g0));
                const float y1 = bf_hi16(xw[i]) * g1 / (1.f + __expf(-g1));
                xw[i] = pack_bf16x2(y0, y1);
            }
            x = make_uint4(xw[0], xw[1], xw[2], xw[3]);
        }
        *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = x;
    }
}
// acc[tn] += A B for output rows 16 rt .. and columns 32 ch + 16 tn ..; reduction length 64.
//   AT false: A[m][k] = Xs[m][k]   AT true: A[m][k] = Xs[k][m]      (Xs, Ys: [64][LD] bf16 tiles)
//   BT false: B[k][n] = Ys[n][k]   BT true: B[k][n] = Ys[k][n]
template <int LD, bool AT, bool BT>
__device__ __forceinline__ void tile_mma8(f32x4 (&acc)[2], const u16* __restrict__ Xs, const u16* __restrict__ Ys, int rt, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 av = AT ? tr_read8(Xs, LD, ks * 32, rt * 16, lane)
                             : *reinterpret_cast<const bf16x8*>(Xs + (rt * 16 + n) * LD + ks * 32 + kg * 8);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int c0 = ch * 32 + tn * 16;
            const bf16x8 bv = BT ? tr_read8(Ys, LD, ks * 32, c0, lane)
                                 : *reinterpret_cast<const bf16x8*>(Ys + (c0 + n) * LD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(av, bv, acc[tn]);
        }
    }
}
// the same with the A operand (the wave's 16 rows x 64 reduction columns) already in registers: operands that several rounds
// share are read from LDS once
template <int LD>
__device__ __forceinline__ void tile_a8(bf16x8 (&av)[2], const u16* __restrict__ Xs, int rt, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) av[ks] = *reinterpret_cast<const bf16x8*>(Xs + (rt * 16 + n) * LD + ks * 32 + kg * 8);
}
template <int LD, bool BT>
__device__ __forceinline__ void tile_mma8r(f32x4 (&acc)[2], const bf16x8 (&av)[2], const u16* __restrict__ Ys, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int c0 = ch * 32 + tn * 16;
            const bf16x8 bv = BT ? tr_read8(Ys, LD, ks * 32, c0, lane)
                                 : *reinterpret_cast<const bf16x8*>(Ys + (c0 + n) * LD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(av[ks], bv, acc[tn]);
        }
}
__device__ __forceinline__ void zero2(f32x4 (&x)[2]) { x[0] = x[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
// C-layout accumulators (row = 16 rt + 4 kg + r, col = 32 ch + 16 tn + n) -> bf16 LDS tile
template <int LD>
__device__ __forceinline__ void cs8_put(u16* __restrict__ dst, const f32x4 (&x)[2], float mul, int rt, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(rt * 16 + kg * 4 + r) * LD + ch * 32 + tn * 16 + n] = cvt_bf16(mul * x[tn][r]);
}
// one element of a score tile -> its hi (and, HL, lo) plane
template <int LD, int HL>
__device__ __forceinline__ void cs8_put_score(u16* __restrict__ tiles, int row, int col, float x) {
    const int idx = row * LD + col;
    const u16 h = cvt_bf16(x);
    tiles[idx] = h;
    if constexpr (HL) tiles[tile_elems<LD>() + idx] = cvt_bf16(x - bf16_to_f32(h));
}

#ifndef CSF_TOK4_CPW
#define CSF_TOK4_CPW 8   // most chunks per workgroup of k_csf_bwd_tok4 (the launch takes the largest power of two that leaves a workgroup per CU)
#endif
// (K = 256: the chunk walk's addresses do not fit beside four rings of summary tiles -- 256 VGPRs and 13-42 spilled; single-bf16 summaries
//  at K <= 128: two workgroups per CU overlap each other and have 128 VGPRs each.  Both keep one chunk per workgroup.)
#ifndef CSF_TOK4_WALK_NK
#define CSF_TOK4_WALK_NK 3
#endif
template <int NK, int HL> __host__ __device__ constexpr int csf_tok4_cpw() { return (NK <= CSF_TOK4_WALK_NK && (HL || NK > 2)) ? CSF_TOK4_CPW : 1; }
// row stride of k_csf_bwd_tok4's tiles: 160 bytes unless its 16-17 tiles (K > 128 with hi + lo pairs) would not fit the 160 KB
template <int NK, int HL> __host__ __device__ constexpr int csf_tok4_ld() { return (HL && NK > 2) ? 72 : 80; }
template <int NK, int HL>
__host__ __device__ constexpr int csf_tok4_smem() {
    constexpr int P = HL ? 2 : 1, NB = (NK > 2 || HL) ? 2 : 1;
    return (P + 2 + 2 * NB * P + NK + (NB > 1 ? 1 : 0)) * tile_elems<csf_tok4_ld<NK, HL>()>() * 2 + 32;
}

// k_csf_bwd_tok4: dQ, dK, dV and diag(dmix) of a.cpw <= csf_tok4_cpw<NK, HL>() consecutive chunks.      grid (ceil(n / a.cpw), bh), 512 threads
// (one workgroup per CU at the h16 / hi + lo arithmetic: a chunk's first tiles are requested during the chunk before it -- behind the
//  last V slice and in step 3, in the slots that held fillers -- so that only the workgroup's first chunk waits for a cold request)
//   dA = tril(dO V^T), A = tril(Q K^T);  dQ = scale (dO P^T + m_ii dA K);  dK = V dS^T + scale m_ii dA^T Q;
//   dV = K dS + scale m_ii A^T dO;  dmix_ii = scale sum(A . dA)                              (autograd of naive.py:71-78)
// V slices outermost, the chunk's K tiles resident in LDS, NK = K / 64 exact (no guard in the unrolled K loop); a ring of NK
// register slots per summary set (and plane) keeps NK rounds of P / dS tiles in flight, the same slots carry the Q / K tiles
// before step 1 and the Q tiles again for step 3.  DBUF (K > 128, or HL: one workgroup per CU whatever the LDS use): a second
// set of P / dS buffers and a dV staging tile of its own, so that a round is commit -> ONE barrier -> refill -> multiply.
template <int NK, int HL>
__global__ __launch_bounds__(NT4, (NK <= 2 && !HL) ? 4 : 2) void k_csf_bwd_tok4(const CsTokArgs a) {
    constexpr int P = HL ? 2 : 1, MP = cs_mplanes(HL), LD = csf_tok4_ld<NK, HL>(), CT = tile_elems<LD>();   // P: LDS planes, MP: planes in memory
    constexpr bool H16 = HL == 2;
    constexpr bool DBUF = NK > 2 || HL;
    constexpr int NB = DBUF ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* As = reinterpret_cast<u16*>(smem_raw);   // [P] m_ii scale tril(Q K^T) [c][c'];  step 3: m_ii tril(dO V^T) [c][c']
    u16* X1 = As + P * CT;                        // Q slice / dO slice
    u16* X2 = X1 + CT;                            // V slice / Q slice
    u16* B1 = X2 + CT;                            // [NB][P] P slice; output staging
    u16* B2 = B1 + NB * P * CT;                   // [NB][P] dS slice; output staging
    u16* KT = B2 + NB * P * CT;                   // the chunk's K tiles [NK]
    u16* Vst = DBUF ? KT + NK * CT : B1;          // dV staging
    float* red = reinterpret_cast<float*>(KT + (NK + (DBUF ? 1 : 0)) * CT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int rt = wave & 3, ch = wave >> 2;
    constexpr bool WALK = csf_tok4_cpw<NK, HL>() > 1;
    const int c_first = WALK ? blockIdx.x * a.cpw : blockIdx.x, c_last = WALK ? min(c_first + a.cpw, a.n) - 1 : c_first;   // the workgroup's chunks
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int V = a.V;   // K == 64 NK
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    const CsLayout L = cs_layout(a.n, (long)64 * NK * V, MP);

    uint4 rP[NK][MP], rdS[NK][MP], nG, nV;
    float mP[H16 ? NK : 1], mdS[H16 ? NK : 1];   // h16: the multiplier of the thread's row strip of each tile in flight
    // a summary tile from its ring slot into its LDS planes (h16: decoded into hi + lo while it is written)
    auto commit_sum = [&](u16* dst, const uint4 (&r)[MP], float m) __attribute__((always_inline)) {
        if constexpr (H16) {
            uint4 hi, lo;
            h16_split8(r[0], m, hi, lo);
            cs8_commit_state<LD>(dst, hi, tid);
            cs8_commit_state<LD>(dst + CT, lo, tid);
        } else {
#pragma unroll
            for (int p = 0; p < MP; ++p) cs8_commit_state<LD>(dst + p * CT, r[p], tid);
        }
    };
    {   // the first chunk's Q / K tiles and first dO / V slices; every later chunk's travel during the chunk before it
        const long p0 = (long)c_first * CS;
        const int rv = (int)min((long)CS, a.T - p0);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            cs8_issue_tok(rP[kk][0], qb + kk * 64, a.q.sn, p0, rv, tid);
            cs8_issue_tok(rdS[kk][0], kb + kk * 64, a.k.sn, p0, rv, tid);
        }
        cs8_issue_tok(nG, gb, a.dout.sn, p0, rv, tid);
        cs8_issue_tok(nV, vb, a.v.sn, p0, rv, tid);
    }
    for (int ci = c_first; ci <= c_last; ++ci) {
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    // the chunk whose first tiles are requested while this one is worked on (the workgroup's last chunk: itself again, dropped --
    // no load behind a branch)
    const long p0n = (long)min(ci + 1, c_last) * CS;
    const int rvn = (int)min((long)CS, a.T - p0n);
    const u16* Pb = reinterpret_cast<const u16*>(a.P) + bh * L.bhs + ci * L.cst;
    const u16* dSb = reinterpret_cast<const u16*>(a.dS) + bh * L.bhs + ci * L.cst;
    const float mii = a.mix[(long)ci * a.ldmix + ci];
    // ---- step 1: A = tril(Q K^T); the K tiles stay ----
    f32x4 accA[2];
    zero2(accA);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        cs8_commit_tok<LD>(X1, rP[kk][0], rv, tid);
        cs8_commit_tok<LD>(KT + kk * CT, rdS[kk][0], rv, tid);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < MP; ++p) cs8_issue_state(rP[kk][p], Pb + cs_tile_off(kk * 64, 0, V, L.ts) + p * CTE, tid);
#pragma unroll
        for (int p = 0; p < MP; ++p) cs8_issue_state(rdS[kk][p], dSb + cs_tile_off(kk * 64, 0, V, L.ts) + p * CTE, tid);
        if constexpr (H16) {
            mP[kk] = cs8_issue_mult(Pb + cs_tile_off(kk * 64, 0, V, L.ts), tid);
            mdS[kk] = cs8_issue_mult(dSb + cs_tile_off(kk * 64, 0, V, L.ts), tid);
        }
        tile_mma8<LD, false, false>(accA, X1, KT + kk * CT, rt, ch, lane);
        __syncthreads();
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            const float av = col <= row ? accA[tn][r] : 0.f;
            accA[tn][r] = av;   // kept for the diagonal term
            cs8_put_score<LD, HL>(As, row, col, mii * a.scale * av);
        }

    // ---- step 2: per V slice: dA, dV, and the dQ / dK partials of every K slice ----
    // The wave's rows of dO (and, K > 128, of V and of the K tiles) are A operands of several rounds: read from LDS once.
    constexpr bool VREG = NK > 2 || HL, KREG = HL ? NK <= 2 : NK > 2;   // (HL: one workgroup per CU, 256 VGPRs)
    bf16x8 aK[KREG ? NK : 1][2];
    if constexpr (KREG) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) tile_a8<LD>(aK[kk], KT + kk * CT, rt, lane);
    }
    f32x4 accQ[NK][2], accK[NK][2], accdA[2];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        zero2(accQ[kk]);
        zero2(accK[kk]);
    }
    zero2(accdA);
    int rr = 0;   // round counter (DBUF: its parity picks the P / dS buffers)
    for (int vs = 0; vs < V; vs += 64) {
        f32x4 accV[2];
        zero2(accV);
        bf16x8 aG[2], aV[2];
        const bool last = vs + 64 >= V;   // (uniform)
        cs8_commit_tok<LD>(X1, nG, rv, tid);
        cs8_commit_tok<LD>(X2, nV, rv, tid);
        // No load of the loop sits behind a branch: hipcc loses count of the loads in flight at every join and waits for ALL of
        // them (s_waitcnt vmcnt(0) before each refill -- the ring then holds one round, whatever its depth).  Behind the last V
        // slice the dO / V slots and the dS slots receive the NEXT chunk's first slices and K tiles, the P slots step 3's Q tiles.
        const int vn = last ? vs : vs + 64;
        const int rowt = tid >> 3, colt = (tid & 7) * 8;
        const long rown = p0n + (rowt < rvn ? rowt : 0);
        {
            const long row = p0 + (rowt < rv ? rowt : 0);
            // (one chunk per workgroup: fillers -- the chunk's first Q tile, lines that step 3 wants anyway)
            const u16* fill = qb + row * a.q.sn + colt;
            nG = gld<uint4>(last ? (WALK ? gb + rown * a.dout.sn + colt : fill) : gb + vn + row * a.dout.sn + colt);
            nV = gld<uint4>(last ? (WALK ? vb + rown * a.v.sn + colt : fill) : vb + vn + row * a.v.sn + colt);
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            u16* B1c = B1 + (DBUF && (rr & 1) ? P * CT : 0);
            u16* B2c = B2 + (DBUF && (rr & 1) ? P * CT : 0);
            ++rr;
            commit_sum(B1c, rP[kk], mP[H16 ? kk : 0]);
            commit_sum(B2c, rdS[kk], mdS[H16 ? kk : 0]);
            __syncthreads();
            {
                const int r = tid >> 3, c = (tid & 7) * 8;
                const u16* qsrc = qb + kk * 64 + (p0 + (r < rv ? r : 0)) * a.q.sn + c;   // step 3's Q tile
                const u16* psrc = Pb + cs_tile_off(kk * 64, vn, V, L.ts) + tid * 8;
                const u16* ssrc = dSb + cs_tile_off(kk * 64, vn, V, L.ts) + tid * 8;
#pragma unroll
                for (int p = 0; p < MP; ++p) rP[kk][p] = gld<uint4>(last ? qsrc : psrc + p * CTE);
#pragma unroll
                for (int p = 0; p < MP; ++p)   // (behind the last slice: the next chunk's K tile; a second plane's slot gets a filler)
                    rdS[kk][p] = gld<uint4>(last ? ((WALK && p == 0) ? kb + kk * 64 + rown * a.k.sn + colt : qsrc) : ssrc + p * CTE);
                if constexpr (H16) {   // (filler: a word of the same line, never used)
                    mP[kk] = gld<float>(last ? reinterpret_cast<const float*>(qsrc) : reinterpret_cast<const float*>(psrc - tid * 8 + CTE) + (tid >> 7));
                    mdS[kk] = gld<float>(last ? reinterpret_cast<const float*>(qsrc) : reinterpret_cast<const float*>(ssrc - tid * 8 + CTE) + (tid >> 7));
                }
            }
            if (kk == 0) {
                tile_a8<LD>(aG, X1, rt, lane);
                if constexpr (VREG) tile_a8<LD>(aV, X2, rt, lane);
                tile_mma8r<LD, false>(accdA, aG, X2, ch, lane);                                     // dO V^T
#pragma unroll
                for (int p = 0; p < P; ++p) tile_mma8<LD, true, true>(accV, As + p * CT, X1, rt, ch, lane);   // A^T dO
            }
            bf16x8 aKk[2];
            if constexpr (!KREG && HL) tile_a8<LD>(aKk, KT + kk * CT, rt, lane);   // (one read for both planes)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                tile_mma8r<LD, false>(accQ[kk], aG, B1c + p * CT, ch, lane);                        // dO P^T
                if constexpr (VREG) tile_mma8r<LD, false>(accK[kk], aV, B2c + p * CT, ch, lane);    // V dS^T
                else                tile_mma8<LD, false, false>(accK[kk], X2, B2c + p * CT, rt, ch, lane);
                if constexpr (KREG) tile_mma8r<LD, true>(accV, aK[kk], B2c + p * CT, ch, lane);     // K dS
                else if constexpr (HL) tile_mma8r<LD, true>(accV, aKk, B2c + p * CT, ch, lane);
                else                tile_mma8<LD, false, true>(accV, KT + kk * CT, B2c + p * CT, rt, ch, lane);
            }
            if constexpr (!DBUF) __syncthreads();
        }
        cs8_put<LD>(Vst, accV, 1.f, rt, ch, lane);
        __syncthreads();
        cs8_store_tok<LD>(mbase(a.dv) + vs, a.dv.sn, p0, rv, Vst, tid);
        if constexpr (!DBUF) __syncthreads();
    }

    // ---- step 3: dA tile (over the dead A tile: its last reader is at least one barrier back), diagonal term, the m_ii parts
    //      of dQ / dK ----
    u16* dAs = As;
    float dsum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            const float dv = col <= row ? accdA[tn][r] : 0.f;
            dsum += accA[tn][r] * dv;
            cs8_put_score<LD, HL>(dAs, row, col, mii * dv);
        }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        cs8_commit_tok<LD>(X2, rP[kk][0], rv, tid);
        if constexpr (WALK) cs8_issue_tok(rP[kk][0], qb + kk * 64, a.q.sn, p0n, rvn, tid);   // the next chunk's Q tile
        __syncthreads();
        f32x4 acc3[2];
        zero2(acc3);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            tile_mma8<LD, false, true>(accQ[kk], dAs + p * CT, KT + kk * CT, rt, ch, lane);   // dA K
            tile_mma8<LD, true, true>(acc3, dAs + p * CT, X2, rt, ch, lane);                  // dA^T Q
        }
        cs8_put<LD>(B1, accQ[kk], a.scale, rt, ch, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) accK[kk][i] += a.scale * acc3[i];
        cs8_put<LD>(B2, accK[kk], 1.f, rt, ch, lane);
        __syncthreads();
        cs8_store_tok<LD>(mbase(a.dq) + kk * 64, a.dq.sn, p0, rv, B1, tid);
        cs8_store_tok<LD>(mbase(a.dk) + kk * 64, a.dk.sn, p0, rv, B2, tid);
        __syncthreads();
    }
    }   // (chunks)
}

// k_csf_out4: O_i = scale (Q_i P_i + m_ii tril(Q_i K_i^T) V_i)                         (naive.py:71-78)
//   grid (ceil(n / CPW), bh, V / (64 NV)), 512 threads; NV = V slices per workgroup (template: the j loop carries no runtime guard)
// A round (K slice ki, V slice j) commits slot j of a register ring to one of two LDS buffers, passes ONE barrier, refills the
// slot with the same V slice of the next K slice (behind the last K slice: with the V rows of slice j for the second phase) and
// multiplies; the Q and K tiles of the next K slice travel during the NV rounds of the current one.  No load of a round sits
// behind a branch (hipcc then waits for ALL loads in flight before each staging write); the prologue issues its loads in the
// order in which the loop re-issues them.  A workgroup walks CPW consecutive chunks: behind a chunk's last K slice the Q / K
// slots are refilled with the NEXT chunk's first tiles and, in the second phase, the P slots with its first P tiles.
// EPI: the fla layer's FusedRMSNormGated before the store (mhla_causal_normgate_fwd; the row sums of squares cross the two
// column halves through LDS); needs the workgroup to own every V slice of the head (V = 64 NV <= 256, one chunk per workgroup).
// LDS: Q tiles [2] (second phase: output staging), K tiles [2] (second phase: V tiles), P tiles [2 buffers][planes]; the score
// tile m_ii tril(Q K^T) goes into the P buffer that the last round does not read.
#ifndef CSF_OUT4_CPW_
#define CSF_OUT4_CPW_ 4
#endif
constexpr int CSF_OUT4_CPW = CSF_OUT4_CPW_;   // chunks per workgroup of k_csf_out4 (plain variant)
// NH: halves of the head's V channels that one workgroup of the fused-epilogue variant walks (V = 64 NV NH): with NH = 2 (V = 512)
// the first half's outputs wait as fp32 in an LDS stash (64 KB) until the row sums of squares of the whole head are known.
constexpr int CSF_OUT4_LD = 80;   // (8 tiles of 10 KB with hi + lo pairs: two workgroups fill the 160 KB exactly)
template <int NV, bool EPI, int HL, int NH = 1>
__host__ __device__ constexpr int csf_out4_smem() {
    return (4 + 2 * (HL ? 2 : 1)) * tile_elems<CSF_OUT4_LD>() * 2 + (EPI ? 2 * 64 * 4 : 0) + (NH > 1 ? (NH - 1) * 64 * 64 * NV * 4 : 0);
}

// (HL with four V slices per workgroup -- the fused epilogue at V = 256 -- holds 32 ring and 32 accumulator registers beside the
//  operands: one workgroup per CU on 256 VGPRs instead of 20 spilled registers at 128)
template <int NV, bool EPI, int HL, int NH = 1>
#ifndef CSF_OUT4_NV4_WAVES
#define CSF_OUT4_NV4_WAVES 4   // four slices with hi + lo pairs: two workgroups per CU at 128 VGPRs and 4 spilled registers (98 us at C5)
#endif                         // beat one workgroup at 134 (107 us) and two slices per workgroup, which read Q and K twice (108-112 us)
__global__ __launch_bounds__(NT4, NH > 1 ? 2 : (HL && NV == 4) ? (EPI ? 2 : CSF_OUT4_NV4_WAVES) : 4) void k_csf_out4(const CsOutArgs a) {   // (NH > 1: 138 KB of LDS, one workgroup per CU anyway)
    constexpr int P = HL ? 2 : 1, MP = cs_mplanes(HL), LD = CSF_OUT4_LD, CT = tile_elems<LD>();   // P: LDS planes, MP: planes in memory
    constexpr bool H16 = HL == 2;
    static_assert(NH == 1 || EPI, "only the fused-epilogue variant walks several halves of the head");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Qs = reinterpret_cast<u16*>(smem_raw);
    u16* Ks = Qs + 2 * CT;
    u16* Ps = Ks + 2 * CT;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int rt = wave & 3, ch = wave >> 2;
    constexpr int CPW = EPI ? 1 : CSF_OUT4_CPW;
    const int c0 = blockIdx.x * CPW, c1 = min(a.n, c0 + CPW), bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int V = a.V, nks = a.K / 64;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const CsLayout L = cs_layout(a.n, (long)a.K * V, MP);
    float ss[4] = {0.f, 0.f, 0.f, 0.f};   // (EPI) row sums of squares over the halves walked so far
#pragma unroll 1
    for (int hv = 0; hv < NH; ++hv) {
    const int vbase = (NH > 1 ? hv : (int)blockIdx.z) * 64 * NV;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh + vbase;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh + vbase;
    if (hv > 0) __syncthreads();   // the previous half's V tiles and score tile are dead
    const int tr = tid >> 3, tc = (tid & 7) * 8;
    // the thread's token row in chunk c (rows past the sequence: the chunk's first row, zeroed on commit)
    auto row_of = [&](int c) { const long p = (long)c * CS; return p + (tr < (int)min((long)CS, a.T - p) ? tr : 0); };
    // the score tile's buffer: the P buffer that the last round (nks NV - 1) does not read
    u16* Ao = Ps + ((((nks * NV - 1) & 1) ^ 1)) * P * CT;

    uint4 rQ, rK, rP[NV][MP];
    float mP[H16 ? NV : 1];   // h16: the multiplier of the thread's row strip of each P tile in flight
    {
        const long trow0 = row_of(c0);
        rQ = gld<uint4>(qb + trow0 * a.q.sn + tc);
        rK = gld<uint4>(kb + trow0 * a.k.sn + tc);
        __builtin_amdgcn_sched_barrier(0);
        const u16* Pb0 = reinterpret_cast<const u16*>(a.P) + bh * L.bhs + c0 * L.cst;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int p = 0; p < MP; ++p) cs8_issue_state(rP[j][p], Pb0 + cs_tile_off(0, vbase + 64 * j, V, L.ts) + p * CTE, tid);
        if constexpr (H16) {
#pragma unroll
            for (int j = 0; j < NV; ++j) mP[j] = cs8_issue_mult(Pb0 + cs_tile_off(0, vbase + 64 * j, V, L.ts), tid);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int ci = c0; ci < c1; ++ci) {
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const u16* Pb = reinterpret_cast<const u16*>(a.P) + bh * L.bhs + ci * L.cst;
    const long trow = row_of(ci);
    const int cn = ci + 1 < c1 ? ci + 1 : ci;   // next chunk (behind the last one: this chunk again -- hot lines, never used)
    const long trown = row_of(cn);
    const u16* Pbn = reinterpret_cast<const u16*>(a.P) + bh * L.bhs + cn * L.cst;
    if (ci > c0) __syncthreads();   // the previous chunk's staging tiles are dead
    const float mii = gld<float>(a.mix + (long)ci * a.ldmix + ci);   // (requested here: a wait for it later would drain the ring)
    f32x4 accO[NV][2], accA[2];
#pragma unroll
    for (int j = 0; j < NV; ++j) zero2(accO[j]);
    zero2(accA);
    int ki = 0;
    do {   // (K >= 64: no branch around the loop for the Q / K loads to sink under)
        const bool lastk = ki + 1 >= nks;   // (uniform)
        const int kn = lastk ? ki : ki + 1;
        u16* Qc = Qs + (ki & 1) * CT;
        u16* Kc = Ks + (ki & 1) * CT;
        bf16x8 aQ[2];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Pc = Ps + ((ki * NV + j) & 1) * P * CT;
            if (j == 0) {
                cs8_commit_tok<LD>(Qc, rQ, rv, tid);
                cs8_commit_tok<LD>(Kc, rK, rv, tid);
            }
            if constexpr (H16) {   // (decoded into the hi + lo planes while it is written)
                uint4 hi, lo;
                h16_split8(rP[j][0], mP[j], hi, lo);
                cs8_commit_state<LD>(Pc, hi, tid);
                cs8_commit_state<LD>(Pc + CT, lo, tid);
            } else {
#pragma unroll
                for (int p = 0; p < MP; ++p) cs8_commit_state<LD>(Pc + p * CT, rP[j][p], tid);
            }
            __syncthreads();
            if (j == 0) {   // the next K slice's tiles -- behind the last K slice: the next chunk's first
                const long rown = lastk ? trown : trow;
                const int coln = lastk ? 0 : kn * 64;
                rQ = gld<uint4>(qb + rown * a.q.sn + coln + tc);
                rK = gld<uint4>(kb + rown * a.k.sn + coln + tc);
            }
            {
                const u16* psrc = Pb + cs_tile_off(kn * 64, vbase + 64 * j, V, L.ts) + tid * 8;
                const u16* vsrc = vb + 64 * j + trow * a.v.sn + tc;   // second phase's V rows (lo slot: the same lines, never used)
#pragma unroll
                for (int p = 0; p < MP; ++p) rP[j][p] = gld<uint4>(lastk ? vsrc : psrc + p * CTE);
                if constexpr (H16) mP[j] = gld<float>(lastk ? reinterpret_cast<const float*>(vsrc) : reinterpret_cast<const float*>(psrc - tid * 8 + CTE) + (tid >> 7));
            }
            if (j == 0) {
                tile_a8<LD>(aQ, Qc, rt, lane);
                tile_mma8r<LD, false>(accA, aQ, Kc, ch, lane);   // Q K^T
            }
#pragma unroll
            for (int p = 0; p < P; ++p) tile_mma8r<LD, true>(accO[j], aQ, Pc + p * CT, ch, lane);     // Q P
        }
    } while (++ki < nks);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            cs8_put_score<LD, HL>(Ao, row, col, col <= row ? mii * accA[tn][r] : 0.f);
        }
    __syncthreads();   // the last round's tiles are dead, the score tile is complete
    if constexpr (!EPI) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Vc = Ks + (j & 1) * CT;
            u16* Oc = Qs + (j & 1) * CT;
            cs8_commit_tok<LD>(Vc, rP[j][0], rv, tid);
            __syncthreads();
#pragma unroll
            for (int p = 0; p < MP; ++p) cs8_issue_state(rP[j][p], Pbn + cs_tile_off(0, vbase + 64 * j, V, L.ts) + p * CTE, tid);   // the next chunk's first P tiles
            if constexpr (H16) mP[j] = cs8_issue_mult(Pbn + cs_tile_off(0, vbase + 64 * j, V, L.ts), tid);
#pragma unroll
            for (int p = 0; p < P; ++p) tile_mma8<LD, false, true>(accO[j], Ao + p * CT, Vc, rt, ch, lane);        // tril(QK^T) V
            cs8_put<LD>(Oc, accO[j], a.scale, rt, ch, lane);
            __syncthreads();
            cs8_store_tok<LD>(ob + 64 * j, a.o.sn, p0, rv, Oc, tid);
        }
    } else {
        float* red = reinterpret_cast<float*>(Ps + 2 * P * CT);   // [2 column halves][64 rows] sums of squares
        float* stash = red + 2 * 64;                               // NH > 1: [NH - 1][NV][2][4][512 threads] fp32 outputs of earlier halves
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Vc = Ks + (j & 1) * CT;
            cs8_commit_tok<LD>(Vc, rP[j][0], rv, tid);
            __syncthreads();
#pragma unroll
            for (int p = 0; p < P; ++p) tile_mma8<LD, false, true>(accO[j], Ao + p * CT, Vc, rt, ch, lane);
        }
        // row sums of squares over the head's V channels: lane holds rows 16 rt + 4 kg + r, columns 64 j + 32 ch + 16 tn + n
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = a.scale * accO[j][tn][r];
                    ss[r] += x * x;
                }
        if (hv + 1 < NH) {   // (uniform) more of the head to come: this half's outputs wait in the stash (every thread its own values)
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stash[(((hv * NV + j) * 2 + tn) * 4 + r) * NT4 + tid] = accO[j][tn][r];
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = ss[r];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
            if (n == 0) red[ch * 64 + rt * 16 + kg * 4 + r] = x;
        }
        __syncthreads();   // (also: every product of the second phase is done, the V tiles are dead)
        float rstd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r;
            rstd[r] = rsqrtf((red[row] + red[64 + row]) / (float)V + a.neps);
        }
        // the halves, last one first (its outputs are still in the accumulators; the earlier ones come back from the stash)
#pragma unroll 1
        for (int he = NH - 1; he >= 0; --he) {
            const int vb_e = (NH > 1 ? he : (int)blockIdx.z) * 64 * NV;
            u16* ob_e = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh + vb_e;
            u16* yb = (u16*)a.y.ptr + b * a.y.sb + h * a.y.sh + vb_e;
            const u16* gb = a.gate.ptr ? (const u16*)a.gate.ptr + b * a.gate.sb + h * a.gate.sh + vb_e : nullptr;
            if (he < NH - 1) {
                __syncthreads();   // the staging tiles of the half stored before are dead
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accO[j][tn][r] = stash[(((he * NV + j) * 2 + tn) * 4 + r) * NT4 + tid];
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                u16* Oc = Qs + (j & 1) * CT;
                u16* Yc = Ks + (j & 1) * CT;
                if (a.o.ptr) {   // training: the operator's own output is kept for the norm's backward
                    cs8_put<LD>(Oc, accO[j], a.scale, rt, ch, lane);
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const float w = a.nw ? gld<float>(a.nw + vb_e + 64 * j + ch * 32 + tn * 16 + n) : 1.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) accO[j][tn][r] *= a.scale * rstd[r] * w;
                }
                cs8_put<LD>(Yc, accO[j], 1.f, rt, ch, lane);
                __syncthreads();
                if (a.o.ptr) cs8_store_tok<LD>(ob_e + 64 * j, a.o.sn, p0, rv, Oc, tid);
                cs8_store_tok_gate<LD>(yb + 64 * j, a.y.sn, gb ? gb + 64 * j : nullptr, a.gate.sn, p0, rv, Yc, tid);
            }
        }
    }
    }   // chunks of the workgroup
    }   // halves of the head (fused epilogue with V = 512)
}

// -------------------------------------------------------------------------------------------------
// k_csf_state2: out[bh][ci][kk][v] = mul * sum_{c in chunk ci} X[c][kk] Y[c][v]      (bf16 hi [+ lo] planes, tile-major)
//   forward: X = K, Y = V (S_j, naive.py:60);  backward: X = Q, Y = dO, mul = scale (dP_i)
// grid (ceil(n / ST2_CPW), bh, blocks): one workgroup per [128 x 256] block of the summary (the whole summary at K = 128,
// V = 256) walks ST2_CPW consecutive chunks.  All of the block's token rows of a chunk are requested at once (48 KB in flight),
// the next chunk's rows travel in registers while the current one is multiplied; every wave multiplies its 32 summary rows
// against all V columns (product formed transposed: a lane owns 4 consecutive v) and streams them out through a wave-private
// staging strip, eight full 128-byte rows per store instruction.
// -------------------------------------------------------------------------------------------------
struct CsfStateArgs {
    View x, y;
    u16* out;
    int H, n, K, V;
    long T;
    float mul;
};
constexpr int ST2_KW = 128, ST2_VW = 256, ST2_LDX = ST2_KW + 8, ST2_LDY = ST2_VW + 8;
#ifndef ST2_T
#define ST2_T 512   // threads: eight waves with one 16-row strip of the summary each (four waves with two: 100 -> 92-94 us for the two launches at C5,
                    // 177 -> 162 at K = 256, V = 512 -- twice the waves per CU at the same tiles; two chunks per workgroup instead of four: no gain)
#endif
#ifndef ST2_CPW_
#define ST2_CPW_ 4
#endif
template <int HL> __host__ __device__ constexpr int csf_state2_smem() { return (CS * ST2_LDX + CS * ST2_LDY + (ST2_T / 64) * cs_mplanes(HL) * 16 * CLD) * 2; }
constexpr int ST2_CPW = ST2_CPW_;

template <int HL>
__global__ __launch_bounds__(ST2_T, 2) void k_csf_state2(const CsfStateArgs a) {
    constexpr int P = cs_mplanes(HL);   // planes written
    constexpr int XP = 4 * 256 / ST2_T, YP = 8 * 256 / ST2_T, XR = 64 / XP, YR = 64 / YP, RRS = 8 / (ST2_T / 64);   // staging passes, rows per pass, strips per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + CS * ST2_LDX;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    u16* Ws = Ys + CS * ST2_LDY + wave * P * 16 * CLD;   // [P][16][CLD]
    const int c0 = blockIdx.x * ST2_CPW, c1 = min(a.n, c0 + ST2_CPW), bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int nvb = (a.V + ST2_VW - 1) / ST2_VW, kb = blockIdx.z / nvb, vb = blockIdx.z - kb * nvb;
    const int k0 = kb * ST2_KW, v0 = vb * ST2_VW, kw = min(ST2_KW, a.K - k0), vw = min(ST2_VW, a.V - v0);
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh + k0;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh + v0;
    const CsLayout L = cs_layout(a.n, (long)a.K * a.V, P);

    // X: 4 passes of 16 rows x 16 pieces; Y: 8 passes of 8 rows x 32 pieces (pieces past the block's width, rows past the chunk: zeros)
    const int xc = (tid & 15) * 8, yc = (tid & 31) * 8;
    const bool xok = xc < kw, yok = yc < vw;
    uint4 xr[XP], yr[YP];
    auto issue = [&](int ci, bool filler) {   // (no load behind a branch: rows past the sequence's end read the chunk's first row;
        const long p0 = (long)ci * CS;        //  the filler behind the last chunk reads that one row with every pass)
        const int rv = filler ? 0 : (int)min((long)CS, a.T - p0);
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int row = (tid >> 4) + XR * p;
            xr[p] = gld_stream16(xb + (p0 + (row < rv ? row : 0)) * a.x.sn + (xok ? xc : 0));
        }
#pragma unroll
        for (int p = 0; p < YP; ++p) {
            const int row = (tid >> 5) + YR * p;
            yr[p] = gld_stream16(yb + (p0 + (row < rv ? row : 0)) * a.y.sn + (yok ? yc : 0));
        }
    };
    issue(c0, false);
    for (int ci = c0; ci < c1; ++ci) {
        const int rv = (int)min((long)CS, a.T - (long)ci * CS);
        u16* ob = a.out + bh * L.bhs + ci * L.cst;
        if (ci > c0) __syncthreads();   // the previous chunk's tiles are dead
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int row = (tid >> 4) + XR * p;
            const bool ok = row < rv && xok;
            *reinterpret_cast<uint4*>(Xs + row * ST2_LDX + xc) = make_uint4(ok ? xr[p].x : 0u, ok ? xr[p].y : 0u, ok ? xr[p].z : 0u, ok ? xr[p].w : 0u);
        }
#pragma unroll
        for (int p = 0; p < YP; ++p) {
            const int row = (tid >> 5) + YR * p;
            const bool ok = row < rv && yok;
            *reinterpret_cast<uint4*>(Ys + row * ST2_LDY + yc) = make_uint4(ok ? yr[p].x : 0u, ok ? yr[p].y : 0u, ok ? yr[p].z : 0u, ok ? yr[p].w : 0u);
        }
        __syncthreads();
        issue(min(ci + 1, c1 - 1), ci + 1 >= c1);
#pragma unroll
        for (int rr = 0; rr < RRS; ++rr) {
            const int rt = wave * RRS + rr;             // 16 summary rows kk = k0 + 16 rt ..
            if (rt * 16 < kw) {
                // the product is formed transposed (m = v, n = kk): a lane ends up with four consecutive v of one summary row
                const bf16x8 xa0 = tr_read8(Xs, ST2_LDX, 0, rt * 16, lane), xa1 = tr_read8(Xs, ST2_LDX, 32, rt * 16, lane);
                for (int vt = 0; vt * 64 < vw; ++vt) {
                    f32x4 acc[4];
                    zero4(acc);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        acc[tn] = mfma_bf16(tr_read8(Ys, ST2_LDY, 0, vt * 64 + tn * 16, lane), xa0, acc[tn]);
                        acc[tn] = mfma_bf16(tr_read8(Ys, ST2_LDY, 32, vt * 64 + tn * 16, lane), xa1, acc[tn]);
                    }
                    if constexpr (HL == 2) {
                        // h16: the strip's multiplier from its largest magnitude (the wave holds the whole 16 x 64 strip), the payload as fp16
                        float mx = 0.f;
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(acc[tn][r]));
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                        const float hm = h16_mult_from_max(fabsf(a.mul) * mx), hs = a.mul * h16_inv(hm);
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            *reinterpret_cast<uint2*>(Ws + nl * CLD + tn * 16 + kg * 4) = make_uint2(h16_pack2(hs * acc[tn][0], hs * acc[tn][1]), h16_pack2(hs * acc[tn][2], hs * acc[tn][3]));
                        if (lane == 0)
                            gst<float>(reinterpret_cast<float*>(ob + cs_tile_off(k0 + rt * 16, v0 + vt * 64, a.V, L.ts) + CTE) + (rt & 3), hm);
                    } else {
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        unsigned h0, h1, l0, l1;
                        split_pack2(a.mul * acc[tn][0], a.mul * acc[tn][1], h0, l0);
                        split_pack2(a.mul * acc[tn][2], a.mul * acc[tn][3], h1, l1);
                        *reinterpret_cast<uint2*>(Ws + nl * CLD + tn * 16 + kg * 4) = make_uint2(h0, h1);
                        if constexpr (HL == 1) *reinterpret_cast<uint2*>(Ws + 16 * CLD + nl * CLD + tn * 16 + kg * 4) = make_uint2(l0, l1);
                    }
                    }
                    wave_lds_fence();
                    // a store instruction covers eight full 128-byte rows (two half rows per lane pair made it 16 half lines)
                    const int r = lane >> 3, c = (lane & 7) * 8;
                    u16* d = ob + cs_tile_off(k0 + rt * 16, v0 + vt * 64, a.V, L.ts) + ((rt * 16) & 63) * CS + r * CS + c;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const uint4 o0 = *reinterpret_cast<const uint4*>(Ws + p * 16 * CLD + r * CLD + c);
                        const uint4 o1 = *reinterpret_cast<const uint4*>(Ws + p * 16 * CLD + (r + 8) * CLD + c);
                        gst<uint4>(d + p * CTE, o0);
                        gst<uint4>(d + p * CTE + 8 * CS, o1);
                    }
                    wave_lds_fence();
                }
            }
        }
    }
}

}  // namespace fast
}  // namespace mhla
