// Causal chunk-mixing MHLA kernels (fla), generic fp32-compute path.
//
// Reference: naive_chunk_simple_mhla_fixed, mhla_nlp/fla/ops/mhla/naive.py:39-82.  Chunk C = 64.
//   forward : k_bm_state<MODE 2>: S_j = K_j^T V_j ;  k_mix<0,1>: P_i = sum_{j<i} m_ij S_j ;
//             k_cs_out: O_i = scale (Q_i P_i + m_ii tril(Q_i K_i^T) V_i)
//   backward: k_bm_state<MODE 2>: dP_i = scale Q_i^T dO_i ; k_mix<1,1>: dS_j = sum_{i>j} m_ij dP_i ;
//             k_dw<1> + k_dw_reduce<1>: dmix ; k_cs_bwd_tok: dQ, dK, dV, diag(dmix).
// Every contraction is a 64x64x64 fp32-MFMA tile product; K and V are walked in slices of 64.
#pragma once
#include "blockmix.hpp"

namespace mhla {

constexpr int CS = 64;               // chunk length
constexpr int CS_LDX = 66;           // x-major tiles [64][66]
constexpr int CS_LDK = 80;           // k-major tiles [64][80]
constexpr int CS_LDO = 68;           // staging

struct CsOutArgs {
    View q, k, v;
    MView o;
    const float* mix;
    int ldmix;
    const float* P;   // [bh][n][K][V]
    int H, n, K, V;
    long T;
    float scale;
    // fused epilogue of the fla layer (k_csf_out<ST, true> only; layers/mhla.py:351-355, fused_norm_gate.py:77-99):
    // y = o * rsqrt(mean(o^2 over V) + neps) * nw * g * sigmoid(g); o itself is stored too when o.ptr is set (training)
    MView y;
    View gate;          // ptr null: no gate
    const float* nw;    // [V] or null
    float neps;
};
constexpr int CS_OUT_SMEM_FLOATS = 3 * CS * CS_LDX + CS * CS_LDK + CS * CS_LDO;

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_cs_out(const CsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                  // [64 c][66]
    float* Ks = Qs + CS * CS_LDX;      // [64 c'][66]
    float* As = Ks + CS * CS_LDX;      // [64 c][66]
    float* Ps = As + CS * CS_LDX;      // [64 kk][80]  P slice, later V slice [64 c][80]
    float* Os = Ps + CS * CS_LDK;      // [64][68]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int v0 = blockIdx.z * 64, vv = min(64, a.V - v0);
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const T* qb = (const T*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const T* kb = (const T*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const T* vb = (const T*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    T* ob = (T*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const float* Pi = a.P + ((long)bh * a.n + ci) * a.K * a.V;
    const bool vec_ok = (a.V & 3) == 0;

    f32x4 accO[4], accA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accO[i] = accA[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ks = 0; ks < a.K; ks += 64) {
        const int kv = min(64, a.K - ks);
        load_tile<T, 64, false>(Qs, CS_LDX, qb + ks, a.q.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        load_tile<T, 64, false>(Ks, CS_LDX, kb + ks, a.k.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        load_mat_f32<64>(Ps, CS_LDK, Pi + (long)ks * a.V + v0, a.V, kv, 64, vv, tid, NTHREADS, vec_ok);
        __syncthreads();
        ab_accum<4, 4, true>(accA, Qs, CS_LDX, Ks, CS_LDX, 16, 64, wave, lane);
        ab_accum<4, 4, false>(accO, Qs, CS_LDX, Ps, CS_LDK, 16, 64, wave, lane);
        __syncthreads();
    }
    const float mii = a.mix[(long)ci * a.ldmix + ci];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave + 4 * i, tm = t >> 2, tn = t & 3, col = tn * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + kq * 4 + r;
            As[row * CS_LDX + col] = col <= row ? mii * accA[i][r] : 0.f;
        }
    }
    load_tile<T, 64, false>(Ps, CS_LDK, vb + v0, a.v.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
    __syncthreads();
    ab_accum<4, 4, false>(accO, As, CS_LDX, Ps, CS_LDK, 16, 64, wave, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave + 4 * i, tm = t >> 2, tn = t & 3, col = tn * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(tm * 16 + kq * 4 + r) * CS_LDO + col] = a.scale * accO[i][r];
    }
    __syncthreads();
    store_tile<T, 64>(ob + v0, a.o.sn, nullptr, p0, Os, CS_LDO, rv, vv, tid, NTHREADS);
}

struct CsTokArgs {
    View q, k, v, dout;
    MView dq, dk, dv;
    const float* mix;
    int ldmix;
    const float* P;    // [bh][n][K][V]
    const float* dS;   // [bh][n][K][V]
    float* diag;       // [bh][n]
    int H, n, K, V;
    long T;
    float scale;
    int cpw = 1;       // k_csf_bwd_tok4: consecutive chunks per workgroup (its grid is ceil(n / cpw) x bh)
};
constexpr int CS_TOK_SMEM_FLOATS = 6 * CS * CS_LDX + CS * CS_LDO + 8;

template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_cs_bwd_tok(const CsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // m_ii scale tril(Q K^T)   [c][c']
    float* dAs = As + CS * CS_LDX;     // m_ii tril(dO V^T)        [c][c']
    float* X1 = dAs + CS * CS_LDX;
    float* X2 = X1 + CS * CS_LDX;
    float* B1 = X2 + CS * CS_LDX;
    float* B2 = B1 + CS * CS_LDX;
    float* Os = B2 + CS * CS_LDX;      // [64][68]
    float* red = Os + CS * CS_LDO;     // [8]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const int K = a.K, V = a.V;
    auto base = [&](const View& w) { return (const T*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (T*)w.ptr + b * w.sb + h * w.sh; };
    const T *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    const float* Pi = a.P + ((long)bh * a.n + ci) * K * V;
    const float* dSi = a.dS + ((long)bh * a.n + ci) * K * V;
    const bool vec_ok = (V & 3) == 0;
    const float mii = a.mix[(long)ci * a.ldmix + ci];

    f32x4 acc1[4], acc2[4];
    auto zero = [](f32x4(&x)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto stage = [&](const f32x4(&x)[4], float mul) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = wave + 4 * i, tm = t >> 2, tn = t & 3, col = tn * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(tm * 16 + kq * 4 + r) * CS_LDO + col] = mul * x[i][r];
        }
    };

    // ---- step 1: A = tril(Q K^T), dA = tril(dO V^T), diag = scale * sum(A . dA) ----
    zero(acc1);
    zero(acc2);
    for (int ks = 0; ks < K; ks += 64) {
        const int kv = min(64, K - ks);
        load_tile<T, 64, false>(X1, CS_LDX, qb + ks, a.q.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        load_tile<T, 64, false>(X2, CS_LDX, kb + ks, a.k.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        __syncthreads();
        ab_accum<4, 4, true>(acc1, X1, CS_LDX, X2, CS_LDX, 16, 64, wave, lane);
        __syncthreads();
    }
    for (int vs = 0; vs < V; vs += 64) {
        const int vv = min(64, V - vs);
        load_tile<T, 64, false>(X1, CS_LDX, gb + vs, a.dout.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
        load_tile<T, 64, false>(X2, CS_LDX, vb + vs, a.v.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
        __syncthreads();
        ab_accum<4, 4, true>(acc2, X1, CS_LDX, X2, CS_LDX, 16, 64, wave, lane);
        __syncthreads();
    }
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave + 4 * i, tm = t >> 2, tn = t & 3, col = tn * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + kq * 4 + r;
            const bool keep = col <= row;
            const float av = keep ? acc1[i][r] : 0.f, dv = keep ? acc2[i][r] : 0.f;
            dsum += av * dv;
            As[row * CS_LDX + col] = mii * a.scale * av;
            dAs[row * CS_LDX + col] = mii * dv;
        }
    }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (red[0] + red[1] + red[2] + red[3]);

    // ---- step 2: dQ, dK per K slice ----
    for (int ks = 0; ks < K; ks += 64) {
        const int kv = min(64, K - ks);
        f32x4 acc3[4];
        zero(acc1);   // dO P^T + m_ii dA K
        zero(acc2);   // V dS^T
        zero(acc3);   // m_ii dA^T Q
        for (int vs = 0; vs < V; vs += 64) {
            const int vv = min(64, V - vs);
            load_tile<T, 64, false>(X1, CS_LDX, gb + vs, a.dout.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
            load_tile<T, 64, false>(X2, CS_LDX, vb + vs, a.v.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
            load_mat_f32<64>(B1, CS_LDX, Pi + (long)ks * V + vs, V, kv, 64, vv, tid, NTHREADS, vec_ok);
            load_mat_f32<64>(B2, CS_LDX, dSi + (long)ks * V + vs, V, kv, 64, vv, tid, NTHREADS, vec_ok);
            __syncthreads();
            ab_accum<4, 4, true>(acc1, X1, CS_LDX, B1, CS_LDX, 16, 64, wave, lane);
            ab_accum<4, 4, true>(acc2, X2, CS_LDX, B2, CS_LDX, 16, 64, wave, lane);
            __syncthreads();
        }
        load_tile<T, 64, false>(X1, CS_LDX, kb + ks, a.k.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        load_tile<T, 64, false>(X2, CS_LDX, qb + ks, a.q.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
        __syncthreads();
        ab_accum<4, 4, false>(acc1, dAs, CS_LDX, X1, CS_LDX, 16, 64, wave, lane);
        ab_accum<4, 4, false, true>(acc3, dAs, CS_LDX, X2, CS_LDX, 16, 64, wave, lane);
        stage(acc1, a.scale);
        __syncthreads();
        store_tile<T, 64>(mbase(a.dq) + ks, a.dq.sn, nullptr, p0, Os, CS_LDO, rv, kv, tid, NTHREADS);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] += a.scale * acc3[i];
        stage(acc2, 1.f);
        __syncthreads();
        store_tile<T, 64>(mbase(a.dk) + ks, a.dk.sn, nullptr, p0, Os, CS_LDO, rv, kv, tid, NTHREADS);
        __syncthreads();
    }

    // ---- step 3: dV per V slice ----
    for (int vs = 0; vs < V; vs += 64) {
        const int vv = min(64, V - vs);
        zero(acc1);
        for (int ks = 0; ks < K; ks += 64) {
            const int kv = min(64, K - ks);
            load_tile<T, 64, false>(X1, CS_LDX, kb + ks, a.k.sn, nullptr, p0, rv, CS, kv, 0.f, tid, NTHREADS);
            load_mat_f32<64>(B1, CS_LDX, dSi + (long)ks * V + vs, V, kv, 64, vv, tid, NTHREADS, vec_ok);
            __syncthreads();
            ab_accum<4, 4, false>(acc1, X1, CS_LDX, B1, CS_LDX, 16, 64, wave, lane);
            __syncthreads();
        }
        load_tile<T, 64, false>(X2, CS_LDX, gb + vs, a.dout.sn, nullptr, p0, rv, CS, vv, 0.f, tid, NTHREADS);
        __syncthreads();
        ab_accum<4, 4, false, true>(acc1, As, CS_LDX, X2, CS_LDX, 16, 64, wave, lane);
        stage(acc1, 1.f);
        __syncthreads();
        store_tile<T, 64>(mbase(a.dv) + vs, a.dv.sn, nullptr, p0, Os, CS_LDO, rv, vv, tid, NTHREADS);
        __syncthreads();
    }
}

}  // namespace mhla
