// Per-head RMSNorm (x optional swish gate) -- the step right after the MHLA operator in the fla layer
// (mhla_nlp/fla/modules/fused_norm_gate.py:77-99, used at mhla_nlp/fla/layers/mhla.py:351-355) and in
// Wan's MHLA_Video_Uni (g_norm [x SiLU(g)], mhla_videogen/diffusion/model/wan/mhla_utils.py:357-362).
// One wave per row (token, head); fp32 math; HBM-bound streaming kernel with 4-wide vector I/O.
#pragma once
#include "common.cuh"

namespace mhla {

struct NormArgs {
    const void* x;
    long ldx;
    const void* g;
    long ldg;
    const float* w;
    void* y;
    long ldy;
    float* rstd;
    const void* dy;
    long lddy;
    void* dx;
    long lddx;
    void* dg;
    long lddg;
    float* dwp;
    long rows;
    int D;
    float eps;
};

// NV: vec4 per lane (D <= 256 * NV)
template <typename T, int NV, bool GATE>
__global__ __launch_bounds__(256) void k_rmsnorm_gate_fwd(const NormArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
        const T* xr = (const T*)a.x + row * a.ldx;
        f32x4 xv[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < a.D) xv[i] = Io<T>::ld4(xr + c);
#pragma unroll
            for (int t = 0; t < 4; ++t) ss += xv[i][t] * xv[i][t];
        }
        ss = wave_sum(ss);
        const float rstd = 1.f / sqrtf(ss / (float)a.D + a.eps);
        if (a.rstd && lane == 0) a.rstd[row] = rstd;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < a.D) {
                f32x4 y = xv[i] * rstd;
                if (a.w) y *= *reinterpret_cast<const f32x4*>(a.w + c);
                if (GATE) {
                    f32x4 gv = Io<T>::ld4((const T*)a.g + row * a.ldg + c);
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t] *= gv[t] / (1.f + __expf(-gv[t]));
                }
                Io<T>::st4((T*)a.y + row * a.ldy + c, y);
            }
        }
    }
}


// Narrow rows (D <= 4 * LPR, LPR = 16 or 32 lanes per row): 64 / LPR rows per wave, so that every lane carries data
// (Wan's per-head norm has D = 128: the one-row-per-wave kernel would leave half the wave idle).
template <typename T, int LPR, bool GATE>
__global__ __launch_bounds__(256) void k_rmsnorm_gate_fwd_sub(const NormArgs a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, c = (lane % LPR) * 4;
    const long nw = (long)gridDim.x * 4 * RPW;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RPW; row0 < a.rows; row0 += nw) {
        const long row = row0 + sub;
        const bool live = row < a.rows && c < a.D;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (live) xv = Io<T>::ld4((const T*)a.x + row * a.ldx + c);
        float ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = 1.f / sqrtf(ss / (float)a.D + a.eps);
        if (a.rstd && row < a.rows && (lane % LPR) == 0) a.rstd[row] = rstd;
        if (live) {
            f32x4 y = xv * rstd;
            if (a.w) y *= *reinterpret_cast<const f32x4*>(a.w + c);
            if (GATE) {
                const f32x4 gv = Io<T>::ld4((const T*)a.g + row * a.ldg + c);
#pragma unroll
                for (int t = 0; t < 4; ++t) y[t] *= gv[t] / (1.f + __expf(-gv[t]));
            }
            Io<T>::st4((T*)a.y + row * a.ldy + c, y);
        }
    }
}

template <typename T, int NV, bool GATE>
__global__ __launch_bounds__(256) void k_rmsnorm_gate_bwd(const NormArgs a) {
    __shared__ float red[4][NV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    f32x4 dwacc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) dwacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
        f32x4 xv[NV], uv[NV], sv[NV], dyv[NV], gv[NV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            xv[i] = dyv[i] = gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < a.D) {
                xv[i] = Io<T>::ld4((const T*)a.x + row * a.ldx + c);
                dyv[i] = Io<T>::ld4((const T*)a.dy + row * a.lddy + c);
                if (GATE) gv[i] = Io<T>::ld4((const T*)a.g + row * a.ldg + c);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) ss += xv[i][t] * xv[i][t];
        }
        ss = wave_sum(ss);
        const float rstd = 1.f / sqrtf(ss / (float)a.D + a.eps);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            f32x4 w = {1.f, 1.f, 1.f, 1.f};
            if (a.w && c < a.D) w = *reinterpret_cast<const f32x4*>(a.w + c);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float s = 1.f;
                if (GATE) s = gv[i][t] / (1.f + __expf(-gv[i][t]));
                sv[i][t] = s;
                const float xhat = xv[i][t] * rstd;
                uv[i][t] = dyv[i][t] * w[t] * s;
                dot += uv[i][t] * xhat;
                dwacc[i][t] += dyv[i][t] * xhat * s;
            }
        }
        dot = wave_sum(dot) / (float)a.D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < a.D) {
                f32x4 dx, dg;
                f32x4 w = {1.f, 1.f, 1.f, 1.f};
                if (a.w) w = *reinterpret_cast<const f32x4*>(a.w + c);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float xhat = xv[i][t] * rstd;
                    dx[t] = rstd * (uv[i][t] - xhat * dot);
                    if (GATE) {
                        const float sg = 1.f / (1.f + __expf(-gv[i][t]));
                        dg[t] = dyv[i][t] * xhat * w[t] * sg * (1.f + gv[i][t] * (1.f - sg));
                    }
                }
                Io<T>::st4((T*)a.dx + row * a.lddx + c, dx);
                if (GATE) Io<T>::st4((T*)a.dg + row * a.lddg + c, dg);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) red[wave][(lane + 64 * i) * 4 + t] = dwacc[i][t];
    __syncthreads();
    for (int c = threadIdx.x; c < a.D; c += 256)
        a.dwp[(long)blockIdx.x * a.D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// -------------------------------------------------------------------------------------------------
// q / k prologue of Wan's MHLA_Video_Uni (mhla_videogen/diffusion/model/wan/mhla_utils.py:268-272 after the
// .float() at :308):  y = relu(rmsnorm_C(x) * w) + eps over the whole channel dim C = H * D of a token, x in the
// dtype of the projection (bf16 / fp16 / fp32), y fp32.  One wave per token row, the row stays in registers
// (C <= 8 * 64 * NV); w == nullptr skips the norm (qk_norm = False: relu(x) + eps).
// -------------------------------------------------------------------------------------------------
struct PrologueArgs {
    const void* x;
    long ldx;
    const float* w;
    float* y;
    long ldy;
    long rows;
    int C;
    float norm_eps, eps;
    int norm;
};

template <typename T, int NV>
__global__ __launch_bounds__(256) void k_qk_prologue(const PrologueArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
        const T* xr = (const T*)a.x + row * a.ldx;
        f32x4 xv[NV][2];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 8;
            xv[i][0] = xv[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < a.C) {
                xv[i][0] = Io<T>::ld4(xr + c);
                xv[i][1] = Io<T>::ld4(xr + c + 4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) ss += xv[i][0][t] * xv[i][0][t] + xv[i][1][t] * xv[i][1][t];
        }
        float rstd = 1.f;
        if (a.norm) {
            ss = wave_sum(ss);
            rstd = 1.f / sqrtf(ss / (float)a.C + a.norm_eps);
        }
        float* yr = a.y + row * a.ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < a.C) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 y = xv[i][u] * rstd;
                    if (a.w) y *= *reinterpret_cast<const f32x4*>(a.w + c + 4 * u);
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t] = fmaxf(y[t], 0.f) + a.eps;
                    *reinterpret_cast<f32x4*>(yr + c + 4 * u) = y;
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// q / k prologue of the fla layer: feature map (mhla_nlp/fla/layers/mhla.py:297-299: relu / identity / elu + 1) followed
// by the NeoX-style rotary embedding (:311; mhla_nlp/fla/modules/rotary.py:45-135: halves (x[i], x[i + K/2]) rotate by
// the token's angle, fp32 math on cos / sin tables kept in the activation dtype), in one pass; BWD applies the
// transposed rotation to the upstream gradient and the feature map's derivative (from the saved input).
//   forward : a = f(x[i]), b = f(x[i + K/2]);  y[i] = a c - b s;  y[i + K/2] = b c + a s
//   backward: da = g[i] c + g[i + K/2] s;  db = -g[i] s + g[i + K/2] c;  dx = (da f'(x[i]), db f'(x[i + K/2]))
// One thread: 4 + 4 elements of one (token, head) row.
// -------------------------------------------------------------------------------------------------
struct FmRotArgs {
    View x;         // forward: input; backward: upstream gradient
    View xs;        // backward: the forward's input (feature-map derivative)
    MView y;
    const void* cos;   // [>= t_off + T][K/2], activation dtype
    const void* sin;
    long ldt;
    int B, T, H, K;
    int fmap;       // 0 identity, 1 relu, 2 elu + 1
    long t_off;
};

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void k_fmap_rotary(const FmRotArgs a) {
    const int G = a.K / 8;   // groups of 4 per half
    const long gid = (long)blockIdx.x * 256 + threadIdx.x, total = (long)a.B * a.T * a.H * G;
    if (gid >= total) return;
    const int g = (int)(gid % G);
    long r = gid / G;
    const int h = (int)(r % a.H);
    r /= a.H;
    const int t = (int)(r % a.T), b = (int)(r / a.T);
    const int i = g * 4, half = a.K / 2;
    const T* xp = (const T*)a.x.ptr + b * a.x.sb + (long)t * a.x.sn + h * a.x.sh + i;
    f32x4 x0 = Io<T>::ld4(xp), x1 = Io<T>::ld4(xp + half);
    const f32x4 c = Io<T>::ld4((const T*)a.cos + (a.t_off + t) * a.ldt + i), s = Io<T>::ld4((const T*)a.sin + (a.t_off + t) * a.ldt + i);
    f32x4 y0, y1;
    if (!BWD) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (a.fmap == 1) { x0[u] = fmaxf(x0[u], 0.f); x1[u] = fmaxf(x1[u], 0.f); }
            else if (a.fmap == 2) { x0[u] = x0[u] > 0.f ? x0[u] + 1.f : __expf(x0[u]); x1[u] = x1[u] > 0.f ? x1[u] + 1.f : __expf(x1[u]); }
        }
        y0 = x0 * c - x1 * s;
        y1 = x1 * c + x0 * s;
    } else {
        y0 = x0 * c + x1 * s;
        y1 = x1 * c - x0 * s;
        if (a.fmap) {
            const T* sp = (const T*)a.xs.ptr + b * a.xs.sb + (long)t * a.xs.sn + h * a.xs.sh + i;
            const f32x4 s0 = Io<T>::ld4(sp), s1 = Io<T>::ld4(sp + half);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (a.fmap == 1) { y0[u] = s0[u] > 0.f ? y0[u] : 0.f; y1[u] = s1[u] > 0.f ? y1[u] : 0.f; }
                else { y0[u] *= s0[u] > 0.f ? 1.f : __expf(s0[u]); y1[u] *= s1[u] > 0.f ? 1.f : __expf(s1[u]); }
            }
        }
    }
    T* yp = (T*)a.y.ptr + b * a.y.sb + (long)t * a.y.sn + h * a.y.sh + i;
    Io<T>::st4(yp, y0);
    Io<T>::st4(yp + half, y1);
}

}  // namespace mhla
