// Per-head RMSNorm (x optional swish gate) -- the step right after the MHLA operator in the fla layer
// (mhla_nlp/fla/modules/fused_norm_gate.py:77-99, used at mhla_nlp/fla/layers/mhla.py:351-355) and in
// Wan's MHLA_Video_Uni (g_norm [x SiLU(g)], mhla_videogen/diffusion/model/wan/mhla_utils.py:357-362).
// One wave per row (token, head); fp32 math; HBM-bound streaming kernel with 4-wide vector I/O.
#pragma once
#include "common.hpp"

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

// Narrow rows in the backward (D <= 4 * LPR): 64 / LPR rows per wave, every lane busy; per-workgroup dw partial row.
template <typename T, int LPR, bool GATE>
__global__ __launch_bounds__(256) void k_rmsnorm_gate_bwd_sub(const NormArgs a) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[4 * RPW][LPR * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / LPR, c = (lane % LPR) * 4;
    const long nw = (long)gridDim.x * 4 * RPW;
    const bool col = c < a.D;
    f32x4 w = {1.f, 1.f, 1.f, 1.f};
    if (a.w && col) w = *reinterpret_cast<const f32x4*>(a.w + c);
    f32x4 dwacc = {0.f, 0.f, 0.f, 0.f};
    for (long row0 = ((long)blockIdx.x * 4 + wave) * RPW; row0 < a.rows; row0 += nw) {
        const long row = row0 + sub;
        const bool live = row < a.rows && col;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f}, dyv = xv, gv = xv;
        if (live) {
            xv = Io<T>::ld4((const T*)a.x + row * a.ldx + c);
            dyv = Io<T>::ld4((const T*)a.dy + row * a.lddy + c);
            if (GATE) gv = Io<T>::ld4((const T*)a.g + row * a.ldg + c);
        }
        float ss = xv[0] * xv[0] + xv[1] * xv[1] + xv[2] * xv[2] + xv[3] * xv[3];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = 1.f / sqrtf(ss / (float)a.D + a.eps);
        f32x4 uv, xh;
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float s = GATE ? gv[t] / (1.f + __expf(-gv[t])) : 1.f;
            xh[t] = xv[t] * rstd;
            uv[t] = dyv[t] * w[t] * s;
            dot += uv[t] * xh[t];
            dwacc[t] += dyv[t] * xh[t] * s;
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        dot /= (float)a.D;
        if (live) {
            f32x4 dx, dg;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dx[t] = rstd * (uv[t] - xh[t] * dot);
                if (GATE) {
                    const float sg = 1.f / (1.f + __expf(-gv[t]));
                    dg[t] = dyv[t] * xh[t] * w[t] * sg * (1.f + gv[t] * (1.f - sg));
                }
            }
            Io<T>::st4((T*)a.dx + row * a.lddx + c, dx);
            if (GATE) Io<T>::st4((T*)a.dg + row * a.lddg + c, dg);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) red[wave * RPW + sub][c + t] = dwacc[t];
    __syncthreads();
    for (int cc = threadIdx.x; cc < a.D; cc += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4 * RPW; ++r) s += red[r][cc];
        a.dwp[(long)blockIdx.x * a.D + cc] = s;
    }
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
    // optional second output: y rotated by the token's rope angles (rope_apply, wan/mhla_utils.py:127-156 / :314):
    // consecutive channel pairs of every head; cos / sin [ntok][D/2] fp32, token = row % ntok
    float* yr;
    long ldyr;
    const float* rcos;
    const float* rsin;
    long ldr;
    int ntok, D;
    // backward
    const float* dy;     // gradient w.r.t. y   (may be nullptr)
    long lddy;
    const float* dyr;    // gradient w.r.t. yr  (may be nullptr)
    long lddyr;
    void* dx;            // gradient w.r.t. x, dtype of x
    long lddx;
    float* dwp;          // [gridDim.x][C] partial weight gradients
};

// 4 channel pairs starting at channel c of row `row`: rotate (fwd) or apply the transposed rotation (inv)
__device__ __forceinline__ void prologue_rope8(f32x4& a, f32x4& b, const PrologueArgs& p, long row, int c, bool inv) {
    const long tok = row % p.ntok;
    const int col = (c % p.D) >> 1;
    const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rcos + tok * p.ldr + col);
    f32x4 sn = *reinterpret_cast<const f32x4*>(p.rsin + tok * p.ldr + col);
    if (inv) sn = -sn;
    const f32x4 a0 = a, b0 = b;
    a[0] = a0[0] * cs[0] - a0[1] * sn[0]; a[1] = a0[0] * sn[0] + a0[1] * cs[0];
    a[2] = a0[2] * cs[1] - a0[3] * sn[1]; a[3] = a0[2] * sn[1] + a0[3] * cs[1];
    b[0] = b0[0] * cs[2] - b0[1] * sn[2]; b[1] = b0[0] * sn[2] + b0[1] * cs[2];
    b[2] = b0[2] * cs[3] - b0[3] * sn[3]; b[3] = b0[2] * sn[3] + b0[3] * cs[3];
}

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
                f32x4 y[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    y[u] = xv[i][u] * rstd;
                    if (a.w) y[u] *= *reinterpret_cast<const f32x4*>(a.w + c + 4 * u);
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[u][t] = fmaxf(y[u][t], 0.f) + a.eps;
                    *reinterpret_cast<f32x4*>(yr + c + 4 * u) = y[u];
                }
                if (a.yr) {
                    prologue_rope8(y[0], y[1], a, row, c, false);
                    *reinterpret_cast<f32x4*>(a.yr + row * a.ldyr + c) = y[0];
                    *reinterpret_cast<f32x4*>(a.yr + row * a.ldyr + c + 4) = y[1];
                }
            }
        }
    }
}

// rstd[row] = 1 / sqrt(mean_C(x[row]^2) + norm_eps): the one number per token that the RMSNorm of the q / k prologue needs.  With it the
// operator's kernels apply relu(x rstd w) + eps while they load the 16-bit projection (split.hpp PRO): the fp32 q / k tensors of
// k_qk_prologue are never written (mhla_blockmix_wan_pro_fwd).  One wave per token row, 16-byte loads.
struct RstdArgs {
    const void* x;
    long ldx;
    float* rstd;
    long rows;
    int C;
    float norm_eps;
};
template <typename T>
__global__ __launch_bounds__(256) void k_rms_rstd(const RstdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
        const T* xr = (const T*)a.x + row * a.ldx;
        float ss = 0.f;
        for (int c = lane * 8; c < a.C; c += 512) {
            const f32x4 x0 = Io<T>::ld4(xr + c), x1 = Io<T>::ld4(xr + c + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) ss += x0[t] * x0[t] + x1[t] * x1[t];
        }
        ss = wave_sum(ss);
        if (lane == 0) a.rstd[row] = 1.f / sqrtf(ss / (float)a.C + a.norm_eps);
    }
}

// Backward of the prologue: g = (dy + R^T dyr) . [u > 0] with u = x rstd w the pre-activation;
//   norm:  dx = rstd (g w - xhat mean_C(g w xhat)),  dw[c] += g xhat      (xhat = x rstd)
//   else:  dx = g
// One wave per row; per-workgroup partial dw rows (fixed order: deterministic), summed by the caller.
template <typename T, int NV>
__global__ __launch_bounds__(256) void k_qk_prologue_bwd(const PrologueArgs a) {
    __shared__ float red[4][NV * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    f32x4 dwacc[NV][2];
#pragma unroll
    for (int i = 0; i < NV; ++i) dwacc[i][0] = dwacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
        const T* xr = (const T*)a.x + row * a.ldx;
        f32x4 xv[NV][2], gv[NV][2];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 8;
            xv[i][0] = xv[i][1] = gv[i][0] = gv[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < a.C) {
                xv[i][0] = Io<T>::ld4(xr + c);
                xv[i][1] = Io<T>::ld4(xr + c + 4);
                if (a.dyr) {
                    gv[i][0] = *reinterpret_cast<const f32x4*>(a.dyr + row * a.lddyr + c);
                    gv[i][1] = *reinterpret_cast<const f32x4*>(a.dyr + row * a.lddyr + c + 4);
                    prologue_rope8(gv[i][0], gv[i][1], a, row, c, true);
                }
                if (a.dy) {
                    gv[i][0] += *reinterpret_cast<const f32x4*>(a.dy + row * a.lddy + c);
                    gv[i][1] += *reinterpret_cast<const f32x4*>(a.dy + row * a.lddy + c + 4);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) ss += xv[i][0][t] * xv[i][0][t] + xv[i][1][t] * xv[i][1][t];
        }
        float rstd = 1.f;
        if (a.norm) {
            ss = wave_sum(ss);
            rstd = 1.f / sqrtf(ss / (float)a.C + a.norm_eps);
        }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 8;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 w = {1.f, 1.f, 1.f, 1.f};
                if (a.w && c < a.C) w = *reinterpret_cast<const f32x4*>(a.w + c + 4 * u);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float xhat = xv[i][u][t] * rstd;
                    const float g = xhat * w[t] > 0.f ? gv[i][u][t] : 0.f;     // relu mask on the pre-activation
                    dwacc[i][u][t] += g * xhat;
                    gv[i][u][t] = g * w[t];
                    dot += gv[i][u][t] * xhat;
                }
            }
        }
        if (a.norm) dot = wave_sum(dot) / (float)a.C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < a.C) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 dx = gv[i][u];
                    if (a.norm)
#pragma unroll
                        for (int t = 0; t < 4; ++t) dx[t] = rstd * (gv[i][u][t] - xv[i][u][t] * rstd * dot);
                    Io<T>::st4((T*)a.dx + row * a.lddx + c + 4 * u, dx);
                }
            }
        }
    }
    if (a.dwp) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) red[wave][(lane + 64 * i) * 8 + 4 * u + t] = dwacc[i][u][t];
        __syncthreads();
        for (int c = threadIdx.x; c < a.C; c += 256)
            a.dwp[(long)blockIdx.x * a.C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
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
