// LePE: the depthwise K x K convolution over V that the DiT / ViT hosts add to the operator's output
// (mhla_dit/mhla/mhla.py:169, 246-247, 271-273: nn.Conv2d(dim, dim, 3, 1, 1, groups=dim);
//  mhla_image_classification/models/modules/attention/mhla.py:169: 5 x 5), computed directly on the token-major,
// block-major layout the operator uses: token n = m * S + s with block m = (py, px) on a pl x pl grid and in-block
// offset s = (by, bx) on a bl x bl grid is pixel (py * bl + by, px * bl + bx).  No NCHW permutes, no im2col.
// Channels are contiguous, so a thread owns a few consecutive channels of one token and walks the K*K neighbours
// (L1 / L2 hits: every token row is read K*K times by neighbouring threads).  HBM-bound streaming kernels.
#pragma once
#include "common.hpp"

namespace mhla {

struct LepeArgs {
    const void* x;      // [B, N, C] (strides xsb, xsn)   forward: v            backward-data: dout
    long xsb, xsn;
    const float* w;     // [K*K][C] fp32 (tap-major)
    const float* bias;  // [C] or nullptr
    const void* add;    // optional [B, N, C] tensor added to the result (the operator's output), or nullptr
    long asb, asn;
    void* y;            // [B, N, C]
    long ysb, ysn;
    int B, pl, bl, C, K;
    int flip;           // 1: correlate with the flipped kernel (gradient w.r.t. the input)
};

__device__ __forceinline__ void lepe_decode(int n, int pl, int bl, int& yy, int& xx) {
    const int S = bl * bl, m = n / S, s = n - m * S;
    const int py = m / pl, px = m - py * pl, by = s / bl, bx = s - by * bl;
    yy = py * bl + by;
    xx = px * bl + bx;
}
__device__ __forceinline__ int lepe_token(int yy, int xx, int pl, int bl) {
    const int py = yy / bl, by = yy - py * bl, px = xx / bl, bx = xx - px * bl;
    return (py * pl + px) * bl * bl + by * bl + bx;
}
// Block / in-block coordinates of a token and of its neighbours without divisions per tap: one axis at a time, a
// displacement d moves the in-block coordinate and carries into the block coordinate.
struct LepePos { int py, px, by, bx; };
__device__ __forceinline__ LepePos lepe_pos(int n, int pl, int bl) {
    const int S = bl * bl, m = n / S, s = n - m * S;
    LepePos p;
    p.py = m / pl; p.px = m - p.py * pl; p.by = s / bl; p.bx = s - p.by * bl;
    return p;
}
// coordinate (block b, offset o) displaced by d; returns false when it leaves the image
__device__ __forceinline__ bool lepe_shift(int b, int o, int d, int pl, int bl, int& b2, int& o2) {
    o2 = o + d;
    b2 = b;
    while (o2 < 0) { o2 += bl; --b2; }
    while (o2 >= bl) { o2 -= bl; ++b2; }
    return b2 >= 0 && b2 < pl;
}

// grid (ceil(N * C/8 / 256), B); a thread: 8 channels of one token.
// Every tap's row is loaded UNCONDITIONALLY (taps that leave the image read the token itself and enter with weight zero): with
// `continue` around the loads hipcc waited for each load before it issued the next (s_waitcnt vmcnt(0) at every join) -- K * K
// dependent round trips per token; now the K * K row loads and weight loads of a token travel together.
template <typename T, int K>
__global__ __launch_bounds__(256, 4) void k_lepe2d(const LepeArgs a) {
    const int CG = a.C / 8, N = a.pl * a.pl * a.bl * a.bl;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * CG) return;
    const int n = (int)(gid / CG), c = (int)(gid - (long)n * CG) * 8, b = blockIdx.y;
    const LepePos pos = lepe_pos(n, a.pl, a.bl);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (a.bias) {
        acc0 = *reinterpret_cast<const f32x4*>(a.bias + c);
        acc1 = *reinterpret_cast<const f32x4*>(a.bias + c + 4);
    }
    const T* xb = (const T*)a.x + b * a.xsb + c;
    constexpr int R = K / 2;
    const int S = a.bl * a.bl;
    f32x4 ad0 = {0.f, 0.f, 0.f, 0.f}, ad1 = ad0;
    if (a.add) {
        const T* p = (const T*)a.add + b * a.asb + (long)n * a.asn + c;
        ad0 = Io<T>::ld4(p);
        ad1 = Io<T>::ld4(p + 4);
    }
    constexpr int RB = 1;   // tap rows per batch of loads: one row (all nine taps at once: 131 VGPRs, three waves per SIMD, slower)
#pragma unroll 1   // (a real loop: unrolled, hipcc hoists every row's loads to the top and spills)
    for (int d0 = 0; d0 < K; d0 += RB) {
        f32x4 x0[RB * K], x1[RB * K], w0[RB * K], w1[RB * K];
        bool ok[RB * K];
#pragma unroll
        for (int dr = 0; dr < RB; ++dr) {
            const int dy = d0 + dr;
            int py2, by2;
            const bool oky = lepe_shift(pos.py, pos.by, dy - R, a.pl, a.bl, py2, by2);
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                int px2, bx2;
                const int t = dr * K + dx;
                ok[t] = lepe_shift(pos.px, pos.bx, dx - R, a.pl, a.bl, px2, bx2) && oky;
                const int tap = a.flip ? (K - 1 - dy) * K + (K - 1 - dx) : dy * K + dx;
                const int nb = ok[t] ? (py2 * a.pl + px2) * S + by2 * a.bl + bx2 : n;
                const T* p = xb + (long)nb * a.xsn;
                x0[t] = Io<T>::ld4(p);
                x1[t] = Io<T>::ld4(p + 4);
                w0[t] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c);
                w1[t] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c + 4);
            }
        }
#pragma unroll
        for (int t = 0; t < RB * K; ++t) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};   // (a select, not a multiplication by 0: the filler row may hold inf / NaN)
            acc0 += (ok[t] ? x0[t] : z) * w0[t];
            acc1 += (ok[t] ? x1[t] : z) * w1[t];
        }
    }
    acc0 += ad0;
    acc1 += ad1;
    T* yp = (T*)a.y + b * a.ysb + (long)n * a.ysn + c;
    Io<T>::st4(yp, acc0);
    Io<T>::st4(yp + 4, acc1);
}

// ---- runs of four tokens (3 x 3, 16-bit tensors, block_len % 4 == 0, 16-byte aligned rows) --------------------------------------
// Four consecutive tokens of a block row are neighbours in the image and in memory.  A thread that owns 8 channels of such a run
// reads a 3 x 6 window of rows (18 loads of 16 bytes) and its 9 x 8 weights ONCE for four outputs: 9 load instructions per token
// where k_lepe2d issues 36 (two 8-byte halves of every tap's row and weights per token) -- the kernels are bound by load issue, not
// by HBM (57 MB in 26 us at the DiT-XL/2 shape).
template <typename T> __device__ __forceinline__ void lepe_unpack8(uint4 r, f32x4& a, f32x4& b);
template <> __device__ __forceinline__ void lepe_unpack8<bf16_t>(uint4 r, f32x4& a, f32x4& b) {
    a = f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    b = f32x4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
template <> __device__ __forceinline__ void lepe_unpack8<f16_t>(uint4 r, f32x4& a, f32x4& b) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 h = __builtin_bit_cast(h8, r);
    a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    b = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
}
// the 3 x 6 window of a run starting at token n0 = (block (py, px), row by, column bx0): packed rows + inside-the-image flags
template <typename T>
__device__ __forceinline__ void lepe_window(uint4 (&win)[3][6], bool (&ok)[3][6], const T* __restrict__ xb, long xsn, int n0,
                                            const LepePos& pos, int pl, int bl) {
    const int S = bl * bl;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int py2, by2;
        const bool oky = lepe_shift(pos.py, pos.by, dy - 1, pl, bl, py2, by2);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            int px2, bx2;
            ok[dy][j] = lepe_shift(pos.px, pos.bx, j - 1, pl, bl, px2, bx2) && oky;
            const int nb = ok[dy][j] ? (py2 * pl + px2) * S + by2 * bl + bx2 : n0;   // (outside: the run's first row, selected away)
            win[dy][j] = gld<uint4>(xb + (long)nb * xsn);
        }
    }
}

// grid (ceil(N / 4 * C / 8 / 256), B); a thread: 8 channels of four consecutive tokens
template <typename T>
__global__ __launch_bounds__(256, 2) void k_lepe2d_run4(const LepeArgs a) {
    const int CG = a.C / 8, N = a.pl * a.pl * a.bl * a.bl;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)(N / 4) * CG) return;
    const int n0 = (int)(gid / CG) * 4, c = (int)(gid % CG) * 8, b = blockIdx.y;
    const LepePos pos = lepe_pos(n0, a.pl, a.bl);
    const T* xb = (const T*)a.x + b * a.xsb + c;
    uint4 win[3][6];
    bool ok[3][6];
    lepe_window<T>(win, ok, xb, a.xsn, n0, pos, a.pl, a.bl);
    f32x4 w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tap = a.flip ? 8 - t : t;
        w0[t] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c);
        w1[t] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c + 4);
    }
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (a.bias) {
        b0 = *reinterpret_cast<const f32x4*>(a.bias + c);
        b1 = *reinterpret_cast<const f32x4*>(a.bias + c + 4);
    }
    uint4 addv[4];
    if (a.add) {
#pragma unroll
        for (int t = 0; t < 4; ++t) addv[t] = gld<uint4>((const T*)a.add + b * a.asb + (long)(n0 + t) * a.asn + c);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 acc0 = b0, acc1 = b1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                f32x4 x0, x1;
                lepe_unpack8<T>(win[dy][t + dx], x0, x1);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};   // (a select, not a multiplication by 0: the filler row may hold inf / NaN)
                acc0 += (ok[dy][t + dx] ? x0 : z) * w0[dy * 3 + dx];
                acc1 += (ok[dy][t + dx] ? x1 : z) * w1[dy * 3 + dx];
            }
        if (a.add) {
            f32x4 a0, a1;
            lepe_unpack8<T>(addv[t], a0, a1);
            acc0 += a0;
            acc1 += a1;
        }
        T* yp = (T*)a.y + b * a.ysb + (long)(n0 + t) * a.ysn + c;
        Io<T>::st4(yp, acc0);
        Io<T>::st4(yp + 4, acc1);
    }
}

// Weight / bias gradient: dw[tap][c] = sum_{b, n} dout[b, n, c] x[b, nbr(n, tap), c], db[c] = sum dout[b, n, c].
// grid (ceil(C / (8 CH)) / 4 rounded up, slices); a wave = 8 channel groups of CH channels x 8 token lanes; each thread walks
// every 8th token of its slice, the 8 token lanes are summed by shuffles; partials part[slice][K*K + 1][C] (row K*K = bias)
// are summed in a fixed order by k_lepe2d_wgrad_reduce: deterministic.
struct LepeWgradArgs {
    const void* x;      // v
    long xsb, xsn;
    const void* g;      // dout
    long gsb, gsn;
    float* part;        // [slices][K*K + 1][C]
    int B, pl, bl, C, K, slices;
};

template <typename T, int K, int CH>   // CH: channels per thread (4 or 8)
__global__ __launch_bounds__(256) void k_lepe2d_wgrad(const LepeWgradArgs a) {
    constexpr int NV = CH / 4, NA = K * K + 1;
    const int N = a.pl * a.pl * a.bl * a.bl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane >> 3;
    const int c = ((blockIdx.x * 4 + wave) * 8 + (lane & 7)) * CH, slice = blockIdx.y;
    const bool live = c < a.C;
    const long total = (long)a.B * N, per = (total + a.slices - 1) / a.slices;
    const long t0 = slice * per, t1 = min(total, t0 + per);
    f32x4 acc[NA][NV];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int u = 0; u < NV; ++u) acc[i][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (long t = t0 + tl; t < t1; t += 8) {
            const int b = (int)(t / N), n = (int)(t - (long)b * N);
            const LepePos pos = lepe_pos(n, a.pl, a.bl);
            f32x4 g[NV];
            const T* gp = (const T*)a.g + b * a.gsb + (long)n * a.gsn + c;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                g[u] = Io<T>::ld4(gp + 4 * u);
                acc[K * K][u] += g[u];
            }
            const T* xb = (const T*)a.x + b * a.xsb + c;
            const int S = a.bl * a.bl;
            // all K * K neighbour rows requested before any is used (taps outside the image: the token itself, weight zero) --
            // behind `if (inside)` every load was waited for before the next was issued: 80 dependent round trips per thread
            f32x4 xv[K * K][NV];
            float mk[K * K];
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
                int py2, by2;
                const bool oky = lepe_shift(pos.py, pos.by, dy - K / 2, a.pl, a.bl, py2, by2);
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    int px2, bx2;
                    const bool ok = lepe_shift(pos.px, pos.bx, dx - K / 2, a.pl, a.bl, px2, bx2) && oky;
                    const int nb = ok ? (py2 * a.pl + px2) * S + by2 * a.bl + bx2 : n;
                    const T* p = xb + (long)nb * a.xsn;
                    mk[dy * K + dx] = ok ? 1.f : 0.f;
#pragma unroll
                    for (int u = 0; u < NV; ++u) xv[dy * K + dx][u] = Io<T>::ld4(p + 4 * u);
                }
            }
#pragma unroll
            for (int i = 0; i < K * K; ++i)
#pragma unroll
                for (int u = 0; u < NV; ++u) acc[i][u] += g[u] * (mk[i] != 0.f ? xv[i][u] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    // sum the 8 token lanes (lane bits 3..5)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[i][u][t];
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                acc[i][u][t] = v;
            }
    if (live && tl == 0) {
        float* out = a.part + (long)slice * NA * a.C + c;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int u = 0; u < NV; ++u) *reinterpret_cast<f32x4*>(out + (long)i * a.C + 4 * u) = acc[i][u];
    }
}

// dwb[(K*K + 1)][C] = sum over slices, fixed order: 64 elements x 4 slice lanes per workgroup, then a 4-way LDS sum
__global__ __launch_bounds__(256) void k_lepe2d_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dwb, int rows_c, int slices) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, pl = threadIdx.x >> 6, i = blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < rows_c) {
        const int per = (slices + 3) / 4, p0 = pl * per, p1 = min(slices, p0 + per);
        for (int p = p0; p < p1; ++p) s += part[(long)p * rows_c + i];
    }
    red[pl][el] = s;
    __syncthreads();
    if (pl == 0 && i < rows_c) dwb[i] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// -------------------------------------------------------------------------------------------------
// 3-D LePE of the Wan host: nn.Conv3d(dim, dim, (3,3,3), 1, (1,1,1), groups=dim) over V laid out as a video
// (wan/mhla_utils.py:199-201, 349-352: 'b (f h w) c -> b c f h w' and back).  Wan's tokens are in raster order
// n = (f * H + h) * W + w, channels contiguous: same scheme as the 2-D kernels, 27 taps, no permutes.
// -------------------------------------------------------------------------------------------------
struct Lepe3dArgs {
    const void* x;      // [B, N, C] (strides xsb, xsn)   forward: v            backward-data: dout
    long xsb, xsn;
    const float* w;     // [27][C] fp32 (tap-major: tap = (df * 3 + dh) * 3 + dw)
    const float* bias;  // [C] or nullptr
    const void* add;    // optional [B, N, C] tensor added to the result, or nullptr
    long asb, asn;
    void* y;
    long ysb, ysn;
    int B, F, H, W, C;
    int flip;           // 1: correlate with the flipped kernel (gradient w.r.t. the input)
};

// grid (ceil(N * C/8 / 256), B); a thread: 8 channels of one token
template <typename T>
__global__ __launch_bounds__(256, 3) void k_lepe3d(const Lepe3dArgs a) {
    const int CG = a.C / 8, HW = a.H * a.W, N = a.F * HW;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * CG) return;
    const int n = (int)(gid / CG), c = (int)(gid - (long)n * CG) * 8, b = blockIdx.y;
    const int f = n / HW, hw = n - f * HW, h = hw / a.W, w = hw - h * a.W;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (a.bias) {
        acc0 = *reinterpret_cast<const f32x4*>(a.bias + c);
        acc1 = *reinterpret_cast<const f32x4*>(a.bias + c + 4);
    }
    const T* xb = (const T*)a.x + b * a.xsb + c;
    // one line of three taps at a time, its loads unconditional (taps outside the video: the token itself, selected away)
    f32x4 ad0 = {0.f, 0.f, 0.f, 0.f}, ad1 = ad0;
    if (a.add) {
        const T* p = (const T*)a.add + b * a.asb + (long)n * a.asn + c;
        ad0 = Io<T>::ld4(p);
        ad1 = Io<T>::ld4(p + 4);
    }
#pragma unroll 1   // (real loops: unrolled, hipcc hoists all 27 taps' loads to the top and spills)
    for (int df = 0; df < 3; ++df) {
        const int f2 = f + df - 1;
        const bool okf = f2 >= 0 && f2 < a.F;
#pragma unroll 1
        for (int dh = 0; dh < 3; ++dh) {
            const int h2 = h + dh - 1;
            const bool okh = okf && h2 >= 0 && h2 < a.H;
            f32x4 x0[3], x1[3], w0[3], w1[3];
            bool ok[3];
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int w2 = w + dw - 1;
                ok[dw] = okh && w2 >= 0 && w2 < a.W;
                const int tap = a.flip ? 26 - ((df * 3 + dh) * 3 + dw) : (df * 3 + dh) * 3 + dw;
                const T* p = xb + (long)(ok[dw] ? (f2 * a.H + h2) * a.W + w2 : n) * a.xsn;
                x0[dw] = Io<T>::ld4(p);
                x1[dw] = Io<T>::ld4(p + 4);
                w0[dw] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c);
                w1[dw] = *reinterpret_cast<const f32x4*>(a.w + (long)tap * a.C + c + 4);
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc0 += (ok[dw] ? x0[dw] : z) * w0[dw];
                acc1 += (ok[dw] ? x1[dw] : z) * w1[dw];
            }
        }
    }
    acc0 += ad0;
    acc1 += ad1;
    T* yp = (T*)a.y + b * a.ysb + (long)n * a.ysn + c;
    Io<T>::st4(yp, acc0);
    Io<T>::st4(yp + 4, acc1);
}

// Weight / bias gradient, as k_lepe2d_wgrad: a wave = 8 channel groups of 4 channels x 8 token lanes, every 8th token of the
// slice per lane, shuffle sum over the token lanes; part[slice][28][C] (row 27 = bias) summed by k_lepe2d_wgrad_reduce.
struct Lepe3dWgradArgs {
    const void* x;      // v
    long xsb, xsn;
    const void* g;      // dout
    long gsb, gsn;
    float* part;        // [slices][28][C]
    int B, F, H, W, C, slices;
};

template <typename T>
__global__ __launch_bounds__(256) void k_lepe3d_wgrad(const Lepe3dWgradArgs a) {
    constexpr int NA = 28;
    const int HW = a.H * a.W, N = a.F * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane >> 3;
    const int c = ((blockIdx.x * 4 + wave) * 8 + (lane & 7)) * 4, slice = blockIdx.y;
    const bool live = c < a.C;
    const long total = (long)a.B * N, per = (total + a.slices - 1) / a.slices;
    const long t0 = slice * per, t1 = min(total, t0 + per);
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (long t = t0 + tl; t < t1; t += 8) {
            const int b = (int)(t / N), n = (int)(t - (long)b * N);
            const int f = n / HW, hw = n - f * HW, h = hw / a.W, w = hw - h * a.W;
            const f32x4 g = Io<T>::ld4((const T*)a.g + b * a.gsb + (long)n * a.gsn + c);
            acc[27] += g;
            const T* xb = (const T*)a.x + b * a.xsb + c;
#pragma unroll
            for (int df = 0; df < 3; ++df) {   // one frame plane at a time: its nine loads unconditional and in flight together
                const int f2 = f + df - 1;
                const bool okf = f2 >= 0 && f2 < a.F;
                f32x4 xv[9];
                float mk[9];
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const int h2 = h + dh - 1;
                    const bool okh = okf && h2 >= 0 && h2 < a.H;
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const int w2 = w + dw - 1;
                        const bool ok = okh && w2 >= 0 && w2 < a.W;
                        mk[dh * 3 + dw] = ok ? 1.f : 0.f;
                        xv[dh * 3 + dw] = Io<T>::ld4(xb + (long)(ok ? (f2 * a.H + h2) * a.W + w2 : n) * a.xsn);
                    }
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[df * 9 + t] += g * (mk[t] != 0.f ? xv[t] : f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = acc[i][t];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[i][t] = v;
        }
    if (live && tl == 0) {
        float* out = a.part + (long)slice * NA * a.C + c;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(out + (long)i * a.C) = acc[i];
    }
}

}  // namespace mhla
