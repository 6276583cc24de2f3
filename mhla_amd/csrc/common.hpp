// Shared device helpers for the gfx950 MHLA kernels (wave64, MFMA, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mhla {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct bf16_t { unsigned short v; };
struct f16_t { _Float16 v; };

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {   // round-to-nearest-even, NaN preserved
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// hardware conversions (gfx950: v_cvt_pk_bf16_f32, round-to-nearest-even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short cvt_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// Loads / stores through an explicit global (address space 1) pointer.  Pointers that come out of an argument struct or a
// select are generic to the compiler: it then emits flat_load, which counts on vmcnt AND lgkmcnt and may complete out of
// order, so every use is preceded by s_waitcnt vmcnt(0) lgkmcnt(0) -- software pipelines of loads collapse into
// load-all / wait-all.  global_load counts on vmcnt only, in order, and the compiler schedules with vmcnt(N).
#define MHLA_GLOBAL_AS __attribute__((address_space(1)))
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T gld(const void* p) { return *(const MHLA_GLOBAL_AS T*)(p); }
template <typename T> __device__ __forceinline__ void gst(void* p, T v) { *(MHLA_GLOBAL_AS T*)(p) = v; }
// uint4 / uint2 are class types (no address-space-qualified copy): go through the native vector types
template <> __device__ __forceinline__ uint4 gld<uint4>(const void* p) {
    const u32x4_t v = *(const MHLA_GLOBAL_AS u32x4_t*)(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ uint2 gld<uint2>(const void* p) {
    const u32x2_t v = *(const MHLA_GLOBAL_AS u32x2_t*)(p);
    return make_uint2(v[0], v[1]);
}
template <> __device__ __forceinline__ void gst<uint4>(void* p, uint4 v) {
    *(MHLA_GLOBAL_AS u32x4_t*)(p) = u32x4_t{v.x, v.y, v.z, v.w};
}
template <> __device__ __forceinline__ void gst<uint2>(void* p, uint2 v) {
    *(MHLA_GLOBAL_AS u32x2_t*)(p) = u32x2_t{v.x, v.y};
}

// Streaming (touch-once) variants: the nontemporal hint marks the lines evict-first, so that token rows which one workgroup
// reads once do not displace the block summaries that later kernels (or other workgroups) read again.  MHLA_NT=0 at build
// time disables them.
#ifndef MHLA_NT
#define MHLA_NT 1
#endif
__device__ __forceinline__ uint4 gld_stream16(const void* p) {
#if MHLA_NT
    const u32x4_t v = __builtin_nontemporal_load((const MHLA_GLOBAL_AS u32x4_t*)(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return gld<uint4>(p);
#endif
}
__device__ __forceinline__ void gst_stream16(void* p, uint4 v) {
#if MHLA_NT
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, (MHLA_GLOBAL_AS u32x4_t*)(p));
#else
    gst<uint4>(p, v);
#endif
}

// LDS-DMA: 16 bytes per lane from `gsrc` (per lane) to LDS byte address `lds_dst` + 16 lane (`lds_dst` wave-uniform).  Written as
// asm so that hipcc does not see an LDS write in flight: with the builtin it puts `s_waitcnt vmcnt(0)` in front of every
// transposed LDS read and every LDS write that follows, which drains the copies the kernel wants to keep in flight.  The copy is
// therefore NOT in the compiler's wait bookkeeping either: the kernel counts it itself (s_waitcnt vmcnt(N), then a barrier,
// then the reads).  M0 holds the destination and is restored in the same statement (it is compiler-reserved).
// Used by ONE kernel (sp::k_sp_mixr_dma), whose wait sites count their outstanding copies in comments next to each wait_vmcnt<N>.
// Validated with AMD clang 22.0.0git (roc-7.2.0, HIP 7.2.26015): the GPU suite, and bit for bit against the compiler-managed form --
// build with -DMHLA_GLDS_BUILTIN=1 (tools/build_variant.sh glds -DMHLA_GLDS_BUILTIN=1; tools/glds_check.py compares the two
// libraries' results on the 256 x 16 shape), where the copy is the builtin and hipcc inserts its own (conservative) waits.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
#ifdef MHLA_GLDS_BUILTIN
    __builtin_amdgcn_global_load_lds((const MHLA_GLOBAL_AS void*)gsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_dst, 16, 0, 0);
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// 4-element vector I/O per dtype (16 B for f32, 8 B for 16-bit types).  (Nontemporal variants of these loads were measured on
// the split-operand / generic kernels in round 2: 1-11 % slower on every shape, so these paths keep regular loads.)
template <typename T> struct Io;
template <> struct Io<float> {
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Io<bf16_t> {
    static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        f32x4 v;
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        return v;
    }
    static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
        uint2 r;
        r.x = pack_bf16x2(v[0], v[1]);
        r.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = r;
    }
};
template <> struct Io<f16_t> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld4(const f16_t* p) {
        const h4 r = *reinterpret_cast<const h4*>(p);
        f32x4 v; v[0] = (float)r[0]; v[1] = (float)r[1]; v[2] = (float)r[2]; v[3] = (float)r[3];
        return v;
    }
    static __device__ __forceinline__ void st4(f16_t* p, f32x4 v) {
        h4 r; r[0] = (_Float16)v[0]; r[1] = (_Float16)v[1]; r[2] = (_Float16)v[2]; r[3] = (_Float16)v[3];
        *reinterpret_cast<h4*>(p) = r;
    }
};

// What a 16-bit store of x loses: x - fl_T(x), as bf16 (the "lo" half of a result kept at fp32 grade next to its stored value).
// The block-mix forward writes it for O beside the output tensor when the backward will want the row dots dO . O at the
// reference's fp32 accuracy (the rounding of O alone costs them 2e-3, which small block counts do not average away).
template <typename T> __device__ __forceinline__ float stored_value(float x);
template <> __device__ __forceinline__ float stored_value<float>(float x) { return x; }
template <> __device__ __forceinline__ float stored_value<bf16_t>(float x) { return bf16_to_f32(cvt_bf16(x)); }
template <> __device__ __forceinline__ float stored_value<f16_t>(float x) { return (float)(_Float16)x; }
template <typename T> __device__ __forceinline__ uint2 store_residual4(f32x4 x) {
    return make_uint2(pack_bf16x2(x[0] - stored_value<T>(x[0]), x[1] - stored_value<T>(x[1])),
                      pack_bf16x2(x[2] - stored_value<T>(x[2]), x[3] - stored_value<T>(x[3])));
}

// h16: a 2-byte storage format of block / chunk summaries -- an fp16 payload x one power-of-two multiplier per group (a block row of
// the block-mixing operator, split.hpp; a 16 x 64 strip of a chunk tile of the causal operator, causal_bf16.hpp): 11 significand bits.
// decode multiplier of a row whose largest magnitude is mx: 2^(floor(log2 mx) - 14), exponent field clamped to [1, 240]
__device__ __forceinline__ float h16_mult_from_max(float mx) {
    const unsigned e = (__float_as_uint(mx) >> 23) & 0xffu;
    return __uint_as_float((e < 15u ? 1u : (e > 254u ? 240u : e - 14u)) << 23);
}
// ... of a row bounded by 2^15 beta: the power of two >= beta, field clamped to [1, 253]
__device__ __forceinline__ float h16_mult_from_bound(float beta) {
    unsigned f = (__float_as_uint(beta) + 0x7fffffu) >> 23;
    f = f < 1u ? 1u : (f > 253u ? 253u : f);
    return __uint_as_float(f << 23);
}
__device__ __forceinline__ float h16_inv(float m) { return __uint_as_float(0x7f000000u - __float_as_uint(m)); }   // 1 / m, exact (field 254 - f)
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned h16_pack2(float a, float b) {
    const h16x2_t h = {(_Float16)a, (_Float16)b};   // (round to nearest even)
    return __builtin_bit_cast(unsigned, h);
}
// 8 values -> their 16-byte payload piece (inv = 1 / m)
__device__ __forceinline__ uint4 h16_pack8(const f32x4& a, const f32x4& b, float inv) {
    return make_uint4(h16_pack2(a[0] * inv, a[1] * inv), h16_pack2(a[2] * inv, a[3] * inv), h16_pack2(b[0] * inv, b[1] * inv), h16_pack2(b[2] * inv, b[3] * inv));
}
// the bf16 hi / lo operands of 8 stored values: x = payload * m, hi = its top 16 bits, lo = x - hi (<= 3 significant bits: exact)
__device__ __forceinline__ void h16_split8(const uint4& pay, float m, uint4& hi, uint4& lo) {
    const unsigned w[4] = {pay.x, pay.y, pay.z, pay.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const h16x2_t p = __builtin_bit_cast(h16x2_t, w[j]);
        const float x0 = (float)p[0] * m, x1 = (float)p[1] * m;
        const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
        h[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        const float d0 = x0 - __uint_as_float(u0 & 0xffff0000u), d1 = x1 - __uint_as_float(u1 & 0xffff0000u);
        l[j] = __builtin_amdgcn_perm(__float_as_uint(d1), __float_as_uint(d0), 0x07060302u);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Token-major view [B, N, H, D]; element strides; D contiguous.
struct View {
    const void* ptr;
    long sb, sn, sh;
};
struct MView {
    void* ptr;
    long sb, sn, sh;
};

// LDS leading dimensions (in floats) for fp32 16x16x4 MFMA operand reads with ds_read_b32:
//  k-major tile  T[k][x]  (lanes 0-15 walk x, lane>>4 walks k):  ld % 32 == 16  -> conflict-free
//  x-major tile  T[x][k]  (lanes 0-15 walk rows x, lane>>4 walks k): ld == 2*odd -> conflict-free
__host__ __device__ constexpr int ld_kmajor(int dp) { return ((dp + 15) / 32) * 32 + 16; }
__host__ __device__ constexpr int ld_xmajor(int dp) { return dp + 2; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), in every lane: four rotate-and-add steps in the VALU (row_ror:8/4/2/1)
// instead of four ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
    return v;
}

// Row of block-major position p in memory.
__device__ __forceinline__ long tok_row(const int* __restrict__ idx, long p) { return idx ? (long)idx[p] : p; }

// Load a [rows x (DP/4 vec4)] tile of a token view into LDS as fp32.
//   dst[r*ld + c]  r < rows_fill (rows >= rows_valid zero), c < DP (cols >= cols_valid zero)
// base points at (b, token 0, h, d0).  `p0` is the block-major position of row 0.
template <typename T, int DP, bool RELU>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, int ld, const T* __restrict__ base, long sn,
                                          const int* __restrict__ idx, long p0, int rows_valid, int rows_fill,
                                          int cols_valid, float eps, int tid, int nthreads) {
    constexpr int CV = DP / 4, U = 4;
    // all global loads of a batch first, LDS stores after: interleaved load/store pairs are serialised by the compiler
    for (int v0 = tid; v0 < rows_fill * CV; v0 += nthreads * U) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * nthreads, r = v / CV, c = (v - r * CV) * 4;
            x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < rows_fill * CV && r < rows_valid && c < cols_valid) {
                x[u] = Io<T>::ld4(base + tok_row(idx, p0 + r) * sn + c);
                if (RELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[u][i] = fmaxf(x[u][i], 0.f) + eps;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * nthreads, r = v / CV, c = (v - r * CV) * 4;
            if (v < rows_fill * CV) {
                float* d = dst + r * ld + c;
                if ((ld & 3) == 0) {
                    *reinterpret_cast<f32x4*>(d) = x[u];
                } else {
                    *reinterpret_cast<f32x2*>(d) = f32x2{x[u][0], x[u][1]};
                    *reinterpret_cast<f32x2*>(d + 2) = f32x2{x[u][2], x[u][3]};
                }
            }
        }
    }
}

// Load a [rows x cols] fp32 matrix (row stride lds_src) into LDS, zero padded to [rows_fill x DP].
template <int DP>
__device__ __forceinline__ void load_mat_f32(float* __restrict__ dst, int ld, const float* __restrict__ src,
                                             long src_ld, int rows_valid, int rows_fill, int cols_valid, int tid,
                                             int nthreads, bool vec_ok) {
    constexpr int CV = DP / 4, U = 4;
    for (int v0 = tid; v0 < rows_fill * CV; v0 += nthreads * U) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * nthreads, r = v / CV, c = (v - r * CV) * 4;
            x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < rows_fill * CV && r < rows_valid && c < cols_valid) {
                const float* s = src + (long)r * src_ld + c;
                if (vec_ok && c + 3 < cols_valid) {
                    x[u] = *reinterpret_cast<const f32x4*>(s);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c + i < cols_valid) x[u][i] = s[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = v0 + u * nthreads, r = v / CV, c = (v - r * CV) * 4;
            if (v < rows_fill * CV) {
                float* d = dst + r * ld + c;
                if ((ld & 3) == 0) {
                    *reinterpret_cast<f32x4*>(d) = x[u];
                } else {
                    *reinterpret_cast<f32x2*>(d) = f32x2{x[u][0], x[u][1]};
                    *reinterpret_cast<f32x2*>(d + 2) = f32x2{x[u][2], x[u][3]};
                }
            }
        }
    }
}

// Store a staged fp32 tile (ld multiple of 4) to a token view with dtype conversion.
template <typename T, int DP>
__device__ __forceinline__ void store_tile(T* __restrict__ base, long sn, const int* __restrict__ idx, long p0,
                                           const float* __restrict__ src, int ld, int rows_valid, int cols_valid,
                                           int tid, int nthreads) {
    constexpr int CV = DP / 4;
    for (int v = tid; v < rows_valid * CV; v += nthreads) {
        const int r = v / CV, c = (v - r * CV) * 4;
        if (c < cols_valid) {
            f32x4 x = *reinterpret_cast<const f32x4*>(src + r * ld + c);
            Io<T>::st4(base + tok_row(idx, p0 + r) * sn + c, x);
        }
    }
}

}  // namespace mhla
