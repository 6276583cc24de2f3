// C ABI, the operator's neighbours: per-head RMSNorm x gate, LePE depthwise convolutions, q/k prologues, feature map + rotary.
#include "capi_common.hpp"
#include "epilogue.hpp"
#include "lepe.hpp"

using namespace mhla;
using namespace mhla::capi;

namespace {

int norm_check(const void* x, const void* y, int64_t rows, int D, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (rows <= 0 || D <= 0 || (D & 3) || D > 512) return fail(MHLA_EINVAL, "rows=%lld D=%d: need D %% 4 == 0 and D <= 512", (long long)rows, D);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    return MHLA_OK;
}
int norm_grid(int64_t rows) {   // backward: one dw partial row per workgroup, so the grid is capped
    int64_t g = (rows + 3) / 4;
    return (int)(g < 8192 ? g : 8192);
}
int norm_fwd_grid(int64_t rows, int rows_per_wave) {   // forward: a wave per row group, no grid-stride serialisation
    int64_t g = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
    return (int)(g < (1 << 20) ? g : (1 << 20));
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// per-head RMSNorm x gate
// ---------------------------------------------------------------------------------------------
int64_t mhla_rmsnorm_gate_dw_rows(int64_t rows) { return norm_grid(rows); }

int mhla_rmsnorm_gate_fwd(const void* x, int64_t ldx, const void* g, int64_t ldg, const float* w, void* y, int64_t ldy,
                          float* rstd, int64_t rows, int D, float eps, int dtype, void* stream) {
    RC(norm_check(x, y, rows, D, dtype));
    if ((ldx | ldy | (g ? ldg : 0)) & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    NormArgs a{};
    a.x = x; a.ldx = ldx; a.g = g; a.ldg = ldg; a.w = w; a.y = y; a.ldy = ldy; a.rstd = rstd; a.rows = rows; a.D = D; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(norm_fwd_grid(rows, 1));
    DISPATCH_T(dtype, {
        if (D <= 64) {
            const dim3 g4(norm_fwd_grid(rows, 4));
            if (g) RC(launch(k_rmsnorm_gate_fwd_sub<ET, 16, true>, g4, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd_sub<ET, 16, false>, g4, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else if (D <= 128) {
            const dim3 g2(norm_fwd_grid(rows, 2));
            if (g) RC(launch(k_rmsnorm_gate_fwd_sub<ET, 32, true>, g2, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd_sub<ET, 32, false>, g2, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else if (D <= 256) {
            if (g) RC(launch(k_rmsnorm_gate_fwd<ET, 1, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd<ET, 1, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        } else {
            if (g) RC(launch(k_rmsnorm_gate_fwd<ET, 2, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
            else   RC(launch(k_rmsnorm_gate_fwd<ET, 2, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_fwd", a));
        }
    });
    return MHLA_OK;
}

// ---------------------------------------------------------------------------------------------
// LePE depthwise convolution on the block-major token layout
// ---------------------------------------------------------------------------------------------
static int lepe_check(const void* x, const void* y, int B, int pl, int bl, int C, int K, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (B <= 0 || pl <= 0 || bl <= 0 || C <= 0 || (C & 7)) return fail(MHLA_EINVAL, "B=%d pieces_len=%d block_len=%d C=%d: need positive sizes and C %% 8 == 0", B, pl, bl, C);
    if (K != 3 && K != 5) return fail(MHLA_ENOTSUP, "kernel size %d: 3 (DiT) and 5 (ViT) are supported", K);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (B > 65535) return fail(MHLA_ENOTSUP, "B=%d exceeds grid limit 65535", B);
    return MHLA_OK;
}
constexpr int LEPE_SLICES = 128;

int mhla_lepe2d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias, const void* add,
                int64_t add_sb, int64_t add_sn, void* y, int64_t y_sb, int64_t y_sn, int B, int pieces_len, int block_len,
                int C, int K, int flip, int dtype, void* stream) {
    RC(lepe_check(x, y, B, pieces_len, block_len, C, K, dtype));
    if (!w_taps) return fail(MHLA_EINVAL, "w_taps null");
    if ((x_sb | x_sn | y_sb | y_sn | (add ? (add_sb | add_sn) : 0)) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    LepeArgs a{x, (long)x_sb, (long)x_sn, w_taps, bias, add, (long)add_sb, (long)add_sn, y, (long)y_sb, (long)y_sn, B, pieces_len, block_len, C, K, flip ? 1 : 0};
    const long N = (long)pieces_len * pieces_len * block_len * block_len, work = N * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    // runs of four tokens: 3 x 3, 16-bit tensors, whole runs inside a block row, 16-byte aligned rows (lepe.hpp)
    const bool al16 = !((x_sb | x_sn | y_sb | y_sn | (add ? (add_sb | add_sn) : 0)) & 7) &&
                      !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)(add ? add : x)) & 15);
    if (K == 3 && dtype != MHLA_F32 && block_len % 4 == 0 && al16) {
        const long work4 = N / 4 * (C / 8);
        if (dtype == MHLA_BF16) RC(launch(k_lepe2d_run4<bf16_t>, dim3((unsigned)((work4 + 255) / 256), B), dim3(256), 0, st, "k_lepe2d_run4", a));
        else                    RC(launch(k_lepe2d_run4<f16_t>, dim3((unsigned)((work4 + 255) / 256), B), dim3(256), 0, st, "k_lepe2d_run4", a));
        return MHLA_OK;
    }
    DISPATCH_T(dtype, {
        if (K == 3) RC(launch(k_lepe2d<ET, 3>, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, "k_lepe2d", a));
        else        RC(launch(k_lepe2d<ET, 5>, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, "k_lepe2d", a));
    });
    return MHLA_OK;
}

size_t mhla_lepe2d_wgrad_ws_bytes(int C, int K) { return (size_t)LEPE_SLICES * (K * K + 1) * C * 4; }

int mhla_lepe2d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn, float* dwb,
                      void* ws, size_t ws_bytes, int B, int pieces_len, int block_len, int C, int K, int dtype,
                      void* stream) {
    RC(lepe_check(x, dout, B, pieces_len, block_len, C, K, dtype));
    if (!dwb || !ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "dwb / workspace null or workspace not 16-byte aligned");
    if (ws_bytes < mhla_lepe2d_wgrad_ws_bytes(C, K)) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, mhla_lepe2d_wgrad_ws_bytes(C, K));
    if ((x_sb | x_sn | g_sb | g_sn) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    LepeWgradArgs a{x, (long)x_sb, (long)x_sn, dout, (long)g_sb, (long)g_sn, (float*)ws, B, pieces_len, block_len, C, K, LEPE_SLICES};
    hipStream_t st = (hipStream_t)stream;
    // a workgroup covers 4 waves x 8 channel groups of CH channels (CH = 8 for K = 3, 4 for K = 5)
    DISPATCH_T(dtype, {
        if (K == 3) RC(launch(k_lepe2d_wgrad<ET, 3, 8>, dim3((C + 255) / 256, LEPE_SLICES), dim3(256), 0, st, "k_lepe2d_wgrad", a));
        else        RC(launch(k_lepe2d_wgrad<ET, 5, 4>, dim3((C + 127) / 128, LEPE_SLICES), dim3(256), 0, st, "k_lepe2d_wgrad", a));
    });
    const int rows_c = (K * K + 1) * C;
    RC(launch(k_lepe2d_wgrad_reduce, dim3((rows_c + 63) / 64), dim3(256), 0, st, "k_lepe2d_wgrad_reduce", (const float*)ws, dwb, rows_c, LEPE_SLICES));
    return MHLA_OK;
}

static int lepe3d_check(const void* x, const void* y, int B, int F, int H, int W, int C, int dtype) {
    if (!x || !y) return fail(MHLA_EINVAL, "null pointer");
    if (B <= 0 || F <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return fail(MHLA_EINVAL, "B=%d F=%d H=%d W=%d C=%d: need positive sizes and C %% 8 == 0", B, F, H, W, C);
    if ((long)F * H * W > (1L << 30)) return fail(MHLA_ENOTSUP, "F*H*W = %ld tokens exceed 2^30", (long)F * H * W);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (B > 65535) return fail(MHLA_ENOTSUP, "B=%d exceeds grid limit 65535", B);
    return MHLA_OK;
}

int mhla_lepe3d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias, const void* add,
                int64_t add_sb, int64_t add_sn, void* y, int64_t y_sb, int64_t y_sn, int B, int F, int H, int W, int C,
                int flip, int dtype, void* stream) {
    RC(lepe3d_check(x, y, B, F, H, W, C, dtype));
    if (!w_taps) return fail(MHLA_EINVAL, "w_taps null");
    if ((x_sb | x_sn | y_sb | y_sn | (add ? (add_sb | add_sn) : 0)) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    Lepe3dArgs a{x, (long)x_sb, (long)x_sn, w_taps, bias, add, (long)add_sb, (long)add_sn, y, (long)y_sb, (long)y_sn, B, F, H, W, C, flip ? 1 : 0};
    const long work = (long)F * H * W * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { RC(launch(k_lepe3d<ET>, dim3((unsigned)((work + 255) / 256), B), dim3(256), 0, st, "k_lepe3d", a)); });
    return MHLA_OK;
}

size_t mhla_lepe3d_wgrad_ws_bytes(int C) { return (size_t)LEPE_SLICES * 28 * C * 4; }

int mhla_lepe3d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn, float* dwb,
                      void* ws, size_t ws_bytes, int B, int F, int H, int W, int C, int dtype, void* stream) {
    RC(lepe3d_check(x, dout, B, F, H, W, C, dtype));
    if (!dwb || !ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "dwb / workspace null or workspace not 16-byte aligned");
    if (ws_bytes < mhla_lepe3d_wgrad_ws_bytes(C)) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, mhla_lepe3d_wgrad_ws_bytes(C));
    if ((x_sb | x_sn | g_sb | g_sn) & 3) return fail(MHLA_EINVAL, "strides must be multiples of 4 elements");
    Lepe3dWgradArgs a{x, (long)x_sb, (long)x_sn, dout, (long)g_sb, (long)g_sn, (float*)ws, B, F, H, W, C, LEPE_SLICES};
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, { RC(launch(k_lepe3d_wgrad<ET>, dim3((C + 127) / 128, LEPE_SLICES), dim3(256), 0, st, "k_lepe3d_wgrad", a)); });
    const int rows_c = 28 * C;
    RC(launch(k_lepe2d_wgrad_reduce, dim3((rows_c + 63) / 64), dim3(256), 0, st, "k_lepe2d_wgrad_reduce", (const float*)ws, dwb, rows_c, LEPE_SLICES));
    return MHLA_OK;
}

static int prologue_check(const void* x, int64_t rows, int C, int dtype, int64_t ldx) {
    if (!x) return fail(MHLA_EINVAL, "null pointer");
    if (rows <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * 8) return fail(MHLA_EINVAL, "rows=%lld C=%d: need C %% 8 == 0 and C <= 4096", (long long)rows, C);
    if (ldx & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    return MHLA_OK;
}
static int prologue_rope_check(const float* cos, const float* sin, int64_t ld_tab, int ntok, int D, int C) {
    if (!cos || !sin) return fail(MHLA_EINVAL, "rope tables null");
    if (D <= 0 || (D & 7) || C % D) return fail(MHLA_EINVAL, "head dim D=%d must be a multiple of 8 dividing C=%d", D, C);
    if (ntok <= 0 || ld_tab < D / 2 || (ld_tab & 3) || ((uintptr_t)cos | (uintptr_t)sin) % 16)
        return fail(MHLA_EINVAL, "rope tables: ntok=%d, ld=%lld must be >= D/2 and a multiple of 4, tables 16-byte aligned", ntok, (long long)ld_tab);
    return MHLA_OK;
}

int mhla_qk_prologue(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy, int64_t rows, int C, int norm,
                     float norm_eps, float eps, int dtype, void* stream) {
    return mhla_qk_prologue_rope(x, ldx, w, y, ldy, nullptr, 0, nullptr, nullptr, 0, 0, 0, rows, C, norm, norm_eps, eps, dtype, stream);
}

int mhla_qk_prologue_rope(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy, float* y_rope, int64_t ldyr,
                          const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D, int64_t rows, int C,
                          int norm, float norm_eps, float eps, int dtype, void* stream) {
    RC(prologue_check(x, rows, C, dtype, ldx | ldy | (y_rope ? ldyr : 0)));
    if (!y) return fail(MHLA_EINVAL, "null pointer");
    if (y_rope) RC(prologue_rope_check(rope_cos, rope_sin, ld_tab, ntok, D, C));
    PrologueArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.y = y; a.ldy = ldy; a.rows = rows; a.C = C; a.norm_eps = norm_eps; a.eps = eps; a.norm = norm ? 1 : 0;
    a.yr = y_rope; a.ldyr = ldyr; a.rcos = rope_cos; a.rsin = rope_sin; a.ldr = ld_tab; a.ntok = ntok > 0 ? ntok : 1; a.D = D > 0 ? D : 8;
    hipStream_t st = (hipStream_t)stream;
    const int64_t gsz = (rows + 3) / 4;
    const dim3 grid((unsigned)(gsz < 16384 ? gsz : 16384));
    DISPATCH_T(dtype, {
        if (C <= 1024)      RC(launch(k_qk_prologue<ET, 2>, grid, dim3(256), 0, st, "k_qk_prologue", a));
        else if (C <= 2048) RC(launch(k_qk_prologue<ET, 4>, grid, dim3(256), 0, st, "k_qk_prologue", a));
        else                RC(launch(k_qk_prologue<ET, 8>, grid, dim3(256), 0, st, "k_qk_prologue", a));
    });
    return MHLA_OK;
}

int mhla_rms_rstd(const void* x, int64_t ldx, float* rstd, int64_t rows, int C, float norm_eps, int dtype, void* stream) {
    if (!x || !rstd) return fail(MHLA_EINVAL, "null pointer");
    if (rows <= 0 || C <= 0 || (C & 7)) return fail(MHLA_EINVAL, "rows=%lld C=%d: need C %% 8 == 0", (long long)rows, C);
    if ((ldx & 3) || ((uintptr_t)x) % 8) return fail(MHLA_EINVAL, "row stride must be a multiple of 4, x 8-byte aligned");
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    RstdArgs a{x, (long)ldx, rstd, (long)rows, C, norm_eps};
    const int64_t gsz = (rows + 3) / 4;
    const dim3 grid((unsigned)(gsz < 16384 ? gsz : 16384));
    DISPATCH_T(dtype, { RC(launch(k_rms_rstd<ET>, grid, dim3(256), 0, (hipStream_t)stream, "k_rms_rstd", a)); });
    return MHLA_OK;
}

static int prologue_bwd_grid(int64_t rows) {   // wide rows (C floats of dw partial each): fewer workgroups than the per-head norm
    const int64_t g = (rows + 3) / 4;
    return (int)(g < 2048 ? g : 2048);
}
int64_t mhla_qk_prologue_dw_rows(int64_t rows) { return prologue_bwd_grid(rows); }

int mhla_qk_prologue_bwd(const void* x, int64_t ldx, const float* w, const float* dy, int64_t lddy, const float* dy_rope,
                         int64_t lddyr, const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D,
                         void* dx, int64_t lddx, float* dw_partial, int64_t rows, int C, int norm, float norm_eps, int dtype,
                         void* stream) {
    RC(prologue_check(x, rows, C, dtype, ldx | lddx | (dy ? lddy : 0) | (dy_rope ? lddyr : 0)));
    if (!dx || (!dy && !dy_rope)) return fail(MHLA_EINVAL, "dx null or no upstream gradient");
    if (C > 2048) return fail(MHLA_ENOTSUP, "backward supports C <= 2048 (C=%d)", C);
    if (dy_rope) RC(prologue_rope_check(rope_cos, rope_sin, ld_tab, ntok, D, C));
    if (w && !dw_partial) return fail(MHLA_EINVAL, "dw_partial null");
    PrologueArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.rows = rows; a.C = C; a.norm_eps = norm_eps; a.norm = norm ? 1 : 0;
    a.rcos = rope_cos; a.rsin = rope_sin; a.ldr = ld_tab; a.ntok = ntok > 0 ? ntok : 1; a.D = D > 0 ? D : 8;
    a.dy = dy; a.lddy = lddy; a.dyr = dy_rope; a.lddyr = lddyr; a.dx = dx; a.lddx = lddx; a.dwp = w ? dw_partial : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(prologue_bwd_grid(rows));
    DISPATCH_T(dtype, {
        if (C <= 1024) RC(launch(k_qk_prologue_bwd<ET, 2>, grid, dim3(256), 0, st, "k_qk_prologue_bwd", a));
        else           RC(launch(k_qk_prologue_bwd<ET, 4>, grid, dim3(256), 0, st, "k_qk_prologue_bwd", a));
    });
    return MHLA_OK;
}

int mhla_featmap_rotary(mhla_view x, mhla_view x_saved, const void* cos, const void* sin, int64_t ld_tab, int64_t t_offset,
                        mhla_mview y, int B, int T, int H, int K, int feature_map, int backward, int dtype, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || (K & 7)) return fail(MHLA_EINVAL, "B=%d T=%d H=%d K=%d: need positive sizes and K %% 8 == 0", B, T, H, K);
    if (feature_map < 0 || feature_map > 2) return fail(MHLA_EINVAL, "feature_map %d: 0 identity, 1 relu, 2 elu+1", feature_map);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    CHECK_VIEW(x); CHECK_VIEW(y);
    if (!cos || !sin || ld_tab < K / 2 || (ld_tab & 3) || t_offset < 0) return fail(MHLA_EINVAL, "cos/sin tables null, ld < K/2, ld %% 4 != 0 or negative offset");
    if (backward && feature_map) CHECK_VIEW(x_saved);
    FmRotArgs a{cv(x), cv(x_saved), cmv(y), cos, sin, (long)ld_tab, B, T, H, K, feature_map, (long)t_offset};
    const long total = (long)B * T * H * (K / 8);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((total + 255) / 256));
    DISPATCH_T(dtype, {
        if (backward) RC(launch(k_fmap_rotary<ET, true>, grid, dim3(256), 0, st, "k_fmap_rotary<bwd>", a));
        else          RC(launch(k_fmap_rotary<ET, false>, grid, dim3(256), 0, st, "k_fmap_rotary", a));
    });
    return MHLA_OK;
}

int mhla_rmsnorm_gate_bwd(const void* x, int64_t ldx, const void* g, int64_t ldg, const float* w, const void* dy,
                          int64_t lddy, void* dx, int64_t lddx, void* dg, int64_t lddg, float* dw_partial, int64_t rows,
                          int D, float eps, int dtype, void* stream) {
    RC(norm_check(x, dx, rows, D, dtype));
    if (!dy || !dw_partial || (g && !dg)) return fail(MHLA_EINVAL, "null pointer");
    if ((ldx | lddy | lddx | (g ? (ldg | lddg) : 0)) & 3) return fail(MHLA_EINVAL, "row strides must be multiples of 4");
    NormArgs a{};
    a.x = x; a.ldx = ldx; a.g = g; a.ldg = ldg; a.w = w; a.dy = dy; a.lddy = lddy; a.dx = dx; a.lddx = lddx;
    a.dg = dg; a.lddg = lddg; a.dwp = dw_partial; a.rows = rows; a.D = D; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(norm_grid(rows));
    DISPATCH_T(dtype, {
        if (D <= 64) {
            if (g) RC(launch(k_rmsnorm_gate_bwd_sub<ET, 16, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd_sub<ET, 16, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else if (D <= 128) {
            if (g) RC(launch(k_rmsnorm_gate_bwd_sub<ET, 32, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd_sub<ET, 32, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else if (D <= 256) {
            if (g) RC(launch(k_rmsnorm_gate_bwd<ET, 1, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd<ET, 1, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        } else {
            if (g) RC(launch(k_rmsnorm_gate_bwd<ET, 2, true>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
            else   RC(launch(k_rmsnorm_gate_bwd<ET, 2, false>, grid, dim3(256), 0, st, "k_rmsnorm_gate_bwd", a));
        }
    });
    return MHLA_OK;
}

}  // extern "C"
