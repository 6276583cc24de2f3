// The packed-fp32 nondeterminism of DESIGN.md section 5 reproduced WITHOUT libmhla_hip.so: this file launches the library's
// own k_fs_dw kernel (csrc/fused.hpp: dW partials on MFMA + transpose reads in workgroups 0..7 of every (b,h), dz = W^T dn as a
// packed-fp32 [64 x 64] x [64 x 16] product on LDS tiles in the extra workgroups) on synthetic inputs, REPS times, and compares
// dz bit for bit with the first repetition.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -DWZ_VARIANT=0 -o pk_fma_repro pk_fma_repro.hip && ./pk_fma_repro     (differences)
//   ... -DWZ_VARIANT=2 (the shipped loop) or =1: none;   ... -DWZ_VARIANT=0 -Xclang -target-feature -Xclang -packed-fp32-ops
//   (the shipped flag set: no v_pk_*_f32): none;   ... -DWZ_VARIANT=0 -DFS_DW_PROBE_NO_DW_ROLE=1 (dW workgroups compiled out) or =2
//   (compiled in, same registers, but returning at once): none -- the dW role's waves must really run beside the dz ones.
//   Measured round 3 (MI355X, ROCm 7.2): 10 700-11 100 of 524 288 dz values differ per repetition, ALL of them at even s (the low
//   halves of the packed pairs) and in rows r % 8 >= 4 (lanes 16-31 / 48-63 of the multiplying waves).
// Finding (round 3): with packed fp32 ops the dz loop compiles to v_pk_fma_f32 whose broadcast operand pair comes from
// ds_read2_b32 behind partial s_waitcnt lgkmcnt(N) waits; the LOW halves of the results of lanes 16-31 / 48-63 then differ
// from run to run.  Waiting for lgkmcnt(0) before the multiply-adds (WZ_VARIANT=1) or reading the W elements one by one
// (WZ_VARIANT=2, ds_read_b32) removes the differences with packed math still on.
#include "../../mhla_amd/csrc/fused.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace mhla::fast;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20, bh = 128, M = 64, S = 64, njg = 8;
    const size_t nstate = (size_t)bh * njg * FE * IT, nrow = (size_t)bh * M * S;
    std::vector<unsigned short> hst(nstate);
    std::vector<float> hdn(nrow), hW(M * M);
    srand(3);
    for (auto& v : hst) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));
    for (auto& v : hdn) v = ((float)(rand() & 0xffff) / 65536.f - 0.5f) * 1e-3f;
    for (auto& v : hW) v = (float)(rand() & 0xffff) / 65536.f * 0.05f;
    unsigned short *dg, *kv; float *dn, *z, *dwp, *W, *dz; int* done;
    CK(hipMalloc(&dg, nstate * 2)); CK(hipMalloc(&kv, nstate * 2)); CK(hipMalloc(&dn, nrow * 4)); CK(hipMalloc(&z, nrow * 4));
    CK(hipMalloc(&dwp, (size_t)bh * DW_SPLIT * 4096 * 4)); CK(hipMalloc(&W, M * M * 4)); CK(hipMalloc(&dz, nrow * 4)); CK(hipMalloc(&done, bh * 4 * 4 + 16));
    CK(hipMemcpy(dg, hst.data(), nstate * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(kv, hst.data(), nstate * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dn, hdn.data(), nrow * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(z, hdn.data(), nrow * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), M * M * 4, hipMemcpyHostToDevice));
#if defined(FS_DW_PROBE_NO_DW_ROLE) && FS_DW_PROBE_NO_DW_ROLE == 2
    FsDwArgs a{dg, kv, dn, z, nullptr, M, S, njg, W, M, dz, done, 4};
#else
    FsDwArgs a{dg, kv, dn, z, dwp, M, S, njg, W, M, dz, done, 4};
#endif
    std::vector<float> first(nrow), cur(nrow);
    long total = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(dz, 0xff, nrow * 4));
        hipLaunchKernelGGL(k_fs_dw<0>, dim3(DW_SPLIT + (S + WZ_C - 1) / WZ_C, bh), dim3(FT8), FS_DW_SMEM, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cur.data(), dz, nrow * 4, hipMemcpyDeviceToHost));
        if (r == 0) { first = cur; continue; }
        long d = 0, low = 0, upper = 0;
        for (size_t i = 0; i < nrow; ++i)
            if (memcmp(&cur[i], &first[i], 4)) { ++d; low += (i % 2) == 0; upper += ((i / S) % 8) >= 4; }
        if (d) printf("rep %d: %ld of %zu dz values differ (%ld at even s = low halves of the packed pairs, %ld in rows r %% 8 >= 4 = lanes 16-31 / 48-63)\n", r, d, nrow, low, upper);
        total += d;
    }
    printf("pk_fma_repro: %d repetitions, %ld differing values in total\n", reps, total);
    return total ? 1 : 0;
}
