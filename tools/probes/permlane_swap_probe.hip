// v_permlane16_swap_b32 semantics on gfx950 (used by pair_pieces in mhla_amd/csrc/fused.hpp):
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/permlane_swap_probe tools/probes/permlane_swap_probe.hip && tools/probes/permlane_swap_probe
// Expected: r[0] = {a.row0, b.row0, a.row2, b.row2}, r[1] = {a.row1, b.row1, a.row3, b.row3}  (rows of 16 lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d;
    unsigned h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4, c = l & 15;
        const unsigned e0 = (row & 1) ? 100 + (row - 1) * 16 + c : l;           // odd rows of r[0]: b's even row below
        const unsigned e1 = (row & 1) ? 100 + l : (row + 1) * 16 + c;           // even rows of r[1]: a's odd row above
        if (h[l] != e0 || h[64 + l] != e1) ok = 0;
    }
    printf("r0:"); for (int l = 0; l < 64; l += 8) printf(" %u", h[l]);
    printf("\nr1:"); for (int l = 0; l < 64; l += 8) printf(" %u", h[64 + l]);
    printf("\npermlane16_swap semantics %s\n", ok ? "AS EXPECTED" : "DIFFERENT");
    return ok ? 0 : 1;
}
