#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// LDS holds a row-major [R=32][C=64] matrix of shorts, value = r*100 + c
__global__ void probe(short* out, int mode) {
    __shared__ short lds[32 * 64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    int l = threadIdx.x;
    // candidate addressing: within 16-lane group g: lane li -> row = g*4 + li/4 ... (mode 0)
    int g = l >> 4, li = l & 15;
    int row, col;
    if (mode == 0) { row = g * 4 + (li >> 2); col = (li & 3) * 4; }
    else { row = li; col = g * 4; }   // mode 1: lane -> its own row, 4 contiguous cols
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2 * 2);
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
