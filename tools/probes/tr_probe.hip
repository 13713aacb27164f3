// Empirical lane map of ds_read_b64_tr_b16 and of the 16x16x32 bf16 MFMA operand/result layouts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_kernel(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;     // element (row r, col c) = r*64 + c
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    int row, col;
    if (mode == 0) { row = 4 * g + (i >> 2); col = 4 * (i & 3); }          // lane 4q+p -> row q, cols 4p..4p+3
    else           { row = 16 * g + i;       col = 0; }                    // lane i -> row i, cols 0..3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}

__global__ void mfma_kernel(const float* A, const float* B, float* C) {     // A [16][32], B [32][16] row-major
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[i * 32 + 8 * g + j]; b[j] = (__bf16)B[(8 * g + j) * 16 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];             // assumed: row 4g+r, col i
}

int main() {
    int* d; hipMalloc(&d, 256 * 4); int h[256];
    for (int mode = 0; mode < 2; ++mode) {
        tr_kernel<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("tr mode %d (value = row*64+col)\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (%d,%d)", h[l*4+e] / 64, h[l*4+e] % 64); printf("\n"); if (l == 19) { l = 47; } }
    }
    float hA[512], hB[512], hC[256], ref[256]; float *dA, *dB, *dC;
    for (int x = 0; x < 512; ++x) { hA[x] = (float)((x * 7 + 3) % 11 - 5); hB[x] = (float)((x * 5 + 1) % 13 - 6); }
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[r*32+k] * hB[k*16+c]; ref[r*16+c] = s; }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dC); hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int x = 0; x < 256; ++x) bad += hC[x] != ref[x];
    printf("mfma 16x16x32 layout check: %d mismatches\n", bad);
    return 0;
}
