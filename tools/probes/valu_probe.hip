// Issue-rate probe on gfx950: cycles per wave-instruction for the VALU/LDS ops the decode kernel uses,
// at 1, 2 and 4 waves per SIMD (256/512/1024-thread blocks on one CU), independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define N 64
template <int MODE>
__global__ void probe(float* out, unsigned long long* cyc, const uint32_t* in) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i & 255];
    __syncthreads();
    float acc[8]; uint32_t a = in[threadIdx.x & 255], b = in[(threadIdx.x + 7) & 255];
    for (int i = 0; i < 8; ++i) acc[i] = (float)i;
    f32x2 pa[4] = {{1,2},{3,4},{5,6},{7,8}}; f32x2 pb = {__uint_as_float(a), __uint_as_float(b)};
    u32x4 sink = {0,0,0,0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) {            // 8 independent v_dot2c
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(*(bf16x2_t*)&a, *(bf16x2_t*)&b, acc[i], false);
        } else if (MODE == 1) {     // 8 independent v_fma
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(__uint_as_float(a), acc[i], __uint_as_float(b));
        } else if (MODE == 2) {     // 4 independent v_pk_fma (8 fmas)
#pragma unroll
            for (int i = 0; i < 4; ++i) pa[i] = __builtin_elementwise_fma(pa[i], pb, pb);
        } else if (MODE == 3) {     // 8 broadcast ds_read_b128 + consume
#pragma unroll
            for (int i = 0; i < 8; ++i) { u32x4 v = *(volatile u32x4*)&lds[(it * 32 + i * 4) & 4095]; sink ^= v; }
        } else if (MODE == 4) {     // 8 per-lane ds_read_b128 (row stride 128 B, swizzled) + consume
#pragma unroll
            for (int i = 0; i < 8; ++i) { int l = threadIdx.x & 63; u32x4 v = *(volatile u32x4*)&lds[((l * 32) + ((i ^ ((l >> 1) & 7)) * 4)) & 4095]; sink ^= v; }
        } else if (MODE == 5) {     // dependent chain of 8 v_dot2c
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_fdot2_f32_bf16(*(bf16x2_t*)&a, *(bf16x2_t*)&b, acc[0], false);
        } else if (MODE == 6) {     // 8 v_exp
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i]; for (int i = 0; i < 4; ++i) s += pa[i][0] + pa[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_iter) {
    float* out; unsigned long long* cyc; uint32_t* in;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 4096); (void)hipMalloc(&in, 1024);
    uint32_t hin[256]; for (int i = 0; i < 256; ++i) hin[i] = 0x3f803f80u + i; (void)hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) {
        probe<MODE><<<1, threads>>>(out, cyc, in); (void)hipDeviceSynchronize();
        probe<MODE><<<1, threads>>>(out, cyc, in); (void)hipDeviceSynchronize();
        unsigned long long h[16]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("  %d waves/SIMD: %6.2f cyc/instr/wave", threads / 256, mx / (N * per_iter));
    }
    printf("\n");
}
int main() {
    run<0>("v_dot2c_f32_bf16 (8 indep)", 8); run<5>("v_dot2c_f32_bf16 (dependent)", 8); run<1>("v_fma_f32 (8 indep)", 8);
    run<2>("v_pk_fma_f32 (4 indep, per instr)", 4); run<6>("v_exp_f32 (8 indep)", 8);
    run<3>("ds_read_b128 broadcast (8)", 8); run<4>("ds_read_b128 per-lane row (8)", 8);
    return 0;
}
