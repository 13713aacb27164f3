// Probe: cost of a device-wide barrier inside one persistent kernel (256..1024 workgroups) on MI355X, against a chain of
// dependent empty kernels.  Every spin is bounded: on timeout the workgroup raises a flag and leaves, so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000u) { *fail = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void barrier_loop(unsigned* ctr, unsigned* fail, float* data, int rounds, int touch) {
    const unsigned nwg = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (touch) {   // each workgroup writes a line and, after the barrier, reads its neighbour's: checks visibility too
            data[(size_t)blockIdx.x * 64 + (threadIdx.x & 63)] = (float)(r + 1);
        }
        if (!grid_barrier(ctr, (unsigned)(r + 1) * nwg, fail)) return;
        if (touch) {
            const float v = __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + 37) % nwg) * 64 + (threadIdx.x & 63)]);
            if (v < (float)(r + 1)) *fail = 2;
        }
    }
}

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) *p = 1.f; }

int main() {
    unsigned *ctr, *fail; float* data;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&fail, 4)); CHECK(hipMalloc(&data, 1024 * 64 * 4));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int rounds = 2000;
    for (int touch = 0; touch < 2; ++touch)
        for (int nwg : {256, 512, 1024}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemsetAsync(ctr, 0, 4, s)); CHECK(hipMemsetAsync(fail, 0, 4, s));
                CHECK(hipEventRecord(a, s));
                hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(256), 0, s, ctr, fail, data, rounds, touch);
                CHECK(hipEventRecord(b, s)); CHECK(hipStreamSynchronize(s));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            unsigned f; CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("grid barrier  workgroups %4d  touch %d : %.3f us per barrier  (fail flag %u)\n", nwg, touch, best * 1e3 / rounds, f);
            if (f == 1) { printf("timeout: not all workgroups co-resident; stopping\n"); return 0; }
        }
    // chain of dependent empty kernels in a graph
    for (int n : {64, 512}) {
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (float*)nullptr);
        CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(a, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(b, s)); CHECK(hipStreamSynchronize(s));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("graph chain of %d empty 256-WG kernels: %.3f us per kernel\n", n, best * 1e3 / n);
    }
    return 0;
}
