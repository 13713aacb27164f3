// Probe: how fast does the MI355X dispatcher start workgroups?  N workgroups stamp s_memrealtime (100 MHz) on entry and leave
// (optionally after a fixed busy time), for several workgroup shapes (threads, static LDS, VGPR footprint).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LDS_BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void stamp_kernel(long long* t_start, long long* t_end, int busy_ticks) {
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { lds[0] = 1; t_start[blockIdx.x] = t0; }
    if (busy_ticks > 0) while (__builtin_amdgcn_s_memrealtime() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime() + lds[0] - 1;
}

template <int LDS_BYTES, int THREADS>
void run(const char* name, int nwg, int busy_ticks, long long* ds, long long* de, hipStream_t s) {
    std::vector<long long> hs(nwg), he(nwg);
    double best_span = 1e18, best_last_start = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL((stamp_kernel<LDS_BYTES, THREADS>), dim3(nwg), dim3(THREADS), 0, s, ds, de, busy_ticks);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(hs.data(), ds, nwg * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(he.data(), de, nwg * 8, hipMemcpyDeviceToHost));
        const long long t0 = *std::min_element(hs.begin(), hs.end());
        const double last_start = (*std::max_element(hs.begin(), hs.end()) - t0) * 0.01;
        const double span = (*std::max_element(he.begin(), he.end()) - t0) * 0.01;
        if (span < best_span) { best_span = span; best_last_start = last_start; }
    }
    printf("%-34s WGs %5d busy %5.1f us: last start +%.2f us, span %.2f us  (%.0f WG/us)\n", name, nwg, busy_ticks * 0.01, best_last_start, best_span,
           nwg / std::max(best_last_start, 0.01));
}

int main() {
    long long *ds, *de; CHECK(hipMalloc(&ds, 16384 * 8)); CHECK(hipMalloc(&de, 16384 * 8));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (int busy : {0, 300}) {
        for (int nwg : {256, 608, 1024, 4096}) {
            run<0, 64>("64 thr, no LDS", nwg, busy, ds, de, s);
            run<0, 256>("256 thr, no LDS", nwg, busy, ds, de, s);
            run<32768, 256>("256 thr, 32 KB LDS", nwg, busy, ds, de, s);
            run<65536, 256>("256 thr, 64 KB LDS", nwg, busy, ds, de, s);
            run<32768, 512>("512 thr, 32 KB LDS", nwg, busy, ds, de, s);
            run<0, 1024>("1024 thr, no LDS", nwg, busy, ds, de, s);
        }
    }
    return 0;
}
