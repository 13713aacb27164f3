// Probe: one-way latency of a flag hand-off between two workgroups, same XCD vs another XCD, by store flavour.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/handoff_probe tools/probes/handoff_probe.hip && tools/probes/handoff_probe
// Ping-pong: block A stores flag[0] = i, block B polls it (sc1 load), stores flag[64] = i, A polls that.  Time per round trip / 2.
// A = block 0; B = block `peer` (8 = same XCD under round-robin placement, 1 = the next XCD); all other blocks exit at once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool PLAIN>
__global__ void pingpong(unsigned* flags, unsigned* xcc_out, unsigned long long* t_out, int peer, int iters) {
    const int b = blockIdx.x;
    if (b != 0 && b != peer) return;
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc_out[b == 0 ? 0 : 1] = xcc;
    volatile unsigned* vf = flags;
    unsigned long long t0 = 0, t1 = 0;
    if (b == 0) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 1; i <= iters; ++i) {
            if (PLAIN) vf[0] = i; else st_sc1(flags, i);
            unsigned spins = 0;
            while (ld_sc1(flags + 64) != (unsigned)i && ++spins < (1u << 22)) {}
        }
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        t_out[0] = t1 - t0;
    } else {
        for (int i = 1; i <= iters; ++i) {
            unsigned spins = 0;
            while (ld_sc1(flags) != (unsigned)i && ++spins < (1u << 22)) {}
            if (PLAIN) vf[64] = i; else st_sc1(flags + 64, i);
        }
    }
}

int main() {
    unsigned *flags, *xcc;
    unsigned long long* t;
    hipMalloc(&flags, 4096); hipMalloc(&xcc, 64); hipMalloc(&t, 64);
    const int iters = 2000;
    for (int peer : {8, 16, 1, 2, 7}) {
        for (int plain = 0; plain < 2; ++plain) {
            hipMemset(flags, 0, 4096);
            if (plain) hipLaunchKernelGGL(pingpong<true>, dim3(64), dim3(64), 0, 0, flags, xcc, t, peer, iters);
            else hipLaunchKernelGGL(pingpong<false>, dim3(64), dim3(64), 0, 0, flags, xcc, t, peer, iters);
            hipDeviceSynchronize();
            unsigned hx[2]; unsigned long long ht;
            hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
            printf("peer block %2d  xcc A=%u B=%u  %-5s stores: one-way %.3f us\n", peer, hx[0], hx[1], plain ? "plain" : "sc1", ht * 0.01 / iters / 2);
        }
    }
    return 0;
}
