// One-shot all-reduce of the row-parallel projections' partial sums over IPC-mapped peer buffers (xGMI), with the residual add
// that follows it fused in.  Replaces, at decode sizes, the dist.all_reduce of RowParallelLinear.forward
// (nanovllm/layers/linear.py:185-190) and the add of add_rms_forward (nanovllm/layers/layernorm.py:35-36); SURVEY.md section 8f-3.
//
// Why not the ring: a decode step all-reduces [<=64, hidden] bf16 twice per layer (57 KB at hidden 896, 229 KB at 3584).  xGMI is
// point-to-point (7 links per GPU); a ring does 2(p-1) dependent hops over ONE link at a time and is latency-bound at these
// sizes.  One-shot: every rank reads the p-1 peers' partials over p-1 DISTINCT links at once and reduces locally — one
// signalling hop plus one remote read, whatever p is.
//
// Protocol (one process per GPU; every rank owns a STAGING buffer and a FLAG table, both fine-grained device memory that the
// peers map through hipIpc handles):
//   epoch e = state[0] + 1 (device-resident: a replayed HIP graph needs no host-side counter), slot = e & 1
//   1. copy this rank's partial into its own staging slot                       (plain 16-byte stores)
//   2. system-scope RELEASE fence, workgroup barrier, then lane p stores e into PEER p's flag table  (remote 4-byte store)
//   3. lane p polls its LOCAL table for peer p's flag >= e (bounded spin), system-scope ACQUIRE fence, workgroup barrier
//   4. read every rank's staging slot (own included), sum in fp32 IN RANK ORDER (every rank computes the same bits), round to
//      bf16 once; epilogue NONE: store the sum;  RESIDUAL_ADD: residual = bf16(residual + sum) and, optionally, the updated rows
//      again in MFMA-fragment order for the next streaming GEMM (what nvh_residual_add_pack did after RCCL)
//   5. the last workgroup to finish publishes state[0] = e
// No trailing barrier: staging is double-buffered by epoch parity.  A peer can be at most ONE call ahead (its call e+1 passes
// step 3 only with this rank's flag e+1, which is sent after this rank finished call e), so while this rank reads slot e & 1 a fast
// peer writes only slot (e+1) & 1; flags are compared with >=, so a flag already overwritten by e+1 still releases the wait.
// Workgroups are independent (each owns a contiguous range of 16-byte chunks and its own flag per peer): no grid-wide sync.
// Everything is stream-ordered and allocation-free: legal under HIP-graph capture.  A spin that runs out (a dead peer) sets
// state[2] and writes NaN instead of hanging the device.  The LATE side of such an event is caught as well: a rank that finds a peer's
// flag at e + 2 or beyond knows that peer gave up on call e (or e + 1) and has since reused the slot of parity e & 1 — the bytes it is
// about to read may be a later call's — and fails the same way (NaN rows, state[2]) instead of summing them.  state[3], when non-zero,
// replaces the poll limit (tests force the time-out with it); the host reads state[2] wherever it synchronises anyway
// (nanovllm_hip/distributed.py raise_if_failed, called at the token readback).
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

constexpr int AR_THREADS = 256;
constexpr unsigned kArSpinLimit = 1u << 22;        // x ~1 us of s_sleep: seconds, then give up

__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(AR_THREADS) void allreduce_oneshot_kernel(const AllReduceArgs a) {
    __shared__ int lds_fail;
    const int tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x;
    const uint32_t e = a.state[0] + 1u;                            // written by the previous launch's last workgroup
    const uint32_t limit = a.state[3] ? a.state[3] : kArSpinLimit;
    const size_t slot = (size_t)(e & 1u) * a.slot_bytes;
    const int chunks_per_row = a.hidden / 8;                       // 16-byte chunks
    const int64_t total = (int64_t)a.rows * chunks_per_row;
    const int64_t per = (total + nblk - 1) / nblk;
    const int64_t c0 = blk * per, c1 = min(total, c0 + per);
    if (tid == 0) lds_fail = 0;

    // 1. my partial -> my staging slot (row-major, dense)
    u32x4* const mine = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.stage[a.rank]) + slot);
    for (int64_t c = c0 + tid; c < c1; c += AR_THREADS) {
        const int row = (int)(c / chunks_per_row), col = (int)(c - (int64_t)row * chunks_per_row);
        mine[c] = *reinterpret_cast<const u32x4*>(a.x + row * a.x_stride + col * 8);
    }
    // 2. publish: everything this workgroup stored is visible system-wide before any of its flags
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid < a.world && tid != a.rank) st_sys(a.flags[tid] + (size_t)a.rank * AR_MAX_BLOCKS + blk, e);
    // 3. wait for the same workgroup of every peer (local polls)
    if (tid < a.world && tid != a.rank) {
        const uint32_t* f = a.flags[a.rank] + (size_t)tid * AR_MAX_BLOCKS + blk;
        unsigned spins = 0;
        int32_t ahead;
        while ((ahead = (int32_t)(ld_sys(f) - e)) < 0) {
            if (++spins > limit) { lds_fail = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (ahead >= 2) lds_fail = 1;                              // the peer is two calls on: it has given up on this one and reused the slot
    }
    // acquire on BOTH sides of the barrier: the polling lanes' fence orders their own later loads; the waves that do not poll must
    // invalidate AFTER the barrier tells them the flags were seen (their earlier invalidate proves nothing about lines fetched since)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const bool failed = lds_fail != 0;                             // workgroup-uniform
    if (failed && tid == 0) st_sys(a.state + 2, e);

    // 4. reduce in rank order
    for (int64_t c = c0 + tid; c < c1; c += AR_THREADS) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int p = 0; p < a.world; ++p) {
            const u32x4 v = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.stage[p]) + slot)[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[2 * k] += bf16_lo(v[k]);
                acc[2 * k + 1] += bf16_hi(v[k]);
            }
        }
        if (failed) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_nanf("");
        }
        const int row = (int)(c / chunks_per_row), col = (int)(c - (int64_t)row * chunks_per_row);
        u32x4 s;
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = pack_bf16x2(acc[2 * k], acc[2 * k + 1]);
        uint16_t* const o = a.out + row * a.out_stride + col * 8;
        if (a.epi == AR_EPI_RESIDUAL_ADD) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(o);
            u32x4 t;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = pack_bf16x2(bf16_lo(r[k]) + bf16_lo(s[k]), bf16_hi(r[k]) + bf16_hi(s[k]));
            *reinterpret_cast<u32x4*>(o) = t;
            if (a.packed) *reinterpret_cast<u32x4*>(a.packed + pack_index(row, col * 8, a.hidden)) = t;
        } else {
            *reinterpret_cast<u32x4*>(o) = s;
        }
    }
    // 5. the last workgroup of this launch publishes the epoch for the next one
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(a.state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (uint32_t)nblk - 1) {
            __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

int allreduce_blocks(int rows, int hidden) {
    // a workgroup moves 4 KiB per sweep of its 256 threads; 57 KB .. 460 KB messages: 8 .. 32 workgroups, each with its own flags
    const int64_t bytes = (int64_t)rows * hidden * 2;
    int n = (int)((bytes + 8191) / 8192);
    return n < 1 ? 1 : (n > AR_MAX_BLOCKS ? AR_MAX_BLOCKS : n);
}

int launch_allreduce_oneshot(const AllReduceArgs& a, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(blocks), dim3(AR_THREADS), 0, stream, a);
    return check_launch("allreduce_oneshot");
}

}  // namespace nvh
