// Row-wise fused elementwise ops around the attention call (engine widening; not part of the attention parity bar):
//   add_rmsnorm  residual-add + RMSNorm in one launch   — nanovllm/layers/layernorm.py:17-41 (rms_forward / add_rms_forward)
//   silu_mul     SiLU(gate) * up                          — nanovllm/layers/activation.py:11-14
// Both are HBM-bound row streams (2-3 x rows x hidden x 2 bytes); at decode they are launch-latency bound, which is
// exactly why they are fused: eager PyTorch spends 5-6 launches per layer on them.  One workgroup per row, 16-byte
// accesses, fp32 math with the reference's rounding points:
//   x32 = float(x) + float(residual); residual_out = bf16(x32); var = mean(x32^2);
//   out = bf16( float(bf16(x32 * rsqrt(var + eps))) * float(weight) )          (layernorm.py:35-40)
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

constexpr int ROW_THREADS = 256;
constexpr int MAX_CHUNKS = 4;                 // 16-byte chunks per thread: hidden <= 256*4*8 = 8192

__device__ __forceinline__ float block_sum(float v, float* lds) {
    v += pair_in_row<1>(v);
    v += pair_in_row<2>(v);
    v += pair_in_row<4>(v);
    v += pair_in_row<8>(v);
    v = sum_xor16(v);
    v = sum_xor32(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

// out = rmsnorm(x [+ residual]) * w ; residual (in/out, nullable) receives bf16(x + residual)
__global__ __launch_bounds__(ROW_THREADS) void add_rmsnorm_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                                   uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
                                                                   float eps, int hidden, int64_t x_stride, int64_t out_stride, int64_t res_stride) {
    __shared__ float lds[4];
    const int row = blockIdx.x;
    const int chunks = hidden / 8;
    const uint16_t* xr = x + row * x_stride;
    uint16_t* rr = residual ? residual + row * res_stride : nullptr;
    float v[MAX_CHUNKS][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i) {
        const int c = threadIdx.x + i * ROW_THREADS;
        if (c < chunks) {
            const u32x4 xv = *reinterpret_cast<const u32x4*>(xr + c * 8);
            u32x4 rv = {0, 0, 0, 0};
            if (rr) rv = *reinterpret_cast<const u32x4*>(rr + c * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[i][2 * k] = bf16_lo(xv[k]) + (rr ? bf16_lo(rv[k]) : 0.f);
                v[i][2 * k + 1] = bf16_hi(xv[k]) + (rr ? bf16_hi(rv[k]) : 0.f);
            }
            if (rr) {
                u32x4 ro;
#pragma unroll
                for (int k = 0; k < 4; ++k) ro[k] = pack_bf16x2(v[i][2 * k], v[i][2 * k + 1]);
                *reinterpret_cast<u32x4*>(rr + c * 8) = ro;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ss += v[i][k] * v[i][k];
        }
    }
    const float inv = rsqrtf(block_sum(ss, lds) / hidden + eps);
#pragma unroll
    for (int i = 0; i < MAX_CHUNKS; ++i) {
        const int c = threadIdx.x + i * ROW_THREADS;
        if (c < chunks) {
            const u32x4 wv = *reinterpret_cast<const u32x4*>(w + c * 8);
            u32x4 ov;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = (float)(__bf16)(v[i][2 * k] * inv) * bf16_lo(wv[k]);
                const float b = (float)(__bf16)(v[i][2 * k + 1] * inv) * bf16_hi(wv[k]);
                ov[k] = pack_bf16x2(a, b);
            }
            *reinterpret_cast<u32x4*>(out + row * out_stride + c * 8) = ov;
        }
    }
}

// out[row, 0:inter] = silu(gu[row, 0:inter]) * gu[row, inter:2*inter]
__global__ __launch_bounds__(256) void silu_mul_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ gu, int n_rows, int inter,
                                                        int64_t gu_stride, int64_t out_stride) {
    const int chunks = inter / 8;
    const int64_t total = (int64_t)n_rows * chunks;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / chunks), c = (int)(idx - (int64_t)row * chunks);
        const uint16_t* g = gu + row * gu_stride + c * 8;
        const u32x4 gv = *reinterpret_cast<const u32x4*>(g);
        const u32x4 uv = *reinterpret_cast<const u32x4*>(g + inter);
        u32x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g0 = bf16_lo(gv[k]), g1 = bf16_hi(gv[k]);
            // F.silu on bf16 computes in fp32 and rounds to bf16; the product with `up` rounds again
            const float s0 = (float)(__bf16)(g0 / (1.f + __expf(-g0))), s1 = (float)(__bf16)(g1 / (1.f + __expf(-g1)));
            ov[k] = pack_bf16x2(s0 * bf16_lo(uv[k]), s1 * bf16_hi(uv[k]));
        }
        *reinterpret_cast<u32x4*>(out + row * out_stride + c * 8) = ov;
    }
}

// residual[row, :] = bf16(residual + y) and the same rows in MFMA-fragment order (pack_index) for the next streaming GEMM:
// the step between a tensor-parallel all-reduce (layers/linear.py:185-190) and the next projection; the add is the one of
// add_rms_forward (layernorm.py:35-36), the norm itself rides in the next GEMM (folded weights).
__global__ __launch_bounds__(256) void residual_add_pack_kernel(uint16_t* __restrict__ residual, const uint16_t* __restrict__ y,
                                                                 uint16_t* __restrict__ packed, int n_rows, int hidden,
                                                                 int64_t res_stride, int64_t y_stride) {
    const int chunks = hidden / 8;
    const int64_t total = (int64_t)n_rows * chunks;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / chunks), c = (int)(idx - (int64_t)row * chunks);
        uint16_t* r = residual + row * res_stride + c * 8;
        const u32x4 rv = *reinterpret_cast<const u32x4*>(r);
        const u32x4 yv = *reinterpret_cast<const u32x4*>(y + row * y_stride + c * 8);
        u32x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = pack_bf16x2(bf16_lo(rv[k]) + bf16_lo(yv[k]), bf16_hi(rv[k]) + bf16_hi(yv[k]));
        *reinterpret_cast<u32x4*>(r) = ov;
        // 8 consecutive columns of one row are one 16-byte slot of the fragment order
        if (packed) *reinterpret_cast<u32x4*>(packed + pack_index(row, c * 8, hidden)) = ov;
    }
}

// Greedy sampling: argmax over each row of bf16 logits (nanovllm/layers/sampler.py at temperature 0 reduces to this).
// One 1024-thread workgroup per row streams the row with 16-byte loads; ties resolve to the lowest index.
// scheduler.postprocess (append the token) + prepare_decode for the NEXT step (engine/model_runner.py:244-269), on the device:
// the row's workgroup owns the row's metadata, so there is nothing to synchronise
__device__ __forceinline__ void advance_row(const AdvanceArgs& adv, int row, int token) {
    const int ctx = adv.context_lens[row];
    if (ctx > 0) {                                             // padding rows (ctx 0, slot -1) stay as they are
        adv.tokens_log[adv.row_steps[row] * adv.log_stride + row] = token;
        adv.row_steps[row] += 1;
        adv.input_ids[row] = token;
        adv.positions[row] += 1;
        adv.context_lens[row] = ctx + 1;
        const int last = ctx;                                  // index of the token the next step stores: new_ctx - 1
        adv.slot_mapping[row] = adv.block_tables[row * adv.bt_stride + last / adv.block_size] * adv.block_size + last % adv.block_size;
    }
}

// Final step of the fused lm_head + argmax: per row, the best of `groups` candidates (value desc, column asc), then the
// advance bookkeeping.  One 256-thread workgroup per row.
__global__ __launch_bounds__(256) void argmax_candidates_kernel(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx,
                                                                 int groups, int64_t cand_stride, int vocab, const AdvanceArgs adv) {
    __shared__ float lds_v[4];
    __shared__ int lds_i[4];
    const int row = blockIdx.x;
    // the row's metadata travels under the candidate loads (thread 0 owns it): the bookkeeping behind the arg-max is then stores only,
    // and they are issued AFTER the embedding row's loads, so that the two dependent chains of the tail (token -> embedding row,
    // context -> block table -> slot) overlap instead of following each other
    int m_ctx = 0, m_blk = 0;
    int64_t m_steps = 0, m_pos = 0, m_tok = 0;
    if (threadIdx.x == 0) {
        m_ctx = adv.context_lens[row];
        m_steps = adv.row_steps[row];
        m_pos = adv.positions[row];
        m_tok = adv.input_ids[row];
        if (m_ctx > 0) m_blk = adv.block_tables[row * adv.bt_stride + m_ctx / adv.block_size];
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int gi = threadIdx.x; gi < groups; gi += 256) {
        const float v = cand_val[(int64_t)gi * cand_stride + row];
        const int i = cand_idx[(int64_t)gi * cand_stride + row];
        if (argmax_better(v, i, best, bidx)) { best = v; bidx = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (argmax_better(ov, oi, best, bidx)) { best = ov; bidx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { lds_v[threadIdx.x >> 6] = best; lds_i[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (argmax_better(lds_v[w], lds_i[w], best, bidx)) { best = lds_v[w]; bidx = lds_i[w]; }
        // a valid token whatever the candidates held (the index feeds an embedding-row address below): the ordering above never
        // lets the (-inf, INT_MAX) identity survive a real candidate; the clamp covers corrupt candidate indices as well
        bidx = bidx < 0 ? 0 : (bidx >= vocab ? vocab - 1 : bidx);
        if (adv.embed) lds_i[0] = m_ctx > 0 ? bidx : (int)m_tok;             // padding rows keep their token
    }
    constexpr int EMB = 4;                                                   // 16-byte chunks of the embedding row per thread: hidden <= 8192
    u32x4 ev[EMB];
    const uint16_t* src = nullptr;
    if (adv.embed) {
        // the next step's embedding lookup (layers/embed_head.py:34-45 at tp = 1) by the workgroup that has just chosen the token
        __syncthreads();
        src = adv.embed + (int64_t)lds_i[0] * adv.hidden;
#pragma unroll
        for (int u = 0; u < EMB; ++u) {
            const int c = (threadIdx.x + 256 * u) * 8;
            if (c < adv.hidden) ev[u] = *reinterpret_cast<const u32x4*>(src + c);
        }
    }
    if (threadIdx.x == 0 && m_ctx > 0) {                                     // advance_row from the prefetched metadata: stores only
        adv.tokens_log[m_steps * adv.log_stride + row] = bidx;
        adv.row_steps[row] = m_steps + 1;
        adv.input_ids[row] = bidx;
        adv.positions[row] = m_pos + 1;
        adv.context_lens[row] = m_ctx + 1;
        adv.slot_mapping[row] = m_blk * adv.block_size + m_ctx % adv.block_size;
    }
    if (adv.embed) {
#pragma unroll
        for (int u = 0; u < EMB; ++u) {
            const int c = (threadIdx.x + 256 * u) * 8;
            if (c < adv.hidden) {
                *reinterpret_cast<u32x4*>(adv.hidden_out + (int64_t)row * adv.hidden_stride + c) = ev[u];
                // eight consecutive columns of one row are contiguous in fragment order as well
                if (adv.hidden_packed) *reinterpret_cast<u32x4*>(adv.hidden_packed + pack_index(row, c, adv.hidden)) = ev[u];
            }
        }
        for (int c = (threadIdx.x + 256 * EMB) * 8; c < adv.hidden; c += 256 * 8) {      // rows wider than 8192 (none today)
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + c);
            *reinterpret_cast<u32x4*>(adv.hidden_out + (int64_t)row * adv.hidden_stride + c) = v;
            if (adv.hidden_packed) *reinterpret_cast<u32x4*>(adv.hidden_packed + pack_index(row, c, adv.hidden)) = v;
        }
    }
}

__global__ __launch_bounds__(1024) void argmax_rows_kernel(int64_t* __restrict__ out, const uint16_t* __restrict__ x, int n, int64_t stride,
                                                            const AdvanceArgs adv) {
    __shared__ float lds_v[16];
    __shared__ int lds_i[16];
    const uint16_t* row = x + blockIdx.x * stride;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int chunks = n / 8;
    for (int c = threadIdx.x; c < chunks; c += 1024) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + c * 8));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = bf16_lo(v[k]), hi = bf16_hi(v[k]);
            if (argmax_better(lo, c * 8 + 2 * k, best, bidx)) { best = lo; bidx = c * 8 + 2 * k; }
            if (argmax_better(hi, c * 8 + 2 * k + 1, best, bidx)) { best = hi; bidx = c * 8 + 2 * k + 1; }
        }
    }
    for (int i = chunks * 8 + threadIdx.x; i < n; i += 1024) {              // tail when n % 8 != 0
        const float f = bf16_lo((uint32_t)row[i]);
        if (argmax_better(f, i, best, bidx)) { best = f; bidx = i; }
    }
    // wave reduce (torch.argmax order: NaN first, value desc, index asc), then across the 16 waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bidx, off, 64);
        if (argmax_better(ov, oi, best, bidx)) { best = ov; bidx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds_v[wave] = best; lds_i[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (argmax_better(lds_v[w], lds_i[w], best, bidx)) { best = lds_v[w]; bidx = lds_i[w]; }
        bidx = bidx < 0 ? 0 : (bidx >= n ? n - 1 : bidx);             // n >= 1 (host): always a valid column
        if (out) out[blockIdx.x] = bidx;
        if (adv.input_ids) advance_row(adv, blockIdx.x, bidx);
    }
}

}  // namespace

int launch_argmax_rows(int64_t* out, const void* x, int n_rows, int n, int64_t stride, const AdvanceArgs& adv, hipStream_t stream) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(n_rows), dim3(1024), 0, stream, out, (const uint16_t*)x, n, stride, adv);
    return check_launch("argmax_rows");
}

int launch_residual_add_pack(void* residual, const void* y, void* packed, int n_rows, int hidden, int64_t res_stride, int64_t y_stride,
                             hipStream_t stream) {
    if (n_rows == 0) return 0;
    const int64_t total = (int64_t)n_rows * (hidden / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(residual_add_pack_kernel, dim3(blocks), dim3(256), 0, stream, (uint16_t*)residual, (const uint16_t*)y,
                       (uint16_t*)packed, n_rows, hidden, res_stride, y_stride);
    return check_launch("residual_add_pack");
}

int launch_argmax_candidates(const float* cand_val, const int32_t* cand_idx, int groups, int64_t cand_stride, int n_rows, int vocab,
                             const AdvanceArgs& adv, hipStream_t stream) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(argmax_candidates_kernel, dim3(n_rows), dim3(256), 0, stream, cand_val, cand_idx, groups, cand_stride, vocab, adv);
    return check_launch("argmax_candidates");
}

int launch_add_rmsnorm(void* out, const void* x, void* residual, const void* w, float eps, int n_rows, int hidden,
                       int64_t x_stride, int64_t out_stride, int64_t res_stride, hipStream_t stream) {
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(n_rows), dim3(ROW_THREADS), 0, stream, (uint16_t*)out, (const uint16_t*)x,
                       (uint16_t*)residual, (const uint16_t*)w, eps, hidden, x_stride, out_stride, res_stride);
    return check_launch("add_rmsnorm");
}

int launch_silu_mul(void* out, const void* gate_up, int n_rows, int inter, int64_t gu_stride, int64_t out_stride, hipStream_t stream) {
    if (n_rows == 0) return 0;
    const int64_t total = (int64_t)n_rows * (inter / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(blocks), dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)gate_up, n_rows, inter,
                       gu_stride, out_stride);
    return check_launch("silu_mul");
}

int max_rmsnorm_hidden(void) { return ROW_THREADS * MAX_CHUNKS * 8; }

}  // namespace nvh
