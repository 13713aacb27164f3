// Variable-length causal prefill attention on MFMA tiles (gfx950).
// Replaces flash_attn_varlen_func at nanovllm/layers/attention.py:93-96, both modes:
//   PAGED = false : K/V are this step's packed rows            (oracle attention_sdpa.py:65-119)
//   PAGED = true  : K/V live in the paged cache via block_tables (prefix-cache hit, attention.py:90-91;
//                   bottom-right aligned causal mask; parity unpinned by the reference)
//
// Dense contraction -> MFMA (v_mfma_f32_16x16x32_bf16), flash-style online softmax, fp32 accumulate.
// Workgroup = 4 waves = a tile of BM = 64 query rows of one (sequence, q head); wave w owns rows
// 16w..16w+15.  K/V tiles of BN = 32 keys are staged through LDS (16-byte coalesced global loads,
// rows padded by 16 B so the 16 lanes of a ds_read_b128 group land on 16 distinct bank slots).
//
// Orientation (what makes the softmax and the P operand lane-local):
//   S^T = K Q^T     A = K rows (ds_read_b128), B = Q rows (registers, loaded once).
//                   C layout: lane l, reg r holds S^T[key 4(l>>4)+r][query l&15]
//                   -> a query's scores sit in 4 registers x 4 lane groups: row max/sum are 3 local ops
//                   plus a 2-step cross-row reduction; the rescale factor is one scalar per lane.
//   O^T = V^T P^T   B = P^T straight from the S^T accumulators of the two 16-key halves (k-slot j<4 ->
//                   key 4g+j, j>=4 -> key 16+4g+j-4, g = l>>4); A = V^T in the SAME k order, produced by
//                   ds_read_b64_tr_b16 (hardware transpose of a 4-key x 16-dim block) from row-major V.
//                   P enters as hi + lo bf16 halves (two MFMAs per tile) so it keeps ~16 mantissa bits.
//                   C layout: lane l, reg r holds O^T[dim 16t+4(l>>4)+r][query l&15] -> 4 contiguous
//                   output dims per lane.
// Algorithmic flops per launch: sum_seq 4*D*H*(causal pairs); bytes: Tq*H*D*2*2 + Tk*KVH*D*2*2.
#include "common.h"
#include "kernels.h"

namespace nvh {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;      // query rows per workgroup
constexpr int BN = 32;      // keys per LDS tile

template <int D, bool PAGED>
__global__ __launch_bounds__(256) void prefill_varlen_kernel(const PrefillArgs a) {
    constexpr int ROWB = D * 2 + 16;                 // padded LDS row, bytes
    constexpr int STEPS = D / 32;                    // k-steps of the QK^T contraction
    constexpr int DT = D / 16;                       // 16-dim output tiles
    constexpr int CPR = D / 8;                       // 16-byte chunks per row
    __shared__ __attribute__((aligned(16))) unsigned char lds_k[BN * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char lds_v[BN * ROWB];

    const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int q_beg = a.cu_q[b], q_end = a.cu_q[b + 1];
    const int k_beg = a.cu_k[b], k_end = a.cu_k[b + 1];
    const int sq = q_end - q_beg, sk = k_end - k_beg;
    const int q0 = qt * BM;
    if (q0 >= sq) return;
    const int kh = head / (a.h / a.kvh);
    const int shift = sk - sq;                       // bottom-right alignment: query r sees keys <= r + shift

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15;                        // query column of this lane
    const int lg = lane >> 4;                        // lane group: k-block of the operands / key rows of C

    // ---- Q fragments (B operand of S^T): Q[row][32*step + 8*lg .. +8]
    const int my_q = q0 + wave * 16 + lq;            // query index inside the sequence
    const bool q_ok = my_q < sq;
    bf16x8 qf[STEPS];
    {
        const uint16_t* qp = a.q + (int64_t)(q_beg + (q_ok ? my_q : 0)) * a.q_row_stride + (int64_t)head * D + lg * 8;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            u32x4 raw = q_ok ? *reinterpret_cast<const u32x4*>(qp + st * 32) : u32x4{0, 0, 0, 0};
            qf[st] = *reinterpret_cast<bf16x8*>(&raw);
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;            // per query column; l_run is this lane group's share

    // keys any row of this workgroup can see: 0 .. min(sk, q0 + BM + shift) - 1
    int kv_end = q0 + BM + shift;
    if (kv_end > sk) kv_end = sk;
    const int q_pos = my_q + shift;                  // last key this lane's query may see

    for (int kv0 = 0; kv0 < kv_end; kv0 += BN) {
        // ---- stage K and V tiles [BN][D] into LDS (zero rows past the sequence end)
        __syncthreads();                             // previous tile fully consumed
#pragma unroll
        for (int c = tid; c < BN * CPR; c += 256) {
            const int r = c / CPR, ch = c - r * CPR;
            const int kidx = kv0 + r;
            u32x4 kd = {0, 0, 0, 0}, vd = {0, 0, 0, 0};
            if (kidx < sk) {
                int64_t koff, voff;
                if constexpr (PAGED) {
                    const int blk = kidx / a.block_size;
                    const int bid = a.block_tables[b * a.bt_row_stride + blk];
                    koff = (((int64_t)bid * a.block_size + (kidx - blk * a.block_size)) * a.kvh + kh) * D + ch * 8;
                    voff = koff;
                } else {
                    koff = (int64_t)(k_beg + kidx) * a.k_row_stride + (int64_t)kh * D + ch * 8;
                    voff = (int64_t)(k_beg + kidx) * a.v_row_stride + (int64_t)kh * D + ch * 8;
                }
                kd = *reinterpret_cast<const u32x4*>(a.k + koff);
                vd = *reinterpret_cast<const u32x4*>(a.v + voff);
            }
            *reinterpret_cast<u32x4*>(lds_k + r * ROWB + ch * 16) = kd;
            *reinterpret_cast<u32x4*>(lds_v + r * ROWB + ch * 16) = vd;
        }
        __syncthreads();

        // ---- S^T for the two 16-key halves
        f32x4 st_acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(lds_k + (16 * t + lq) * ROWB + (st * 32 + lg * 8) * 2);
                st_acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[st], st_acc[t], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (log2 domain)
        float sv[8];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * t + 4 * lg + r;
                const bool ok = key <= q_pos && key < sk;
                const float x = ok ? st_acc[t][r] * a.scale_log2 : -INFINITY;
                sv[4 * t + r] = x;
                mx = fmaxf(mx, x);
            }
        mx = max_xor16(mx);
        mx = max_xor32(mx);
        const float m_new = fmaxf(m_run, mx);
        // rows that have seen no key yet keep m = -inf; use 0 as the reference point so exp2 stays finite
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - m_use);            // m_run = -inf -> 0
        float psum = 0.f;
        float pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pv[i] = fast_exp2(sv[i] - m_use);
            psum += pv[i];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] *= alpha;
        // P as hi + lo bf16 (~16 mantissa bits): a single bf16 P costs ~4e-3 abs at |o| ~ 3, over the 1e-3 parity bar
        bf16x8 pf, pf_lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pf[i] = (__bf16)pv[i];
            pf_lo[i] = (__bf16)(pv[i] - (float)pf[i]);
        }

        // ---- O^T += V^T P^T
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            // lane 4q+p of its 16-lane group addresses key row (4*lg + q), dims 16t + 4p .. +3
            const int vq = lq >> 2, vp = lq & 3;
            const unsigned char* base = lds_v + (4 * lg + vq) * ROWB + (16 * t + 4 * vp) * 2;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + 16 * ROWB));
            const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // whole-register concat, no repack
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf_lo, o[t], 0, 0, 0);
        }
    }

    // ---- finalise: total row sum over the 4 lane groups, normalise, store 4 contiguous dims per tile
    l_run = sum_xor16(l_run);
    l_run = sum_xor32(l_run);
    if (!q_ok) return;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    const int64_t orow = ((int64_t)(q_beg + my_q) * a.h + head) * D;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int d0 = 16 * t + 4 * lg;
        const f32x4 r = o[t] * inv;
        if (a.out_f32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + orow + d0) = r;
        } else {
            u32x2 pk = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + orow + d0) = pk;
        }
    }
}

template <int D>
int launch_d(const PrefillArgs& a, hipStream_t stream) {
    dim3 grid((a.max_seqlen_q + BM - 1) / BM, a.h, a.batch);
    if (a.block_tables) hipLaunchKernelGGL((prefill_varlen_kernel<D, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((prefill_varlen_kernel<D, false>), grid, dim3(256), 0, stream, a);
    return check_launch("prefill_varlen");
}

}  // namespace

int launch_prefill_varlen(const PrefillArgs& a, hipStream_t stream) {
    if (a.batch == 0 || a.max_seqlen_q == 0) return 0;
    return a.hd == 64 ? launch_d<64>(a, stream) : launch_d<128>(a, stream);
}

}  // namespace nvh
