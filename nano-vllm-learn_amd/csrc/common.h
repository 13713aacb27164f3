// Shared device helpers for the gfx950 attention kernels: bf16 packing, wave64 cross-lane
// primitives (DPP within a 16-lane row, permlane swaps across rows), error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvh {

constexpr int kWave = 64;                       // CDNA wavefront width; never 32
constexpr float kLog2e = 1.4426950408889634f;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short bf16x8_s __attribute__((ext_vector_type(8)));   // MFMA A/B fragment: 8 bf16 in 4 VGPRs

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xFFFF0000u); }

// round-to-nearest-even, NaN preserved (plain casts lower to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t r;
    r.x = (__bf16)lo;
    r.y = (__bf16)hi;
    return *reinterpret_cast<uint32_t*>(&r);
}

// acc + a.lo*b.lo + a.hi*b.hi with bf16 operands packed two per dword (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf16x2_t*>(&a), *reinterpret_cast<bf16x2_t*>(&b), acc, false);
}

// ---- cross-lane -------------------------------------------------------------------------------
// DPP controls (GFX9 encoding): quad_perm = p0 | p1<<2 | p2<<4 | p3<<6; row_ror:n = 0x120+n;
// row_mirror = 0x140 (lane i <-> 15-i of a 16-lane row); row_half_mirror = 0x141 (i <-> 7-i of 8).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// value of the lane whose index differs in bit log2(MASK), for MASK in {1,2,4,8} (inside one 16-lane row).
// MASK 4 uses row_half_mirror and MASK 8 row_ror:8... those pair lane i with 7-i / i+8: both connect the
// two halves a sum/max butterfly needs, so they are only valid inside symmetric all-reduces (below).
template <int MASK>
__device__ __forceinline__ float pair_in_row(float x) {
    if constexpr (MASK == 1) return dpp_f32<0xB1>(x);          // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp_f32<0x4E>(x);     // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) return dpp_f32<0x141>(x);    // row_half_mirror
    else return dpp_f32<0x128>(x);                              // row_ror:8
}

// ---- cross-row exchanges (gfx950 v_permlane16_swap / v_permlane32_swap) -----------------------------------
// v_permlane32_swap vdst, src : lanes 32..63 of vdst <-> lanes 0..31 of src.
// v_permlane16_swap vdst, src : odd 16-lane rows of vdst <-> even rows of src.
// With vdst == src == x both halves of every pair end up in the two results, so a symmetric
// reduction of the two results is an all-reduce over the pair (rows r, r^1 / lanes l, l^32).
// one v_max_f32 (IEEE maxNum: the non-NaN operand wins): plain fmaxf makes hipcc canonicalise both operands first (a v_max x, x each)
// whenever it cannot prove they are already quiet — three VALU instructions instead of one on every step of a max chain
__device__ __forceinline__ float max2(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float max_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// Reduce-scatter steps: ONE swap + ONE add folds TWO values over the pair.
//   fold32(a, b): lanes 0..31 get a[l] + a[l+32], lanes 32..63 get b[l-32] + b[l].
//   fold16(a, b): even rows get a summed over (row, row+1), odd rows get b summed over (row-1, row).
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float sum_xor16(float x) { return fold16(x, x); }
__device__ __forceinline__ float sum_xor32(float x) { return fold32(x, x); }

// all-reduce (sum) over the LANES lanes that share lane/LANES; LANES in {8,16}
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    x += pair_in_row<1>(x);
    x += pair_in_row<2>(x);
    x += pair_in_row<4>(x);
    if constexpr (LANES == 16) x += pair_in_row<8>(x);
    return x;
}

// all-reduce (max) over the lanes that share lane%LANES (the 64/LANES token slots of a wave); LANES in {8,16}
template <int LANES>
__device__ __forceinline__ float slot_max(float x) {
    if constexpr (LANES == 8) x = fmaxf(x, pair_in_row<8>(x));
    x = max_xor16(x);
    x = max_xor32(x);
    return x;
}

// arg-max ordering shared by every greedy kernel: torch.argmax's — NaN counts as the maximum, ties (and several NaNs) go to the
// lowest index.  A candidate (-inf, INT_MAX) is the identity: any real element beats it, so the sentinel never survives a row
// that holds at least one element, even an all -inf or all-NaN row.
__device__ __forceinline__ bool argmax_better(float v, int i, float best, int bidx) {
    const bool vn = v != v, bn = best != best;
    if (vn != bn) return vn;
    if (!vn && v != best) return v > best;
    return i < bidx;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- host-side error plumbing ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

}  // namespace nvh
