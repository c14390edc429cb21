// Declarations shared by the attention translation units (hv_attention.hip: the 8-wave kernel + the C ABI entry points;
// hv_attention_w4.hip: the 4-wave x 64-row kernel with asm-owned accumulator registers).
#pragma once
#include "hv_common.hpp"
#include "../../include/hv_kernels.h"
#include <type_traits>

namespace hv_attn {

constexpr int D = 128;
constexpr int KVT = 64;
constexpr int KV_TILE_BYTES = KVT * D * 2;       // 16 KiB

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t sq, sk, sv, so;   // token strides (elements); head h lives at column h*128
    int n_q, n_kv, n_heads, n_qtiles;
    float scale_log2e;
    // KV split (load balance when the grid is only a few rounds of workgroups): blockIdx.y = split s handles keys
    // [s*split_keys, min(n_kv, (s+1)*split_keys)); partial O (unnormalised, fp32) and (m, l) go to the workspace
    int n_splits, split_keys;
    float* part_o;      // [n_splits][n_q][n_heads][128]
    float* part_ml;     // [n_splits][n_q][n_heads][2]
    int partial;        // 1: always leave the unnormalised partial (ring attention merges K/V chunks later), even unsplit
    // per head: max over the keys of |k|^2 (fp32 bit patterns, written by attn_kmax_kernel; nullable).  With it a wave can bound every
    // score of a query row by |q'| |k|_max (Cauchy-Schwarz) and, when that bound is within 90 (log2 units) of its first tile's row
    // max, run the whole key range against the bound as a STATIC maximum: no row max per tile, never a rescale.
    const unsigned* kmax2;
};
constexpr int KMAX_BYTES = 256;      // head of the workspace: 62 heads x 4 bytes + two mode flags (words 62, 63)
constexpr int KMAX_HEADS = 62;       // word 62: some wave ran against the static bound; word 63: some wave kept the online maximum (tests)

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ float half_swap_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// compile-time loop: the body sees its index as a constant expression (inline-asm "i" operands, register arrays)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// hv_attention_w4.hip: launch of the 4-wave x 64-row kernel (same AttnArgs; grid.x = n_heads * ceil(n_q / 256), grid.y = splits)
int launch_w4(const AttnArgs& a, dim3 grid, hipStream_t stream);

}  // namespace hv_attn
