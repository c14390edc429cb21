// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the HunyuanVideo hot path.
// Wave = 64 lanes everywhere; bf16 values travel as raw 16-bit words and are widened in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HV_OK 0
#define HV_ERR_ARG (-1)
#define HV_ERR_LAUNCH (-2)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_t;  // storage type in HBM / LDS

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf2f_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf2f_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
// round-to-nearest-even f32 -> bf16 (a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950, NaN-safe)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
// round a float to the nearest bf16 value, kept as float (the "bf16-emulated" contract points)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ void unpack8(const u32x4& w, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf2f_lo(w[i]);
        f[2 * i + 1] = bf2f_hi(w[i]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    // tanh(u) = 1 - 2/(exp(2u)+1); exp via exp2
    float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);  // exp(2u)
    // v_rcp_f32 (1 ulp) instead of an IEEE division (a ~10-instruction expansion per element): the result is rounded to bf16 /
    // fp16 right after, 13+ bits coarser than the reciprocal's error
    float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Raising a kernel's dynamic-LDS limit is a per-DEVICE attribute: one process may drive several GPUs, so the "already done"
// flag is kept per device ordinal (hipGetDevice is a thread-local read, no driver call).
struct HvPerDeviceOnce { bool done[64] = {}; };
static inline int hv_set_max_lds(HvPerDeviceOnce& once, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return HV_ERR_LAUNCH;
    if (!once.done[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return HV_ERR_LAUNCH;
        once.done[dev] = true;
    }
    return HV_OK;
}

static inline int hv_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HV_OK : HV_ERR_LAUNCH;
}
