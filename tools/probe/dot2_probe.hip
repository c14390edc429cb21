// probe: semantics of __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__global__ void k(const uint32_t* in, float* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        float acc = 10.0f;
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, in[i]), __builtin_bit_cast(bf16x2_t, 0x3F803F80u), acc, false);
        out[i] = acc;
    }
}
static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main() {
    const int n = 4;
    float a[n] = {1.0f, 0.5f, 3.0f, 1e-3f}, b[n] = {2.0f, 0.25f, -1.0f, 2e-3f};
    uint32_t h[n];
    for (int i = 0; i < n; ++i) h[i] = bf(a[i]) | ((uint32_t)bf(b[i]) << 16);
    uint32_t* d; float* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    float r[n]; hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("lo=%g hi=%g  ->  %g (expected %g)\n", a[i], b[i], r[i], 10.0f + a[i] + b[i]);
    return 0;
}
