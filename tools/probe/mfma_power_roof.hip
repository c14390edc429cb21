// Diagnostic, standalone (hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_power_roof.hip -o tools/probe/bin/mfma_power_roof):
// the sustained rate of a bare v_mfma_f32_32x32x16_bf16 stream - no LDS, no VALU, no memory traffic inside the loop - as a function of
// the DATA in the operand registers.  The part is power-limited under dense MFMA work (1400 W cap): with zeros the stream runs near the
// 2.5 PFLOP/s the clock allows, with random bf16 operands the clock drops.  That rate - not 2.5 PF - is the roof a bf16 attention or
// GEMM kernel on real activations can approach on this part.
// usage: mfma_power_roof [seconds per mode = 6] [waves per SIMD = 1] [16 | 32 = bf16 MFMA shape, 1632 = 16x16x32 f16] [extra: 1 LDS reads, 2 softmax VALU, 3 both, 4 / 8 / 16 = one exp / add / pack per MFMA]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <chrono>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NFRAG = 8;

__global__ __launch_bounds__(256) void mfma_stream(const uint4* __restrict__ a_src, const uint4* __restrict__ b_src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x;       // every wave of the block loads its own fragments
    bf16x8 a[NFRAG], b[NFRAG];
    for (int i = 0; i < NFRAG; ++i) {
        uint4 ua = a_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane];
        uint4 ub = b_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane];
        a[i] = __builtin_bit_cast(bf16x8, ua);
        b[i] = __builtin_bit_cast(bf16x8, ub);
    }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFRAG; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 2 * j) % NFRAG], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[blockIdx.x * 256 + lane] = s;     // never true: keeps the accumulators alive
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the 32x32x16 stream with the attention kernel's FILLERS beside it, results unused (what do they cost in energy?):
// EXTRA & 1: one 16-byte LDS read of random data per two MFMAs (0.5 KB per MFMA per wave, the 4-wave kernel's fragment traffic);
// EXTRA & 2: per MFMA one v_exp_f32 + one v_add_f32, per two MFMAs one v_cvt_pk_bf16_f32, on random data (its softmax VALU).
template <int EXTRA>
__global__ __launch_bounds__(256) void mfma_stream_fill(const uint4* __restrict__ a_src, const uint4* __restrict__ b_src, float* __restrict__ sink, int iters) {
    __shared__ uint4 lds[4096];      // 64 KiB
    const int lane = threadIdx.x;
    bf16x8 a[NFRAG], b[NFRAG];
    for (int i = 0; i < NFRAG; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, a_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
        b[i] = __builtin_bit_cast(bf16x8, b_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
    }
    for (int i = lane; i < 4096; i += 256) lds[i] = a_src[(size_t)blockIdx.x * NFRAG * 256 + (i & 2047)] ^ b_src[(size_t)blockIdx.x * NFRAG * 256 + (i & 2047)];
    __syncthreads();
    const unsigned laddr = (unsigned)(size_t)lds + lane * 16;      // conflict-free: consecutive lanes, consecutive 16-byte words
    float x[4], l = 0.f;
    for (int i = 0; i < 4; ++i) x[i] = -1.f - 0.37f * (float)((lane * 4 + i) % 61);    // exp2 arguments in [-24, -1]
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFRAG; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 2 * j) % NFRAG], acc[j], 0, 0, 0);
                if ((EXTRA & 1) && (j & 1) == 0) {
                    u32x4 t;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(laddr), "n"(((i * 4 + j) * 4096) & 0xffff));
                }
                if (EXTRA & 2) {
                    float e;
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x[j]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(e));
                    if (j & 1) {
                        unsigned pk;
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(e), "v"(l));
                    }
                }
                if (EXTRA & 4) {        // the pieces of the softmax VALU on their own: 4 = one v_exp_f32 per MFMA, 8 = one v_add_f32, 16 = one v_cvt_pk per MFMA
                    float e;
                    asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x[j]));
                }
                if (EXTRA & 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(x[j]));
                if (EXTRA & 16) {
                    unsigned pk;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[j]), "v"(x[(j + 1) & 3]));
                }
            }
        if (EXTRA & 1) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = l;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[blockIdx.x * 256 + lane] = s;
}

// the same stream with v_mfma_f32_16x16x32_bf16 (the GEMM kernel's shape): 16 independent accumulators, 128 MFMAs per iteration
__global__ __launch_bounds__(256) void mfma_stream16(const uint4* __restrict__ a_src, const uint4* __restrict__ b_src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x;
    bf16x8 a[NFRAG], b[NFRAG];
    for (int i = 0; i < NFRAG; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, a_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
        b[i] = __builtin_bit_cast(bf16x8, b_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
    }
    f32x4 acc[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFRAG; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + (j >> 2)) % NFRAG], b[(i + 2 * (j & 3)) % NFRAG], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j)
        for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[blockIdx.x * 256 + lane] = s;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// ... and with v_mfma_f32_16x16x32_f16 (the VAE convs' instruction): the operand bits are random fp16 values
__global__ __launch_bounds__(256) void mfma_stream16h(const uint4* __restrict__ a_src, const uint4* __restrict__ b_src, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x;
    f16x8 a[NFRAG], b[NFRAG];
    for (int i = 0; i < NFRAG; ++i) {
        a[i] = __builtin_bit_cast(f16x8, a_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
        b[i] = __builtin_bit_cast(f16x8, b_src[(size_t)(blockIdx.x * NFRAG + i) * 256 + lane]);
    }
    f32x4 acc[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFRAG; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + (j >> 2)) % NFRAG], b[(i + 2 * (j & 3)) % NFRAG], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j)
        for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[blockIdx.x * 256 + lane] = s;
}

static uint16_t to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 6.0;
    int wps = argc > 2 ? atoi(argv[2]) : 1;
    const bool f16 = argc > 3 && atoi(argv[3]) == 1632;     // third argument 1632: 16x16x32 with fp16 operands
    const bool s16 = argc > 3 && (atoi(argv[3]) == 16 || f16);       // third argument 16: the 16x16x32 shape
    const int extra = argc > 4 ? atoi(argv[4]) : 0;         // fourth argument: 1 = + LDS reads, 2 = + softmax VALU, 3 = both (32x32x16 only)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * wps;      // 256 threads = 4 waves = one per SIMD; wps blocks per CU
    const size_t n16 = (size_t)blocks * NFRAG * 256 * 8;    // bf16 values per operand array
    std::vector<uint16_t> ha(n16), hb(n16);
    uint4 *da, *db;
    float* sink;
    hipMalloc(&da, n16 * 2); hipMalloc(&db, n16 * 2); hipMalloc(&sink, (size_t)blocks * 256 * 4);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    const char* modes[] = {"zeros x zeros", "N(0,1) x N(0,1)  (Q.K^T-like)", "U(0,1) x N(0,1)  (P.V-like)", "N(0,1) x zeros", "1.0 x N(0,1)"};
    printf("%d CUs, %d wave(s) per SIMD, %s%s%s, %.0f s per mode\n", prop.multiProcessorCount, wps, f16 ? "v_mfma_f32_16x16x32_f16" : s16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16",
           extra & 1 ? " + 0.5 KB LDS read per MFMA" : "", extra & 2 ? " + exp/add/pack per MFMA" : extra == 4 ? " + v_exp_f32 per MFMA" : extra == 8 ? " + v_add_f32 per MFMA" : extra == 16 ? " + v_cvt_pk_bf16_f32 per MFMA" : "", seconds);
    for (int m = 0; m < 5; ++m) {
        for (size_t i = 0; i < n16; ++i) {
            float x = 0.f, y = 0.f;
            if (m == 1) { x = nd(rng); y = nd(rng); }
            if (m == 2) { x = ud(rng); y = nd(rng); }
            if (m == 3) { x = nd(rng); }
            if (m == 4) { x = 1.f; y = nd(rng); }
            ha[i] = f16 ? to_f16(x) : to_bf16(x); hb[i] = f16 ? to_f16(y) : to_bf16(y);
        }
        hipMemcpy(da, ha.data(), n16 * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), n16 * 2, hipMemcpyHostToDevice);
        const int iters = s16 ? 20000 : 40000;        // 32 (128) MFMAs per iteration per wave
        const double flop_per_launch = (double)blocks * 4 * iters * (s16 ? 128 * (2.0 * 16 * 16 * 32) : 32 * (2.0 * 32 * 32 * 16));
        auto launch = [&](int n) {
            if (f16) mfma_stream16h<<<blocks, 256>>>(da, db, sink, n);
            else if (s16) mfma_stream16<<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 1) mfma_stream_fill<1><<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 2) mfma_stream_fill<2><<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 3) mfma_stream_fill<3><<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 4) mfma_stream_fill<4><<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 8) mfma_stream_fill<8><<<blocks, 256>>>(da, db, sink, n);
            else if (extra == 16) mfma_stream_fill<16><<<blocks, 256>>>(da, db, sink, n);
            else mfma_stream<<<blocks, 256>>>(da, db, sink, n);
        };
        launch(1000);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        double first = 0, last = 0;
        int n = 0;
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            last = flop_per_launch / ms / 1e9;
            if (n == 0) first = last;
            ++n;
        }
        printf("  %-34s first launch %7.1f TFLOP/s, sustained (launch %d) %7.1f TFLOP/s\n", modes[m], first, n, last);
        fflush(stdout);
    }
    return 0;
}
