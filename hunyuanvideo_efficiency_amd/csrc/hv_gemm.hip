// MFMA GEMM with fused epilogues for the DiT linears (K2, K7, K8, K10, K14-dequantised) and - through the same
// main loop with a gathering A-operand - the VAE's CausalConv3d as an implicit GEMM (K15 + K17):
//     C[M,N] = A[M,K] . W[N,K]^T (+ bias[N])  ->  activation / gated residual / column split / fp32 out
// Both operands are K-contiguous ("NT"), exactly nn.Linear's layout, so every MFMA fragment is one
// 16-byte LDS read.  Design for gfx950:
//   * 256x256x64 tile, 512 threads = 8 waves as 2(M) x 4(N); each wave owns 128x64 outputs as
//     8x4 v_mfma_f32_16x16x32_{bf16,f16} accumulators (128 VGPRs).  The MFMA is issued "swapped"
//     (D = Wfrag . Afrag^T) so that a lane's 4 accumulator registers are 4 consecutive n of one row m;
//     the W rows are dealt to MFMA rows in an interleaved order so that two n-repeats give a lane
//     8 consecutive n -> 16-byte global stores.
//   * operands stream HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered
//     (2 x 64 KiB); LDS images are lane-linear as the DMA requires, with the XOR bank swizzle applied
//     on the SOURCE address and on the read (both-sides rule) -> conflict-free ds_read_b128.
//   * conv mode: row m is an output voxel (t,h,w) of a channels-last [T,H,W,C] tensor and K runs over
//     (tap, channel); the per-lane DMA source address is the tap-shifted voxel with its indices CLAMPED
//     (= replicate padding, incl. the causal 2-frame front pad in T) and optionally halved (= the nearest
//     upsample that precedes the conv in UpsampleCausal3D, first frame not repeated in T).  Padding and
//     upsampling therefore cost no memory pass at all.
//   * 1-D grid, remapped so each XCD gets a contiguous chunk of tiles (shared L2) and 32 co-resident
//     tiles form a 4(M) x 8(N) patch (A and W panels are shared inside an XCD).
// Algorithmic work: 2*M*N*K flop; HBM bytes >= 2*(M*K + N*K + M*N).
#include "hv_common.hpp"
#include "../../include/hv_kernels.h"

namespace {

constexpr int BM = 256, BK = 64;                 // BN is a template parameter: 256 (default) or 128 (N <= 128: VAE 128-channel convs)
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB per A tile (W tile: BN * BK * 2)
template <int BN> constexpr int stage_bytes() { return TILE_BYTES + BN * BK * 2; }
template <int BN> constexpr int lds_bytes() { return 2 * stage_bytes<BN>(); }   // double buffered: 128 KiB (BN=256) / 96 KiB (BN=128)
constexpr int GROUP_M = 4;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct BF16T {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(uint32_t w) { return bf2f_lo(w); }
    static __device__ __forceinline__ float hi(uint32_t w) { return bf2f_hi(w); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf2(a, b); }
    static __device__ __forceinline__ float round(float f) { return rbf(f); }
};
struct F16T {
    typedef f16x8 vec8;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
    }
    static __device__ __forceinline__ float round(float f) { return (float)(_Float16)f; }
};

struct GemmArgs {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    const uint16_t* bias;
    int M, N, K;
    uint16_t* out0; int64_t ld0; int act0;
    int n_split;
    uint16_t* out1; int64_t ld1; int act1;
    const uint16_t* gate;
    const uint16_t* res; int64_t ld_res;
    float* out_f32;            // if set: plain fp32 result (acc + bias) to out_f32[m*ld0 + n], nothing else
    int tiles_m, tiles_n;
    // conv mode (ksz = 3): output grid cT x cH x cW, source tensor sT x sH x sW (differs when upsampling), cin per tap
    int cT, cH, cW, sT, sH, sW, up_t, up_hw, cin;
    int mt, mh, mw, bH, bW;   // strided conv: source coordinate = output coordinate * m; bH/bW: extent the index is clamped to
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_w(int row) { return (((row >> 3) & 3) << 1) | ((row >> 1) & 1); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return gelu_tanh_f(v);
    if (act == 2) return silu_f(v);
    return v;
}

template <typename DT, bool CONV, int BN>
__global__ __launch_bounds__(512, 2) void gemm_kernel(GemmArgs g) {
    constexpr int STAGE_BYTES = stage_bytes<BN>();
    constexpr int NREP = BN / 64;              // n-repeats of 16 per wave (4 waves along N)
    constexpr int WPIECES = BN / 64;           // 64-row DMA pieces of the W tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int nwg = gridDim.x;
    int lin;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int band = lin / (GROUP_M * g.tiles_n), rem = lin % (GROUP_M * g.tiles_n);
    const int band_m0 = band * GROUP_M;
    const int gm = min(GROUP_M, g.tiles_m - band_m0);
    const int tm = band_m0 + rem % gm, tn = rem / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging addresses: 4 DMA pieces per operand per K-tile, each 512 thr x 16 B = 64 rows
    const int srow = tid >> 3, scp = tid & 7;  // row within a 64-row piece, 16-B chunk position in LDS
    // conv: the source row of tap (dt,dh,dw) separates into a frame, a line and a column term.  All nine terms of each staged row
    // are formed ONCE here as byte offsets (replicate/causal padding = the clamps, nearest upsample = the halvings, stride = the
    // coordinate multipliers); the K loop only selects three of them with wave-uniform conditions and adds: no multiplies, clamps or
    // 64-bit arithmetic per K-tile (that arithmetic used to cost about as many issue cycles as the tile's MFMAs at BN = 128).
    uint32_t offT[4][3], offH[4][3], offW[4][3];
    uint32_t tap_off[4] = {0u, 0u, 0u, 0u};       // byte offsets of the current tap's source rows (refreshed when the tap changes)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 64 + srow;
        const int ar = min(m0 + row, g.M - 1);  // clamp: tails read valid rows
        if (CONV) {
            const int vw = (ar % g.cW) * g.mw;
            const int th = ar / g.cW;
            const int vh = (th % g.cH) * g.mh;
            const int vt = (th / g.cH) * g.mt;
            const uint32_t row_bytes = (uint32_t)g.lda * 2u;
            uint32_t aT[3], aH[3], aW[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int ti = max(vt + d - 2, 0);
                if (g.up_t) ti = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
                aT[d] = (uint32_t)ti * (uint32_t)(g.sH * g.sW) * row_bytes;
                aH[d] = (uint32_t)(min(max(vh + d - 1, 0), g.bH - 1) >> g.up_hw) * (uint32_t)g.sW * row_bytes;
                aW[d] = (uint32_t)(min(max(vw + d - 1, 0), g.bW - 1) >> g.up_hw) * row_bytes + (uint32_t)((scp ^ swz_a(row)) << 4);
            }
            offT[i][0] = aT[0]; offT[i][1] = aT[1] - aT[0]; offT[i][2] = aT[2] - aT[1];
            offH[i][0] = aH[0]; offH[i][1] = aH[1] - aH[0]; offH[i][2] = aH[2] - aH[1];
            offW[i][0] = aW[0]; offW[i][1] = aW[1] - aW[0]; offW[i][2] = aW[2] - aW[1];
        }
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;  // this wave's 1 KiB slice of each piece; scalar: the DMA
                                                                        // destination goes through M0, no VALU/readfirstlane per piece
    // Plain GEMM: source = scalar tile base (advanced by scalar adds per K-tile) + invariant per-lane byte offset -> the saddr form
    // of global_load_lds, no 64-bit vector address arithmetic in the K loop.  (Conv mode gathers a different row per tap.)
    const char* a_tile = reinterpret_cast<const char*>(g.A + (int64_t)m0 * g.lda);
    const char* w_tile = reinterpret_cast<const char*>(g.W + (int64_t)n0 * g.ldw);
    uint32_t a_o[4], w_o[WPIECES];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 64 + srow;
        a_o[i] = (uint32_t)(((int64_t)(min(m0 + row, g.M - 1) - m0) * g.lda + ((scp ^ swz_a(row)) << 3)) * 2);
    }
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
        const int row = i * 64 + srow;
        w_o[i] = (uint32_t)(((int64_t)(min(n0 + row, g.N - 1) - n0) * g.ldw + ((scp ^ swz_w(row)) << 3)) * 2);
    }

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave_lds;
        const int koff = kt * BK;
        if (CONV) {
            const int tap = koff / g.cin, c0 = koff - tap * g.cin;
            const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            uint64_t abv = reinterpret_cast<uint64_t>(g.A + c0);
            asm volatile("" : "+s"(abv));
            const char* ab = reinterpret_cast<const char*>(abv);
            if (c0 == 0) {      // a new tap (every cin/64 K-tiles): wave-uniform branch
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    // [0] = term of tap coordinate 0, [1] / [2] = increments to coordinates 1 / 2 (a select is "add or add nothing")
                    tap_off[i] = offT[i][0] + (dt >= 1 ? offT[i][1] : 0u) + (dt >= 2 ? offT[i][2] : 0u) +
                                 offH[i][0] + (dh >= 1 ? offH[i][1] : 0u) + (dh >= 2 ? offH[i][2] : 0u) +
                                 offW[i][0] + (dw >= 1 ? offW[i][1] : 0u) + (dw >= 2 ? offW[i][2] : 0u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t o = tap_off[i];
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ab + o), (lds_ptr_t)(base + i * 8192), 16, 0, 0);
            }
        } else {
            uint64_t abv = reinterpret_cast<uint64_t>(a_tile + koff * 2);
            asm volatile("" : "+s"(abv));      // keep the tile base a scalar pair: stops the compiler folding it into per-lane 64-bit pointers
            const char* ab = reinterpret_cast<const char*>(abv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t o = a_o[i];
                asm volatile("" : "+v"(o));    // opaque per iteration: the zero-extension stays next to the load (saddr + 32-bit voffset form)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ab + o), (lds_ptr_t)(base + i * 8192), 16, 0, 0);
            }
        }
        uint64_t wbv = reinterpret_cast<uint64_t>(w_tile + koff * 2);
        asm volatile("" : "+s"(wbv));
        const char* wb = reinterpret_cast<const char*>(wbv);
#pragma unroll
        for (int i = 0; i < WPIECES; ++i) {
            uint32_t o = w_o[i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wb + o), (lds_ptr_t)(base + TILE_BYTES + i * 8192), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside an operand tile), k-step 0; k-step 1 flips chunk bit 2
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[8], w_off[NREP];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int row = wm * 128 + mi * 16 + fr;
        a_off[mi] = row * 128 + ((fq ^ swz_a(row)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NREP; ++ni) {
        // MFMA row i of n-repeat ni holds W row: (ni>>1)*32 + (i>>2)*8 + (ni&1)*4 + (i&3)
        const int row = wn * (BN / 4) + (ni >> 1) * 32 + (fr >> 2) * 8 + (ni & 1) * 4 + (fr & 3);
        w_off[ni] = row * 128 + ((fq ^ swz_w(row)) << 4);
    }

    f32x4 acc[8][NREP];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < NREP; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = g.K / BK;
    stage(0, 0);
    __syncthreads();  // (the compiler drains vmcnt before the barrier: DMA landed)

    typedef typename DT::vec8 vec8;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage(buf ^ 1, kt + 1);
        const char* At = smem + buf * STAGE_BYTES;
        const char* Wt = At + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 af[8], wf[NREP];
#pragma unroll
            for (int ni = 0; ni < NREP; ++ni) wf[ni] = *reinterpret_cast<const vec8*>(Wt + (w_off[ni] ^ (ks << 6)));
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) af[mi] = *reinterpret_cast<const vec8*>(At + (a_off[mi] ^ (ks << 6)));
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NREP; ++ni) acc[mi][ni] = DT::mfma(wf[ni], af[mi], acc[mi][ni]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds, per (mi, n-repeat pair), 8 consecutive n of row m
    auto unpack = [](const u32x4& w, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = DT::lo(w[i]);
            f[2 * i + 1] = DT::hi(w[i]);
        }
    };
    const int mrow0 = m0 + wm * 128 + fr;
#pragma unroll
    for (int np = 0; np < NREP / 2; ++np) {
        const int n = n0 + wn * (BN / 4) + np * 32 + fq * 8;
        if (n >= g.N) continue;
        float b[8], gt[8];
        if (g.bias) unpack(*reinterpret_cast<const u32x4*>(g.bias + n), b);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = 0.f;
        }
        if (g.gate) unpack(*reinterpret_cast<const u32x4*>(g.gate + n), gt);
        const bool second = n >= g.n_split;
        uint16_t* obase = second ? g.out1 + (n - g.n_split) : g.out0 + n;
        const int64_t ldo = second ? g.ld1 : g.ld0;
        const int act = second ? g.act1 : g.act0;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = mrow0 + mi * 16;
            if (m >= g.M) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mi][2 * np][r] + b[r];
                v[4 + r] = acc[mi][2 * np + 1][r] + b[4 + r];
            }
            if (g.out_f32) {
                float4* o = reinterpret_cast<float4*>(g.out_f32 + (int64_t)m * g.ld0 + n);
                o[0] = make_float4(v[0], v[1], v[2], v[3]);
                o[1] = make_float4(v[4], v[5], v[6], v[7]);
                continue;
            }
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = apply_act(DT::round(v[j]), act);
            }
            if (g.res) {
                float rs[8];
                unpack(*reinterpret_cast<const u32x4*>(g.res + (int64_t)m * g.ld_res + n), rs);
                if (g.gate) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rs[j] + DT::round(DT::round(v[j]) * gt[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rs[j] + DT::round(v[j]);
                }
            }
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = DT::pack(v[2 * i], v[2 * i + 1]);
            *reinterpret_cast<u32x4*>(obase + (int64_t)m * ldo) = w;
        }
    }
}

template <typename DT, bool CONV, int BN>
int launch_bn(GemmArgs& g, hipStream_t stream) {
    static HvPerDeviceOnce once;     // one per template instantiation
    if (hv_set_max_lds(once, (const void*)gemm_kernel<DT, CONV, BN>, lds_bytes<BN>()) != HV_OK) return HV_ERR_LAUNCH;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    gemm_kernel<DT, CONV, BN><<<dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(512), lds_bytes<BN>(), stream>>>(g);
    return hv_check_launch();
}

template <typename DT, bool CONV>
int launch(GemmArgs& g, hipStream_t stream) {
    // narrow outputs (N <= 128: the VAE's 128-channel and output convs, small projections) take the 256x128 tile so that at
    // most half a tile of MFMA work is padding
    return g.N <= 128 ? launch_bn<DT, CONV, 128>(g, stream) : launch_bn<DT, CONV, 256>(g, stream);
}

int fill_common(GemmArgs& g, const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int M, int N, int K,
                void* out0, int64_t ld0, int act0, int n_split, void* out1, int64_t ld1, int act1, const void* gate,
                const void* res, int64_t ld_res) {
    if (!A || !W || !out0 || M < 0 || N <= 0 || K < BK || (K % BK) || (N & 7) || (lda & 7) || (ldw & 7) || (ld0 & 7))
        return HV_ERR_ARG;
    if (n_split <= 0 || n_split > N) n_split = N;
    if (n_split < N && (!out1 || (n_split & 7) || (ld1 & 7))) return HV_ERR_ARG;
    if ((gate && !res) || (res && (ld_res & 7))) return HV_ERR_ARG;
    if (act0 < 0 || act0 > 2 || act1 < 0 || act1 > 2) return HV_ERR_ARG;
    g = GemmArgs{};
    g.A = (const uint16_t*)A; g.lda = lda; g.W = (const uint16_t*)W; g.ldw = ldw; g.bias = (const uint16_t*)bias;
    g.M = M; g.N = N; g.K = K;
    g.out0 = (uint16_t*)out0; g.ld0 = ld0; g.act0 = act0; g.n_split = n_split;
    g.out1 = (uint16_t*)out1; g.ld1 = ld1; g.act1 = act1;
    g.gate = (const uint16_t*)gate; g.res = (const uint16_t*)res; g.ld_res = ld_res;
    return HV_OK;
}

}  // namespace

extern "C" int hv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int M, int N, int K,
                            void* out0, int64_t ld0, int act0, int n_split, void* out1, int64_t ld1, int act1,
                            const void* gate, const void* res, int64_t ld_res, hipStream_t stream) {
    GemmArgs g;
    int rc = fill_common(g, A, lda, W, ldw, bias, M, N, K, out0, ld0, act0, n_split, out1, ld1, act1, gate, res, ld_res);
    if (rc != HV_OK) return rc;
    if (M == 0) return HV_OK;
    return launch<BF16T, false>(g, stream);
}

extern "C" int hv_gemm_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int M, int N, int K,
                           void* out, int64_t ldo, int out_is_f32, const void* res, int64_t ld_res, hipStream_t stream) {
    GemmArgs g;
    int rc = fill_common(g, A, lda, W, ldw, bias, M, N, K, out, ldo, 0, 0, nullptr, 0, 0, nullptr, res, ld_res);
    if (rc != HV_OK) return rc;
    if (out_is_f32) {
        if (res) return HV_ERR_ARG;
        g.out_f32 = (float*)out;
    }
    if (M == 0) return HV_OK;
    return launch<F16T, false>(g, stream);
}

extern "C" int hv_conv3d_causal_f16(const void* x, int64_t ldx, const void* w_taps, const void* bias, void* out, int64_t ldo,
                                    int T, int H, int W, int Cin, int Cout, int up_t, int up_hw, const void* res,
                                    int64_t ld_res, hipStream_t stream) {
    // x: channels-last source [sT, sH, sW, >=Cin] (row stride ldx); output grid T x H x W (T = 1 + 2*(sT-1) when up_t,
    // H = 2*sH, W = 2*sW when up_hw); w_taps: [Cout][27][Cin] f16.
    if (T <= 0 || H <= 0 || W <= 0 || Cin < 64 || (Cin % 64) || Cout <= 0 || (up_t & ~1) || (up_hw & ~1)) return HV_ERR_ARG;
    if ((int64_t)T * H * W > 0x7fffffff) return HV_ERR_ARG;
    if ((up_t && !(T & 1)) || (up_hw && ((H & 1) || (W & 1)))) return HV_ERR_ARG;
    GemmArgs g;
    int rc = fill_common(g, x, ldx, w_taps, (int64_t)27 * Cin, bias, T * H * W, Cout, 27 * Cin, out, ldo, 0, 0, nullptr, 0, 0,
                         nullptr, res, ld_res);
    if (rc != HV_OK) return rc;
    g.cT = T; g.cH = H; g.cW = W; g.cin = Cin; g.up_t = up_t; g.up_hw = up_hw;
    g.sT = up_t ? (T + 1) / 2 : T; g.sH = H >> up_hw; g.sW = W >> up_hw;
    g.mt = g.mh = g.mw = 1; g.bH = H; g.bW = W;
    if ((int64_t)g.sT * g.sH * g.sW * ldx * 2 >= ((int64_t)1 << 32)) return HV_ERR_ARG;   // the gather uses 32-bit byte offsets (4 GiB source)
    return launch<F16T, true>(g, stream);
}

extern "C" int hv_conv3d_causal_strided_f16(const void* x, int64_t ldx, const void* w_taps, const void* bias, void* out, int64_t ldo,
                                            int sT, int sH, int sW, int Cin, int Cout, int stride_t, int stride_h, int stride_w,
                                            hipStream_t stream) {
    // DownsampleCausal3D: the same replicate/causal padding, then Conv3d with stride (1|2 per axis).  Source index of output
    // voxel (t,h,w), tap (dt,dh,dw): (t*st + dt - 2, h*sh + dh - 1, w*sw + dw - 1), clamped into the source grid.
    if (sT <= 0 || sH <= 0 || sW <= 0 || Cin < 64 || (Cin % 64) || Cout <= 0 || stride_t < 1 || stride_t > 2 || stride_h < 1 ||
        stride_h > 2 || stride_w < 1 || stride_w > 2)
        return HV_ERR_ARG;
    const int T = (sT - 1) / stride_t + 1, H = (sH - 1) / stride_h + 1, W = (sW - 1) / stride_w + 1;
    if ((int64_t)sT * sH * sW > 0x7fffffff) return HV_ERR_ARG;
    GemmArgs g;
    int rc = fill_common(g, x, ldx, w_taps, (int64_t)27 * Cin, bias, T * H * W, Cout, 27 * Cin, out, ldo, 0, 0, nullptr, 0, 0,
                         nullptr, nullptr, 0);
    if (rc != HV_OK) return rc;
    g.cT = T; g.cH = H; g.cW = W; g.cin = Cin; g.up_t = 0; g.up_hw = 0;
    g.sT = sT; g.sH = sH; g.sW = sW;
    g.mt = stride_t; g.mh = stride_h; g.mw = stride_w; g.bH = sH; g.bW = sW;
    if ((int64_t)sT * sH * sW * ldx * 2 >= ((int64_t)1 << 32)) return HV_ERR_ARG;
    return launch<F16T, true>(g, stream);
}
