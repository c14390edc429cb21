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
#include <cstdlib>
#include <type_traits>

namespace {

// HV_GEMM_2STAGE=1 in the environment keeps every plain GEMM on the 2-stage main loop.  Read per call (a getenv, no driver
// work) so one process can A/B the two loops and check them against each other bit for bit (tests/test_gpu_gemm_pipeline.py).
inline bool hv_gemm_force_2stage() {
    const char* e = std::getenv("HV_GEMM_2STAGE");
    return e && e[0] == '1';
}

constexpr int BM = 256, BK = 64;                 // BN is a template parameter: 256 (default) or 128 (N <= 128: VAE 128-channel convs)
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB per A tile (W tile: BN * BK * 2)
template <int BN> constexpr int stage_bytes() { return TILE_BYTES + BN * BK * 2; }
template <int BN> constexpr int lds_bytes() { return 2 * stage_bytes<BN>(); }   // double buffered: 128 KiB (BN=256) / 96 KiB (BN=128)
#ifndef HV_GROUP_M
#define HV_GROUP_M 4
#endif
constexpr int GROUP_M = HV_GROUP_M;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct BF16T {
    typedef bf16x8 vec8;
    static constexpr int ESIZE = 2;
    static constexpr bool FP8 = false;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(uint32_t w) { return bf2f_lo(w); }
    static __device__ __forceinline__ float hi(uint32_t w) { return bf2f_hi(w); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf2(a, b); }
    static __device__ __forceinline__ float round(float f) { return rbf(f); }
};
struct F16T {
    typedef f16x8 vec8;
    static constexpr int ESIZE = 2;
    static constexpr bool FP8 = false;
    static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
    }
    static __device__ __forceinline__ float round(float f) { return (float)(_Float16)f; }
};

// FP8-MFMA path: both operands are OCP e4m3fn bytes (a 128-byte LDS row holds 128 k instead of 64), one
// v_mfma_scale_f32_16x16x128_f8f6f4 per (m-rep, n-rep) and K-tile with all block scales = 1 (E8M0 127): 2x the bf16 FLOPs per
// clock on the same bytes.  Outputs are bf16, so the epilogue helpers are BF16T's.  A lane's operand is its two 16-byte chunks
// (fq, 4 + fq) of the row - the same chunks for A and W, so element j of lane group fq meets the same k in both (the only
// requirement of the contraction; checked with exact integer data in tests/test_gpu_fp8_mfma.py).
struct FP8T : BF16T {
    static constexpr int ESIZE = 1;
    static constexpr bool FP8 = true;
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    // 32 bytes of one LDS row = one lane's operand: the two 16-byte chunks are read straight into the halves of an 8-register tuple
    static __device__ __forceinline__ i32x8 load2(const char* p_lo, const char* p_hi) {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(p_lo), hi = *reinterpret_cast<const i32x4*>(p_hi);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ f32x4 mfma8(i32x8 a, i32x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A: fp8 e4m3*/, 0 /*B: fp8 e4m3*/, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
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
    const float* a_scale;      // FP8 path: per-row activation scales [M] (fp32) ...
    const uint16_t* w_scale;   // ... and the per-tensor weight scale (one bf16, fp8_optimization.py `fp8_scale`): y = acc * a_scale[m] * w_scale
    int tiles_m, tiles_n;
    // conv mode (ksz = 3): output grid cT x cH x cW, source tensor sT x sH x sW (differs when upsampling), cin per tap
    int cT, cH, cW, sT, sH, sW, up_t, up_hw, cin;
    int mt, mh, mw, bH, bW;   // strided conv: source coordinate = output coordinate * m; bH/bW: extent the index is clamped to
    // sub-pixel form of (nearest upsample -> conv), see gemm8_kernel<.., SP>: M tiles [sp_tile0[c], sp_tile0[c+1]) belong to output
    // parity class c (sp_rows[c] class-local rows = source-grid voxels), weights [class][N][sp_ntap * cin], tap offsets from sp_tab
    int sp_tile0[9], sp_rows[8], sp_ntap;
    const uint32_t* sp_tab;   // [class][sp_ntap]: (ot + 8) | (oh + 8) << 4 | (ow + 8) << 8, source voxel = clamp(class voxel + offset)
    // GroupNorm statistics of the OUTPUT, taken in the epilogue from the values it stores: gn_partial[b][n][2] = (sum, sum of squares)
    // over the valid rows of 64-row block b (b = tile * 4 + wave row block), one writer per entry, fixed order: run-to-run identical
    float* gn_partial;
};

// output row of class-local voxel m of parity class (pt, ph, pw): the class's voxels are the source grid (frames x sH x sW), output
// (t, h, w) = (2 kt + pt | kt, 2 kh + ph, 2 kw + pw) in the cT x cH x cW output grid
struct SpRowMap {
    int sH, sW, cH, cW, up_t, pt, ph, pw;
    __device__ __forceinline__ int64_t operator()(int m) const {
        const int kw = m % sW, th = m / sW, kh = th % sH, kt = th / sH;
        const int t = up_t ? 2 * kt + pt : kt;
        return ((int64_t)t * cH + 2 * kh + ph) * cW + 2 * kw + pw;
    }
};
struct IdRowMap {
    __device__ __forceinline__ int64_t operator()(int m) const { return m; }
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

// ---- epilogue shared by both main loops: lane holds, per (mi, n-repeat pair), 8 consecutive n of row m
template <typename DT, int MREP, int NREP, int MSTEP = 16, typename RowMap = IdRowMap>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mrow0, int ncol0, int fq,
                                              const RowMap& rowmap = RowMap(), int gn_blk0 = 0) {
    auto unpack = [](const u32x4& w, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = DT::lo(w[i]);
            f[2 * i + 1] = DT::hi(w[i]);
        }
    };
    // mrow0: this lane's first output row (+ MSTEP per m-repeat); ncol0: the wave's first output column (+ 32 per n-repeat pair).
    // Every global read of the epilogue is issued in one batch per 32-column group BEFORE any of it is consumed (row indices
    // clamped instead of branched around, stores guarded): eight residual rows + eight row scales in flight together cost one
    // memory round trip; read one by one behind `if (m < M)` branches they cost sixteen per tile (~2 us each under load), which
    // was 30 us of the 64 us an fp8 K = 3072 tile took.
    int64_t orow[MREP];                     // output (and residual) row of each m-repeat; clamped rows are loaded, never stored
#pragma unroll
    for (int mi = 0; mi < MREP; ++mi) orow[mi] = rowmap(min(mrow0 + mi * MSTEP, g.M - 1));
    float rsc[MREP];
    if (g.a_scale) {
        const float wsc = bf2f(*g.w_scale);
#pragma unroll
        for (int mi = 0; mi < MREP; ++mi) rsc[mi] = g.a_scale[min(mrow0 + mi * MSTEP, g.M - 1)] * wsc;
    } else {
#pragma unroll
        for (int mi = 0; mi < MREP; ++mi) rsc[mi] = 1.0f;
    }
#pragma unroll
    for (int np = 0; np < NREP / 2; ++np) {
        const int n = ncol0 + np * 32 + fq * 8;
        if (n >= g.N) continue;
        float b[8], gt[8];
        u32x4 rraw[MREP];
        if (g.res) {
#pragma unroll
            for (int mi = 0; mi < MREP; ++mi)
                rraw[mi] = *reinterpret_cast<const u32x4*>(g.res + orow[mi] * g.ld_res + n);
        }
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
        // GroupNorm statistics (fp16 outputs only): per 64-row block and per PAIR of adjacent columns, (sum, sum of squares) of the
        // stored fp16 values by v_dot2_f32_f16 on the packed words the store uses - [0..3] sums, [4..7] squares of pairs 0..3
        constexpr bool GN = std::is_same<DT, F16T>::value;
        constexpr int NGB = MREP / 4;             // 64-row blocks this wave covers (4 m-repeats x 16 lanes each)
        float gst[NGB][8];
#pragma unroll
        for (int bq = 0; bq < NGB; ++bq)
#pragma unroll
            for (int j = 0; j < 8; ++j) gst[bq][j] = 0.f;
#pragma unroll
        for (int mi = 0; mi < MREP; ++mi) {
            const int m = mrow0 + mi * MSTEP;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[mi][2 * np][r] * rsc[mi] + b[r];
                v[4 + r] = acc[mi][2 * np + 1][r] * rsc[mi] + b[4 + r];
            }
            if (g.out_f32) {
                if (m < g.M) {
                    float4* o = reinterpret_cast<float4*>(g.out_f32 + orow[mi] * g.ld0 + n);
                    o[0] = make_float4(v[0], v[1], v[2], v[3]);
                    o[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
                continue;
            }
            if (act == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_tanh_f(DT::round(v[j]));
            } else if (act == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = silu_f(DT::round(v[j]));
            }
            if (g.res) {
                float rs[8];
                unpack(rraw[mi], rs);
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
            if (m < g.M) *reinterpret_cast<u32x4*>(obase + orow[mi] * ldo) = w;
            if constexpr (GN) {
                if (g.gn_partial && m < g.M) {
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                    const h2_t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (not __builtin_bit_cast(h2_t, w[i]): on an element of an ext_vector it reads element 0 for every i - hipcc 7.2)
                        const uint32_t wi = w[i];
                        h2_t hv;
                        __builtin_memcpy(&hv, &wi, 4);
                        gst[mi / 4][i] = __builtin_amdgcn_fdot2(hv, one, gst[mi / 4][i], false);
                        gst[mi / 4][4 + i] = __builtin_amdgcn_fdot2(hv, hv, gst[mi / 4][4 + i], false);
                    }
                }
            }
        }
        if constexpr (GN) {
            if (g.gn_partial) {
                // The 16 lanes that share fq hold the rows of these 8 columns (all 16 took the same `n < N` branch).  Butterfly
                // transpose-reduction over lane bits 3..1: each step a lane keeps half of its values and receives the partner's copy
                // of that half (8 -> 4 -> 2 -> 1 values), then one plain exchange over bit 0: 8 shuffles instead of 32.  Lane pair
                // q = fr >> 1 ends up with value q = [square?][pair].  gn_partial[b][n .. n+7][2]: the pair's total goes to its
                // even column, the odd column's slot is zero (a GroupNorm group holds whole pairs: C / groups is even).
                const int fr_ = threadIdx.x & 15;
#pragma unroll
                for (int bq = 0; bq < NGB; ++bq) {
                    float a4[4], a2[2], a1;
                    const bool b3 = fr_ & 8, b2 = fr_ & 4, b1 = fr_ & 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float keep = b3 ? gst[bq][4 + i] : gst[bq][i], send = b3 ? gst[bq][i] : gst[bq][4 + i];
                        a4[i] = keep + __shfl_xor(send, 8, 64);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float keep = b2 ? a4[2 + i] : a4[i], send = b2 ? a4[i] : a4[2 + i];
                        a2[i] = keep + __shfl_xor(send, 4, 64);
                    }
                    {
                        const float keep = b1 ? a2[1] : a2[0], send = b1 ? a2[0] : a2[1];
                        a1 = keep + __shfl_xor(send, 2, 64);
                    }
                    a1 += __shfl_xor(a1, 1, 64);
                    // value held: square = b3, pair = 2*b2 + b1
                    const int pair = (b2 ? 2 : 0) + (b1 ? 1 : 0);
                    float* o = g.gn_partial + ((int64_t)(gn_blk0 + bq) * g.N + n + 2 * pair + (fr_ & 1)) * 2 + (b3 ? 1 : 0);
                    *o = (fr_ & 1) ? 0.f : a1;
                }
            }
        }
    }
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

    gemm_epilogue<DT, 8, NREP>(g, acc, m0 + wm * 128 + fr, n0 + wn * (BN / 4), fq, IdRowMap(), tm * 4 + wm * 2);
}

// =====================================================================================================================
// Plain GEMM, N > 128, K >= 192: the pipelined main loop ("8 phases": two K-tiles x four 16-MFMA phases per iteration).
//
// Why: the 2-stage loop above issues a whole K-tile's DMA, 24 fragment reads and 64 MFMAs per wave between two
// __syncthreads(), each of which drains the LDS-DMA queue (vmcnt(0)): the matrix pipe idles while every wave of the CU reads
// its fragments at once, and again while the next tile's DMA lands (36 % SQ_WAIT_ANY, 43-44 % of the MFMA roof).  Here
//   * a K-tile is cut into four 16 KiB half-tiles in the order they are consumed - Am0 (rows wm*128 + [0,64) of both wave
//     rows), Bn0 (W rows wn*64 + [0,32) of all four wave columns), Bn1, Am1 - and a phase computes one 64 x 32 output
//     quadrant per wave (4 x 2 accumulators x 2 k-steps = 16 MFMAs): Q(m0,n0), Q(m0,n1), Q(m1,n1), Q(m1,n0); the fragments
//     it needs beyond what is already in registers are ONE half-tile's (12 / 4 / 8 / 0 ds_read_b128);
//   * every phase issues one half-tile of LDS-DMA (2 global_load_lds_dwordx4 per thread) for the K-tile TWO ahead into the
//     slot whose last reader finished the phase before, and waits with a COUNTED vmcnt(10): five half-tiles (80 KiB per CU)
//     stay in flight across the raw s_barriers, the DMA queue is never drained inside the loop;
//   * the two wave rows (waves 0-3 / 4-7 = the two waves of every SIMD) run the same program one barrier apart, so in every
//     barrier interval one of them is in its MFMA cluster and the other in its fragment reads + DMA issue: the matrix pipe
//     of each SIMD is fed back to back, and at most four waves read LDS at a time.
// Hazards (guide: "Read a staged buffer one phase AFTER the wait that retires it"; WAR: a slot is re-staged only after a barrier
// that follows its readers' lgkmcnt(0)): with phase index f = 4*tile + p, phase f reads half-tiles <= f+1 and issues half-tile
// f+7; each wave waits, before the barrier that ends its read segment, for its own share of half-tile f+2 (vmcnt(10)) and for
// its fragment reads (lgkmcnt(0)).  Both wave rows pass a barrier between any wait and the first read that depends on it, and
// between the last read of a slot and the DMA that overwrites it (slot of Am0/Bn0: free after phase 1, re-staged in phases
// 2/3; Bn1: after 2, in 4; Am1: after 3, in the next tile's phase 1).
constexpr int HT_BYTES = 16384;            // half-tile: 128 rows x 64 k x 2 B
constexpr int BUF8_BYTES = 4 * HT_BYTES;   // [Am0 | Bn0 | Bn1 | Am1] of one K-tile
constexpr int LDS8_BYTES = 2 * BUF8_BYTES; // 128 KiB

template <typename DT, bool CONV = false, bool SP = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs g) {
    typedef typename DT::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

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
    int m0 = tm * BM;
    const int n0 = tn * 256;
    // sub-pixel mode: the M tile's output parity class and its class-local rows
    int sp_c = 0, sp_M = g.M;
    if constexpr (SP) {
#pragma unroll
        for (int c = 1; c < 8; ++c) sp_c += tm >= g.sp_tile0[c] ? 1 : 0;
        m0 = (tm - g.sp_tile0[sp_c]) * BM;
        sp_M = g.sp_rows[sp_c];
    }
    const int sp_pt = g.up_t ? sp_c >> 2 : 0;

    // ---- staging: thread -> (row srow of a 64-row piece, 16-B chunk position scp); 2 pieces per half-tile
    const int srow = tid >> 3, scp = tid & 7;
    uint32_t a_o[2][2], w_o[2][2];      // [half][piece] invariant per-lane byte offsets from the scalar K-tile base
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int arow = i * 128 + h * 64 + srow;                       // tile row of LDS row (i*64 + srow) of half-tile Am<h>
            if constexpr (CONV) {
                // conv mode: row = output voxel.  Its (t, h, w) source coordinates (already multiplied by the stride) are kept PACKED
                // in one register; the byte offset of the current tap's source row is re-formed from them only when the half-tile
                // stream this row belongs to crosses into a new tap (every cin/64 K-tiles), see conv_tap().
                if constexpr (SP) {
                    // sub-pixel mode: row = voxel (kt, kh, kw) of the SOURCE grid; the frame coordinate is the k of "taps on frames
                    // k-1, k" (odd frames of a temporally upsampled output: k = kt + 1)
                    const int ar = min(m0 + arow, sp_M - 1);
                    const int kw = ar % g.sW, th = ar / g.sW;
                    a_o[h][i] = (uint32_t)(th / g.sH + sp_pt) | ((uint32_t)(th % g.sH) << 8) | ((uint32_t)kw << 20);
                } else {
                const int ar = min(m0 + arow, g.M - 1);
                const int vw = (ar % g.cW) * g.mw, th = ar / g.cW;
                a_o[h][i] = (uint32_t)((th / g.cH) * g.mt) | ((uint32_t)((th % g.cH) * g.mh) << 8) | ((uint32_t)vw << 20);
                }
            } else
            a_o[h][i] = (uint32_t)((int64_t)(min(m0 + arow, g.M - 1) - m0) * g.lda * DT::ESIZE + ((scp ^ swz_a(srow)) << 4));
            const int lr = i * 64 + srow;                                   // LDS row of half-tile Bn<h>
            const int wrow = (lr >> 5) * 64 + h * 32 + (lr & 31);           // wave column (lr>>5), its W rows [h*32, h*32+32)
            w_o[h][i] = (uint32_t)((int64_t)(min(n0 + wrow, g.N - 1) - n0) * g.ldw * DT::ESIZE + ((scp ^ swz_w(lr)) << 4));
        }
    const int wave_lds = wave * 1024;
    const char* a_tile = CONV ? reinterpret_cast<const char*>(g.A) : reinterpret_cast<const char*>(g.A) + (int64_t)m0 * g.lda * DT::ESIZE;
    const char* w_tile = reinterpret_cast<const char*>(g.W) + ((int64_t)n0 + (SP ? (int64_t)sp_c * g.N : 0)) * g.ldw * DT::ESIZE;
    // conv mode: byte offsets of the CURRENT tap's source rows, one set per A half-tile stream (Am0 and Am1 are staged in
    // different phases for different K-tiles, so each stream crosses tap boundaries on its own)
    uint32_t tap_off[2][2] = {{0u, 0u}, {0u, 0u}};
    const uint32_t a_chunk = (uint32_t)((scp ^ swz_a(srow)) << 4);
    const int lg_cin = CONV ? 31 - __builtin_clz((unsigned)g.cin) : 0;       // cin is a power of two >= 256 here (host-side dispatch)
    auto conv_tap = [&](int h, int tap) {      // wave-uniform tap (dt, dh, dw): replicate / causal padding = the clamps, upsample = the halvings
        const uint32_t row_bytes = (uint32_t)g.lda * 2u;
        if constexpr (SP) {
            const uint32_t e = g.sp_tab[sp_c * g.sp_ntap + tap];      // wave-uniform: a scalar load, once per cin/64 K-tiles
            const int ot = (int)(e & 15u) - 8, oh = (int)((e >> 4) & 15u) - 8, ow = (int)((e >> 8) & 15u) - 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t c = a_o[h][i];
                const int ti = max((int)(c & 255u) + ot, 0);
                const int hi = min(max((int)((c >> 8) & 4095u) + oh, 0), g.sH - 1);
                const int wi = min(max((int)(c >> 20) + ow, 0), g.sW - 1);
                tap_off[h][i] = (uint32_t)((ti * g.sH + hi) * g.sW + wi) * row_bytes + a_chunk;
            }
            return;
        }
        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t c = a_o[h][i];
            int ti = max((int)(c & 255u) + dt - 2, 0);
            if (g.up_t) ti = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
            const int hi = min(max((int)((c >> 8) & 4095u) + dh - 1, 0), g.bH - 1) >> g.up_hw;
            const int wi = min(max((int)(c >> 20) + dw - 1, 0), g.bW - 1) >> g.up_hw;
            tap_off[h][i] = (uint32_t)((ti * g.sH + hi) * g.sW + wi) * row_bytes + a_chunk;
        }
    };

    // half-tile J (0 Am0, 1 Bn0, 2 Bn1, 3 Am1) of K-tile `tile` into buffer `buf` (wave-uniform: the destination goes through M0)
    auto stage = [&](auto Jc, int buf, int tile) {
        constexpr int J = decltype(Jc)::value;
        char* dst = smem + buf * BUF8_BYTES + J * HT_BYTES + wave_lds;
        constexpr bool isA = (J == 0 || J == 3);
        constexpr int h = (J == 3 || J == 2) ? 1 : 0;
        uint64_t bv;
        if constexpr (CONV && isA) {
            const int tap = (tile * BK) >> lg_cin, c0 = tile * BK - (tap << lg_cin);
            if (c0 == 0) conv_tap(h, tap);       // a new tap for this stream: once per cin/64 K-tiles
            bv = reinterpret_cast<uint64_t>(a_tile + c0 * 2);
        } else {
            bv = reinterpret_cast<uint64_t>((isA ? a_tile : w_tile) + (int64_t)tile * (BK * 2));
        }
        asm volatile("" : "+s"(bv));
        const char* b = reinterpret_cast<const char*>(bv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t o = isA ? (CONV ? tap_off[h][i] : a_o[h][i]) : w_o[h][i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b + o), (lds_ptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };

    // ---- fragment read offsets inside a half-tile (k-step 0; k-step 1 flips chunk bit 2 = byte bit 6)
    const int fr = lane & 15, fq = lane >> 4;
    const int a_rd = (wm * 64 + fr) * 128 + ((fq ^ swz_a(fr)) << 4);                         // + mi * 2048
    const int w_lr = wn * 32 + (fr >> 2) * 8 + (fr & 3);                                     // + (ni & 1) * 4 rows = 512 B
    const int w_rd = w_lr * 128 + ((fq ^ swz_w(w_lr)) << 4);

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments: 16-bit types keep the two k-steps of a row apart ([..][ks], one MFMA each); fp8 reads both 16-byte chunks into
    // ONE 8-register operand (a 128-deep MFMA).  Same bytes, same addresses, same registers.
    typedef typename std::conditional<DT::FP8, __attribute__((ext_vector_type(8))) int, vec8>::type frag_t;
    constexpr int NKS = DT::FP8 ? 1 : 2;
    frag_t af[4][NKS], wf0[2][NKS], wf1[2][NKS];

    auto readA = [&](auto BUFc, auto MHc) {
        constexpr int BUF = decltype(BUFc)::value, MH = decltype(MHc)::value;
        const char* base = smem + BUF * BUF8_BYTES + (MH ? 3 : 0) * HT_BYTES;
        if constexpr (DT::FP8) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi][0] = DT::load2(base + (a_rd + mi * 2048), base + ((a_rd ^ 64) + mi * 2048));
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[mi][ks] = *reinterpret_cast<const vec8*>(base + ((a_rd ^ (ks << 6)) + mi * 2048));
        }
    };
    auto readB = [&](auto BUFc, auto NHc, frag_t (&wf)[2][NKS]) {
        constexpr int BUF = decltype(BUFc)::value, NH = decltype(NHc)::value;
        const char* base = smem + BUF * BUF8_BYTES + (1 + NH) * HT_BYTES;
        if constexpr (DT::FP8) {
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) wf[nl][0] = DT::load2(base + (w_rd + nl * 512), base + ((w_rd ^ 64) + nl * 512));
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) wf[nl][ks] = *reinterpret_cast<const vec8*>(base + ((w_rd ^ (ks << 6)) + nl * 512));
        }
    };
    auto mma = [&](auto MHc, auto NHc, const frag_t (&wf)[2][NKS]) {
        constexpr int MH = decltype(MHc)::value, NH = decltype(NHc)::value;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) {
                    if constexpr (DT::FP8) acc[MH * 4 + mi][NH * 2 + nl] = DT::mfma8(wf[nl][ks], af[mi][ks], acc[MH * 4 + mi][NH * 2 + nl]);
                    else acc[MH * 4 + mi][NH * 2 + nl] = DT::mfma(wf[nl][ks], af[mi][ks], acc[MH * 4 + mi][NH * 2 + nl]);
                }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // one phase: [fragment reads | DMA issue | counted waits] barrier [16 MFMAs] barrier.   VM = vmcnt literal of this phase.
#define HV_PHASE_SYNC(VM)                                            \
    asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_setprio(1);
#define HV_PHASE_SYNC_NOVM()                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_setprio(1);
#define HV_PHASE_END()                                                \
    __builtin_amdgcn_s_setprio(0);                                    \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);

    // KIND 0: steady (every phase issues, vmcnt(10)); 1: tile nkt-2 (only phase 1 issues; 10, 8, 6, 4); 2: tile nkt-1 (2, 0, 0, 0)
    auto tile_fn = [&](auto BUFc, auto KINDc, int t) {
        constexpr int BUF = decltype(BUFc)::value, KIND = decltype(KINDc)::value;
        using B = std::integral_constant<int, BUF>;
        if constexpr (DT::FP8) {
            // fp8 schedule: ONE set of W fragments (8 registers each), B(n0) re-read in phase 4 (the 8-register operands leave no room
            // to keep it: 28 instead of 24 fragment reads per K-tile).  The Bn0 slot then stays live until phase 4, so the DMA order
            // is Am0(t+2), Bn1(t+2), Am1(t+2) in phases 2-4 and Bn0(t+1) in phase 1 of the NEXT... i.e. phase 1 issues Bn0 of
            // tile t+1.  Everything tile t+1 needs is older than or equal to Bn0(t+1), which has exactly three younger half-tiles
            // at the end of phase 4: one counted vmcnt(6) per K-tile covers all of it.
            readB(B{}, I0{}, wf0);
            __builtin_amdgcn_sched_barrier(0);
            readA(B{}, I0{});
            if constexpr (KIND <= 1) stage(I1{}, BUF ^ 1, t + 1);
            HV_PHASE_SYNC_NOVM()
            mma(I0{}, I0{}, wf0);
            HV_PHASE_END()
            readB(B{}, I1{}, wf0);
            if constexpr (KIND == 0) stage(I0{}, BUF, t + 2);
            HV_PHASE_SYNC_NOVM()
            mma(I0{}, I1{}, wf0);
            HV_PHASE_END()
            readA(B{}, I1{});
            if constexpr (KIND == 0) stage(I2{}, BUF, t + 2);
            HV_PHASE_SYNC_NOVM()
            mma(I1{}, I1{}, wf0);
            HV_PHASE_END()
            readB(B{}, I0{}, wf0);
            if constexpr (KIND == 0) stage(I3{}, BUF, t + 2);
            if constexpr (KIND == 0) { HV_PHASE_SYNC(6) } else { HV_PHASE_SYNC(0) }
            mma(I1{}, I0{}, wf0);
            HV_PHASE_END()
            return;
        }
        // phase 1: Q(m0, n0)
        readB(B{}, I0{}, wf0);
        __builtin_amdgcn_sched_barrier(0);
        readA(B{}, I0{});
        if constexpr (KIND <= 1) stage(I3{}, BUF ^ 1, t + 1);
        if constexpr (KIND <= 1) { HV_PHASE_SYNC(10) } else { HV_PHASE_SYNC(2) }
        mma(I0{}, I0{}, wf0);
        HV_PHASE_END()
        // phase 2: Q(m0, n1)
        readB(B{}, I1{}, wf1);
        if constexpr (KIND == 0) stage(I0{}, BUF, t + 2);
        if constexpr (KIND == 0) { HV_PHASE_SYNC(10) } else if constexpr (KIND == 1) { HV_PHASE_SYNC(8) } else { HV_PHASE_SYNC(0) }
        mma(I0{}, I1{}, wf1);
        HV_PHASE_END()
        // phase 3: Q(m1, n1)
        readA(B{}, I1{});
        if constexpr (KIND == 0) stage(I1{}, BUF, t + 2);
        if constexpr (KIND == 0) { HV_PHASE_SYNC(10) } else if constexpr (KIND == 1) { HV_PHASE_SYNC(6) } else { HV_PHASE_SYNC(0) }
        mma(I1{}, I1{}, wf1);
        HV_PHASE_END()
        // phase 4: Q(m1, n0)   (B(n0) still in registers from phase 1)
        if constexpr (KIND == 0) stage(I2{}, BUF, t + 2);
        if constexpr (KIND == 0) { HV_PHASE_SYNC(10) } else if constexpr (KIND == 1) { HV_PHASE_SYNC(4) } else { HV_PHASE_SYNC(0) }
        mma(I1{}, I0{}, wf0);
        HV_PHASE_END()
    };

    // ---- prologue: half-tiles 0..6 (K-tile 0 and Am0, Bn0, Bn1 of K-tile 1); phase 0 reads half-tiles 0 and 1.
    // K-tile t lives in buffer (t + par) & 1 with par = (number of steady tiles) & 1, so that the unrolled pair loop and the two
    // tail tiles always see buffers (0, 1): an odd steady count is absorbed by ONE peeled tile in front, not by a second copy of the tail.
    const int nkt = g.K * DT::ESIZE / (BK * 2);      // 128-byte K-tiles; >= 3 (host-side dispatch)
    const int ns = nkt - 2;        // steady tiles 0 .. nkt-3
    const int par = ns & 1;
    if constexpr (DT::FP8) {     // issue order of the fp8 schedule: Am0, Bn1, Am1, Bn0 per K-tile; K-tile 0 complete = three younger half-tiles
        stage(I0{}, par, 0); stage(I2{}, par, 0); stage(I3{}, par, 0); stage(I1{}, par, 0);
        stage(I0{}, par ^ 1, 1); stage(I2{}, par ^ 1, 1); stage(I3{}, par ^ 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        stage(I0{}, par, 0); stage(I1{}, par, 0); stage(I2{}, par, 0); stage(I3{}, par, 0);
        stage(I0{}, par ^ 1, 1); stage(I1{}, par ^ 1, 1); stage(I2{}, par ^ 1, 1);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // the second wave row runs one barrier interval behind the first
    __builtin_amdgcn_sched_barrier(0);

    int t = 0;
    if (par) {
        tile_fn(I1{}, I0{}, 0);
        t = 1;
    }
    for (; t < ns; t += 2) {
        tile_fn(I0{}, I0{}, t);
        tile_fn(I1{}, I0{}, t + 1);
    }
    tile_fn(I0{}, I1{}, t);
    tile_fn(I1{}, I2{}, t + 1);
    if (wm == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count of the staggered rows
#undef HV_PHASE_SYNC
#undef HV_PHASE_SYNC_NOVM
#undef HV_PHASE_END
    // The epilogue's global reads (bias, gate, residual rows, fp8 row scales) must not be hoisted into the K loop's tail: every
    // wait there is a COUNTED vmcnt that assumes the only vector-memory operations in flight are the LDS-DMAs (an extra plain
    // load issued behind them would be among "the N youngest" and let a needed half-tile stay in flight: a race that showed up as
    // wrong N-tail tiles of the fp8 kernel).  The compiler does hoist such loads across an asm with a "memory" clobber (kernel-
    // argument pointers are invariant to it), so the pointers themselves are made opaque here, after the last MFMA phase.
    // ... and every accumulator is pinned HERE, under the full EXEC mask: the epilogue consumes acc[mi][ni] only inside the
    // divergent `column < N` / `row < M` regions, and the compiler sinks pure instructions towards their uses - it moved the last
    // K-tile's v_mfma_scale_* (an intrinsic it does not treat as convergent) into those regions, where they ran with a partial EXEC
    // mask: wrong N-tail tiles in the fp8 kernel only, plus ~200 spilled fragment registers kept alive for the sunk MFMAs.
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" : : "v"(acc[mi][ni]));     // a USE, not a redefinition: no effect on the loop's allocation
    GemmArgs ge = g;
    asm volatile("s_waitcnt vmcnt(0)" : "+s"(ge.bias), "+s"(ge.gate), "+s"(ge.res), "+s"(ge.a_scale), "+s"(ge.w_scale), "+s"(ge.out0),
                 "+s"(ge.out1), "+s"(ge.out_f32), "+s"(ge.gn_partial) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP) {
        ge.M = sp_M;
        const SpRowMap rm{g.sH, g.sW, g.cH, g.cW, g.up_t, sp_pt, (sp_c >> 1) & 1, sp_c & 1};
        gemm_epilogue<DT, 8, 4, 16, SpRowMap>(ge, acc, m0 + wm * 128 + fr, n0 + wn * 64, fq, rm, tm * 4 + wm * 2);
    } else
    gemm_epilogue<DT, 8, 4>(ge, acc, m0 + wm * 128 + fr, n0 + wn * 64, fq, IdRowMap(), tm * 4 + wm * 2);
}

template <typename DT, bool CONV = false>
int launch_gemm8(GemmArgs& g, hipStream_t stream) {
    static HvPerDeviceOnce once;
    if (hv_set_max_lds(once, (const void*)gemm8_kernel<DT, CONV>, LDS8_BYTES) != HV_OK) return HV_ERR_LAUNCH;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + 255) / 256;
    gemm8_kernel<DT, CONV><<<dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(512), LDS8_BYTES, stream>>>(g);
    return hv_check_launch();
}

int launch_gemm8_subpixel(GemmArgs& g, hipStream_t stream) {      // g.tiles_m / sp_* filled by the caller
    static HvPerDeviceOnce once;
    if (hv_set_max_lds(once, (const void*)gemm8_kernel<F16T, true, true>, LDS8_BYTES) != HV_OK) return HV_ERR_LAUNCH;
    g.tiles_n = (g.N + 255) / 256;
    gemm8_kernel<F16T, true, true><<<dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(512), LDS8_BYTES, stream>>>(g);
    return hv_check_launch();
}

// =====================================================================================================================
// Pipelined 256 x 128 tile for the VAE's 128-channel convolutions (Cout <= 128: 39 % of the tiled decode ran on the 2-stage
// 256 x 128 loop at ~0.8 PF).  Same ideas as gemm8_kernel - raw barriers, counted vmcnt, the two wave rows one barrier apart -
// re-proportioned for the narrower tile:
//   * 8 waves as 4 (M) x 2 (N), each 64 x 64 outputs = 4 x 4 accumulators (64 VGPRs; ~140 in all);
//   * a K-tile is three 16-KiB units in consumption order: Am0 (rows wm*64 + [0,32) of all four wave rows), B (all 128 W rows),
//     Am1; two phases per K-tile of 16 MFMAs each: (m0 x all n) after reading B (8) + A(m0) (4) fragments, (m1 x all n) after 4 more;
//   * THREE K-tile buffers (144 KiB): every unit is issued two K-tiles ahead - phase 1 issues Am0 and B, phase 2 issues Am1 of
//     tile t+2 into the buffer tile t-1 just left - so 4-5 units (64-80 KiB) are in flight behind vmcnt(10) / vmcnt(8);
//   * K-tile count of a 3x3x3 conv is 27*cin/64, always a multiple of 3: the loop is unrolled by three tiles (static LDS offsets).
// Arithmetic intensity is 85 flop per staged byte (256 x 256: 128), so the LDS-fill rate (L2 -> LDS, 66-73 GB/s per CU) caps this
// tile near 1.5 PF.  conv mode only (plain N <= 128 GEMMs are tiny); gather offsets per tap from packed coordinates as in gemm8.
constexpr int C128_UNIT = 16384, C128_BUF = 3 * C128_UNIT, C128_LDS = 3 * C128_BUF;   // 144 KiB

__global__ __launch_bounds__(512, 2) void conv128_kernel(GemmArgs g) {
    typedef F16T DT;
    typedef DT::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;          // waves 0-3 (wm 0,1) / 4-7 (wm 2,3) = the two waves of every SIMD

    const int nwg = gridDim.x;
    int lin;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;       // tiles_n == 1 for every 128-channel layer: consecutive M tiles per XCD
    const int m0 = tm * BM, n0 = tn * 128;

    const int srow = tid >> 3, scp = tid & 7;
    uint32_t a_o[2][2], w_o[2], tap_off[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = i * 64 + srow;                                    // LDS row of the unit
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (lr >> 5) * 64 + h * 32 + (lr & 31);           // tile row: wave row (lr>>5), its rows [h*32, h*32+32)
            const int ar = min(m0 + arow, g.M - 1);
            const int vw = (ar % g.cW) * g.mw, th = ar / g.cW;
            a_o[h][i] = (uint32_t)((th / g.cH) * g.mt) | ((uint32_t)((th % g.cH) * g.mh) << 8) | ((uint32_t)vw << 20);
        }
        w_o[i] = (uint32_t)((int64_t)(min(n0 + lr, g.N - 1) - n0) * g.ldw * 2 + ((scp ^ swz_w(lr)) << 4));
    }
    const uint32_t a_chunk = (uint32_t)((scp ^ swz_a(srow)) << 4);
    const int wave_lds = wave * 1024;
    const char* a_base = reinterpret_cast<const char*>(g.A);
    const char* w_tile = reinterpret_cast<const char*>(g.W) + (int64_t)n0 * g.ldw * 2;
    const int lg_cin = 31 - __builtin_clz((unsigned)g.cin);
    auto conv_tap = [&](int h, int tap) {
        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        const uint32_t row_bytes = (uint32_t)g.lda * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t c = a_o[h][i];
            int ti = max((int)(c & 255u) + dt - 2, 0);
            if (g.up_t) ti = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
            const int hi = min(max((int)((c >> 8) & 4095u) + dh - 1, 0), g.bH - 1) >> g.up_hw;
            const int wi = min(max((int)(c >> 20) + dw - 1, 0), g.bW - 1) >> g.up_hw;
            tap_off[h][i] = (uint32_t)((ti * g.sH + hi) * g.sW + wi) * row_bytes + a_chunk;
        }
    };
    // unit U (0 Am0, 1 B, 2 Am1) of K-tile `tile` into buffer `buf`
    auto stage = [&](auto Uc, int buf, int tile) {
        constexpr int U = decltype(Uc)::value;
        char* dst = smem + buf * C128_BUF + U * C128_UNIT + wave_lds;
        uint64_t bv;
        if constexpr (U != 1) {
            constexpr int h = U >> 1;
            const int tap = (tile * BK) >> lg_cin, c0 = tile * BK - (tap << lg_cin);
            if (c0 == 0) conv_tap(h, tap);
            bv = reinterpret_cast<uint64_t>(a_base + c0 * 2);
        } else {
            bv = reinterpret_cast<uint64_t>(w_tile + (int64_t)tile * (BK * 2));
        }
        asm volatile("" : "+s"(bv));
        const char* b = reinterpret_cast<const char*>(bv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t o = U == 1 ? w_o[i] : tap_off[U >> 1][i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b + o), (lds_ptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    const int a_rd = (wm * 32 + fr) * 128 + ((fq ^ swz_a(fr)) << 4);                         // + mi' * 2048 (mi' = 0, 1)
    const int w_lr = wn * 64 + (fr >> 2) * 8 + (fr & 3);                                     // + (ni>>1)*32 rows + (ni&1)*4 rows
    const int w_rd = w_lr * 128 + ((fq ^ swz_w(w_lr)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec8 af[2][2], wf[4][2];

    auto readA = [&](auto BUFc, auto MHc) {
        constexpr int BUF = decltype(BUFc)::value, MH = decltype(MHc)::value;
        const char* base = smem + BUF * C128_BUF + (MH ? 2 : 0) * C128_UNIT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi][ks] = *reinterpret_cast<const vec8*>(base + ((a_rd ^ (ks << 6)) + mi * 2048));
    };
    auto readB = [&](auto BUFc) {
        constexpr int BUF = decltype(BUFc)::value;
        const char* base = smem + BUF * C128_BUF + C128_UNIT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni][ks] = *reinterpret_cast<const vec8*>(base + ((w_rd ^ (ks << 6)) + (ni >> 1) * 4096 + (ni & 1) * 512));
    };
    auto mma = [&](auto MHc) {
        constexpr int MH = decltype(MHc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[MH * 2 + mi][ni] = DT::mfma(wf[ni][ks], af[mi][ks], acc[MH * 2 + mi][ni]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#define HV_C_SYNC(VM)                                                 \
    asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_setprio(1);
#define HV_C_END()                                                    \
    __builtin_amdgcn_s_setprio(0);                                    \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);
    // KIND 0 steady (issues for tile t+2; vmcnt 10 / 8), 1 = tile nkt-2 (no issue; 6 / 2), 2 = tile nkt-1 (0 / 0)
    auto tile_fn = [&](auto BUFc, auto KINDc, int t) {
        constexpr int BUF = decltype(BUFc)::value, KIND = decltype(KINDc)::value;
        constexpr int NB = (BUF + 2) % 3;                 // buffer of tile t+2 = the one tile t-1 just left
        using B = std::integral_constant<int, BUF>;
        readB(B{});
        __builtin_amdgcn_sched_barrier(0);
        readA(B{}, I0{});
        if constexpr (KIND == 0) { stage(I0{}, NB, t + 2); stage(I1{}, NB, t + 2); }
        if constexpr (KIND == 0) { HV_C_SYNC(10) } else if constexpr (KIND == 1) { HV_C_SYNC(6) } else { HV_C_SYNC(0) }
        mma(I0{});
        HV_C_END()
        readA(B{}, I1{});
        if constexpr (KIND == 0) stage(I2{}, NB, t + 2);
        if constexpr (KIND == 0) { HV_C_SYNC(8) } else if constexpr (KIND == 1) { HV_C_SYNC(2) } else { HV_C_SYNC(0) }
        mma(I1{});
        HV_C_END()
    };

    const int nkt = g.K / BK;      // a multiple of 3, >= 27 (host-side dispatch)
    stage(I0{}, 0, 0); stage(I1{}, 0, 0); stage(I2{}, 0, 0);
    stage(I0{}, 1, 1); stage(I1{}, 1, 1); stage(I2{}, 1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wm >= 2) __builtin_amdgcn_s_barrier();      // the second wave row group runs one barrier interval behind the first
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    for (; t + 3 < nkt; t += 3) {                  // tiles 0 .. nkt-4 (steady: t + 2 <= nkt - 1 for each)
        tile_fn(I0{}, I0{}, t);
        tile_fn(I1{}, I0{}, t + 1);
        tile_fn(I2{}, I0{}, t + 2);
    }
    tile_fn(I0{}, I0{}, t);                        // tile nkt-3, still steady
    tile_fn(I1{}, I1{}, t + 1);
    tile_fn(I2{}, I2{}, t + 2);
    if (wm < 2) __builtin_amdgcn_s_barrier();
#undef HV_C_SYNC
#undef HV_C_END
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" : : "v"(acc[mi][ni]));
    GemmArgs ge = g;
    asm volatile("s_waitcnt vmcnt(0)" : "+s"(ge.bias), "+s"(ge.gate), "+s"(ge.res), "+s"(ge.a_scale), "+s"(ge.w_scale), "+s"(ge.out0),
                 "+s"(ge.out1), "+s"(ge.out_f32), "+s"(ge.gn_partial) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    gemm_epilogue<DT, 4, 4>(ge, acc, m0 + wm * 64 + fr, n0 + wn * 64, fq, IdRowMap(), tm * 4 + wm);
}

int launch_conv128(GemmArgs& g, hipStream_t stream) {
    static HvPerDeviceOnce once;
    if (hv_set_max_lds(once, (const void*)conv128_kernel, C128_LDS) != HV_OK) return HV_ERR_LAUNCH;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + 127) / 128;
    conv128_kernel<<<dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(512), C128_LDS, stream>>>(g);
    return hv_check_launch();
}

// =====================================================================================================================
// 256 x 128 conv tile with W-SHIFT REUSE of the staged activations (cW divides 256, unit stride along W).
//
// conv128_kernel re-gathers the 256 activation rows for each of the 27 taps although the three taps (dt, dh, 0..2) read the
// same source rows displaced by one output position: staged per K-tile it moves 48 KiB for 4.2 MFLOP and is bound by the
// L2 -> LDS fill (52 GB/s per CU measured, 1.0-1.16 PF).  Here the K loop runs in GROUPS of three K-tiles - one (dt, dh) row of
// taps and one 64-channel chunk - that share ONE staged activation block: S[j] = the source row of output j for the centre tap
// dw = 1, and the fragment reads of tap dw take row j + dw - 1 of the same block.  A tile starts on a multiple of cW and cW
// divides 256, so j - 1 / j + 1 only leave the block where the conv clamps anyway (w = 0 / w = cW - 1): those lanes read row j
// itself (replicate padding), decided once per lane when the fragment addresses are formed.  Per group 32 KiB of activations +
// 3 x 16 KiB of weights are staged instead of 144 KiB: 153 flop per staged byte (was 85).
// The LDS array (fragment reads 256 B/clk + the DMA's writes) is as busy as the matrix pipe with 64 x 64 outputs per wave, and a
// 16-row fragment read displaced by one LDS row is a 2-way bank conflict under any XOR swizzle (no pairing of the 16 rows
// survives all three displacements), so the rows are also PERMUTED: tile row j sits at LDS row (j & 3) * 64 + (j >> 2), a
// lane's four m-repeats are the four consecutive tile rows 4*fr .. 4*fr + 3 (block b = LDS rows b*64 ..), and "row j + 1" of
// block b IS block b + 1 in the same lane's registers.  The three taps of a group then need six fragment blocks - blocks 0..3,
// "row j - 1 of block 0" (block 3 one LDS row up) and "row j + 1 of block 3" (block 0 one row down) - instead of twelve, only
// the last two displaced: 12 A-fragment reads per group instead of 24 (+ 24 of W).
// The sum over K is taken in the order (dt, dh, channel chunk, dw) instead of (tap, channel chunk): a fixed order, so results
// are run-to-run identical, but not bit-identical to conv128_kernel's (fp32 accumulation, fp16 output).
//
// LDS (128 KiB): weights in a ring of four 16-KiB K-tile slots, each issued THREE K-tiles ahead; activations double-buffered
// per group (2 x 32 KiB, rows linear), the halves of group g+1 issued in phases 1 and 5 of group g.  Two phases per K-tile,
// split by N (phase 1: the wave's 64 x 32 left half after 8 A + 4 W fragment reads, phase 2: the right half after 4 more W reads),
// so the A fragments of a tap are read once.  Counted waits (two loads per thread and 16-KiB unit):
//     issue   P1a W(t+3) | P2a A1(g+1) | P1b W(t+3) | P2b -        | P1c W(t+3) | P2c A0(g+2)
//     vmcnt       -      |     8       |     -      |  8           |     -      |  6          (before the barrier ending the phase)
// 4-5 units (64-80 KiB per CU) stay in flight across the raw barriers; the two wave groups (waves 0-3 / 4-7) run one barrier
// apart as in gemm8_kernel.  WAR: W(t+3) goes to the slot of W(t-1), last read in phase 2 of tile t-1; A0(g+2) / A1(g+1) go to
// the buffer of group g / g-1 whose last fragment read is in phase 1 of the group's third tile - each at least one barrier
// (for the other wave group: its lgkmcnt(0) + one barrier) before the issue.
constexpr int CS_SLOT = 16384, CS_A0 = 4 * CS_SLOT, CS_ABUF = 32768, CS_LDS = CS_A0 + 2 * CS_ABUF;   // 128 KiB

// NARROW (Cout <= 32: conv_out, 128 -> 3 padded to 8): only the waves that own columns 0..63 (wn = 0) read fragments and issue
// MFMAs, and only for their first 32 columns - 16 MFMAs per K-tile instead of 32 per wave and none for the other four waves; the
// staging (the bound of this variant) and the barrier protocol are unchanged.
template <bool NARROW>
__device__ __forceinline__ void conv128s_body(GemmArgs g) {
    typedef F16T DT;
    typedef DT::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = gridDim.x;
    int lin;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = lin * BM;          // tiles_n == 1

    const int srow = tid >> 3, scp = tid & 7;
    uint32_t a_o[2][2], w_o[2], tap_off[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int R = u * 128 + i * 64 + srow;                          // LDS row R holds tile row 4 * (R & 63) + (R >> 6)
            const int ar = min(m0 + (((R & 63) << 2) | (R >> 6)), g.M - 1);
            const int vw = ar % g.cW, th = ar / g.cW;
            a_o[u][i] = (uint32_t)((th / g.cH) * g.mt) | ((uint32_t)((th % g.cH) * g.mh) << 8) | ((uint32_t)vw << 20);
        }
        const int lr = i * 64 + srow;
        w_o[i] = (uint32_t)((int64_t)min(lr, g.N - 1) * g.ldw * 2 + ((scp ^ swz_w(lr)) << 4));
    }
    const uint32_t a_chunk = (uint32_t)((scp ^ swz_a(srow)) << 4);
    const int wave_lds = wave * 1024;
    const char* a_base = reinterpret_cast<const char*>(g.A);
    const char* w_base = reinterpret_cast<const char*>(g.W);
    const int lg_cpk = 31 - __builtin_clz((unsigned)g.cin) - 6;      // log2(K-tiles per tap)
    const int cpk_mask = (1 << lg_cpk) - 1;
    auto conv_row = [&](int u, int dt, int dh) {                     // gather offsets of the centre tap (dt, dh, 1)
        const uint32_t row_bytes = (uint32_t)g.lda * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t c = a_o[u][i];
            int ti = max((int)(c & 255u) + dt - 2, 0);
            if (g.up_t) ti = ti == 0 ? 0 : 1 + ((ti - 1) >> 1);
            const int hi = min(max((int)((c >> 8) & 4095u) + dh - 1, 0), g.bH - 1) >> g.up_hw;
            const int wi = (int)(c >> 20) >> g.up_hw;
            tap_off[u][i] = (uint32_t)((ti * g.sH + hi) * g.sW + wi) * row_bytes + a_chunk;
        }
    };
    auto stageA = [&](auto Uc, int grp) {                            // half U of group grp's activation block
        constexpr int U = decltype(Uc)::value;
        char* dst = smem + CS_A0 + (grp & 1) * CS_ABUF + U * 16384 + wave_lds;
        const int dtdh = grp >> lg_cpk, cc = grp & cpk_mask;
        if (cc == 0) conv_row(U, dtdh / 3, dtdh % 3);
        uint64_t bv = reinterpret_cast<uint64_t>(a_base + cc * (BK * 2));
        asm volatile("" : "+s"(bv));
        const char* b = reinterpret_cast<const char*>(bv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t o = tap_off[U][i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b + o), (lds_ptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };
    auto stageB = [&](int tile) {                                    // weights of K-tile `tile` (loop order) into slot tile & 3
        char* dst = smem + (tile & 3) * CS_SLOT + wave_lds;
        const int grp = tile / 3, dw = tile - grp * 3;
        const int kt = ((((grp >> lg_cpk) * 3 + dw) << lg_cpk) + (grp & cpk_mask));      // K-tile index in the weight layout
        uint64_t bv = reinterpret_cast<uint64_t>(w_base + (int64_t)kt * (BK * 2));
        asm volatile("" : "+s"(bv));
        const char* b = reinterpret_cast<const char*>(bv);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t o = w_o[i];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b + o), (lds_ptr_t)(dst + i * 8192), 16, 0, 0);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    // fragment addresses: this lane's tile rows are j = wm*64 + 4*fr + b, b = 0..3, at LDS rows b*64 + wm*16 + fr; row j - 1 of
    // b = 0 is block 3 one LDS row up, row j + 1 of b = 3 is block 0 one LDS row down - unless the conv clamps there (w = 0 /
    // w = cW - 1, which only b = 0 / b = 3 can be since 4 | cW): then the lane reads its own row j again
    int a_rd[6];
    {
        const int rb = wm * 16 + fr, jw = (m0 + wm * 64 + 4 * fr) % g.cW;
        auto addr = [&](int R) { return CS_A0 + R * 128 + ((fq ^ swz_a(R)) << 4); };
#pragma unroll
        for (int b = 0; b < 4; ++b) a_rd[b] = addr(b * 64 + rb);
        a_rd[4] = addr(jw == 0 ? rb : 192 + rb - 1);
        a_rd[5] = addr(jw + 3 == g.cW - 1 ? 192 + rb : rb + 1);
    }
    const int w_lr = wn * 64 + (fr >> 2) * 8 + (fr & 3);                                     // + (ni>>1)*32 rows + (ni&1)*4 rows
    const int w_rd = w_lr * 128 + ((fq ^ swz_w(w_lr)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec8 ab[4][2], ax[2], wf[2][2];       // blocks 0..3 (kept for the three taps of a group), the extra block of tap 0 / tap 2

    auto readA = [&](vec8 (&dst)[2], int a) {
        if (NARROW && wn != 0) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dst[ks] = *reinterpret_cast<const vec8*>(smem + (a ^ (ks << 6)));
    };
    auto readB = [&](auto NHc, int wslot) {
        constexpr int NH = decltype(NHc)::value;
        if constexpr (NARROW) {
            if (NH != 0 || wn != 0) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) wf[nl][ks] = *reinterpret_cast<const vec8*>(smem + ((w_rd ^ (ks << 6)) + wslot) + nl * 512);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* base = smem + ((w_rd ^ (ks << 6)) + wslot) + NH * 4096;
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) wf[nl][ks] = *reinterpret_cast<const vec8*>(base + nl * 512);
        }
    };
    auto mma = [&](auto DWc, auto NHc) {
        constexpr int DW = decltype(DWc)::value, NH = decltype(NHc)::value;
        if constexpr (NARROW) {
            if (NH != 0 || wn != 0) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int b = mi + DW - 1;
                    const vec8& a = b < 0 || b > 3 ? ax[ks] : ab[b < 0 ? 0 : b > 3 ? 3 : b][ks];
#pragma unroll
                    for (int nl = 0; nl < 2; ++nl) acc[mi][nl] = DT::mfma(wf[nl][ks], a, acc[mi][nl]);     // columns 8 fq + 4 nl + r: 0..7 for fq = 0
                }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int b = mi + DW - 1;                 // block holding row j + DW - 1 of output row j = ... + mi
                const vec8& a = b < 0 || b > 3 ? ax[ks] : ab[b < 0 ? 0 : b > 3 ? 3 : b][ks];
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) acc[mi][NH * 2 + nl] = DT::mfma(wf[nl][ks], a, acc[mi][NH * 2 + nl]);
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#define HV_S_SYNC_NOVM()                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_setprio(1);
#define HV_S_SYNC(VM)                                                 \
    asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_setprio(1);
#define HV_S_END()                                                    \
    __builtin_amdgcn_s_setprio(0);                                    \
    __builtin_amdgcn_sched_barrier(0);                                \
    __builtin_amdgcn_s_barrier();                                     \
    __builtin_amdgcn_sched_barrier(0);
    // KIND 0: steady group; 1: the last but one (no A0 of group g+2); 2: the last (nothing left to issue)
    auto tile_fn = [&](auto DWc, auto KINDc, int grp) {
        constexpr int DW = decltype(DWc)::value, KIND = decltype(KINDc)::value;
        const int t = grp * 3 + DW;
        const int wslot = (t & 3) * CS_SLOT, abuf = (grp & 1) * CS_ABUF;
        // A fragment reads of a group: tap 0 phase 1: row j-1 of block 0 + blocks 0..2 (8 reads), phase 2: block 3; tap 2 phase 1:
        // row j+1 of block 3 (into the registers tap 0's extra block left) - 12 ds_read_b128 per group instead of 24
        if constexpr (DW == 0) {
            readA(ax, a_rd[4] + abuf);
            readA(ab[0], a_rd[0] + abuf);
            readA(ab[1], a_rd[1] + abuf);
            readA(ab[2], a_rd[2] + abuf);
        }
        if constexpr (DW == 2) readA(ax, a_rd[5] + abuf);
        readB(I0{}, wslot);
        if constexpr (KIND < 2) stageB(t + 3);
        HV_S_SYNC_NOVM()
        mma(DWc, I0{});
        HV_S_END()
        if constexpr (DW == 0) readA(ab[3], a_rd[3] + abuf);
        readB(I1{}, wslot);
        if constexpr (KIND < 2 && DW == 0) stageA(I1{}, grp + 1);
        if constexpr (KIND == 0 && DW == 2) stageA(I0{}, grp + 2);
        if constexpr (KIND == 0) {
            if constexpr (DW == 2) { HV_S_SYNC(6) } else { HV_S_SYNC(8) }
        } else if constexpr (KIND == 1) {
            if constexpr (DW == 2) { HV_S_SYNC(4) } else { HV_S_SYNC(8) }
        } else {
            if constexpr (DW == 0) { HV_S_SYNC(2) } else { HV_S_SYNC(0) }
        }
        mma(DWc, I1{});
        HV_S_END()
    };

    const int ngrp = g.K / (3 * BK);      // >= 2 (host-side dispatch)
    stageA(I0{}, 0); stageB(0); stageA(I1{}, 0); stageB(1); stageB(2); stageA(I0{}, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wm >= 2) __builtin_amdgcn_s_barrier();      // the second wave group runs one barrier interval behind the first
    __builtin_amdgcn_sched_barrier(0);
    int grp = 0;
    for (; grp < ngrp - 2; ++grp) {
        tile_fn(I0{}, I0{}, grp);
        tile_fn(I1{}, I0{}, grp);
        tile_fn(I2{}, I0{}, grp);
    }
    tile_fn(I0{}, I1{}, grp);
    tile_fn(I1{}, I1{}, grp);
    tile_fn(I2{}, I1{}, grp);
    ++grp;
    tile_fn(I0{}, I2{}, grp);
    tile_fn(I1{}, I2{}, grp);
    tile_fn(I2{}, I2{}, grp);
    if (wm < 2) __builtin_amdgcn_s_barrier();
#undef HV_S_SYNC
#undef HV_S_SYNC_NOVM
#undef HV_S_END
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) asm volatile("" : : "v"(acc[mi][ni]));
    GemmArgs ge = g;
    asm volatile("s_waitcnt vmcnt(0)" : "+s"(ge.bias), "+s"(ge.gate), "+s"(ge.res), "+s"(ge.a_scale), "+s"(ge.w_scale), "+s"(ge.out0),
                 "+s"(ge.out1), "+s"(ge.out_f32), "+s"(ge.gn_partial) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    gemm_epilogue<DT, 4, 4, 1>(ge, acc, m0 + wm * 64 + 4 * fr, wn * 64, fq, IdRowMap(), lin * 4 + wm);
}

__global__ __launch_bounds__(512, 2) void conv128s_kernel(GemmArgs g) { conv128s_body<false>(g); }
__global__ __launch_bounds__(512, 2) void conv128s_narrow_kernel(GemmArgs g) { conv128s_body<true>(g); }

int launch_conv128s(GemmArgs& g, hipStream_t stream) {
    static HvPerDeviceOnce once;
    if (hv_set_max_lds(once, (const void*)conv128s_kernel, CS_LDS) != HV_OK) return HV_ERR_LAUNCH;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = 1;
    if (g.N <= 32) {
        static HvPerDeviceOnce once_n;
        if (hv_set_max_lds(once_n, (const void*)conv128s_narrow_kernel, CS_LDS) != HV_OK) return HV_ERR_LAUNCH;
        conv128s_narrow_kernel<<<dim3((unsigned)g.tiles_m), dim3(512), CS_LDS, stream>>>(g);
        return hv_check_launch();
    }
    conv128s_kernel<<<dim3((unsigned)g.tiles_m), dim3(512), CS_LDS, stream>>>(g);
    return hv_check_launch();
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
    if (g.N <= 128) {
        if constexpr (CONV && std::is_same<DT, F16T>::value) {
            // the pipelined 256 x 128 conv tile: cin a power of two >= 128 (a tap spans >= 2 K-tiles), 27*cin/64 K-tiles (a multiple of 3),
            // coordinates within the packing
            const bool pow2 = (g.cin & (g.cin - 1)) == 0;
            if (pow2 && g.cin >= 128 && (g.K / BK) % 3 == 0 && g.K / BK >= 6 && g.cT * g.mt < 256 && g.bH <= 4096 && g.bW <= 4096 &&
                !hv_gemm_force_2stage()) {
                // unit stride along W, clamp extent = output width, output rows of a tile = whole W rows: the W-shift-reuse kernel
                static const bool no_shift = getenv("HV_CONV_NOSHIFT") != nullptr;
                if (g.mw == 1 && g.bW == g.cW && g.cW <= 256 && 256 % g.cW == 0 && g.cW % 4 == 0 && !no_shift) return launch_conv128s(g, stream);
                return launch_conv128(g, stream);
            }
        }
        return launch_bn<DT, CONV, 128>(g, stream);
    }
    if constexpr (!CONV) {
        if (g.K >= 3 * BK && !hv_gemm_force_2stage()) return launch_gemm8<DT>(g, stream);     // pipelined main loop
    } else if constexpr (std::is_same<DT, F16T>::value) {
        // conv: the pipelined loop re-forms its gather offsets per tap from packed coordinates (no room for the 36 separable offsets
        // next to the deeper pipeline's fragments), which pays when a tap spans >= 4 K-tiles: cin a power of two >= 256 - the 256-
        // and 512-channel layers that make up the BN = 256 convs of the decoder; coordinates must fit the packing (t < 256, h, w < 4096)
        const bool pow2 = (g.cin & (g.cin - 1)) == 0;
        if (pow2 && g.cin >= 256 && g.cT * g.mt < 256 && g.bH <= 4096 && g.bW <= 4096 && !hv_gemm_force_2stage())
            return launch_gemm8<DT, true>(g, stream);
    }
    return launch_bn<DT, CONV, 256>(g, stream);
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

extern "C" int hv_gemm_fp8(const void* A_q, int64_t lda, const float* a_row_scale, const void* W_q, int64_t ldw, const void* w_scale_bf16,
                           const void* bias, int M, int N, int K, void* out0, int64_t ld0, int act0, int n_split, void* out1,
                           int64_t ld1, int act1, const void* gate, const void* res, int64_t ld_res, hipStream_t stream) {
    // K % 128 == 0 and K >= 384 (three 128-byte K-tiles: the pipelined loop is the only fp8 main loop); operand row strides % 16
    if (!a_row_scale || !w_scale_bf16 || K < 384 || (K % 128) || (lda & 15) || (ldw & 15)) return HV_ERR_ARG;
    GemmArgs g;
    int rc = fill_common(g, A_q, lda, W_q, ldw, bias, M, N, K, out0, ld0, act0, n_split, out1, ld1, act1, gate, res, ld_res);
    if (rc != HV_OK) return rc;
    g.a_scale = a_row_scale;
    g.w_scale = (const uint16_t*)w_scale_bf16;
    if (M == 0) return HV_OK;
    return launch_gemm8<FP8T>(g, stream);
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
                                    int64_t ld_res, float* gn_partial, int64_t gn_partial_floats, hipStream_t stream) {
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
    if (gn_partial) {
        if (gn_partial_floats < hv_gn_partial_rows(g.M) * (int64_t)Cout * 2) return HV_ERR_ARG;
        g.gn_partial = gn_partial;
    }
    return launch<F16T, true>(g, stream);
}

extern "C" int64_t hv_gn_partial_rows(int64_t M) { return 4 * ((M + BM - 1) / BM); }

extern "C" int64_t hv_subpixel_gn_partial_rows(int sT, int sH, int sW, int up_t) {
    int64_t tiles = 0;
    for (int c = 0; c < (up_t ? 8 : 4); ++c) tiles += ((int64_t)((up_t && (c >> 2)) ? sT - 1 : sT) * sH * sW + BM - 1) / BM;
    return tiles * 4;
}

extern "C" int hv_conv3d_upsampled_subpixel_f16(const void* x, int64_t ldx, const void* w_sub, const void* tap_table, int ntap,
                                               const void* bias, void* out, int64_t ldo, int sT, int sH, int sW, int Cin, int Cout,
                                               int up_t, float* gn_partial, int64_t gn_partial_floats, hipStream_t stream) {
    // (nearest upsample x2 in H, W [and causally in T: 2 sT - 1 frames] -> causal 3x3x3 conv) in its sub-pixel form: the outputs of one
    // parity class (t, h, w mod 2) see each source voxel through a fixed set of taps, so the class is a conv over the SOURCE grid with
    // 2 (pre-summed) taps per upsampled axis - 8 (or 3*2*2 = 12) taps instead of 27.  x: channels-last source [sT, sH, sW, >= Cin];
    // out: channels-last [T2, 2 sH, 2 sW, Cout]; w_sub: [classes][Cout][ntap * Cin] f16, classes = 8 (up_t; index pt*4 + ph*2 + pw) or
    // 4 (ph*2 + pw); tap_table (device, int32 [classes][ntap]): source offset of each tap of each class, (ot+8) | (oh+8)<<4 | (ow+8)<<8.
    // The caller builds weights and table (vae_ops.subpixel_weights): which taps are summed, and whether the rounding residue of a
    // sum is carried as an extra "lo" tap, is its choice - this kernel only needs offsets.
    if (!x || !w_sub || !tap_table || !out || sT <= 0 || sH <= 0 || sW <= 0 || ntap < 1 || ntap > 27 || (up_t & ~1)) return HV_ERR_ARG;
    if (Cin < 256 || (Cin & (Cin - 1)) || Cout <= 128 || (Cout & 7)) return HV_ERR_ARG;       // the pipelined conv tile's shapes
    const int T2 = up_t ? 2 * sT - 1 : sT, H2 = 2 * sH, W2 = 2 * sW, ncls = up_t ? 8 : 4;
    if ((int64_t)T2 * H2 * W2 > 0x7fffffff || sT + 1 >= 256 || sH > 4096 || sW > 4096) return HV_ERR_ARG;
    GemmArgs g;
    int rc = fill_common(g, x, ldx, w_sub, (int64_t)ntap * Cin, bias, sT * sH * sW, Cout, ntap * Cin, out, ldo, 0, 0, nullptr, 0, 0,
                         nullptr, nullptr, 0);
    if (rc != HV_OK) return rc;
    g.cT = T2; g.cH = H2; g.cW = W2; g.cin = Cin; g.up_t = up_t; g.up_hw = 1;
    g.sT = sT; g.sH = sH; g.sW = sW; g.mt = g.mh = g.mw = 1; g.bH = H2; g.bW = W2;
    if ((int64_t)sT * sH * sW * ldx * 2 >= ((int64_t)1 << 32)) return HV_ERR_ARG;
    g.sp_ntap = ntap; g.sp_tab = (const uint32_t*)tap_table;
    int tiles = 0;
    for (int c = 0; c < 8; ++c) {
        const int frames = c >= ncls ? 0 : (up_t && (c >> 2)) ? sT - 1 : sT;       // odd output frames: 1, 3, .. 2 sT - 3
        g.sp_tile0[c] = c < ncls ? tiles : 0x7fffffff;
        g.sp_rows[c] = frames * sH * sW;
        tiles += (g.sp_rows[c] + BM - 1) / BM;
    }
    g.sp_tile0[8] = tiles;
    // an empty class (sT = 1: no odd frames) must not capture tiles: give it the start of its successor, which the class search skips past
    if (tiles == 0) return HV_OK;
    g.tiles_m = tiles;
    if (gn_partial) {
        if (gn_partial_floats < (int64_t)tiles * 4 * Cout * 2) return HV_ERR_ARG;
        g.gn_partial = gn_partial;
    }
    return launch_gemm8_subpixel(g, stream);
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
