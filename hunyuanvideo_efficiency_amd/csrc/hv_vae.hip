// HBM-bound kernels of the 3D causal VAE decode (K16, K18 softmax, K19, K20) on channels-last fp16 activations
// [voxel = (t*H + h)*W + w][C].  The convolutions are hv_conv3d_causal_f16 (hv_gemm.hip).
#include "hv_common.hpp"
#include "../../include/hv_kernels.h"

namespace {

typedef _Float16 h16;
__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(h16, v); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (h16)f); }
__device__ __forceinline__ float rh(float f) { return (float)(h16)f; }
__device__ __forceinline__ void unpack8h(const u32x4& w, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = h2f((uint16_t)(w[i] & 0xFFFFu));
        f[2 * i + 1] = h2f((uint16_t)(w[i] >> 16));
    }
}
__device__ __forceinline__ u32x4 pack8h(const float (&f)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2h(f[2 * i]) | ((uint32_t)f2h(f[2 * i + 1]) << 16);
    return w;
}

// ---- GroupNorm pass 1: per-workgroup partial (sum, sumsq) per channel.  256 threads = (C/8) channel-threads x
// (2048/C) row lanes; partial[blk][c][2] fp32.
__global__ __launch_bounds__(256) void gn_partial_kernel(const uint16_t* __restrict__ x, int64_t ldx, int64_t M, int C,
                                                          float* __restrict__ partial, int rows_per_blk) {
    extern __shared__ float red[];   // [row lanes][C][2]
    const int cthreads = C >> 3;
    const int ct = threadIdx.x % cthreads, rl = threadIdx.x / cthreads, nrl = 256 / cthreads;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = min(M, r0 + rows_per_blk);
    if (rl < nrl) {
        // pointer walk with a fixed stride, four independent 16-B loads in flight per thread (the one-load-per-iteration form with a
        // 64-bit multiply per row read at 2.6 TB/s)
        const uint16_t* xp = x + (r0 + rl) * ldx + ct * 8;
        const int64_t xs = (int64_t)nrl * ldx;
        int64_t r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl, xp += 4 * xs) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(xp + u * xs);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8h(w[u], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s[j] += v[j];
                    q[j] += v[j] * v[j];
                }
            }
        }
        for (; r < r1; r += nrl, xp += xs) {
            float v[8];
            unpack8h(*reinterpret_cast<const u32x4*>(xp), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                q[j] += v[j] * v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(rl * C + ct * 8 + j) * 2] = s[j];
            red[(rl * C + ct * 8 + j) * 2 + 1] = q[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int l = 0; l < nrl; ++l) {
            a += red[(l * C + c) * 2];
            b += red[(l * C + c) * 2 + 1];
        }
        partial[((int64_t)blockIdx.x * C + c) * 2] = a;
        partial[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}

// ---- GroupNorm pass 2: fold the partials [nrow][C][2] in fp64 -> per-channel affine y = x*sc[c] + sh[c].  One block per group
// (a single block folding all groups took 90-160 us on the large activations: more than the HBM pass over the activation it
// follows); partial rows come from gn_partial_kernel (<= 1024 rows) or from a conv epilogue (one row per 64 output rows).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int64_t nrow, int C, int groups,
                                                           int64_t M, float eps, const uint16_t* __restrict__ w,
                                                           const uint16_t* __restrict__ b, float* __restrict__ affine) {
    __shared__ double ws[4], wq[4];
    const int cpg = C / groups, g = blockIdx.x;
    const float* p = partial + (int64_t)g * cpg * 2;
    double s = 0.0, q = 0.0;
    if ((cpg & 1) == 0) {
        for (int64_t r = threadIdx.x; r < nrow; r += 256) {
            const float4* row = reinterpret_cast<const float4*>(p + r * C * 2);       // cpg even, C*2 floats per row: 16-B aligned
            float fs = 0.f, fq = 0.f;
            for (int c = 0; c < cpg / 2; ++c) {
                const float4 v = row[c];
                fs += v.x + v.z;
                fq += v.y + v.w;
            }
            s += (double)fs;
            q += (double)fq;
        }
    } else {
        for (int64_t r = threadIdx.x; r < nrow; r += 256)
            for (int c = 0; c < cpg; ++c) {
                s += (double)p[(r * C + c) * 2];
                q += (double)p[(r * C + c) * 2 + 1];
            }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wq[threadIdx.x >> 6] = q; }
    __syncthreads();
    s = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    q = (wq[0] + wq[1]) + (wq[2] + wq[3]);
    const double n = (double)M * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += 256) {
        const float wc = h2f(w[c]), bc = h2f(b[c]);
        affine[2 * c] = rstd * wc;
        affine[2 * c + 1] = bc - (float)mean * rstd * wc;
    }
}

// ---- the same fold in two levels for the long partial lists a conv epilogue leaves (one row per 64 output rows: 66,560 rows for the
// 65 x 256 x 256 activations): block (g, s) folds every S-th 256-row slab of group g's columns into one fp64 pair, gn_finalize2 folds
// the S pairs in index order.  Fixed assignment and order: run-to-run identical.
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ partial, int64_t nrow, int C, int groups,
                                                       double* __restrict__ tmp) {
    __shared__ double ws[4], wq[4];
    const int cpg = C / groups, g = blockIdx.x, S = gridDim.y;
    const float* p = partial + (int64_t)g * cpg * 2;
    double s = 0.0, q = 0.0;
    for (int64_t r = (int64_t)blockIdx.y * 256 + threadIdx.x; r < nrow; r += (int64_t)S * 256) {
        const float4* row = reinterpret_cast<const float4*>(p + r * C * 2);           // cpg even (host-checked): 16-B pieces
        float fs = 0.f, fq = 0.f;
        for (int c = 0; c < cpg / 2; ++c) {
            const float4 v = row[c];
            fs += v.x + v.z;
            fq += v.y + v.w;
        }
        s += (double)fs;
        q += (double)fq;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tmp[((int64_t)g * S + blockIdx.y) * 2] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        tmp[((int64_t)g * S + blockIdx.y) * 2 + 1] = (wq[0] + wq[1]) + (wq[2] + wq[3]);
    }
}

// The same first level with COALESCED reads, for C / 2 a power of two <= 256 (every shipped width): a block folds every S-th run of
// 256 / (C/2) whole rows - a thread owns one 16-byte piece (two channels x two moments) of its row lane, consecutive lanes read a
// contiguous row - then the row lanes and the pieces of a group are summed in index order.  (gn_fold_kernel reads 8 * C/groups bytes
// per row at a stride of 8 C: 132 us for the 66,560 partial rows of the largest activation; this one 20 us.)
__global__ __launch_bounds__(256) void gn_fold_rows_kernel(const float* __restrict__ partial, int64_t nrow, int C, int groups,
                                                            double* __restrict__ tmp) {
    __shared__ double sm[256][2];
    const int cv = C >> 1, ct = threadIdx.x % cv, rl = threadIdx.x / cv, nrl = 256 / cv, S = gridDim.x;
    double s = 0.0, q = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * nrl + rl; r < nrow; r += (int64_t)S * nrl) {
        const float4 v = reinterpret_cast<const float4*>(partial + r * C * 2)[ct];
        s += (double)(v.x + v.z);
        q += (double)(v.y + v.w);
    }
    sm[threadIdx.x][0] = s, sm[threadIdx.x][1] = q;
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int g = threadIdx.x, pg = cv / groups;       // pieces per group
        double gs = 0.0, gq = 0.0;
        for (int l = 0; l < nrl; ++l)
            for (int c = g * pg; c < (g + 1) * pg; ++c) {
                gs += sm[l * cv + c][0];
                gq += sm[l * cv + c][1];
            }
        tmp[((int64_t)g * S + blockIdx.x) * 2] = gs;
        tmp[((int64_t)g * S + blockIdx.x) * 2 + 1] = gq;
    }
}

__global__ __launch_bounds__(64) void gn_finalize2_kernel(const double* __restrict__ tmp, int S, int C, int groups, int64_t M, float eps,
                                                           const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                           float* __restrict__ affine) {
    const int cpg = C / groups, g = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < S; ++i) {          // every lane folds all S pairs in the same order: no cross-lane step, same bits in every lane
        s += tmp[((int64_t)g * S + i) * 2];
        q += tmp[((int64_t)g * S + i) * 2 + 1];
    }
    const double n = (double)M * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += 64) {
        const float wc = h2f(w[c]), bc = h2f(b[c]);
        affine[2 * c] = rstd * wc;
        affine[2 * c + 1] = bc - (float)mean * rstd * wc;
    }
}

// ---- GroupNorm pass 3: y = [silu](x*sc + sh) -> fp16.  HBM-bound (read + write of the activation): a thread owns ONE 8-channel
// chunk for the whole launch - its 8 (scale, shift) pairs live in registers - and walks rows with a fixed 32-bit element stride;
// a block covers 256 / (C/8) consecutive rows per step, so every wave-instruction moves whole contiguous rows (16 B per lane).
// (The first version re-derived (row, chunk) from a flat 64-bit index per vector and fetched 16 affine scalars per vector from
// global memory: 1.4 TB/s.)
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y,
                                                        int64_t ldy, int64_t M, int C, const float* __restrict__ affine) {
    const int cvec = C >> 3;                       // chunk-threads per row (C <= 2048 -> cvec <= 256, a power of two for the VAE widths)
    const int ct = threadIdx.x % cvec, rl = threadIdx.x / cvec, nrl = 256 / cvec;
    if (rl >= nrl) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float4 a = *reinterpret_cast<const float4*>(affine + 2 * (ct * 8 + j));
        sc[j] = a.x, sh[j] = a.y, sc[j + 1] = a.z, sh[j + 1] = a.w;
    }
    // A block owns 4 * nrl CONSECUTIVE rows (16 KiB of a 128-channel activation): four independent 16-B loads per thread, one block
    // step apart, all inside one contiguous piece.  (Grid-stride over 4096 blocks with the four loads 16 MB apart ran at 4.4 TB/s;
    // torch's elementwise kernels, which walk memory this way, at 6.2 on the same traffic.)
    const int64_t r0 = (int64_t)blockIdx.x * (4 * nrl) + rl;
    const uint16_t* xp = x + r0 * ldx + ct * 8;
    uint16_t* yp = y + r0 * ldy + ct * 8;
    const int64_t xs = (int64_t)nrl * ldx, ys = (int64_t)nrl * ldy;
    u32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (r0 + u * nrl < M) w[u] = *reinterpret_cast<const u32x4*>(xp + u * xs);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (r0 + u * nrl >= M) break;
        float v[8], o[8];
        unpack8h(w[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = v[j] * sc[j] + sh[j];
            o[j] = SILU ? silu_f(t) : t;
        }
        *reinterpret_cast<u32x4*>(yp + u * ys) = pack8h(o);
    }
}

// ---- conv_out (DecoderCausal3D.conv_out, 128 -> 3; vae.py:283-294) as 27 per-tap PLANES + a gather-sum.
// A 3x3x3 conv with Cout <= 4 is HBM-bound work (10 kFLOP per 256-byte voxel), but as an implicit GEMM it stages every activation
// nine times for a 256 x 8 output tile (conv128s_narrow_kernel: 2.1 ms per 65x256x256 tile, after a 0.39 ms GroupNorm pass that
// writes what it reads).  Here the GroupNorm affine + SiLU in front of it and the channel contraction run in ONE streaming pass
//     planes[tap][voxel][c] = sum_ch  w[c][ch][tap] * silu(x[voxel][ch] * scale[ch] + shift[ch])        (fp32, c < 3)
// - every activation read once, normalised in registers exactly as gn_apply_kernel does (fp32 affine, SiLU, one rounding to fp16),
// contracted by v_mfma_f32_16x16x32_f16 with the whole weight set (28 fragments) resident in registers - and a second pass sums the
// 27 planes at the tap-shifted voxels (replicate padding in H/W, causal in T = clamped indices, the conv kernels' gather rule):
//     out[v][c] = bias[c] + sum_tap planes[tap][shift_tap(v)][c]         in the fixed order tap = 0..26, fp32, one rounding to fp16.
// Operand roles: A = weights (row i = 4 * tap' + c of a 16-row block: 4 taps x (3 channels + 1 zero row)), B = activations (column
// j = voxel of a 16-voxel group), so a lane's four accumulator registers are the channels of ONE (tap, voxel): a 12-byte store into
// a plane that is contiguous over voxels.  Traffic per voxel: 256 B read + 324 B written, then 324 B read + 16 B written.
constexpr int CO4_NB = 7;                 // 16-row blocks of (tap, channel) rows: 4 taps each, 27 taps -> 7

template <bool SILU>
__global__ __launch_bounds__(256, 2) void conv_cout4_planes_kernel(const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ affine,
                                                                 const u32x4* __restrict__ w_frag, float* __restrict__ planes, int64_t M, int nks) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    u32x4 wf[CO4_NB][4];                  // [row block][k step]: rows = lane & 15, k = 8 * (lane >> 4) .. + 7 of the step
#pragma unroll
    for (int nb = 0; nb < CO4_NB; ++nb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[nb][ks] = ks < nks ? w_frag[(nb * 4 + ks) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    // the affine in LDS as [k step][lane >> 4][scale x 8 | shift x 8]: a lane reads its 16 floats of a k step with four 16-byte reads
    // (64 registers otherwise: with them the kernel holds one wave per SIMD)
    __shared__ __attribute__((aligned(16))) float aff_s[4 * 4 * 16];
    if (affine) {
        for (int i = threadIdx.x; i < 4 * 4 * 16; i += 256) {
            const int ks = i >> 6, q = (i >> 4) & 3, j = i & 7, ch = ks * 32 + q * 8 + j;
            aff_s[i] = ks < nks ? affine[2 * ch + ((i >> 3) & 1)] : 0.f;
        }
        __syncthreads();
    }
    const int64_t ngrp = (M + 15) >> 4;
    auto load = [&](int64_t g, u32x4 (&raw)[4]) {
        const uint16_t* xp = x + min(g * 16 + fr, M - 1) * ldx + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < nks) raw[ks] = *reinterpret_cast<const u32x4*>(xp + ks * 32);
    };
    u32x4 cur[4], nxt[4];
    if (wave < ngrp) load(wave, cur);
    for (int64_t g = wave; g < ngrp; g += nwave) {
        if (g + nwave < ngrp) load(g + nwave, nxt);        // the next group's rows are in flight while this one is normalised and contracted
        f32x4 acc[CO4_NB];
#pragma unroll
        for (int nb = 0; nb < CO4_NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks >= nks) break;
            u32x4 b = cur[ks];
            if (affine) {
                float v[8], o[8], sc[8], sh[8];
                const f32x4* ap = reinterpret_cast<const f32x4*>(aff_s + (ks * 4 + fq) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[j] = ap[0][j], sc[4 + j] = ap[1][j], sh[j] = ap[2][j], sh[4 + j] = ap[3][j];
                unpack8h(b, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = v[j] * sc[j] + sh[j];
                    o[j] = SILU ? silu_f(t) : t;
                }
                b = pack8h(o);
            }
#pragma unroll
            for (int nb = 0; nb < CO4_NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[nb][ks]), __builtin_bit_cast(f16x8, b), acc[nb], 0, 0, 0);
        }
        const int64_t v = g * 16 + fr;
        if (v < M) {
#pragma unroll
            for (int nb = 0; nb < CO4_NB; ++nb) {
                const int tap = nb * 4 + fq;
                if (tap < 27) {
                    float* o = planes + ((int64_t)tap * M + v) * 3;
                    o[0] = acc[nb][0], o[1] = acc[nb][1], o[2] = acc[nb][2];
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cur[ks] = nxt[ks];
    }
}

// out[v][0..7] = fp16(bias[c] + sum_tap planes[tap][src_tap(v)][c]) for c < 3, zeros behind: one thread per voxel, 27 independent
// 12-byte loads (contiguous over the lanes of a wave wherever the shift does not clamp)
__global__ __launch_bounds__(256) void conv_cout4_gather_kernel(const float* __restrict__ planes, const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                                 int64_t ldo, int T, int H, int W, int cout) {
    const int64_t M = (int64_t)T * H * W;
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= M) return;
    const int w = (int)(v % W), h = (int)((v / W) % H), t = (int)(v / ((int64_t)W * H));
    float a[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = c < cout ? h2f(bias[c]) : 0.f;
    float p[27][3];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        const int ti = max(t + dt - 2, 0), hi = min(max(h + dh - 1, 0), H - 1), wi = min(max(w + dw - 1, 0), W - 1);
        const float* src = planes + ((int64_t)tap * M + ((int64_t)ti * H + hi) * W + wi) * 3;
        p[tap][0] = src[0], p[tap][1] = src[1], p[tap][2] = src[2];
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] += p[tap][c];
    u32x4 o = {0u, 0u, 0u, 0u};
    o[0] = (uint32_t)f2h(a[0]) | ((uint32_t)f2h(a[1]) << 16);
    o[1] = (uint32_t)f2h(a[2]);
    if (ldo >= 8) *reinterpret_cast<u32x4*>(out + v * ldo) = o;
    else {
        for (int c = 0; c < cout; ++c) out[v * ldo + c] = f2h(a[c]);
    }
}

// ---- row softmax: P[r][c] = softmax_c(scale * S[r][c]) for c < valid(r), 0 for valid(r) <= c < cols_pad; fp32 in, fp16 out.
// valid(r) = cols, or with causal_block > 0 the frame-causal mask of prepare_causal_attention_mask (unet_causal_3d_blocks.py:38-46):
// min(cols, (r / causal_block + 1) * causal_block) - a query of frame f sees the keys of frames <= f.  VEC = 4: float4 loads and
// 8-byte stores when strides and extents allow (the VAE's frames are multiples of 4 tokens).
template <int VEC>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_, uint16_t* __restrict__ P,
                                                            int64_t ldp, int rows, int cols, int cols_pad, float scale,
                                                            int causal_block) {
    __shared__ float redm[4], reds[4];
    const int r = blockIdx.x;
    int valid = cols;
    if (causal_block > 0) valid = min((int)min((int64_t)cols, ((int64_t)(r / causal_block) + 1) * causal_block), cols);
    const float* s = S + (int64_t)r * lds_;
    uint16_t* p = P + (int64_t)r * ldp;
    float m = -INFINITY;
    if constexpr (VEC == 4) {
        for (int c = threadIdx.x * 4; c < valid; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(s + c);
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    } else {
        for (int c = threadIdx.x; c < valid; c += 256) m = fmaxf(m, s[c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])) * scale;
    float sum = 0.f;
    if constexpr (VEC == 4) {
        for (int c = threadIdx.x * 4; c < valid; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(s + c);
            sum += (__expf(v.x * scale - m) + __expf(v.y * scale - m)) + (__expf(v.z * scale - m) + __expf(v.w * scale - m));
        }
    } else {
        for (int c = threadIdx.x; c < valid; c += 256) sum += __expf(s[c] * scale - m);
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (reds[0] + reds[1] + reds[2] + reds[3]);
    if constexpr (VEC == 4) {
        for (int c = threadIdx.x * 4; c < cols_pad; c += 1024) {
            uint2 o = make_uint2(0u, 0u);
            if (c < valid) {
                const float4 v = *reinterpret_cast<const float4*>(s + c);
                o.x = (uint32_t)f2h(__expf(v.x * scale - m) * inv) | ((uint32_t)f2h(__expf(v.y * scale - m) * inv) << 16);
                o.y = (uint32_t)f2h(__expf(v.z * scale - m) * inv) | ((uint32_t)f2h(__expf(v.w * scale - m) * inv) << 16);
            }
            *reinterpret_cast<uint2*>(p + c) = o;
        }
    } else {
        for (int c = threadIdx.x; c < cols_pad; c += 256) p[c] = c < valid ? f2h(__expf(s[c] * scale - m) * inv) : (uint16_t)0;
    }
}

// ---- transpose [R][C] (row stride lds) -> [C][R] (row stride ldd), 16-bit elements, 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ src, int64_t lds_, uint16_t* __restrict__ dst,
                                                           int64_t ldd, int R, int C) {
    __shared__ uint16_t tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(int64_t)(r0 + i) * lds_ + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(int64_t)(c0 + i) * ldd + r0 + tx] = tile[tx][i];
}

// ---- latent crop: z fp32 [C,T,H,W] (strides sc,st,sh,sw) region -> channels-last fp16 [t][h][w][Cpad], zero padded channels
__global__ __launch_bounds__(256) void latent_tile_kernel(const float* __restrict__ z, int64_t sc, int64_t st, int64_t sh,
                                                           int64_t sw, int C, int T, int H, int W, int Cpad,
                                                           uint16_t* __restrict__ out) {
    const int64_t n = (int64_t)T * H * W * Cpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        int64_t v = i / Cpad;
        const int w = (int)(v % W);
        v /= W;
        const int h = (int)(v % H);
        const int t = (int)(v / H);
        out[i] = c < C ? f2h(z[c * sc + t * st + h * sh + w * sw]) : (uint16_t)0;
    }
}

// ---- generic strided 4-D fp16 ops on [C,T,H,W] index space
struct View4 { int64_t s[4]; };
__device__ __forceinline__ void idx4(int64_t i, const int* dims, int (&ix)[4]) {
    ix[3] = (int)(i % dims[3]); i /= dims[3];
    ix[2] = (int)(i % dims[2]); i /= dims[2];
    ix[1] = (int)(i % dims[1]);
    ix[0] = (int)(i / dims[1]);
}

struct Dims4 { int d[4]; };

// b[idx] = f16( f16(a[idx shifted] * (1 - y/E)) + f16(b[idx] * (y/E)) ), y = idx[axis] in [0,E)   (blend_v/h/t)
__global__ __launch_bounds__(256) void blend_kernel(const uint16_t* __restrict__ a, View4 sa, uint16_t* __restrict__ b, View4 sb,
                                                     Dims4 dims, int axis, int extent) {
    const int64_t n = (int64_t)dims.d[0] * dims.d[1] * dims.d[2] * dims.d[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int ix[4];
        idx4(i, dims.d, ix);
        const int y = ix[axis];
        const float wa = (float)(1.0 - (double)y / (double)extent), wb = (float)((double)y / (double)extent);
        const int64_t oa = ix[0] * sa.s[0] + ix[1] * sa.s[1] + ix[2] * sa.s[2] + ix[3] * sa.s[3];
        const int64_t ob = ix[0] * sb.s[0] + ix[1] * sb.s[1] + ix[2] * sb.s[2] + ix[3] * sb.s[3];
        b[ob] = f2h(rh(h2f(a[oa]) * wa) + rh(h2f(b[ob]) * wb));
    }
}

__global__ __launch_bounds__(256) void copy4d_kernel(const uint16_t* __restrict__ src, View4 ss, uint16_t* __restrict__ dst, View4 sd,
                                                      Dims4 dims) {
    const int64_t n = (int64_t)dims.d[0] * dims.d[1] * dims.d[2] * dims.d[3];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int ix[4];
        idx4(i, dims.d, ix);
        dst[ix[0] * sd.s[0] + ix[1] * sd.s[1] + ix[2] * sd.s[2] + ix[3] * sd.s[3]] =
            src[ix[0] * ss.s[0] + ix[1] * ss.s[1] + ix[2] * ss.s[2] + ix[3] * ss.s[3]];
    }
}

// out_f32 = clamp(f16(f16(x/2) + 0.5), 0, 1)   (pipeline_hunyuan_video.py:1090-1092)
__global__ __launch_bounds__(256) void postprocess_kernel(const uint16_t* __restrict__ x, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = rh(rh(h2f(x[i]) * 0.5f) + 0.5f);
        out[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256); }

// ---------------------------------------------------------------------------------------------------------------------
// Temporal resampling of a channels-last activation [T, HW, C] (the fork's "t_ops": unet_causal_3d_blocks.py:657-672,
// 764-783,884-907).  mode 0: replicate-pad k-1 frames in front, then average k frames with stride s (avg_pool3d (k,1,1)/(s,1,1),
// fp32 accumulation, one fp16 rounding); mode 1: nearest-neighbour repeat of every frame `s` times (F.interpolate (s,1,1)).
__global__ __launch_bounds__(256) void temporal_resample_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                                 int64_t ldo, int t_out, int64_t hw, int c8, int mode, int k, int s) {
    const int64_t total = (int64_t)t_out * hw * c8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c8);
        const int64_t row = idx / c8, p = row % hw;
        const int t = (int)(row / hw);
        const uint16_t* src = x + p * ldx + c * 8;
        uint16_t* dst = out + row * ldo + c * 8;
        if (mode == 1) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src + (int64_t)(t / s) * hw * ldx);
        } else {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < k; ++i) {
                const int ti = max(t * s + i - (k - 1), 0);
                const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)ti * hw * ldx);
                const _Float16* h = reinterpret_cast<const _Float16*>(&v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)h[j];
            }
            uint4 o;
            _Float16* oh = reinterpret_cast<_Float16*>(&o);
            const float inv = 1.0f / (float)k;
#pragma unroll
            for (int j = 0; j < 8; ++j) oh[j] = (_Float16)(acc[j] * inv);
            *reinterpret_cast<uint4*>(dst) = o;
        }
    }
}

}  // namespace

extern "C" int hv_groupnorm_affine_f16(const void* x, int64_t ldx, int64_t M, int C, int groups, float eps, const void* weight,
                                       const void* bias, float* partial_ws, int64_t partial_ws_floats, float* affine_out,
                                       hipStream_t stream) {
    if (!x || !weight || !bias || !partial_ws || !affine_out || M <= 0 || C < 8 || (C & 7) || C > 2048 || groups <= 0 ||
        groups > 64 || (groups & (groups - 1)) || (C % groups) || (ldx & 7))
        return HV_ERR_ARG;
    const int cthreads = C / 8;
    if (cthreads > 256 || (256 % cthreads)) return HV_ERR_ARG;
    const int nrl = 256 / cthreads;
    int nblk = (int)((M + 2047) / 2048);
    if (nblk > 1024) nblk = 1024;
    if ((int64_t)nblk * C * 2 > partial_ws_floats) nblk = (int)(partial_ws_floats / (2 * C));
    if (nblk < 1) return HV_ERR_ARG;
    const int rows_per_blk = (int)((M + nblk - 1) / nblk);
    gn_partial_kernel<<<dim3(nblk), dim3(256), (size_t)nrl * C * 2 * sizeof(float), stream>>>((const uint16_t*)x, ldx, M, C,
                                                                                            partial_ws, rows_per_blk);
    gn_finalize_kernel<<<dim3(groups), dim3(256), 0, stream>>>(partial_ws, nblk, C, groups, M, eps, (const uint16_t*)weight,
                                                               (const uint16_t*)bias, affine_out);
    return hv_check_launch();
}

extern "C" int hv_groupnorm_finalize_f16(const float* partial, int64_t partial_floats, int64_t nrow, int64_t M, int C, int groups,
                                         float eps, const void* weight, const void* bias, float* affine_out, hipStream_t stream) {
    // partial: [nrow][C][2] (sum, sum of squares) rows as written by a conv epilogue (hv_conv3d_causal_f16 `gn_partial`), followed by
    // HV_GN_FOLD_WS_FLOATS floats of workspace; M = rows of the activation the statistics cover.  The epilogue credits a pair of
    // adjacent columns to the even one: a group must hold whole pairs.
    if (!partial || !weight || !bias || !affine_out || nrow <= 0 || M <= 0 || C < 8 || (C & 7) || C > 2048 || groups <= 0 ||
        groups > 64 || (groups & (groups - 1)) || (C % groups) || ((C / groups) & 1))
        return HV_ERR_ARG;
    // the fp64 fold scratch lives behind the partials in the caller's buffer: refuse a buffer that does not hold it
    if (partial_floats < ((nrow * C * 2 + 1) & ~(int64_t)1) + HV_GN_FOLD_WS_FLOATS) return HV_ERR_ARG;
    int S = (int)((nrow + 1023) / 1024);                                     // >= 4 rows per thread and slab set
    if (S > HV_GN_FOLD_WS_FLOATS / (64 * 4)) S = HV_GN_FOLD_WS_FLOATS / (64 * 4);       // groups <= 64, 2 doubles = 4 floats each
    if (S < 1) S = 1;
    double* tmp = reinterpret_cast<double*>(const_cast<float*>(partial) + ((nrow * C * 2 + 1) & ~(int64_t)1));
    const int cv = C >> 1;
    if (cv <= 256 && (256 % cv) == 0 && (cv % groups) == 0) {
        const int nrl = 256 / cv;
        int Sr = (int)((nrow + 4 * nrl - 1) / (4 * nrl));                  // >= 4 rows per thread
        if (Sr > HV_GN_FOLD_WS_FLOATS / (64 * 4)) Sr = HV_GN_FOLD_WS_FLOATS / (64 * 4);
        S = Sr < 1 ? 1 : Sr;
        gn_fold_rows_kernel<<<dim3(S), dim3(256), 0, stream>>>(partial, nrow, C, groups, tmp);
    } else {
        gn_fold_kernel<<<dim3(groups, S), dim3(256), 0, stream>>>(partial, nrow, C, groups, tmp);
    }
    gn_finalize2_kernel<<<dim3(groups), dim3(64), 0, stream>>>(tmp, S, C, groups, M, eps, (const uint16_t*)weight,
                                                               (const uint16_t*)bias, affine_out);
    return hv_check_launch();
}

extern "C" int hv_groupnorm_apply_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int C, const float* affine,
                                      int silu, hipStream_t stream) {
    if (!x || !y || !affine || M <= 0 || C < 8 || (C & 7) || C > 2048 || (ldx & 7) || (ldy & 7)) return HV_ERR_ARG;
    const int nrl = 256 / (C >> 3);                                  // rows per block step
    const int64_t nblk = (M + 4 * nrl - 1) / (4 * nrl);             // four steps per block
    if (nblk > 0x7fffffff) return HV_ERR_ARG;
    const dim3 grid((unsigned)nblk);
    if (silu)
        gn_apply_kernel<true><<<grid, dim3(256), 0, stream>>>((const uint16_t*)x, ldx, (uint16_t*)y, ldy, M, C, affine);
    else
        gn_apply_kernel<false><<<grid, dim3(256), 0, stream>>>((const uint16_t*)x, ldx, (uint16_t*)y, ldy, M, C, affine);
    return hv_check_launch();
}

extern "C" int64_t hv_conv3d_cout4_planes_floats(int64_t M) { return M > 0 ? 27 * M * 3 : 0; }

extern "C" int hv_conv3d_cout4_f16(const void* x, int64_t ldx, const float* affine, int silu, const void* w_frag, const void* bias, void* out,
                                   int64_t ldo, int T, int H, int W, int Cin, int Cout, float* planes, int64_t planes_floats, hipStream_t stream) {
    if (!x || !w_frag || !bias || !out || !planes || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin > 128 || (Cin & 31) || Cout <= 0 || Cout > 3 ||
        ldx < Cin || (ldx & 7) || ldo < Cout || (ldo >= 8 && (ldo & 7)) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)w_frag & 15))
        return HV_ERR_ARG;
    const int64_t M = (int64_t)T * H * W;
    if (planes_floats < hv_conv3d_cout4_planes_floats(M)) return HV_ERR_ARG;
    const int64_t ngrp = (M + 15) >> 4;
    const int64_t want = (ngrp + 3) / 4;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);      // 4 waves per block, each walks 16-voxel groups grid-stride
    if (silu)
        conv_cout4_planes_kernel<true><<<dim3(blocks), dim3(256), 0, stream>>>((const uint16_t*)x, ldx, affine, (const u32x4*)w_frag, planes, M, Cin / 32);
    else
        conv_cout4_planes_kernel<false><<<dim3(blocks), dim3(256), 0, stream>>>((const uint16_t*)x, ldx, affine, (const u32x4*)w_frag, planes, M, Cin / 32);
    if (int rc = hv_check_launch(); rc != HV_OK) return rc;
    const int64_t gblk = (M + 255) / 256;
    if (gblk > 0x7fffffff) return HV_ERR_ARG;
    conv_cout4_gather_kernel<<<dim3((unsigned)gblk), dim3(256), 0, stream>>>(planes, (const uint16_t*)bias, (uint16_t*)out, ldo, T, H, W, Cout);
    return hv_check_launch();
}

extern "C" int hv_softmax_rows_f32_f16(const float* S, int64_t ld_s, void* P, int64_t ld_p, int rows, int cols, int cols_pad,
                                       float scale, int causal_block, hipStream_t stream) {
    if (!S || !P || rows <= 0 || cols <= 0 || cols_pad < cols || causal_block < 0) return HV_ERR_ARG;
    const bool vec = !(ld_s & 3) && !(ld_p & 3) && !(cols & 3) && !(cols_pad & 3) && !(causal_block & 3) &&
                     !((uintptr_t)S & 15) && !((uintptr_t)P & 7);
    if (vec)
        softmax_rows_kernel<4><<<dim3(rows), dim3(256), 0, stream>>>(S, ld_s, (uint16_t*)P, ld_p, rows, cols, cols_pad, scale, causal_block);
    else
        softmax_rows_kernel<1><<<dim3(rows), dim3(256), 0, stream>>>(S, ld_s, (uint16_t*)P, ld_p, rows, cols, cols_pad, scale, causal_block);
    return hv_check_launch();
}

extern "C" int hv_transpose_16b(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C, hipStream_t stream) {
    if (!src || !dst || R <= 0 || C <= 0) return HV_ERR_ARG;
    transpose16_kernel<<<dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream>>>((const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, R, C);
    return hv_check_launch();
}

extern "C" int hv_vae_latent_tile_f16(const float* z, int64_t sc, int64_t st, int64_t sh, int64_t sw, int C, int T, int H, int W,
                                      int Cpad, void* out, hipStream_t stream) {
    if (!z || !out || C <= 0 || T <= 0 || H <= 0 || W <= 0 || Cpad < C) return HV_ERR_ARG;
    latent_tile_kernel<<<dim3(grid_for((int64_t)T * H * W * Cpad)), dim3(256), 0, stream>>>(z, sc, st, sh, sw, C, T, H, W, Cpad, (uint16_t*)out);
    return hv_check_launch();
}

extern "C" int hv_vae_blend_f16(const void* a, const int64_t* a_strides, void* b, const int64_t* b_strides, const int* dims,
                                int axis, int extent, hipStream_t stream) {
    if (!a || !b || !a_strides || !b_strides || !dims || axis < 0 || axis > 3 || extent <= 0 || dims[axis] > extent) return HV_ERR_ARG;
    View4 sa, sb;
    Dims4 d;
    for (int i = 0; i < 4; ++i) { sa.s[i] = a_strides[i]; sb.s[i] = b_strides[i]; d.d[i] = dims[i]; if (dims[i] <= 0) return HV_ERR_ARG; }
    blend_kernel<<<dim3(grid_for((int64_t)d.d[0] * d.d[1] * d.d[2] * d.d[3])), dim3(256), 0, stream>>>((const uint16_t*)a, sa, (uint16_t*)b, sb, d, axis, extent);
    return hv_check_launch();
}

extern "C" int hv_copy4d_16b(const void* src, const int64_t* src_strides, void* dst, const int64_t* dst_strides, const int* dims,
                             hipStream_t stream) {
    if (!src || !dst || !src_strides || !dst_strides || !dims) return HV_ERR_ARG;
    View4 ss, sd;
    Dims4 d;
    for (int i = 0; i < 4; ++i) { ss.s[i] = src_strides[i]; sd.s[i] = dst_strides[i]; d.d[i] = dims[i]; if (dims[i] <= 0) return HV_ERR_ARG; }
    copy4d_kernel<<<dim3(grid_for((int64_t)d.d[0] * d.d[1] * d.d[2] * d.d[3])), dim3(256), 0, stream>>>((const uint16_t*)src, ss, (uint16_t*)dst, sd, d);
    return hv_check_launch();
}

extern "C" int hv_vae_postprocess_f16_f32(const void* x, float* out, int64_t n, hipStream_t stream) {
    if (!x || !out || n <= 0) return HV_ERR_ARG;
    postprocess_kernel<<<dim3(grid_for(n)), dim3(256), 0, stream>>>((const uint16_t*)x, out, n);
    return hv_check_launch();
}

extern "C" int hv_temporal_resample_f16(const void* x, int64_t ldx, void* out, int64_t ldo, int T_in, int64_t HW, int C, int mode,
                                        int k, int s, hipStream_t stream) {
    if (!x || !out || T_in <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldo & 7) || s < 1 || (mode != 0 && mode != 1) ||
        (mode == 0 && k < 1))
        return HV_ERR_ARG;
    const int t_out = mode == 1 ? T_in * s : (T_in - 1) / s + 1;
    const int64_t total = (int64_t)t_out * HW * (C / 8);
    temporal_resample_kernel<<<dim3(grid_for(total)), dim3(256), 0, stream>>>((const uint16_t*)x, ldx, (uint16_t*)out, ldo, t_out, HW, C / 8,
                                                                             mode, k, s);
    return hv_check_launch();
}
