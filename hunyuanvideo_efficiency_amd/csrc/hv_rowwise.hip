// HBM-bound row-wise kernels of the DiT block: LayerNorm + adaLN modulate (K1), per-head RMSNorm +
// 3D RoPE applied in place on the fused QKV rows (K3+K4, which also removes the concat K5), the
// small-M linear (GEMV) used for every `vec`-only projection (K9/K11), and the token/latent
// reshuffles (K10 gather, unpatchify, Euler step K13).  All loads/stores are 16 B per lane.
#include "hv_common.hpp"
#include "../../include/hv_kernels.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm(no affine) * mul + add.   mode 0: mul = bf16(1 + scale[d]), add = shift[d]  (adaLN modulate)
//                                      mode 1: mul = weight[d],          add = bias[d]   (affine LN)
// One wave per row, the row lives in registers (D <= 4096, D % 8 == 0).  Two-pass mean/variance in fp32.
template <int MAXC>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                      const bf16_t* __restrict__ mul, bf16_t* __restrict__ out,
                                                      int64_t M, int D, int64_t ldx, int64_t ldo, float eps, int mode) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nchunk = D >> 3;
    const bf16_t* xr = x + row * ldx;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            u32x4 w = *reinterpret_cast<const u32x4*>(xr + ch * 8);
            unpack8(w, v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[c][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    bf16_t* orow = out + row * ldo;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float m[8], a[8], o[8];
            if (mul) {
                unpack8(*reinterpret_cast<const u32x4*>(mul + ch * 8), m);
                if (mode == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) m[j] = rbf(1.0f + m[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = 1.0f;
            }
            if (add) {
                unpack8(*reinterpret_cast<const u32x4*>(add + ch * 8), a);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * m[j] + a[j];
            *reinterpret_cast<u32x4*>(orow + ch * 8) = pack8(o);
        }
    }
}

extern "C" int hv_ln_modulate_bf16(const void* x, const void* shift_or_bias, const void* scale_or_weight, void* out,
                                   int64_t M, int D, int64_t ldx, int64_t ldo, float eps, int mode, hipStream_t stream) {
    if (!x || !out || M < 0 || D <= 0 || (D & 7) || D > 4096 || (ldx & 7) || (ldo & 7) || (mode != 0 && mode != 1))
        return HV_ERR_ARG;
    if (M == 0) return HV_OK;
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    const bf16_t *xp = (const bf16_t*)x, *ap = (const bf16_t*)shift_or_bias, *mp = (const bf16_t*)scale_or_weight;
    if (D <= 512)
        ln_mod_kernel<1><<<grid, block, 0, stream>>>(xp, ap, mp, (bf16_t*)out, M, D, ldx, ldo, eps, mode);
    else if (D <= 2048)
        ln_mod_kernel<4><<<grid, block, 0, stream>>>(xp, ap, mp, (bf16_t*)out, M, D, ldx, ldo, eps, mode);
    else if (D <= 3072)
        ln_mod_kernel<6><<<grid, block, 0, stream>>>(xp, ap, mp, (bf16_t*)out, M, D, ldx, ldo, eps, mode);
    else
        ln_mod_kernel<8><<<grid, block, 0, stream>>>(xp, ap, mp, (bf16_t*)out, M, D, ldx, ldo, eps, mode);
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// FP8-MFMA path (opt-in; BASELINE.json config 4): per-token (row) dynamic quantisation of a GEMM A operand to OCP e4m3fn,
//   s_row = max|y_row| / 448,  q = e4m3(clamp(y / s_row, +-448))  (round to nearest even),  y_row ~= q * s_row.
// The row scale is applied by the GEMM epilogue (hv_gemm_fp8) together with the per-tensor weight scale of the reference's FP8
// checkpoints (fp8_optimization.py:85-100).  pack4: four fp32 -> one dword of 4 e4m3 bytes (v_cvt_pk_fp8_f32 x2).
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float clamp448(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }

// K1 with an fp8 output: y = bf16(LN(x) * bf16(1 + scale) + shift) exactly as ln_mod_kernel mode 0 (the bf16 rounding of the
// reference contract is kept, so the ONLY new error is the e4m3 rounding), then the per-row quantisation above.
template <int MAXC>
__global__ __launch_bounds__(256) void ln_mod_fp8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                          const bf16_t* __restrict__ mul, uint8_t* __restrict__ out,
                                                          float* __restrict__ row_scale, int64_t M, int D, int64_t ldx,
                                                          int64_t ldo, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nchunk = D >> 3;
    const bf16_t* xr = x + row * ldx;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            unpack8(*reinterpret_cast<const u32x4*>(xr + ch * 8), v[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[c][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float m[8], a[8];
            if (mul) {
                unpack8(*reinterpret_cast<const u32x4*>(mul + ch * 8), m);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = rbf(1.0f + m[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = 1.0f;
            }
            if (add) unpack8(*reinterpret_cast<const u32x4*>(add + ch * 8), a);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = rbf((v[c][j] - mean) * rstd * m[j] + a[j]);
                amax = fmaxf(amax, fabsf(v[c][j]));
            }
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) row_scale[row] = sc;
    uint8_t* orow = out + row * ldo;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            u32x2 w;
            w[0] = pack4_fp8(clamp448(v[c][0] * inv), clamp448(v[c][1] * inv), clamp448(v[c][2] * inv), clamp448(v[c][3] * inv));
            w[1] = pack4_fp8(clamp448(v[c][4] * inv), clamp448(v[c][5] * inv), clamp448(v[c][6] * inv), clamp448(v[c][7] * inv));
            *reinterpret_cast<u32x2*>(orow + ch * 8) = w;
        }
    }
}

extern "C" int hv_ln_modulate_fp8(const void* x, const void* shift, const void* scale, void* out_q, float* out_row_scale, int64_t M,
                                  int D, int64_t ldx, int64_t ldq, float eps, hipStream_t stream) {
    if (!x || !out_q || !out_row_scale || M < 0 || D <= 0 || (D & 7) || D > 4096 || (ldx & 7) || (ldq & 15)) return HV_ERR_ARG;
    if (M == 0) return HV_OK;
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    const bf16_t *xp = (const bf16_t*)x, *ap = (const bf16_t*)shift, *mp = (const bf16_t*)scale;
    if (D <= 512)
        ln_mod_fp8_kernel<1><<<grid, block, 0, stream>>>(xp, ap, mp, (uint8_t*)out_q, out_row_scale, M, D, ldx, ldq, eps);
    else if (D <= 2048)
        ln_mod_fp8_kernel<4><<<grid, block, 0, stream>>>(xp, ap, mp, (uint8_t*)out_q, out_row_scale, M, D, ldx, ldq, eps);
    else if (D <= 3072)
        ln_mod_fp8_kernel<6><<<grid, block, 0, stream>>>(xp, ap, mp, (uint8_t*)out_q, out_row_scale, M, D, ldx, ldq, eps);
    else
        ln_mod_fp8_kernel<8><<<grid, block, 0, stream>>>(xp, ap, mp, (uint8_t*)out_q, out_row_scale, M, D, ldx, ldq, eps);
    return hv_check_launch();
}

// bf16 rows [M, K] (row stride ldx) -> e4m3 rows + row scales; one wave per row, two passes over the row (the second one hits L2:
// a row is at most 30 KiB).  K % 8 == 0.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out,
                                                              int64_t ldo, float* __restrict__ row_scale, int64_t M, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + row * ldx;
    const int nchunk = K >> 3;
    float amax = 0.f;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + ch * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) row_scale[row] = sc;
    uint8_t* orow = out + row * ldo;
    for (int ch = lane; ch < nchunk; ch += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + ch * 8), f);
        u32x2 w;
        w[0] = pack4_fp8(clamp448(f[0] * inv), clamp448(f[1] * inv), clamp448(f[2] * inv), clamp448(f[3] * inv));
        w[1] = pack4_fp8(clamp448(f[4] * inv), clamp448(f[5] * inv), clamp448(f[6] * inv), clamp448(f[7] * inv));
        *reinterpret_cast<u32x2*>(orow + ch * 8) = w;
    }
}

extern "C" int hv_quant_rows_fp8(const void* x, int64_t ldx, void* out_q, int64_t ldq, float* out_row_scale, int64_t M, int K,
                                 hipStream_t stream) {
    if (!x || !out_q || !out_row_scale || M < 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldq & 15)) return HV_ERR_ARG;
    if (M == 0) return HV_OK;
    quant_rows_fp8_kernel<<<dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream>>>((const bf16_t*)x, ldx, (uint8_t*)out_q, ldq,
                                                                                  out_row_scale, M, K);
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// In-place per-head RMSNorm (+ RoPE) of the q and k thirds of fused QKV rows [n_rows, ld], head_dim 128.
//   y = bf16( x_f32 * rsqrt(mean(x^2)+eps) )  -> y = bf16(y * w)     (cast BEFORE the gain, norm_layers.py:56-58)
//   rope rows (row < n_rope): out[2i]   = y[2i]*cos[2i]   - y[2i+1]*sin[2i]
//                             out[2i+1] = y[2i+1]*cos[2i+1] + y[2i]*sin[2i+1]     (fp32, one rounding)
// 16 lanes own one 128-wide head vector (8 elements = 4 RoPE pairs per lane).
// dst != nullptr: out of place with a per-head-block scatter - head hv (0 .. 2H-1 in q|k order) of row r goes to
// dst[(hv / hpb) * dst_bs + r * dst_ld + (hv % hpb) * 128]: the Ulysses send layout [peer][row][heads of that peer], so the pack copy
// that would follow is this kernel's own store.
__global__ __launch_bounds__(256) void qknorm_rope_kernel(bf16_t* __restrict__ qkv, const bf16_t* __restrict__ qw,
                                                           const bf16_t* __restrict__ kw, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, int64_t n_rows, int64_t n_rope,
                                                           int H, int64_t ld, int64_t k_off, float eps, bf16_t* __restrict__ dst,
                                                           int64_t dst_ld, int hpb, int64_t dst_bs) {
    const int64_t row = blockIdx.x;
    const int sub = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;  // 16 groups
    float wq[8], wk[8], c[8], s[8];
    unpack8(*reinterpret_cast<const u32x4*>(qw + sub * 8), wq);
    unpack8(*reinterpret_cast<const u32x4*>(kw + sub * 8), wk);
    const bool rope = row < n_rope;
    if (rope) {
        const float4* cp = reinterpret_cast<const float4*>(cosT + row * 128 + sub * 8);
        const float4* sp = reinterpret_cast<const float4*>(sinT + row * 128 + sub * 8);
        float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        c[0] = c0.x, c[1] = c0.y, c[2] = c0.z, c[3] = c0.w, c[4] = c1.x, c[5] = c1.y, c[6] = c1.z, c[7] = c1.w;
        s[0] = s0.x, s[1] = s0.y, s[2] = s0.z, s[3] = s0.w, s[4] = s1.x, s[5] = s1.y, s[6] = s1.z, s[7] = s1.w;
    }
    for (int hv = grp; hv < 2 * H; hv += 16) {
        const bool is_k = hv >= H;
        const int h = is_k ? hv - H : hv;
        bf16_t* p = qkv + row * ld + (is_k ? k_off : 0) + h * 128 + sub * 8;
        float x[8];
        unpack8(*reinterpret_cast<const u32x4*>(p), x);
        if (dst) p = dst + (int64_t)(hv / hpb) * dst_bs + row * dst_ld + (hv % hpb) * 128 + sub * 8;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
        const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = rbf(rbf(x[j] * r) * (is_k ? wk[j] : wq[j]));
        if (rope) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                o[j] = y[j] * c[j] - y[j + 1] * s[j];
                o[j + 1] = y[j + 1] * c[j + 1] + y[j] * s[j + 1];
            }
            *reinterpret_cast<u32x4*>(p) = pack8(o);
        } else {
            *reinterpret_cast<u32x4*>(p) = pack8(y);
        }
    }
}

extern "C" int hv_qknorm_rope_bf16(void* qkv, const void* q_weight, const void* k_weight, const float* cos_tab,
                                   const float* sin_tab, int64_t n_rows, int64_t n_rope, int n_heads, int head_dim,
                                   int64_t ld, int64_t k_offset, float eps, hipStream_t stream) {
    if (!qkv || !q_weight || !k_weight || head_dim != 128 || n_heads <= 0 || n_rows < 0 || n_rope < 0 ||
        n_rope > n_rows || (ld & 7) || (k_offset & 7) || (n_rope > 0 && (!cos_tab || !sin_tab)))
        return HV_ERR_ARG;
    if (n_rows == 0) return HV_OK;
    qknorm_rope_kernel<<<dim3((unsigned)n_rows), dim3(256), 0, stream>>>((bf16_t*)qkv, (const bf16_t*)q_weight,
                                                                          (const bf16_t*)k_weight, cos_tab, sin_tab, n_rows,
                                                                          n_rope, n_heads, ld, k_offset, eps, nullptr, 0, 1, 0);
    return hv_check_launch();
}

extern "C" int hv_qknorm_rope_scatter_bf16(const void* qkv, const void* q_weight, const void* k_weight, const float* cos_tab,
                                           const float* sin_tab, int64_t n_rows, int64_t n_rope, int n_heads, int head_dim,
                                           int64_t ld, int64_t k_offset, float eps, void* dst, int64_t dst_ld,
                                           int heads_per_block, int64_t dst_block_stride, hipStream_t stream) {
    if (!qkv || !dst || !q_weight || !k_weight || head_dim != 128 || n_heads <= 0 || n_rows < 0 || n_rope < 0 || n_rope > n_rows ||
        (ld & 7) || (k_offset & 7) || (dst_ld & 7) || (dst_block_stride & 7) || heads_per_block <= 0 ||
        (n_rope > 0 && (!cos_tab || !sin_tab)))
        return HV_ERR_ARG;
    if (n_rows == 0) return HV_OK;
    qknorm_rope_kernel<<<dim3((unsigned)n_rows), dim3(256), 0, stream>>>((bf16_t*)const_cast<void*>(qkv), (const bf16_t*)q_weight,
                                                                          (const bf16_t*)k_weight, cos_tab, sin_tab, n_rows, n_rope,
                                                                          n_heads, ld, k_offset, eps, (bf16_t*)dst, dst_ld,
                                                                          heads_per_block, dst_block_stride);
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Small-M linear: out[m][n] = act_out( sum_k act_in(x[m][k]) * W[n][k] + b[n] ), M <= 4.  One wave per output
// column n, W streamed once (HBM-bound).  act flags: bit0 = SiLU on the input (rounded to bf16 as the reference's
// bf16 SiLU does), bit1 = SiLU on the output (applied to the bf16-rounded sum, rounded again).
template <int MM>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                    const bf16_t* __restrict__ b, const bf16_t* __restrict__ addend,
                                                    bf16_t* __restrict__ out, int N, int K, int64_t ldx, int64_t ldo, int act) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const bf16_t* wr = W + (int64_t)n * K;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        float w[8];
        unpack8(*reinterpret_cast<const u32x4*>(wr + k0), w);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            float xv[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + m * ldx + k0), xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = xv[j];
                if (act & 1) a = rbf(silu_f(a));
                acc[m] += a * w[j];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        float r = wave_sum(acc[m]);
        if (lane == 0) {
            if (b) r += bf2f(b[n]);
            if (act & 2) r = silu_f(rbf(r));
            if (addend) r = rbf(r) + bf2f(addend[m * ldo + n]);   // vec = vec + embedder(...) in bf16 (models.py:621,631)
            out[m * ldo + n] = f2bf(r);
        }
    }
}

extern "C" int hv_linear_smallm_bf16(const void* x, const void* W, const void* bias, const void* addend, void* out, int M,
                                     int N, int K, int64_t ldx, int64_t ldo, int act, hipStream_t stream) {
    if (!x || !W || !out || M < 1 || M > 4 || N < 1 || K < 8 || (K & 7) || (ldx & 7)) return HV_ERR_ARG;
    dim3 grid((N + 3) / 4), block(256);
    const bf16_t *xp = (const bf16_t*)x, *wp = (const bf16_t*)W, *bp = (const bf16_t*)bias, *ap = (const bf16_t*)addend;
    bf16_t* op = (bf16_t*)out;
    switch (M) {
        case 1: gemv_kernel<1><<<grid, block, 0, stream>>>(xp, wp, bp, ap, op, N, K, ldx, ldo, act); break;
        case 2: gemv_kernel<2><<<grid, block, 0, stream>>>(xp, wp, bp, ap, op, N, K, ldx, ldo, act); break;
        case 3: gemv_kernel<3><<<grid, block, 0, stream>>>(xp, wp, bp, ap, op, N, K, ldx, ldo, act); break;
        default: gemv_kernel<4><<<grid, block, 0, stream>>>(xp, wp, bp, ap, op, N, K, ldx, ldo, act); break;
    }
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Latent -> patch rows: A[tok][c*4 + ph*2 + pw] = bf16(x[c][t][2h+ph][2w+pw]), tok = (t*Hp + h)*Wp + w
// (PatchEmbed's Conv3d k=s=(1,2,2) as a K=64 GEMM, embed_layers.py:40-59).  x is fp32 [C,T,H,W] (C*4 == 64).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, bf16_t* __restrict__ A, int C, int T,
                                                        int H, int W) {
    const int Hp = H >> 1, Wp = W >> 1;
    const int64_t ntok = (int64_t)T * Hp * Wp;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (tok, c) per thread
    if (idx >= ntok * C) return;
    const int c = (int)(idx % C);
    const int64_t tok = idx / C;
    const int w = (int)(tok % Wp), h = (int)((tok / Wp) % Hp), t = (int)(tok / ((int64_t)Wp * Hp));
    const float* p = x + (((int64_t)c * T + t) * H + 2 * h) * W + 2 * w;
    float2 r0 = *reinterpret_cast<const float2*>(p);
    float2 r1 = *reinterpret_cast<const float2*>(p + W);
    u32x2 o;
    o[0] = pack_bf2(r0.x, r0.y);
    o[1] = pack_bf2(r1.x, r1.y);
    *reinterpret_cast<u32x2*>(A + tok * (C * 4) + c * 4) = o;
}

extern "C" int hv_patchify_f32_bf16(const float* x, void* A, int C, int T, int H, int W, hipStream_t stream) {
    if (!x || !A || C <= 0 || T <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return HV_ERR_ARG;
    const int64_t n = (int64_t)T * (H / 2) * (W / 2) * C;
    patchify_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(x, (bf16_t*)A, C, T, H, W);
    return hv_check_launch();
}

// unpatchify (models.py:697-710): y[tok][c*4+ph*2+pw] -> out[c][t][2h+ph][2w+pw]  (bf16 -> bf16)
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ out, int C,
                                                          int T, int H, int W, int64_t ldy) {
    const int Hp = H >> 1, Wp = W >> 1;
    const int64_t ntok = (int64_t)T * Hp * Wp;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ntok * C) return;
    const int64_t tok = idx % ntok;  // consecutive threads -> consecutive w: coalesced writes
    const int c = (int)(idx / ntok);
    const int w = (int)(tok % Wp), h = (int)((tok / Wp) % Hp), t = (int)(tok / ((int64_t)Wp * Hp));
    u32x2 v = *reinterpret_cast<const u32x2*>(y + tok * ldy + c * 4);
    bf16_t* p = out + (((int64_t)c * T + t) * H + 2 * h) * W + 2 * w;
    *reinterpret_cast<uint32_t*>(p) = v[0];
    *reinterpret_cast<uint32_t*>(p + W) = v[1];
}

extern "C" int hv_unpatchify_bf16(const void* y, void* out, int C, int T, int H, int W, int64_t ldy, hipStream_t stream) {
    if (!y || !out || C <= 0 || T <= 0 || (H & 1) || (W & 1) || (ldy & 3)) return HV_ERR_ARG;
    const int64_t n = (int64_t)T * (H / 2) * (W / 2) * C;
    unpatchify_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>((const bf16_t*)y, (bf16_t*)out, C, T, H,
                                                                                  W, ldy);
    return hv_check_launch();
}

// Euler step (scheduling_flow_match_discrete.py:236-242): sample_f32 += f32(v_bf16) * dt
__global__ __launch_bounds__(256) void euler_kernel(float* __restrict__ s, const bf16_t* __restrict__ v, float dt, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 a = *reinterpret_cast<float4*>(s + i);
        u32x2 w = *reinterpret_cast<const u32x2*>(v + i);
        a.x += bf2f_lo(w[0]) * dt, a.y += bf2f_hi(w[0]) * dt, a.z += bf2f_lo(w[1]) * dt, a.w += bf2f_hi(w[1]) * dt;
        *reinterpret_cast<float4*>(s + i) = a;
    } else {
        for (int64_t j = i; j < n; ++j) s[j] += bf2f(v[j]) * dt;
    }
}

extern "C" int hv_euler_step_f32(float* sample, const void* model_out_bf16, float dt, int64_t n, hipStream_t stream) {
    if (!sample || !model_out_bf16 || n < 0) return HV_ERR_ARG;
    if (n == 0) return HV_OK;
    euler_kernel<<<dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, stream>>>(sample, (const bf16_t*)model_out_bf16, dt, n);
    return hv_check_launch();
}

// same update with an fp32 velocity (the reference upcasts model_output to fp32, :239; a caller that hands in fp32 keeps it)
__global__ __launch_bounds__(256) void euler_f32_kernel(float* __restrict__ s, const float* __restrict__ v, float dt, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 a = *reinterpret_cast<float4*>(s + i);
        const float4 b = *reinterpret_cast<const float4*>(v + i);
        a.x += b.x * dt, a.y += b.y * dt, a.z += b.z * dt, a.w += b.w * dt;
        *reinterpret_cast<float4*>(s + i) = a;
    } else {
        for (int64_t j = i; j < n; ++j) s[j] += v[j] * dt;
    }
}

extern "C" int hv_euler_step_f32_f32(float* sample, const float* model_out_f32, float dt, int64_t n, hipStream_t stream) {
    if (!sample || !model_out_f32 || n < 0) return HV_ERR_ARG;
    if (n == 0) return HV_OK;
    euler_f32_kernel<<<dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, stream>>>(sample, model_out_f32, dt, n);
    return hv_check_launch();
}

// masked mean over tokens (token_refiner.py:222-228): out[d] = sum_l x[l][d]*mask[l] / sum_l mask[l], fp32 -> bf16 out
__global__ __launch_bounds__(256) void masked_mean_kernel(const bf16_t* __restrict__ x, const int* __restrict__ mask,
                                                           bf16_t* __restrict__ out, int L, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float s = 0.f, c = 0.f;
    for (int l = 0; l < L; ++l) {
        const float m = mask ? (float)mask[l] : 1.0f;
        s += bf2f(x[(int64_t)l * D + d]) * m;
        c += m;
    }
    out[d] = f2bf(s / c);
}

extern "C" int hv_masked_mean_bf16(const void* x, const int* mask, void* out, int L, int D, hipStream_t stream) {
    if (!x || !out || L <= 0 || D <= 0) return HV_ERR_ARG;
    masked_mean_kernel<<<dim3((D + 255) / 256), dim3(256), 0, stream>>>((const bf16_t*)x, mask, (bf16_t*)out, L, D);
    return hv_check_launch();
}

// broadcast one row into n rows (token refiner: fully-masked query rows attend only key 0 -> out = v[0])
__global__ __launch_bounds__(256) void bcast_row_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int64_t n,
                                                         int D, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    dst[(i / D) * ld + (i % D)] = src[i % D];
}

extern "C" int hv_broadcast_row_bf16(const void* src, void* dst, int64_t n_rows, int D, int64_t ld, hipStream_t stream) {
    if (!src || !dst || n_rows < 0 || D <= 0) return HV_ERR_ARG;
    if (n_rows == 0) return HV_OK;
    bcast_row_kernel<<<dim3((unsigned)((n_rows * D + 255) / 256)), dim3(256), 0, stream>>>((const bf16_t*)src, (bf16_t*)dst,
                                                                                          n_rows, D, ld);
    return hv_check_launch();
}

// timestep_embedding (embed_layers.py:93-117): out[i] = cos(t*f_i), out[half+i] = sin(t*f_i), f_i = exp(-ln(P)*i/half),
// computed in fp32 and cast to the MLP's weight dtype (bf16), embed_layers.py:153-155.  t is read from device memory.
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int n_t, int dim, float max_period) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_t * half) return;
    const int b = idx / half, i = idx % half;
    const float f = expf(-logf(max_period) * (float)i / (float)half);
    const float a = t[b] * f;
    out[b * dim + i] = f2bf(cosf(a));
    out[b * dim + half + i] = f2bf(sinf(a));
}

extern "C" int hv_timestep_embedding_bf16(const float* t, void* out, int n_t, int dim, float max_period, hipStream_t stream) {
    if (!t || !out || n_t <= 0 || dim <= 0 || (dim & 1)) return HV_ERR_ARG;
    const int n = n_t * (dim / 2);
    timestep_embedding_kernel<<<dim3((n + 127) / 128), dim3(128), 0, stream>>>(t, (bf16_t*)out, n_t, dim, max_period);
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Batched strided 2-D copy (the pack / unpack steps of the Ulysses head<->token exchange, C1):
//   dst[b*dst_bs + r*dst_ld + c] = src[b*src_bs + r*src_ld + c],  c < cols (cols % 8 == 0), 16 B per lane.
__global__ __launch_bounds__(256) void copy3d_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                      int cvec, int64_t src_bs, int64_t src_ld, int64_t dst_bs, int64_t dst_ld) {
    const int b = blockIdx.z;
    const int64_t total = (int64_t)rows * cvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cvec;
        const int c = (int)(i % cvec) * 8;
        *reinterpret_cast<u32x4*>(dst + b * dst_bs + r * dst_ld + c) =
            *reinterpret_cast<const u32x4*>(src + b * src_bs + r * src_ld + c);
    }
}

extern "C" int hv_copy3d_bf16(const void* src, void* dst, int n_batch, int64_t rows, int cols, int64_t src_batch_stride,
                              int64_t src_ld, int64_t dst_batch_stride, int64_t dst_ld, hipStream_t stream) {
    if (!src || !dst || n_batch < 0 || rows < 0 || cols <= 0 || (cols & 7) || (src_ld & 7) || (dst_ld & 7) ||
        (src_batch_stride & 7) || (dst_batch_stride & 7) || rows > 0x7fffffff)
        return HV_ERR_ARG;
    if (n_batch == 0 || rows == 0) return HV_OK;
    const int64_t total = rows * (cols / 8);
    const unsigned gx = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    copy3d_kernel<<<dim3(gx, 1, (unsigned)n_batch), dim3(256), 0, stream>>>((const bf16_t*)src, (bf16_t*)dst, (int)rows, cols / 8,
                                                                            src_batch_stride, src_ld, dst_batch_stride, dst_ld);
    return hv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// K14: FP8 weight-only path (fp8_optimization.py:50-53,55-80): W_bf16 = bf16( bf16(w_e4m3fn) * scale_bf16 ).
// OCP e4m3fn is the native fp8 of gfx950.  HBM-bound: 1 byte in, 2 bytes out, 16 B stores.
__device__ __forceinline__ float e4m3fn_to_f32(uint32_t b) {
    const uint32_t s = (b & 0x80u) << 24, e = (b >> 3) & 0xFu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;                                   // subnormal: m * 2^-9
    else if (e == 15 && m == 7) v = __uint_as_float(0x7FC00000u);              // NaN (no infinities in e4m3fn)
    else v = __uint_as_float(((e + 120u) << 23) | (m << 20));                  // 2^(e-7) * (1 + m/8)
    return __uint_as_float(__float_as_uint(v) | s);
}

__global__ __launch_bounds__(256) void fp8_dequant_kernel(const uint8_t* __restrict__ w8, const bf16_t* __restrict__ scale,
                                                           bf16_t* __restrict__ out, int64_t n) {
    const float sc = bf2f(scale[0]);
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
        if (i + 8 <= n) {
            const u32x2 p = *reinterpret_cast<const u32x2*>(w8 + i);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = e4m3fn_to_f32((p[j >> 2] >> ((j & 3) * 8)) & 0xFFu) * sc;   // e4m3 -> bf16 is exact
            *reinterpret_cast<u32x4*>(out + i) = pack8(f);
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = f2bf(e4m3fn_to_f32(w8[j]) * sc);
        }
    }
}

extern "C" int hv_fp8_dequant_bf16(const void* w_e4m3fn, const void* scale_bf16, void* out_bf16, int64_t n, hipStream_t stream) {
    if (!w_e4m3fn || !scale_bf16 || !out_bf16 || n <= 0 || (n & 7)) return HV_ERR_ARG;
    const int64_t nv = n / 8;
    const unsigned gx = (unsigned)((nv + 255) / 256 > 16384 ? 16384 : (nv + 255) / 256);
    fp8_dequant_kernel<<<dim3(gx), dim3(256), 0, stream>>>((const uint8_t*)w_e4m3fn, (const bf16_t*)scale_bf16, (bf16_t*)out_bf16, n);
    return hv_check_launch();
}
