/* C ABI of libhv_kernels.so: the MI355X (gfx950) kernels behind the HunyuanVideo denoise / VAE-decode
 * hot path.  Plain pointers + sizes + a hipStream_t; no allocation inside, caller owns every buffer;
 * work is enqueued on `stream` and the call returns immediately.  Return: 0 = HV_OK, -1 = bad
 * argument (nothing launched), -2 = launch failure.  bf16 tensors are raw 16-bit words ("void*").
 *
 * The reference (c976237222/HunyuanVideo_efficiency) has no FFI of its own; its seams are Python call
 * sites.  Each entry point below names the reference arithmetic it replaces
 * (paths relative to /root/reference/hyvideo/).  The Python host side that mirrors the reference
 * interface (the modules/ directory of the hunyuanvideo_efficiency_amd package) binds these symbols with ctypes;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 */
#ifndef HV_KERNELS_H
#define HV_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ABI version of this header (bumped on any signature change). */
int hv_abi_version(void);

/* K1: nn.LayerNorm(elementwise_affine=False, eps) followed by modulate()
 *     (modules/models.py:161-164,182-185,235,246,338; modules/modulate_layers.py:31-49; mlp_layers.py:115-117)
 * mode 0: out = LN(x) * bf16(1 + scale[d]) + shift[d]      (shift/scale may be NULL = absent)
 * mode 1: out = LN(x) * weight[d] + bias[d]                (token_refiner.py:33-35,56-58 affine LayerNorm)
 * x/out: [M, D] bf16 with row strides ldx/ldo (elements); D % 8 == 0, D <= 4096. */
int hv_ln_modulate_bf16(const void* x, const void* shift_or_bias, const void* scale_or_weight, void* out,
                        int64_t M, int D, int64_t ldx, int64_t ldo, float eps, int mode, hipStream_t stream);

/* K3+K4(+K5): per-head RMSNorm of q and k (modules/norm_layers.py:43,56-59) and apply_rotary_emb
 * (modules/posemb_layers.py:133-137,165-171) IN PLACE on fused QKV rows:
 *   row r: q head h at qkv[r*ld + h*128], k head h at qkv[r*ld + k_offset + h*128]  (v untouched).
 * Rows [0, n_rope) are rotated with cos/sin[r][128] (fp32); rows [n_rope, n_rows) are only normalised
 * (text tokens).  Writing in place into the joint [img|txt] QKV buffer removes the torch.cat of
 * models.py:195-197,358-359. head_dim must be 128. */
int hv_qknorm_rope_bf16(void* qkv, const void* q_weight, const void* k_weight, const float* cos_tab,
                        const float* sin_tab, int64_t n_rows, int64_t n_rope, int n_heads, int head_dim,
                        int64_t ld, int64_t k_offset, float eps, hipStream_t stream);

/* The same, OUT OF PLACE with a scatter by head block (sequence parallelism, hyvideo/modules/attenion.py:169-180: the q / k chunk
 * that is normalised here is exchanged by an all-to-all over head groups next): head hv of row r (hv = 0 .. 2*n_heads-1 in the
 * q-then-k order of the source row) is stored at dst[(hv / heads_per_block) * dst_block_stride + r * dst_ld + (hv % heads_per_block)
 * * 128] - the [peer][row][heads of that peer] send layout - so no pack copy follows.  The source row is not modified. */
int hv_qknorm_rope_scatter_bf16(const void* qkv, const void* q_weight, const void* k_weight, const float* cos_tab,
                                const float* sin_tab, int64_t n_rows, int64_t n_rope, int n_heads, int head_dim, int64_t ld,
                                int64_t k_offset, float eps, void* dst, int64_t dst_ld, int heads_per_block,
                                int64_t dst_block_stride, hipStream_t stream);

/* K2/K7/K8/K10: nn.Linear on MFMA with fused epilogue.  C = A[M,K] . W[N,K]^T + bias
 *   (models.py:165,186,231,242,339-341,392-393; mlp_layers.py:53-59,117; embed_layers.py:40-59)
 * Columns [0, n_split) go to out0 (row stride ld0) with activation act0, columns [n_split, N) to
 * out1[., n - n_split] (row stride ld1) with act1 (act: 0 none, 1 GELU-tanh, 2 SiLU) - the single-stream
 * block's linear1 split + mlp_act (models.py:339-341,392).  n_split <= 0 or >= N: no split.
 * If gate != NULL: out = res + bf16(bf16(y) * gate[n])  (x + apply_gate(y, gate), models.py:231-250,393);
 * res may alias out0.  K % 64 == 0, N % 8 == 0, all strides % 8 == 0. */
int hv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int M, int N, int K,
                 void* out0, int64_t ld0, int act0, int n_split, void* out1, int64_t ld1, int act1,
                 const void* gate, const void* res, int64_t ld_res, hipStream_t stream);

/* K9/K11: small-M Linear (M <= 4): out = act_out(W . act_in(x) + b).  act bit0: SiLU on input
 * (ModulateDiT / adaLN: Linear(SiLU(vec)), modulate_layers.py:27-28), bit1: SiLU on output
 * (MLPEmbedder / TimestepEmbedder hidden layer, mlp_layers.py:72-73, embed_layers.py:140-150).
 * addend (nullable, same layout as out): out = bf16(y) + addend, the bf16 `vec = vec + embedder(...)` adds of
 * models.py:621,631 and token_refiner.py:229. */
int hv_linear_smallm_bf16(const void* x, const void* W, const void* bias, const void* addend, void* out, int M,
                          int N, int K, int64_t ldx, int64_t ldo, int act, hipStream_t stream);

/* timestep_embedding (modules/embed_layers.py:93-117,152-155): [cos(t f_i) | sin(t f_i)] in fp32 -> bf16;
 * t: n_t device floats. */
int hv_timestep_embedding_bf16(const float* t, void* out, int n_t, int dim, float max_period, hipStream_t stream);

/* K6/K6': softmax(scale * q k^T) v, bf16, head_dim 128, non-causal, over ONE contiguous key segment
 * (flash_attn_varlen_func / _flash_attn_forward of modules/attenion.py:107-120,181-207: the caller
 * issues one call per cu_seqlens segment).  q/k/v/o: token-major, head h at column h*128 of each row;
 * strides in elements (so q,k,v may point into one fused QKV buffer and o into a wider concat buffer).
 * workspace (nullable): caller-owned scratch, used in two steps:
 *   >= HV_ATTN_MIN_WORKSPACE_BYTES (256) and n_kv >= 4096, n_heads <= 62: a pre-pass leaves max_k |k|^2 per head in its first 256
 *      bytes and the kernel bounds every score of a query row by |q'| |k|_max; a wave whose rows all have that bound within 90
 *      (log2 units) of their first tile's row max runs against it as a STATIC maximum (no row max per tile, never a rescale:
 *      +4 % at S = 119,056), any other wave keeps the online running maximum - same result up to the rounding of P;
 *   >= hv_attn_workspace_bytes(n_q, n_kv, n_heads) and a launch only a few workgroup rounds deep (e.g. 3 heads per rank under
 *      Ulysses-8): the key range is split in two halves processed by separate workgroups and merged (log-sum-exp) by a second
 *      tiny kernel: shorter makespan, same result up to fp32 rounding.
 * Without a workspace the single-pass kernel with the online maximum always runs.  A workspace must not be shared by launches
 * that can run concurrently (different streams). */
#define HV_ATTN_MIN_WORKSPACE_BYTES 256
int hv_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t stride_q, int64_t stride_k,
                     int64_t stride_v, int64_t stride_o, int n_q, int n_kv, int n_heads, int head_dim,
                     float scale, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* bytes of scratch hv_attn_fwd_bf16 can use: the 256-byte bound area + the partials of a 2-way KV split (returned as int64). */
int64_t hv_attn_workspace_bytes(int n_q, int n_kv, int n_heads);
/* crc32 of the generated steady-state iteration (csrc/hv_attention_w4_loop.inc, `#define HV_W4_LOOP_SIGNATURE`) the library's attention
 * kernel was compiled from: lets a host check that the library in front of it is the product build and not a timing experiment. */
int hv_attn_w4_loop_signature(void);        /* the 32 bits of the crc, as int */

/* Ring attention (hybrid Ulysses x Ring, xfuser `ring_degree > 1`: hyvideo/inference.py:171-175, call site
 * modules/attenion.py:169-180): the queries of one rank against ONE K/V chunk.  Leaves the unnormalised partial in slot(s)
 * [slot, slot+splits) of caller-owned buffers part_o = float[n_slots][n_q][n_heads][128], part_ml = float[n_slots][n_q][n_heads][2]
 * (running max in the log2 domain, denominator).  splits = 1 | 2: 2 halves the key range over two workgroup sets (load balance,
 * see hv_attn_suggest_splits) and fills two slots. */
int hv_attn_partial_bf16(const void* q, const void* k, const void* v, int64_t stride_q, int64_t stride_k, int64_t stride_v,
                         int n_q, int n_kv, int n_heads, int head_dim, float scale, void* part_o, void* part_ml,
                         int n_slots, int slot, int splits, hipStream_t stream);

/* Online-softmax merge of all n_slots partials: o = sum_s O_s 2^(m_s-m) / sum_s l_s 2^(m_s-m), written as bf16 rows. */
int hv_attn_merge_bf16(const void* part_o, const void* part_ml, void* o, int64_t stride_o, int n_q, int n_heads, int n_slots,
                       hipStream_t stream);

/* 1 or 2: whether splitting the key range shortens the makespan of this launch shape on 256 CUs. */
int hv_attn_suggest_splits(int n_q, int n_kv, int n_heads);

/* K10 gather: fp32 latent [C,T,H,W] -> bf16 patch rows [T*(H/2)*(W/2), C*4] (embed_layers.py:40-59). */
int hv_patchify_f32_bf16(const float* x, void* A, int C, int T, int H, int W, hipStream_t stream);

/* unpatchify (models.py:697-710): y[tok][c*4+ph*2+pw] (row stride ldy) -> out[C,T,H,W] bf16. */
int hv_unpatchify_bf16(const void* y, void* out, int C, int T, int H, int W, int64_t ldy, hipStream_t stream);

/* K13: FlowMatchDiscreteScheduler.step (scheduling_flow_match_discrete.py:236-242):
 * sample_f32[i] += f32(model_out_bf16[i]) * dt. */
int hv_euler_step_f32(float* sample, const void* model_out_bf16, float dt, int64_t n, hipStream_t stream);

/* K13 with an fp32 velocity (the reference upcasts model_output, :239): sample_f32[i] += model_out_f32[i] * dt. */
int hv_euler_step_f32_f32(float* sample, const float* model_out_f32, float dt, int64_t n, hipStream_t stream);

/* token_refiner.py:222-228: out[d] = sum_l x[l][d]*mask[l] / sum_l mask[l]  (mask NULL = plain mean). */
int hv_masked_mean_bf16(const void* x, const int* mask, void* out, int L, int D, hipStream_t stream);

/* token_refiner.py:155-157: fully masked query rows see only key 0 -> their attention output is v[0]. */
int hv_broadcast_row_bf16(const void* src, void* dst, int64_t n_rows, int D, int64_t ld, hipStream_t stream);

/* C1 pack/unpack: batched strided 2-D copy dst[b][r][c] = src[b][r][c] with independent batch/row strides
 * (elements).  Re-lays tokens x heads for the Ulysses all-to-all (xfuser SeqAllToAll4D reached from
 * modules/attenion.py:169-180); the exchange itself is RCCL (torch.distributed "nccl"). cols % 8 == 0. */
int hv_copy3d_bf16(const void* src, void* dst, int n_batch, int64_t rows, int cols, int64_t src_batch_stride,
                   int64_t src_ld, int64_t dst_batch_stride, int64_t dst_ld, hipStream_t stream);

/* K14: FP8 weight-only Linear (modules/fp8_optimization.py:50-53,55-80): out = bf16(bf16(w_e4m3fn[i]) * scale_bf16),
 * the per-forward dequantisation the reference performs before F.linear; the GEMM then runs on hv_gemm_bf16.
 * OCP e4m3fn (gfx950-native).  n % 8 == 0. */
int hv_fp8_dequant_bf16(const void* w_e4m3fn, const void* scale_bf16, void* out_bf16, int64_t n, hipStream_t stream);

/* ---- FP8-MFMA path (opt-in; BASELINE.json configs[3] "fp8 weight path (CDNA4 fp8 MFMA)").  The reference's FP8 is weight-only
 * (fp8_optimization.py:50-80: e4m3fn weights + per-tensor `fp8_scale`, dequantised to bf16 every forward - that path is
 * hv_fp8_dequant_bf16 + hv_gemm_bf16).  Here the SAME stored weights feed v_mfma_scale_f32_16x16x128_f8f6f4 directly and the
 * activations are quantised per token: q = e4m3(clamp(x / s_row, +-448)), s_row = max|x_row| / 448, so
 *   y[m][n] = (sum_k qa[m][k] * qw[n][k]) * s_row[m] * fp8_scale + bias[n]   followed by hv_gemm_bf16's epilogues.
 * New error vs the reference path: the e4m3 rounding of the activations (2^-4 relative per element, averaging over K);
 * tolerance stated in tests/test_gpu_fp8_mfma.py. */

/* K1 with an fp8 output: y = bf16(LN(x) * bf16(1 + scale) + shift) (hv_ln_modulate_bf16 mode 0), then per-row quantisation:
 * out_q [M, D] e4m3 bytes (row stride ldq bytes, % 16), out_row_scale [M] fp32. */
int hv_ln_modulate_fp8(const void* x, const void* shift, const void* scale, void* out_q, float* out_row_scale, int64_t M,
                       int D, int64_t ldx, int64_t ldq, float eps, hipStream_t stream);

/* per-row quantisation of bf16 rows x[M, K] (row stride ldx elements) -> e4m3 rows + fp32 row scales (A operands that are not
 * LayerNorm outputs: attention output, GELU(MLP) hidden, the single block's [attn | mlp] concat). */
int hv_quant_rows_fp8(const void* x, int64_t ldx, void* out_q, int64_t ldq, float* out_row_scale, int64_t M, int K,
                      hipStream_t stream);

/* C = (A_q . W_q^T) * a_row_scale[m] * w_scale + bias with hv_gemm_bf16's epilogue arguments.  A_q [M, K], W_q [N, K]: e4m3fn
 * bytes (row strides lda / ldw in bytes, % 16); w_scale_bf16: ONE bf16 on the device (the layer's `fp8_scale`).
 * K % 128 == 0, K >= 384. */
int hv_gemm_fp8(const void* A_q, int64_t lda, const float* a_row_scale, const void* W_q, int64_t ldw, const void* w_scale_bf16,
                const void* bias, int M, int N, int K, void* out0, int64_t ld0, int act0, int n_split, void* out1, int64_t ld1,
                int act1, const void* gate, const void* res, int64_t ld_res, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * 3D causal VAE decode (hyvideo/vae).  Activations are fp16, CHANNELS-LAST: row = voxel (t*H + h)*W + w, C contiguous.
 * --------------------------------------------------------------------------------------------------------- */

/* fp16 Linear on MFMA (diffusers Attention to_q/to_k/to_v/to_out of the VAE mid block, unet_causal_3d_blocks.py:579-593;
 * 1x1x1 convs: post_quant_conv autoencoder_kl_causal_3d.py:115 and conv_shortcut unet_causal_3d_blocks.py:339-347).
 * out = A.W^T + bias; out_is_f32: fp32 result (attention scores); res (nullable): out = res + f16(y). */
int hv_gemm_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, int M, int N, int K,
                void* out, int64_t ldo, int out_is_f32, const void* res, int64_t ld_res, hipStream_t stream);

/* K15 (+K17): CausalConv3d 3x3x3 = replicate-pad(W 1,1; H 1,1; T 2,0) + Conv3d (unet_causal_3d_blocks.py:49-75) as an
 * implicit GEMM; with up_t/up_hw the nearest upsample of UpsampleCausal3D.forward (:154-172: first frame x(1,2,2), others
 * x(2,2,2)) that precedes its conv is folded into the gather.  x: source [sT,sH,sW,Cin] (row stride ldx);
 * w_taps: [Cout][27][Cin] fp16 (tap = (dt*3 + dh)*3 + dw); out: [T*H*W, Cout]; res (nullable): out = res + f16(y)
 * (ResnetBlockCausal3D residual :413-415).  Cin % 64 == 0, Cout % 8 == 0 (callers zero-pad 16->64 and 3->8).
 * The source is addressed with 32-bit byte offsets: sT*sH*sW*ldx*2 must be < 4 GiB (HV_ERR_ARG otherwise).
 * gn_partial (nullable, device fp32 [hv_gn_partial_rows(T*H*W)][Cout][2], gn_partial_floats = its size): the epilogue also
 * leaves the GroupNorm statistics of the tensor it stores - (sum, sum of squares) per channel over each 64-row block, every entry
 * written once, in a fixed order - so the GroupNorm that follows (ResnetBlockCausal3D.norm1/norm2 :395-411, conv_norm_out) needs no
 * pass over the activation: hv_groupnorm_finalize_f16 folds them.  HV_ERR_ARG if the buffer is too small. */
int hv_conv3d_causal_f16(const void* x, int64_t ldx, const void* w_taps, const void* bias, void* out, int64_t ldo,
                         int T, int H, int W, int Cin, int Cout, int up_t, int up_hw, const void* res,
                         int64_t ld_res, float* gn_partial, int64_t gn_partial_floats, hipStream_t stream);

/* rows of the gn_partial buffer for an output of M rows: one per 64 output rows, whole 256-row tiles. */
int64_t hv_gn_partial_rows(int64_t M);
/* ... and for hv_conv3d_upsampled_subpixel_f16, whose M tiles are formed per parity class (some blocks cover no row: zeros). */
int64_t hv_subpixel_gn_partial_rows(int sT, int sH, int sW, int up_t);

/* UpsampleCausal3D (unet_causal_3d_blocks.py:154-172: nearest x2 in H, W - and in T with the first frame kept single - then the
 * causal 3x3x3 conv) in its SUB-PIXEL form.  The outputs of one parity class (t, h, w mod 2) read every source voxel through a
 * fixed subset of the 27 taps, so each class is a conv over the SOURCE grid with 2 pre-summed taps per upsampled axis:
 *     H/W, even output 2k:   w0 x[k-1] + (w1+w2) x[k]        odd output 2k+1: (w0+w1) x[k] + w2 x[k+1]      (indices clamped = replicate pad)
 *     T,   even frame  2k:   w0 x[k-1] + (w1+w2) x[k]        odd frame 2k-1:  (w0+w1) x[k-1] + w2 x[k]      (k-1 clamped at 0 = causal pad)
 * 8 taps (12 when only H, W are upsampled) instead of 27: 3.4x (2.25x) fewer flops than hv_conv3d_causal_f16 with up_t/up_hw, and
 * algebraically the same function.  x: source [sT,sH,sW,Cin]; out: [T2, 2 sH, 2 sW, Cout], T2 = 2 sT - 1 (up_t) or sT;
 * w_sub: [classes][Cout][ntap*Cin] fp16, class = pt*4 + ph*2 + pw (8, up_t) or ph*2 + pw (4); tap_table: device int32
 * [classes][ntap], source offset of each tap as (ot+8) | (oh+8)<<4 | (ow+8)<<8.  Weights and table come from the caller
 * (vae_ops.subpixel_weights): a summed weight is rounded to fp16 once ("fast": <= 2 fp16 ulp from the 27-tap form), or its rounding
 * residue rides along as an extra tap with the same offset ("exact": products exact, as in the 27-tap form).
 * Cin a power of two >= 256, Cout > 128 and % 8 == 0 (the decoder's 256- and 512-channel upsamplers).
 * gn_partial: as for hv_conv3d_causal_f16, [hv_subpixel_gn_partial_rows(sT, sH, sW, up_t)][Cout][2]. */
int hv_conv3d_upsampled_subpixel_f16(const void* x, int64_t ldx, const void* w_sub, const void* tap_table, int ntap,
                                     const void* bias, void* out, int64_t ldo, int sT, int sH, int sW, int Cin, int Cout,
                                     int up_t, float* gn_partial, int64_t gn_partial_floats, hipStream_t stream);

/* DecoderCausal3D tail (vae.py:283-294: conv_norm_out -> SiLU -> conv_out): GroupNorm apply + SiLU + the 3x3x3 causal conv to
 * Cout <= 3 channels in two streaming passes instead of an HBM pass and an implicit GEMM that stages every activation nine times:
 *   planes[tap][voxel][c] = sum_ch w[c][ch][tap] * act(x[voxel][ch] * affine[ch][0] + affine[ch][1])      fp32, every voxel read once
 *   out[voxel][c] = fp16(bias[c] + sum_{tap = 0..26} planes[tap][clamped tap-shifted voxel][c])           padding rule of hv_conv3d_causal_f16
 * x: channels-last fp16 [T*H*W][ldx >= Cin]; affine (nullable: x is used as it is): fp32 [Cin][2] (scale, shift) as
 * hv_groupnorm_finalize_f16 / hv_groupnorm_affine_f16 leave it, applied in fp32 and rounded to fp16 once - the arithmetic of
 * hv_groupnorm_apply_f16; silu: 0 | 1.  w_frag: the weights in MFMA fragment order, fp16 [7][4][64][8]: row block nb, k step ks, lane l,
 * element j = w[c][ks*32 + 8*(l>>4) + j][tap] with 4*tap + c = 16*nb + (l & 15), zero for c == 3, tap >= 27 and channels >= Cin
 * (vae_ops.cout4_weight_fragments builds it).  out: [T*H*W][ldo]: ldo >= 8: columns Cout..7 are written as zeros (one 16-byte store).
 * planes: workspace of hv_conv3d_cout4_planes_floats(T*H*W) floats.  Cin % 32 == 0, Cin <= 128, Cout <= 3.  The sum over the taps is
 * taken in a fixed order: results are run-to-run identical (not bit-identical to hv_conv3d_causal_f16: another summation order). */
int hv_conv3d_cout4_f16(const void* x, int64_t ldx, const float* affine, int silu, const void* w_frag, const void* bias, void* out,
                        int64_t ldo, int T, int H, int W, int Cin, int Cout, float* planes, int64_t planes_floats, hipStream_t stream);
int64_t hv_conv3d_cout4_planes_floats(int64_t M);

/* DownsampleCausal3D (VAE encoder, unet_causal_3d_blocks.py:185-247): the same padding as hv_conv3d_causal_f16, then the 3x3x3
 * conv with stride 1|2 per axis (the fork's t_ops `downsample_stride` override, :737-742, changes these strides).
 * x: source [sT,sH,sW,Cin]; out: [T*H*W, Cout] with T = (sT-1)/stride_t + 1, H = (sH-1)/stride_h + 1, W likewise. */
int hv_conv3d_causal_strided_f16(const void* x, int64_t ldx, const void* w_taps, const void* bias, void* out, int64_t ldo,
                                 int sT, int sH, int sW, int Cin, int Cout, int stride_t, int stride_h, int stride_w,
                                 hipStream_t stream);

/* The fork's temporal ops on a channels-last activation x[T_in][HW][C] (unet_causal_3d_blocks.py:657-672,764-783,884-907).
 * mode 0: F.pad(replicate, k-1 frames in front) + F.avg_pool3d((k,1,1), stride (s,1,1)) -> T_out = (T_in-1)/s + 1;
 * mode 1: F.interpolate(scale_factor=(s,1,1), mode="nearest") -> T_out = T_in*s.  C % 8 == 0. */
int hv_temporal_resample_f16(const void* x, int64_t ldx, void* out, int64_t ldo, int T_in, int64_t HW, int C, int mode,
                             int k, int s, hipStream_t stream);

/* K16 pass 1+2: GroupNorm statistics (nn.GroupNorm(32, C, eps 1e-6, affine), unet_causal_3d_blocks.py:302,323) folded
 * into a per-channel affine: affine_out[2c] = rstd_g*w[c], affine_out[2c+1] = b[c] - mean_g*rstd_g*w[c].
 * partial_ws: caller workspace of partial_ws_floats floats (>= 2*C; 2*C*1024 for full parallelism). */
int hv_groupnorm_affine_f16(const void* x, int64_t ldx, int64_t M, int C, int groups, float eps, const void* weight,
                            const void* bias, float* partial_ws, int64_t partial_ws_floats, float* affine_out,
                            hipStream_t stream);

/* K16 pass 2 alone: the same affine from statistics a conv epilogue already took (hv_conv3d_causal_f16 `gn_partial`):
 * partial [nrow][C][2] fp32 followed by HV_GN_FOLD_WS_FLOATS floats of fold workspace IN THE SAME BUFFER (the fp64 fold scratch
 * starts at the next even float index behind the partials), M = rows of the activation they cover.  partial_floats = size of that
 * buffer in floats: HV_ERR_ARG unless it is >= align2(nrow*C*2) + HV_GN_FOLD_WS_FLOATS (a buffer sized for the conv alone is too
 * small and is refused, never overrun).  Folded in fp64 in a fixed order.  The epilogue credits each pair of adjacent channels
 * to the even one, so C / groups must be even (HV_ERR_ARG otherwise). */
#define HV_GN_FOLD_WS_FLOATS 16384
int hv_groupnorm_finalize_f16(const float* partial, int64_t partial_floats, int64_t nrow, int64_t M, int C, int groups, float eps,
                              const void* weight, const void* bias, float* affine_out, hipStream_t stream);

/* K16 pass 3: y = [SiLU](x*affine[2c] + affine[2c+1]) -> fp16 (norm + nonlinearity, unet_causal_3d_blocks.py:361-363,399-405). */
int hv_groupnorm_apply_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int C, const float* affine,
                           int silu, hipStream_t stream);

/* K18: P = softmax(scale * S) row-wise, fp32 -> fp16, columns [valid, cols_pad) zero-filled (K padding of the P.V GEMM).
 * causal_block = 0: valid = cols for every row.  causal_block = HW > 0: the frame-causal mask of prepare_causal_attention_mask
 * (unet_causal_3d_blocks.py:38-46) - row r belongs to frame r / HW and sees the keys of frames <= its own: valid =
 * min(cols, (r / HW + 1) * HW) - so one launch covers all frames of a tile. */
int hv_softmax_rows_f32_f16(const float* S, int64_t ld_s, void* P, int64_t ld_p, int rows, int cols, int cols_pad,
                            float scale, int causal_block, hipStream_t stream);

/* [R][C] -> [C][R] for 16-bit elements (V^T for the P.V GEMM). */
int hv_transpose_16b(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int R, int C, hipStream_t stream);

/* latent crop: fp32 z[C,T,H,W] region (element strides) -> channels-last fp16 [T,H,W,Cpad] (zero-padded channels):
 * the tile slicing of autoencoder_kl_causal_3d.py:443,520 + the fp16 cast of autocast. */
int hv_vae_latent_tile_f16(const float* z, int64_t sc, int64_t st, int64_t sh, int64_t sw, int C, int T, int H, int W,
                           int Cpad, void* out, hipStream_t stream);

/* K19: blend_v / blend_h / blend_t (autoencoder_kl_causal_3d.py:344-360) on strided [C,T,H,W] fp16 views:
 * b[i] = f16(f16(a[i]*(1 - y/extent)) + f16(b[i]*(y/extent))), y = index along `axis`; dims[axis] <= extent. */
int hv_vae_blend_f16(const void* a, const int64_t* a_strides, void* b, const int64_t* b_strides, const int* dims,
                     int axis, int extent, hipStream_t stream);

/* crop + concat of tiles (autoencoder_kl_causal_3d.py:463-465,531-537): strided 4-D copy of 16-bit elements. */
int hv_copy4d_16b(const void* src, const int64_t* src_strides, void* dst, const int64_t* dst_strides, const int* dims,
                  hipStream_t stream);

/* K20 tail: (image / 2 + 0.5).clamp(0, 1) in fp16 then .float() (pipeline_hunyuan_video.py:1090-1092). */
int hv_vae_postprocess_f16_f32(const void* x, float* out, int64_t n, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HV_KERNELS_H */
