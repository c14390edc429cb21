"""GPU parity at PRODUCTION shape (720x1280x129f, d = 3072, 24 heads, S = 119,056; VAE tile 65x256x256 at the shipped widths)
by SAMPLED rows / voxels against the fp32 oracle: the toy-size tests in test_gpu_ops.py / test_gpu_vae.py cannot exercise
32-bit offset arithmetic, the partial last tile at n_kv = 118,811, 1-D grids of thousands of workgroups, XCD remaps with
grids that are not multiples of 8, or the KV split at the real Ulysses-8 shape.  Inputs are hash-generated on the GPU
(synthetic.hashed_uniform), the sampled rows are recomputed on the CPU with oracle/ arithmetic in the same bf16 / fp16
contract, tolerances are the per-op ones of tests/test_gpu_ops.py (1-2 output ulps)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402
from oracle import vae_ref as VR  # noqa: E402

E = R.Prec(True)
DEV = "cuda"
BF16, F16 = torch.bfloat16, torch.float16
S_IMG, S_TXT, N_VALID, D, H = 118800, 256, 11, 3072, 24
S, CU1 = S_IMG + S_TXT, S_IMG + N_VALID


@pytest.fixture(scope="module")
def ops():
    from hunyuanvideo_efficiency_amd import ops as _ops, _lib
    _lib.torch_ops()
    return _ops


def _u(shape, key, scale=1.0, dtype=BF16):
    return (syn.hashed_uniform(shape, key, 11, DEV) * (scale * math.sqrt(3.0))).to(dtype)


def _sample_rows(n, count, key):
    """about `count` distinct row indices: both ends, the tile borders next to them, and hashed ones in between."""
    fixed = [0, 1, 255, 256, n - 257, n - 256, n - 65, n - 64, n - 2, n - 1]
    h = (syn.hashed_uniform((count,), key, 3) * 0.5 + 0.5).clamp(0, 0.999999)
    return torch.tensor(sorted(set(fixed + [int(v * n) for v in h.tolist()])))


def _attn_rows_ref(q_rows, k, v):
    """oracle attention (bf16-emulated contract: P rounded to bf16 before P.V, unrounded row sum) for a few query rows of one
    head over ALL keys.  q_rows [r,128], k/v [n_kv,128] float."""
    return R.sdpa(q_rows[None, :, None, :], k[None, :, None, :], v[None, :, None, :], E)[0, :, 0]


def test_attention_production_shape_sampled_rows(ops):
    """hv_attn_fwd_bf16 as the single-stream block launches it: q|k|v strided out of the fused [S, 9216] rows, n_q = n_kv =
    118,811 (img + 11 valid text: the last KV tile holds 27 keys), 24 heads, output into the [S, 15360] concat buffer.
    64 query rows x heads {0, 11, 23} against CPU softmax over all 118,811 keys.  Queries are scaled so that the softmax is
    peaked (outputs O(1)); with uniform random logits every output would be ~1e-2 and any tolerance vacuous."""
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    qkv[:, :D] = _u((S, D), "fs.q", 4.0)
    qkv[:, D:2 * D] = _u((S, D), "fs.k", 1.0)
    qkv[:, 2 * D:] = _u((S, D), "fs.v", 1.0)
    cat = torch.zeros(S, 5 * D, dtype=BF16, device=DEV)
    ops.attn_fwd(qkv[:CU1, :D], qkv[:CU1, D:2 * D], qkv[:CU1, 2 * D:], cat[:CU1, :D], H)
    ops.attn_fwd(qkv[CU1:, :D], qkv[CU1:, D:2 * D], qkv[CU1:, 2 * D:], cat[CU1:, :D], H)     # pad-text segment (245 rows)
    torch.cuda.synchronize()
    assert float(cat[:, D:].abs().max()) == 0, "attention wrote outside its column range of the concat buffer"
    rows = _sample_rows(CU1, 64, "fs.rows")
    for h in (0, 11, 23):
        c = slice(h * 128, (h + 1) * 128)
        k = qkv[:CU1, D + h * 128:D + (h + 1) * 128].float().cpu()
        v = qkv[:CU1, 2 * D + h * 128:2 * D + (h + 1) * 128].float().cpu()
        ref = _attn_rows_ref(qkv[rows.to(DEV), c].float().cpu(), k, v)
        got = cat[rows.to(DEV), c].float().cpu()
        assert float(ref.abs().max()) > 0.3                      # the comparison is not vacuous
        torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=8e-3)
    # the whole chip busy, every CU on its 44th workgroup in a row: the launch must reproduce itself bit for bit (a hazard between an
    # inline-asm instruction and the v_exp_f32 / MFMA it reads from would show up as run-to-run noise in the row sums)
    first = cat[:CU1, :D].clone()
    ops.attn_fwd(qkv[:CU1, :D], qkv[:CU1, D:2 * D], qkv[:CU1, 2 * D:], cat[:CU1, :D], H)
    torch.cuda.synchronize()
    assert torch.equal(cat[:CU1, :D], first)
    del first
    # second segment: rows [CU1, S) attend among themselves only
    r2 = torch.tensor([0, 100, S - CU1 - 1])
    k = qkv[CU1:, D:D + 128].float().cpu()
    v = qkv[CU1:, 2 * D:2 * D + 128].float().cpu()
    ref = _attn_rows_ref(qkv[CU1:, :128][r2.to(DEV)].float().cpu(), k, v)
    torch.testing.assert_close(cat[CU1:, :128][r2.to(DEV)].float().cpu(), ref, rtol=2 ** -7, atol=8e-3)


def test_attention_ulysses8_shape_kv_split_vs_single_pass(ops):
    """The per-rank launch of BASELINE.json config 3: all 118,811 tokens x 24/8 = 3 heads, where the C ABI takes the 2-way KV
    split + log-sum-exp merge (hv_attn_suggest_splits == 2).  Split == single pass to fp32 merge rounding, and both == oracle on
    sampled rows."""
    from hunyuanvideo_efficiency_amd import _lib
    hp, n = 3, CU1
    assert _lib.host("attn_suggest_splits", n, n, hp) == 2
    w = hp * 128
    q, k, v = _u((n, w), "u8.q", 4.0), _u((n, w), "u8.k"), _u((n, w), "u8.v")
    o_split = torch.empty(n, w, dtype=BF16, device=DEV)
    o_single = torch.empty(n, w, dtype=BF16, device=DEV)
    ops.attn_fwd(q, k, v, o_split, hp, kv_split_workspace=True)
    ops.attn_fwd(q, k, v, o_single, hp, kv_split_workspace=False)
    torch.cuda.synchronize()
    diff = (o_split.float() - o_single.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * max(1.0, float(o_single.float().abs().max())), float(diff.max())   # <= 2 bf16 ulps
    # the halves defer their running max independently, so P is rounded to bf16 at different scales than in the single pass:
    # a third of the outputs may land on the neighbouring bf16 value, none further than 2 ulps (above)
    assert float((diff > 0).float().mean()) < 0.5
    rows = _sample_rows(n, 48, "u8.rows")
    for h in range(hp):
        c = slice(h * 128, (h + 1) * 128)
        ref = _attn_rows_ref(q[rows.to(DEV), c].float().cpu(), k[:, c].float().cpu(), v[:, c].float().cpu())
        torch.testing.assert_close(o_split[rows.to(DEV), c].float().cpu(), ref, rtol=2 ** -7, atol=8e-3)


def test_gemm_production_shapes_sampled_rows(ops):
    """hv_gemm_bf16 as the single-stream block launches it (models.py:339-341,392-393):
    linear1: [119056, 3072] x [21504, 3072]^T, columns [0, 9216) -> qkv, [9216, 21504) -> GELU-tanh -> cat[:, 3072:]
    linear2: [119056, 15360] x [3072, 15360]^T + gate * y + residual, in place on x.  256 sampled rows each."""
    x = _u((S, D), "gp.x", 1.0)
    w1, b1 = _u((7 * D, D), "gp.w1", 1 / math.sqrt(D)), _u((7 * D,), "gp.b1", 0.1)
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    cat = torch.zeros(S, 5 * D, dtype=BF16, device=DEV)
    ops.gemm(x, w1, b1, out=qkv, n_split=3 * D, out1=cat[:, D:], act1=ops.ACT_GELU_TANH)
    torch.cuda.synchronize()
    rows = _sample_rows(S, 256, "gp.rows")
    rd = rows.to(DEV)
    y = E.r(x[rd].float().cpu() @ w1.float().cpu().T + b1.float().cpu())
    torch.testing.assert_close(qkv[rd].float().cpu(), y[:, :3 * D], rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(cat[rd, D:].float().cpu(), R.gelu_tanh(y[:, 3 * D:], E), rtol=2 ** -7, atol=2e-2)
    assert float(cat[:, :D].abs().max()) == 0
    # linear2 (K = 15360) with gate and residual, in place on the residual stream
    cat[:, :D] = _u((S, D), "gp.attn", 1.0)
    w2, b2 = _u((D, 5 * D), "gp.w2", 1 / math.sqrt(5 * D)), _u((D,), "gp.b2", 0.1)
    gate = _u((D,), "gp.gate", 0.5)
    a_rows, x_rows = cat[rd].float().cpu(), x[rd].float().cpu()
    ops.gemm(cat, w2, b2, out=x, gate=gate, res=x)
    torch.cuda.synchronize()
    y2 = E.r(a_rows @ w2.float().cpu().T + b2.float().cpu())
    ref = R.gate_residual(x_rows[None], y2[None], gate.float().cpu()[None], E)[0]
    torch.testing.assert_close(x[rd].float().cpu(), ref, rtol=2 ** -7, atol=2e-2)


def _conv_ref_voxels(x_cl, sT, sH, sW, w_taps, bias, vox, T, Hh, W, up_t, up_hw, res=None):
    """CausalConv3d (+ preceding nearest upsample) at sampled output voxels with the ORACLE's own index semantics: an id
    volume is pushed through oracle.vae_ref.upsample_causal and the replicate / causal padding of causal_conv3d, so which
    source voxel every (output voxel, tap) reads is decided by oracle code, not by a restatement of the kernel's arithmetic."""
    ids = torch.arange(sT * sH * sW, dtype=torch.float32).reshape(1, 1, sT, sH, sW)
    if up_t or up_hw:
        ids = VR.upsample_causal(ids, (2 if up_t else 1, 2 if up_hw else 1, 2 if up_hw else 1))
    assert tuple(ids.shape[2:]) == (T, Hh, W)
    idp = torch.nn.functional.pad(ids, (1, 1, 1, 1, 2, 0), mode="replicate")[0, 0].long()     # [T+2, H+2, W+2]
    cout, _, cin = w_taps.shape
    wt = w_taps.float().cpu()
    out = torch.empty(len(vox), cout)
    t, h, w = vox[:, 0], vox[:, 1], vox[:, 2]
    acc = torch.zeros(len(vox), cout, dtype=torch.float64)
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                src = idp[t + dt, h + dh, w + dw]
                a = x_cl[src.to(x_cl.device), :cin].float().cpu().double()
                acc += a @ wt[:, (dt * 3 + dh) * 3 + dw].double().T
    out = acc.float() + bias.float().cpu()
    out = out.to(F16).float()
    if res is not None:
        lin = (t * Hh + h) * W + w
        out = (res[lin.to(res.device)].float().cpu() + out).to(F16).float()
    return out


def _sample_voxels(T, Hh, W, count, key):
    edge = [(t, h, w) for t in (0, 1, T - 1) for h in (0, 1, Hh - 1) for w in (0, 1, W - 1)]       # every border and corner
    u = (syn.hashed_uniform((count, 3), key, 5) * 0.5 + 0.5).clamp(0, 0.999999)
    rnd = [(int(a * T), int(b * Hh), int(c * W)) for a, b, c in u.tolist()]
    return torch.tensor(sorted(set(edge + rnd)))


@pytest.mark.parametrize("name,cin,cout,T,Hh,W,up", [
    ("128ch_full_res", 128, 128, 65, 256, 256, False),        # up_blocks.3 resnets: the largest activation (1.09 GB), BN = 128 tiles
    ("512ch_fused_upsample", 512, 512, 33, 128, 128, True),   # up_blocks.1 upsampler: (2,2,2) nearest upsample folded into the gather
    ("256_to_128", 256, 128, 65, 256, 256, False),            # up_blocks.3 resnets.0 conv1
])
def test_conv3d_production_shapes_sampled_voxels(name, cin, cout, T, Hh, W, up):
    from hunyuanvideo_efficiency_amd import vae_ops as V
    sT, sH, sW = ((T + 1) // 2, Hh // 2, W // 2) if up else (T, Hh, W)
    x = _u((sT * sH * sW, cin), "cv.x." + name, 1.0, F16)
    wt = _u((cout, 27, cin), "cv.w." + name, 1 / math.sqrt(27 * cin), F16)
    b = _u((cout,), "cv.b." + name, 0.1, F16)
    res = _u((T * Hh * W, cout), "cv.r." + name, 1.0, F16) if not up else None
    out = V.conv3d_causal(x, wt, b, T, Hh, W, cin, cout, up_t=up, up_hw=up, res=res)
    again = V.conv3d_causal(x, wt, b, T, Hh, W, cin, cout, up_t=up, up_hw=up, res=res)
    torch.cuda.synchronize()
    assert torch.equal(out, again)          # counted-vmcnt pipelines: a too-late wait shows up as run-to-run differences at this size
    del again
    vox = _sample_voxels(T, Hh, W, 1000, "cv.v." + name)
    ref = _conv_ref_voxels(x, sT, sH, sW, wt, b, vox, T, Hh, W, up, up, res)
    lin = ((vox[:, 0] * Hh + vox[:, 1]) * W + vox[:, 2]).to(DEV)
    # fp16 output of an fp32-accumulated K = 27*Cin dot product: 1 fp16 ulp at |y| <= 4 is 2^-9 ~ 2e-3 (+ the residual's rounding)
    torch.testing.assert_close(out[lin].float().cpu(), ref, rtol=2 ** -9, atol=4e-3)


def test_decoder_tail_planes_production_shape_sampled_voxels():
    """hv_conv3d_cout4_f16 (conv_norm_out + SiLU + conv_out as one streaming pass + a gather-sum) at the largest tile, 65 x 256 x 256 x
    128 -> 3 (4.26 M voxels: 1.38 GB of fp32 planes, 64-bit plane offsets): sampled output voxels incl. every border against the
    direct definition with the oracle's index semantics (fp64 sum over the 27 taps of the fp16-rounded normalised activations -
    hv_groupnorm_apply_f16's output, checked on its own in test_gpu_vae.py); run-to-run identical."""
    from hunyuanvideo_efficiency_amd import vae_ops as V
    T, Hh, W, c, co = 65, 256, 256, 128, 3
    x = (_u((T * Hh * W, c), "ct.x", 2.0, F16).float() + 0.3).to(F16)
    gw, gb = (1 + _u((c,), "ct.gw", 0.1, F16).float()).to(F16), _u((c,), "ct.gb", 0.1, F16)
    w5 = _u((co, c, 3, 3, 3), "ct.w", 1 / math.sqrt(27 * c), F16)
    b8 = torch.zeros(8, dtype=F16, device=DEV)
    b8[:co] = _u((co,), "ct.b", 0.1, F16)
    aff = V.groupnorm_affine(x, gw, gb)
    wf = V.cout4_weight_fragments(w5)
    out = V.conv_cout4(x, aff, True, wf, b8, T, Hh, W, c, co)
    again = V.conv_cout4(x, aff, True, wf, b8, T, Hh, W, c, co)
    torch.cuda.synchronize()
    assert torch.equal(out, again) and float(out[:, co:].abs().max()) == 0.0
    del again
    h = V.groupnorm_apply(x, aff, True)
    vox = _sample_voxels(T, Hh, W, 1000, "ct.v")
    wt = w5.permute(0, 2, 3, 4, 1).reshape(co, 27, c).contiguous()
    ref = _conv_ref_voxels(h, T, Hh, W, wt, b8[:co], vox, T, Hh, W, False, False)
    lin = ((vox[:, 0] * Hh + vox[:, 1]) * W + vox[:, 2]).to(DEV)
    torch.testing.assert_close(out[lin, :co].float().cpu(), ref, rtol=2 ** -9, atol=2e-3)


@pytest.mark.parametrize("name,c,sT,sH,sW,up_t,mode", [
    ("512ch_t_hw", 512, 17, 64, 64, True, "fast"),          # up_blocks.1 upsampler -> 33 x 128 x 128 (8 parity classes, 8 taps)
    ("256ch_hw", 256, 65, 128, 128, False, "fast"),         # up_blocks.2 upsampler -> 65 x 256 x 256 (4 classes, 12 taps): the largest
    ("512ch_t_hw_exact", 512, 9, 32, 32, True, "exact"),    # residue taps: 15 per class
])
def test_subpixel_upsampler_production_shapes_sampled_voxels(name, c, sT, sH, sW, up_t, mode):
    """hv_conv3d_upsampled_subpixel_f16 at the shipped upsampler shapes against the DIRECT definition (nearest upsample, then the
    27-tap causal conv, fp64 on sampled output voxels incl. every border): the class -> voxel scatter, the per-class tap offsets and
    the 32-bit offset arithmetic at 4.26 M output rows.  `fast` carries one extra fp16 rounding per summed weight (<= 2 ulp)."""
    from hunyuanvideo_efficiency_amd import vae_ops as V
    T, Hh, W = (2 * sT - 1 if up_t else sT), 2 * sH, 2 * sW
    x = _u((sT * sH * sW, c), "sp.x." + name, 1.0, F16)
    w5 = _u((c, c, 3, 3, 3), "sp.w." + name, 1 / math.sqrt(27 * c), F16)
    b = _u((c,), "sp.b." + name, 0.1, F16)
    w_sub, table, ntap = V.subpixel_weights(w5, up_t, mode)
    out, st = V.conv3d_upsampled_subpixel(x, w_sub, table, ntap, b, sT, sH, sW, c, c, up_t, gn_stats=True)
    torch.cuda.synchronize()
    assert out.shape == (T * Hh * W, c)
    vox = _sample_voxels(T, Hh, W, 600, "sp.v." + name)
    wt = w5.permute(0, 2, 3, 4, 1).reshape(c, 27, c).contiguous()
    ref = _conv_ref_voxels(x, sT, sH, sW, wt, b, vox, T, Hh, W, up_t, True)
    lin = ((vox[:, 0] * Hh + vox[:, 1]) * W + vox[:, 2]).to(DEV)
    torch.testing.assert_close(out[lin].float().cpu(), ref, rtol=2 ** -9, atol=4e-3 if mode == "exact" else 6e-3)
    # the epilogue's GroupNorm statistics of this tensor == a separate pass over it
    gw, gb = (1 + _u((c,), "sp.gw", 0.1, F16).float()).to(F16), _u((c,), "sp.gb", 0.1, F16)
    torch.testing.assert_close(V.groupnorm_affine_from_stats(st, gw, gb), V.groupnorm_affine(out, gw, gb), rtol=5e-5, atol=5e-6)


def test_conv_epilogue_statistics_largest_activation():
    """GroupNorm statistics from the epilogue of the 128 -> 128 conv at 65 x 256 x 256 (66,560 partial rows, two-level fp64 fold)
    against the separate statistics pass over the stored tensor."""
    from hunyuanvideo_efficiency_amd import vae_ops as V
    T, Hh, W, c = 65, 256, 256, 128
    x = _u((T * Hh * W, c), "es.x", 1.0, F16)
    wt = _u((c, 27, c), "es.w", 1 / math.sqrt(27 * c), F16)
    b = _u((c,), "es.b", 0.1, F16)
    res = _u((T * Hh * W, c), "es.r", 1.0, F16)
    out, st = V.conv3d_causal(x, wt, b, T, Hh, W, c, c, res=res, gn_stats=True)
    assert st.rows == 4 * ((T * Hh * W + 255) // 256)
    gw, gb = (1 + _u((c,), "es.gw", 0.1, F16).float()).to(F16), _u((c,), "es.gb", 0.1, F16)
    torch.testing.assert_close(V.groupnorm_affine_from_stats(st, gw, gb), V.groupnorm_affine(out, gw, gb), rtol=5e-5, atol=5e-6)


def test_groupnorm_production_shape():
    """GroupNorm(32) + SiLU over the largest activation (65x256x256 x 128 channels): statistics over 17 M elements per group."""
    from hunyuanvideo_efficiency_amd import vae_ops as V
    M, C = 65 * 256 * 256, 128
    x = (_u((M, C), "gn.x", 1.0, F16).float() * 0.7 + 0.3).to(F16)
    w, b = (1 + _u((C,), "gn.w", 0.1, F16).float()).to(F16), _u((C,), "gn.b", 0.1, F16)
    aff = V.groupnorm_affine(x, w, b, 32, 1e-6)
    y = V.groupnorm_apply(x, aff, True)
    torch.cuda.synchronize()
    # statistics in fp64 with torch's own reductions on the GPU (checker arithmetic, chunked to bound memory)
    s1 = torch.zeros(C, dtype=torch.float64, device=DEV)
    s2 = torch.zeros(C, dtype=torch.float64, device=DEV)
    for i in range(0, M, 1 << 20):
        xc = x[i:i + (1 << 20)].double()
        s1 += xc.sum(0)
        s2 += (xc * xc).sum(0)
    n_el = M * (C // 32)
    mean = (s1.reshape(32, -1).sum(1) / n_el).cpu()
    var = (s2.reshape(32, -1).sum(1) / n_el).cpu() - mean ** 2
    rstd = (var + 1e-6).rsqrt()
    rows = _sample_rows(M, 200, "gn.rows").to(DEV)
    xn = (x[rows].float().cpu().reshape(-1, 32, C // 32).double() - mean[None, :, None]) * rstd[None, :, None]
    ref = torch.nn.functional.silu(xn.reshape(-1, C).float() * w.float().cpu() + b.float().cpu())
    torch.testing.assert_close(y[rows].float().cpu(), ref.cpu(), rtol=2 ** -9, atol=3e-3)


def test_decoder_tile_shipped_widths_vs_oracle():
    """One decoder tile at the SHIPPED widths (128, 256, 512, 512) vs oracle.vae_ref.decode_tile in the same fp16 contract.
    The tile is 5x16x16 latent -> 17x128x128 pixels (4.6 TFLOP: the CPU oracle finishes in well under a minute); the full
    17x32x32 tile is the same code path at 16x the voxels, covered above by the sampled-voxel convs and the GroupNorm at
    65x256x256.  K18 (mid-block attention) is PARITY UNPINNED: diffusers' Attention is absent from the reference tree."""
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
    boc = syn.VAE_BLOCK_OUT_CHANNELS
    vae = AutoencoderKLCausal3D(block_out_channels=boc, device=DEV)
    sd = syn.synth_vae_state_dict(boc, seed=0)
    vae.load_state_dict({k: v.to(F16) for k, v in sd.items()}, strict=True)
    sd16 = {k: v.to(F16).float() for k, v in sd.items()}
    z = syn.hashed_uniform((1, 16, 5, 16, 16), "ft.z", 0) * 1.7
    y = vae.decode(z.to(DEV), return_dict=False)[0]
    torch.cuda.synchronize()
    ref = VR.decode_tile(sd16, z, boc, VR.Prec(True))
    assert y.shape == ref.shape == (1, 3, 17, 128, 128)
    err = (y.float().cpu() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) / scale < 1.5e-2, float(err.max()) / scale      # ~30 fp16 layers deep (bar of test_gpu_vae.py)
    assert float(err.mean()) / scale < 1e-3
