"""GPU: the opt-in FP8-MFMA path (BASELINE.json configs[3], "fp8 weight path (CDNA4 fp8 MFMA)") against the
quantisation-aware oracle (oracle/dit_ref.py: fp8_quant_rows, Fp8MfmaPrec).

What is pinned: (1) the operand maps of v_mfma_scale_f32_16x16x128_f8f6f4 as the kernel feeds it (exact small-integer data:
every product and partial sum is exact, the result must be bit-exact); (2) the per-token quantisers bit for bit; (3) the GEMM on
random data given the SAME quantised operands (only fp32 accumulation order differs: 1 bf16 ulp); (4) a whole d=512 model with
FP8 compute == the oracle with the same quantisation points, 3e-2 of range like every model-level bf16 test, and its distance
to the reference's weight-only semantics (the price of quantising activations to e4m3) stays below 1e-1 of range.
Tolerances: e4m3 has a 3-bit mantissa (2^-4 relative per element); a K-long dot product of independently rounded terms
carries ~2^-4 / sqrt(K) relative noise, i.e. 1e-3..3e-3 at K = 3072..512."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

DEV = "cuda"
E = R.Prec(True)
FP8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from hunyuanvideo_efficiency_amd import ops as _ops, _lib
    _lib.torch_ops()
    return _ops


def U(shape, key, scale=1.0):
    return syn.hashed_uniform(shape, key, 19) * (scale * math.sqrt(3.0))


def test_gemm_fp8_exact_integers(ops):
    """A = I-like + asymmetric small integers, W = asymmetric small integers, all exactly representable in e4m3: bit-exact."""
    M, N, K = 512, 512, 512
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 5) % K] = 1.0
    a[:, 3] += (torch.arange(M) % 3).float()
    a[:, K - 1] -= (torch.arange(M) % 2).float()
    w = ((torch.arange(N)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 11 - 5).float()
    ref = a @ w.T
    one = torch.ones(M, dtype=torch.float32, device=DEV)
    ws = torch.ones(1, dtype=torch.bfloat16, device=DEV)
    got = ops.gemm_fp8(a.to(FP8).to(DEV), one, w.to(FP8).to(DEV), ws)
    assert torch.equal(got.float().cpu(), ref)
    # row / tensor scales and bias: powers of two and small integers keep everything exact
    rs = (2.0 ** (torch.arange(M) % 4 - 2).float()).to(DEV)
    b = ((torch.arange(N) % 7) - 3).float().to(torch.bfloat16).to(DEV)
    got = ops.gemm_fp8(a.to(FP8).to(DEV), rs, w.to(FP8).to(DEV), torch.tensor([0.5], dtype=torch.bfloat16, device=DEV), b)
    assert torch.equal(got.float().cpu(), (ref * rs.cpu()[:, None] * 0.5 + b.float().cpu()).to(torch.bfloat16).float())


@pytest.mark.parametrize("M,K", [(5, 256), (1000, 3072), (33, 4096), (300, 15360)])
def test_quantisers_bit_exact(ops, M, K):
    x = U((M, K), "q.x", 2.0).to(torch.bfloat16)
    x[min(3, M - 1)] = 0                                    # an all-zero row: scale 1, zeros
    q_ref, s_ref = R.fp8_quant_rows(x.float())
    q, s = ops.quant_rows_fp8(x.to(DEV))
    assert torch.equal(s.cpu(), s_ref[:, 0])
    assert torch.equal(q.cpu().float(), q_ref)
    if K <= 4096:
        sh, sc = U((K,), "q.sh", 0.3).to(torch.bfloat16), U((K,), "q.sc", 0.3).to(torch.bfloat16)
        y = E.r(R.ln_modulate(x.float()[None], sh.float()[None], sc.float()[None], E))[0]
        q_ref, s_ref = R.fp8_quant_rows(y)
        q, s = ops.ln_modulate_fp8(x.to(DEV), sh.to(DEV), sc.to(DEV))
        # the LayerNorm itself is within 1 bf16 ulp of the oracle (test_gpu_ops.py::test_ln_modulate); where it agrees exactly the
        # quantised bytes must too - compare the dequantised values at the coarser of the two tolerances
        torch.testing.assert_close(s.cpu(), s_ref[:, 0], rtol=2 ** -7, atol=0)
        deq, deq_ref = q.cpu().float() * s.cpu()[:, None], q_ref * s_ref
        assert float((deq - deq_ref).abs().max()) <= float(s_ref.max()) * 32.0 + 1e-6      # one e4m3 step at |q| <= 448
        assert float(((deq - deq_ref).abs() > 1e-6).float().mean()) < 0.05


@pytest.mark.parametrize("M,N,K", [(256, 256, 384), (300, 520, 512), (1000, 768, 1024), (513, 1792, 3072), (2000, 3072, 15360)])
def test_gemm_fp8_random_vs_oracle_same_operands(ops, M, N, K):
    a = U((M, K), f"f.a{K}", 1.5).to(torch.bfloat16)
    w = U((N, K), f"f.w{K}", 1.0)
    b = U((N,), "f.b", 0.1).to(torch.bfloat16)
    wscale = (w.abs().max() / 448.0).to(torch.bfloat16)
    w8 = (w / wscale.float()).clamp(-448, 448).to(FP8)
    aq, asc = ops.quant_rows_fp8(a.to(DEV))
    got = ops.gemm_fp8(aq, asc, w8.to(DEV), wscale.reshape(1).to(DEV), b.to(DEV))
    P = R.Fp8MfmaPrec()
    ref = P.linear(a.float(), P.fp8(w8, wscale), b.float())
    torch.testing.assert_close(got.float().cpu(), ref, rtol=2 ** -7, atol=2e-2)
    # epilogues (same code path as hv_gemm_bf16's): y may legitimately sit 1 bf16 ulp from the oracle's (fp32 summation order),
    # and an epilogue can amplify that relative to a smaller result (y * gate + res), so the epilogues are checked EXACTLY against
    # the oracle's formulas applied to the kernel's own y
    y = got.float().cpu()
    got = ops.gemm_fp8(aq, asc, w8.to(DEV), wscale.reshape(1).to(DEV), b.to(DEV), act=ops.ACT_GELU_TANH)
    torch.testing.assert_close(got.float().cpu(), R.gelu_tanh(y, E), rtol=2 ** -7, atol=2e-2)     # (GELU: exp2-based tanh vs torch's)
    gate, res = U((N,), "f.g", 0.5).to(torch.bfloat16), U((M, N), "f.r").to(torch.bfloat16)
    r_dev = res.to(DEV).clone()
    got = ops.gemm_fp8(aq, asc, w8.to(DEV), wscale.reshape(1).to(DEV), b.to(DEV), out=r_dev, gate=gate.to(DEV), res=r_dev)
    assert torch.equal(got.float().cpu(), R.gate_residual(res.float()[None], y[None], gate.float()[None], E)[0])
    ns = (N // 2) // 8 * 8                  # column split (linear1's qkv | mlp): a multiple of 8
    out0 = torch.zeros(M, ns + 64, dtype=torch.bfloat16, device=DEV)
    out1 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_fp8(aq, asc, w8.to(DEV), wscale.reshape(1).to(DEV), b.to(DEV), out=out0, n_split=ns, out1=out1[:, 8:])
    assert torch.equal(out0[:, :ns].float().cpu(), y[:, :ns]) and torch.equal(out1[:, 8:8 + N - ns].float().cpu(), y[:, ns:])
    assert float(out0[:, ns:].abs().max()) == 0 and float(out1[:, :8].abs().max()) == 0


def test_model_fp8_mfma_vs_quantisation_aware_oracle():
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear, enable_fp8_mfma
    from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    cfg = syn.DiTConfig(hidden_size=512, heads_num=4, mm_double_blocks_depth=1, mm_single_blocks_depth=2)
    model = build_model(cfg, DEV)
    sd_bf16 = {k: p.float().cpu() for k, p in model.state_dict().items()}
    convert_fp8_linear(model, None, torch.bfloat16)
    assert enable_fp8_mfma(model) == 3
    thw, txt_len, n_valid = (5, 16, 16), 32, 11
    x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, thw, txt_len, n_valid, seed=2)
    T, H, W = thw
    cos, sin = get_nd_rotary_pos_embed(cfg.rope_dim_list, [T, H // 2, W // 2], theta=256, use_real=True)
    g = torch.tensor([6016.0])
    t = torch.tensor([997.093])
    with torch.no_grad():
        out = model(x.to(DEV), t.to(DEV), text_states=ts.to(torch.bfloat16).to(DEV), text_mask=tm.to(DEV), text_states_2=ts2.to(DEV),
                    freqs_cos=cos.to(DEV), freqs_sin=sin.to(DEV), guidance=g.to(DEV))["x"].float().cpu()
    assert model._ws.xq is not None, "the FP8-MFMA path did not run"
    P = R.Fp8MfmaPrec()
    sd_q, sd_w = {}, {}
    for k, p in model.state_dict().items():
        sd_q[k] = p.float().cpu()
        sd_w[k] = p.float().cpu()
    for name, layer in model.named_modules():
        if hasattr(layer, "fp8_scale"):
            sd_q[name + ".weight"] = P.fp8(layer.weight.cpu(), layer.fp8_scale.cpu())
            sd_w[name + ".weight"] = (layer.weight.cpu().to(torch.bfloat16) * layer.fp8_scale.cpu()).float()
    args = (cfg, x, t, E.r(ts), tm, ts2, cos, sin, g)
    ref_q = R.dit_forward(sd_q, *args, P)              # our contract: e4m3 activations x e4m3 weights
    ref_w = R.dit_forward(sd_w, *args, E)              # the reference's semantics: weight-only FP8
    ref_b = R.dit_forward(sd_bf16, *args, E)           # no FP8 at all
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(out, ref_q) < 3e-2, rel(out, ref_q)
    assert rel(out, ref_w) < 1e-1, rel(out, ref_w)
    assert rel(ref_w, ref_b) < 0.25                     # (context: what weight-only FP8 itself costs on random-init weights)
