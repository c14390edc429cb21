"""GPU parity: every C-ABI kernel against the oracle (oracle/dit_ref.py, bf16-emulated contract) on the
same seeded inputs.  Tolerances are stated per test: the kernels and the oracle round to bf16 at the same
points, so what remains is fp32 accumulation order (-> at most 1 bf16 ulp = 2^-8 relative after rounding)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

E = R.Prec(True)
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from hunyuanvideo_efficiency_amd import ops as _ops
    from hunyuanvideo_efficiency_amd import _lib
    _lib.load()  # fails loudly if the HIP extension is missing
    return _ops


def U(shape, key, scale=1.0):
    return syn.hashed_uniform(shape, key, 7) * (scale * math.sqrt(3.0))


def bf(x):
    return x.to(torch.bfloat16)


def close(got, ref, rtol=2 ** -7, atol=1e-2):
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("M,D", [(5, 256), (1000, 3072), (33, 4096), (7, 1024)])
def test_ln_modulate(ops, M, D):
    x, sh, sc = bf(U((M, D), "ln.x", 2.0)), bf(U((D,), "ln.sh", 0.3)), bf(U((D,), "ln.sc", 0.3))
    ref = E.r(R.ln_modulate(x.float()[None], sh.float()[None], sc.float()[None], E))[0]
    got = ops.ln_modulate(x.to(DEV), sh.to(DEV), sc.to(DEV))
    close(got, ref, atol=2e-2)   # 1 bf16 ulp at |y| <= 4
    # affine mode (token refiner LayerNorm) and no-modulation mode
    w, b = bf(1 + U((D,), "ln.w", 0.1)), bf(U((D,), "ln.b", 0.1))
    ref = E.r(torch.nn.functional.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-6))
    close(ops.ln_modulate(x.to(DEV), b.to(DEV), w.to(DEV), affine=True), ref, atol=2e-2)
    ref = E.r(torch.nn.functional.layer_norm(x.float(), (D,), None, None, 1e-6))
    close(ops.ln_modulate(x.to(DEV)), ref, atol=2e-2)


@pytest.mark.parametrize("n_rows,n_rope,H", [(37, 30, 2), (300, 300, 24), (64, 0, 3)])
def test_qknorm_rope(ops, n_rows, n_rope, H):
    ld = 3 * H * 128
    qkv = bf(U((n_rows, ld), "qk.x", 1.5))
    qw, kw = bf(1 + U((128,), "qk.qw", 0.1)), bf(1 + U((128,), "qk.kw", 0.1))
    cos, sin = R.rope_tables([max(n_rope, 1), 1, 1], [16, 56, 56], 256.0)
    ang = U((max(n_rope, 1), 64), "qk.ang", 3.0)   # arbitrary angles: exercises every column
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
    q, k, v = qkv.float().reshape(1, n_rows, 3, H, 128).unbind(2)
    qn, kn = R.rms_norm(q, qw.float(), E), R.rms_norm(k, kw.float(), E)
    if n_rope:
        qn = torch.cat([R.apply_rope(qn[:, :n_rope], cos[:n_rope], sin[:n_rope], E), qn[:, n_rope:]], 1)
        kn = torch.cat([R.apply_rope(kn[:, :n_rope], cos[:n_rope], sin[:n_rope], E), kn[:, n_rope:]], 1)
    ref = torch.stack([qn, kn, v], 2).reshape(n_rows, ld)
    got = ops.qknorm_rope_(qkv.to(DEV), qw.to(DEV), kw.to(DEV), cos.to(DEV), sin.to(DEV), n_rope, H, H * 128)
    close(got, ref, atol=2e-2)
    assert torch.equal(got[:, 2 * H * 128:].cpu(), qkv[:, 2 * H * 128:])  # v untouched
    # scatter form (hv_qknorm_rope_scatter_bf16): the same values, out of place, head blocks of hb heads laid out [block][row][hb*128]
    # (the Ulysses send layout), source rows untouched
    for hb in [h for h in (1, 2, H, 2 * H) if (2 * H) % h == 0]:
        nb = 2 * H // hb
        src = qkv.to(DEV)
        buf = torch.full((nb, n_rows, hb * 128), 7.0, dtype=torch.bfloat16, device=DEV)
        out = ops.qknorm_rope_(src, qw.to(DEV), kw.to(DEV), cos.to(DEV), sin.to(DEV), n_rope, H, H * 128, out=buf.permute(1, 0, 2))
        assert torch.equal(src.cpu(), qkv)
        want = got[:, :2 * H * 128].reshape(n_rows, nb, hb * 128).permute(1, 0, 2)
        assert torch.equal(buf, want) and out.data_ptr() == buf.data_ptr()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 520, 128), (1000, 768, 256), (77, 64, 3072), (513, 1792, 256)])
def test_gemm_bias_and_act(ops, M, N, K):
    a, w, b = bf(U((M, K), "g.a")), bf(U((N, K), "g.w", 1 / math.sqrt(K))), bf(U((N,), "g.b", 0.1))
    y = E.r(a.float() @ w.float().T + b.float())
    close(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV)), y, atol=2e-2)
    close(ops.gemm(a.to(DEV), w.to(DEV), None), E.r(a.float() @ w.float().T), atol=2e-2)
    close(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), act=ops.ACT_GELU_TANH), R.gelu_tanh(y, E), atol=2e-2)
    close(ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), act=ops.ACT_SILU), E.r(torch.nn.functional.silu(y)), atol=2e-2)


def test_gemm_exact_integers(ops):
    """A = I-like / asymmetric small-integer operands: every product and sum is exact in bf16/fp32, so the
    result must be bit-exact; catches transposed or permuted fragment maps (guide: A=I with asymmetric B)."""
    M, N, K = 512, 512, 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[:, 0] += (torch.arange(M) % 3).float()
    w = ((torch.arange(N)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 11 - 5).float()
    ref = a @ w.T
    got = ops.gemm(bf(a).to(DEV), bf(w).to(DEV), None)
    assert torch.equal(got.float().cpu(), ref)


def test_gemm_gate_residual_inplace_and_split(ops):
    M, N, K = 700, 512, 192
    a, w, b = bf(U((M, K), "gr.a")), bf(U((N, K), "gr.w", 1 / math.sqrt(K))), bf(U((N,), "gr.b", 0.1))
    gate, res = bf(U((N,), "gr.g", 0.5)), bf(U((M, N), "gr.r"))
    y = E.r(a.float() @ w.float().T + b.float())
    ref = R.gate_residual(res.float()[None], y[None], gate.float()[None], E)[0]
    r_dev = res.to(DEV).clone()
    got = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), out=r_dev, gate=gate.to(DEV), res=r_dev)
    close(got, ref, atol=2e-2)
    # column split: [0,256) plain into a wide buffer, [256,512) GELU into another buffer at an offset
    out0 = torch.zeros(M, 320, dtype=torch.bfloat16, device=DEV)
    out1 = torch.zeros(M, 400, dtype=torch.bfloat16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), out=out0, n_split=256, out1=out1[:, 144:], act1=ops.ACT_GELU_TANH)
    close(out0[:, :256], y[:, :256], atol=2e-2)
    close(out1[:, 144:], R.gelu_tanh(y[:, 256:], E), atol=2e-2)
    assert float(out0[:, 256:].abs().max()) == 0 and float(out1[:, :144].abs().max()) == 0


def test_linear_smallm(ops):
    x, w, b = bf(U((1, 3072), "sm.x")), bf(U((768, 3072), "sm.w", 0.02)), bf(U((768,), "sm.b", 0.1))
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    close(ops.linear_smallm(xd, wd, bd), E.linear(x.float(), w.float(), b.float()), atol=1e-2)
    close(ops.linear_smallm(xd, wd, bd, silu_in=True),
          E.linear(E.r(torch.nn.functional.silu(x.float())), w.float(), b.float()), atol=1e-2)
    close(ops.linear_smallm(xd, wd, bd, silu_out=True),
          E.r(torch.nn.functional.silu(E.linear(x.float(), w.float(), b.float()))), atol=1e-2)
    x3 = bf(U((3, 256), "sm.x3"))
    w3 = bf(U((40, 256), "sm.w3", 0.06))
    close(ops.linear_smallm(x3.to(DEV), w3.to(DEV)), E.linear(x3.float(), w3.float(), None), atol=1e-2)


def _attn_ref(q, k, v):
    return R.sdpa(q.float()[None], k.float()[None], v.float()[None], E)[0]


@pytest.mark.parametrize("n_q,n_kv,H", [(64, 64, 1), (300, 333, 2), (1000, 1500, 3), (17, 5, 2), (513, 4096, 1)])
def test_attention(ops, n_q, n_kv, H):
    q, k, v = (bf(U((n, H, 128), f"at.{nm}")) for n, nm in ((n_q, "q"), (n_kv, "k"), (n_kv, "v")))
    ref = _attn_ref(q, k, v).reshape(n_q, H * 128)
    out = torch.empty(n_q, H * 128, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(q.reshape(n_q, -1).to(DEV), k.reshape(n_kv, -1).to(DEV), v.reshape(n_kv, -1).to(DEV), out, H)
    # outputs are convex combinations of |v| <= 1.73: atol = 2 bf16 ulps at 1.0; rtol = 1 ulp
    close(out, ref, rtol=2 ** -7, atol=8e-3)


def test_attention_strided_fused_buffers(ops):
    """q,k,v read out of one fused [S, 3*H*128] buffer, output written into a wider concat buffer, and the
    two-segment varlen semantics of attenion.py:34-57,107-120 via two calls."""
    H, S, cut = 2, 400, 331
    qkv = bf(U((S, 3 * H * 128), "ats.qkv"))
    cat = torch.zeros(S, H * 128 + 64, dtype=torch.bfloat16, device=DEV)
    d = qkv.to(DEV)
    hd = H * 128
    for lo, hi in ((0, cut), (cut, S)):
        ops.attn_fwd(d[lo:hi, :hd], d[lo:hi, hd:2 * hd], d[lo:hi, 2 * hd:], cat[lo:hi, :hd], H)
    q, k, v = qkv.float().reshape(1, S, 3, H, 128).unbind(2)
    ref = R.attention_varlen(q, k, v, torch.tensor([0, cut, S], dtype=torch.int32), E)[0]
    close(cat[:, :hd], ref, atol=8e-3)
    assert float(cat[:, hd:].abs().max()) == 0


def test_attention_forced_rescale(ops):
    """A key far larger than everything before it appears in a late tile: the online-softmax rescale branch
    must fire there (a bounded-random test never exercises it)."""
    n_q, n_kv, H = 96, 640, 1
    q, k, v = bf(U((n_q, H, 128), "fr.q")), bf(U((n_kv, H, 128), "fr.k", 0.3)), bf(U((n_kv, H, 128), "fr.v"))
    k[500] = q[40] * 2.0          # logit ~ 2*|q|^2/sqrt(128) >> others, for query 40 (and partially others)
    k[70] = q[5] * 1.5
    ref = _attn_ref(q, k, v).reshape(n_q, 128)
    out = torch.empty(n_q, 128, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(q.reshape(n_q, -1).to(DEV), k.reshape(n_kv, -1).to(DEV), v.reshape(n_kv, -1).to(DEV), out, H)
    close(out, ref, rtol=2 ** -7, atol=8e-3)


def test_patchify_unpatchify_euler(ops):
    C, T, H, W = 16, 3, 8, 12
    x = U((C, T, H, W), "pu.x")
    A = ops.patchify(x.to(DEV))
    ref = x.reshape(C, T, H // 2, 2, W // 2, 2).permute(1, 2, 4, 0, 3, 5).reshape(T * (H // 2) * (W // 2), C * 4)
    assert torch.equal(A.float().cpu(), E.r(ref))
    y = bf(U((T * (H // 2) * (W // 2), C * 4), "pu.y"))
    got = ops.unpatchify(y.to(DEV), C, T, H, W)
    ref = R.unpatchify(y.float()[None], T, H // 2, W // 2, C, [1, 2, 2])[0]
    assert torch.equal(got.float().cpu(), ref)
    s, v = U((1, C, T, H, W), "pu.s"), bf(U((1, C, T, H, W), "pu.v"))
    sig = R.flow_sigmas(50, 7.0)
    ref = R.euler_step(s, v, sig, 3)
    got = ops.euler_step_(s.to(DEV).clone(), v.to(DEV), float(sig[4] - sig[3]))
    torch.testing.assert_close(got.cpu(), ref, rtol=0, atol=1e-7)


def test_masked_mean_and_broadcast(ops):
    x = bf(U((32, 512), "mm.x"))
    mask = torch.zeros(32, dtype=torch.int32)
    mask[:11] = 1
    ref = (x.float() * mask[:, None].float()).sum(0) / mask.sum()
    close(ops.masked_mean(x.to(DEV), mask.to(DEV)), E.r(ref), atol=4e-3)
    dst = torch.zeros(5, 600, dtype=torch.bfloat16, device=DEV)
    ops.broadcast_row_(x[3].to(DEV), dst[:, :512])
    assert torch.equal(dst[:, :512].cpu(), x[3][None].expand(5, 512)) and float(dst[:, 512:].abs().max()) == 0


def test_bad_arguments_fail_loudly(ops):
    from hunyuanvideo_efficiency_amd._lib import HVKernelError
    with pytest.raises(HVKernelError):
        ops.ln_modulate(torch.zeros(4, 256, dtype=torch.bfloat16))        # CPU tensor: no fallback
    with pytest.raises(HVKernelError):
        ops.gemm(torch.zeros(4, 72, dtype=torch.bfloat16, device=DEV), torch.zeros(8, 72, dtype=torch.bfloat16, device=DEV))
