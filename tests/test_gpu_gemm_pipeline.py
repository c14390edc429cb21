"""GPU: the pipelined ("8-phase", counted-vmcnt) GEMM main loop against the 2-stage loop it replaces and against the oracle.

Both loops feed every output element the same k-blocks in the same order into the same fp32 MFMA accumulation chain, so their
results must agree BIT FOR BIT on any data - a race in the new loop's LDS-DMA / barrier protocol (a fragment read before its
half-tile landed, a slot re-staged under a reader) shows up as a mismatch that comes and goes, so the comparison is repeated
over many launches, shapes (K-tile counts of both parities, ragged M / N, the production shapes) and while other work runs.
HV_GEMM_2STAGE=1 selects the old loop per call (csrc/hv_gemm.hip)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

DEV = "cuda"
E = R.Prec(True)


@pytest.fixture(scope="module")
def ops():
    from hunyuanvideo_efficiency_amd import ops as _ops, _lib
    _lib.torch_ops()
    return _ops


def _u(shape, key, scale=1.0):
    return (syn.hashed_uniform(shape, key, 13, DEV) * (scale * math.sqrt(3.0))).to(torch.bfloat16)


def _both(ops, *args, **kw):
    os.environ["HV_GEMM_2STAGE"] = "1"
    try:
        ref = ops.gemm(*args, **kw).clone()
    finally:
        os.environ["HV_GEMM_2STAGE"] = "0"
    got = ops.gemm(*args, **kw)
    return got, ref


@pytest.mark.parametrize("M,N,K", [(256, 256, 192), (300, 520, 256), (1000, 768, 320), (513, 1792, 384), (4096, 3072, 3072),
                                    (2000, 9216, 448), (777, 264, 1024), (33, 3072, 15360)])
def test_pipelined_equals_2stage_bitwise_and_oracle(ops, M, N, K):
    a, w, b = _u((M, K), f"p.a{M}"), _u((N, K), f"p.w{N}", 1 / math.sqrt(K)), _u((N,), "p.b", 0.1)
    got, ref = _both(ops, a, w, b)
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    if M * N * K <= 2 ** 33:
        y = E.r(a.float().cpu() @ w.float().cpu().T + b.float().cpu())
        torch.testing.assert_close(got.float().cpu(), y, rtol=2 ** -7, atol=2e-2)


def test_pipelined_exact_integers(ops):
    """A = I-like / asymmetric small-integer operands: every product and sum is exact -> bit-exact result (fragment maps)."""
    M, N, K = 512, 512, 256
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[:, 0] += (torch.arange(M) % 3).float()
    w = ((torch.arange(N)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 11 - 5).float()
    got = ops.gemm(a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), None)
    assert torch.equal(got.float().cpu(), a @ w.T)


def test_pipelined_race_screen_production_shapes(ops):
    """The step's own launches (linear1 with the column split + GELU, linear2 with gate + residual), 6 launches each, compared
    bit for bit with the 2-stage loop; a copy kernel keeps the memory system busy on a second stream meanwhile."""
    S, D = 119056, 3072
    x = _u((S, D), "r.x")
    w1, b1 = _u((7 * D, D), "r.w1", 1 / math.sqrt(D)), _u((7 * D,), "r.b1", 0.1)
    qkv = torch.empty(S, 3 * D, dtype=torch.bfloat16, device=DEV)
    cat = torch.empty(S, 5 * D, dtype=torch.bfloat16, device=DEV)
    os.environ["HV_GEMM_2STAGE"] = "1"
    ops.gemm(x, w1, b1, out=qkv, n_split=3 * D, out1=cat[:, D:], act1=ops.ACT_GELU_TANH)
    os.environ["HV_GEMM_2STAGE"] = "0"
    q_ref, c_ref = qkv.clone(), cat[:, D:].clone()
    side = torch.cuda.Stream()
    junk_a = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    junk_b = torch.empty_like(junk_a)
    for i in range(6):
        qkv.zero_()
        cat.zero_()
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a)
        ops.gemm(x, w1, b1, out=qkv, n_split=3 * D, out1=cat[:, D:], act1=ops.ACT_GELU_TANH)
        torch.cuda.synchronize()
        assert torch.equal(qkv, q_ref) and torch.equal(cat[:, D:], c_ref), f"launch {i} differs from the 2-stage loop"
    del q_ref, c_ref, qkv
    cat[:, :D] = _u((S, D), "r.attn")
    w2, b2, gate = _u((D, 5 * D), "r.w2", 1 / math.sqrt(5 * D)), _u((D,), "r.b2", 0.1), _u((D,), "r.g", 0.5)
    os.environ["HV_GEMM_2STAGE"] = "1"
    ref = ops.gemm(cat, w2, b2, out=torch.empty_like(x), gate=gate, res=x).clone()
    os.environ["HV_GEMM_2STAGE"] = "0"
    for i in range(6):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a)
        got = ops.gemm(cat, w2, b2, out=torch.empty_like(x), gate=gate, res=x)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), f"launch {i} differs from the 2-stage loop"
