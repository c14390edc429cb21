"""GPU: the shipped deferred-max attention kernel (running max in the C operand of the S chain, Q pre-scaled, threshold 2^8,
gap-scheduled body + bounds-checked buffer DMA) against the oracle on inputs that FORCE its rare branch (guide rule 26: a
bounded-random test never takes it):
  * a late key far above everything before it -> raise_max fires mid-stream, for some rows only;
  * a steadily growing max: every tile raises it by < THR (deferred: never rescaled) vs by > THR (rescaled every tile);
  * the first tile all very negative (tile 0 fixes the initial max) and a huge first key (later tiles vanish);
The product library ships ONE attention kernel (no run-time selection); the superseded generations live in tools/attn_variants/ and
are compared on the same cases by tools/attn_variants/check_variants.py when a same-box A/B library is built."""
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
    from hunyuanvideo_efficiency_amd import ops as _ops, _lib
    _lib.torch_ops()
    return _ops


def U(shape, key, scale=1.0):
    return (syn.hashed_uniform(shape, key, 17) * (scale * math.sqrt(3.0))).to(torch.bfloat16)


def _run(ops, q, k, v, H):
    n_q, n_kv = q.shape[0], k.shape[0]
    out = torch.empty(n_q, H * 128, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(q.reshape(n_q, -1).to(DEV), k.reshape(n_kv, -1).to(DEV), v.reshape(n_kv, -1).to(DEV), out, H)
    return out


def _ref(q, k, v):
    return R.sdpa(q.float()[None], k.float()[None], v.float()[None], E)[0].reshape(q.shape[0], -1)


def _check(ops, q, k, v, H, atol=8e-3):
    ref = _ref(q, k, v)
    got = _run(ops, q, k, v, H)
    torch.testing.assert_close(got.float().cpu(), ref, rtol=2 ** -7, atol=atol)


def test_late_spike_raises_max_for_some_rows(ops):
    n_q, n_kv, H = 96, 1000, 1
    q, k, v = U((n_q, H, 128), "s.q"), U((n_kv, H, 128), "s.k", 0.3), U((n_kv, H, 128), "s.v")
    k[700] = q[40] * 2.0          # logit(40, 700) ~ 2|q|^2/sqrt(128) = 68 (log2 domain 98) >> THR above the running max
    k[70] = q[5] * 1.5
    k[999] = q[77] * 1.0          # in the partial last tile
    _check(ops, q, k, v, H)


@pytest.mark.parametrize("step", [0.3, 1.5])
def test_growing_max(ops, step):
    """key j's logit grows linearly with j for every query: per 64-key tile the max rises by 64*step*|q|^2/sqrt(128)/... -
    step 0.3 stays under the threshold (deferred, P up to 2^8), step 1.5 crosses it every tile (rescaled every tile)."""
    n_q, n_kv, H = 64, 1536, 2
    q = U((n_q, H, 128), "g.q", 0.2)
    base = torch.ones(128) / math.sqrt(128.0)
    q = (q.float() + 4.0 * base).to(torch.bfloat16)                 # every query has a common component of norm 4
    ramp = (torch.arange(n_kv).float() * (step / 64.0 * math.sqrt(128.0) / 4.0 * 8.0))[:, None, None]
    k = (U((n_kv, H, 128), "g.k", 0.2).float() + ramp * base).to(torch.bfloat16)      # logit gain per tile ~ 8*step (natural units)
    v = U((n_kv, H, 128), "g.v")
    _check(ops, q, k, v, H)


def test_first_tile_extremes(ops):
    n_q, n_kv, H = 40, 300, 1
    q, v = U((n_q, H, 128), "f.q"), U((n_kv, H, 128), "f.v")
    k = U((n_kv, H, 128), "f.k", 0.3)
    k2 = k.clone()
    k2[:64] = (-q[:1].float() * 3.0).expand(64, H, 128).to(torch.bfloat16)      # tile 0: very negative logits for query 0
    _check(ops, q, k2, v, H)
    k3 = k.clone()
    k3[0] = q[3] * 4.0                                                            # key 0 dominates query 3 completely
    _check(ops, q, k3, v, H)


@pytest.mark.parametrize("n_q,n_kv,H", [(17, 5, 2), (300, 64, 1), (256, 65, 3), (1000, 12345, 2)])
def test_shapes_and_partials(ops, n_q, n_kv, H):
    q, k, v = U((n_q, H, 128), "p.q", 2.0), U((n_kv, H, 128), "p.k"), U((n_kv, H, 128), "p.v")
    _check(ops, q, k, v, H)


def test_run_to_run_identical(ops):
    """The kernel pins inline-asm VALU instructions (row-sum adds, row-max v_max3) next to MFMAs and v_exp_f32 whose results they read;
    the compiler pads hazards only for instructions it can see.  A too-early read shows up as run-to-run different sums (that is how
    the 16x16x32 variant's first schedule was caught), so: same inputs, several runs, identical bits."""
    n_q, n_kv, H = 1000, 12345, 2
    q, k, v = U((n_q, H, 128), "d.q", 2.0), U((n_kv, H, 128), "d.k"), U((n_kv, H, 128), "d.v")
    first = _run(ops, q, k, v, H).clone()
    for _ in range(5):
        assert torch.equal(_run(ops, q, k, v, H), first)


def _modes(ops):
    """(some wave used the static row bound, some wave kept the online maximum) of the last launch that was given a workspace."""
    ws = ops._attn_ws[str(torch.device(DEV, torch.cuda.current_device()))]
    torch.cuda.synchronize()
    w = ws[:256].view(torch.int32).cpu()
    return bool(w[62]), bool(w[63])


def test_static_maximum_mode_and_its_fallback(ops):
    """Long key ranges (>= 4,096 keys, workspace given) first compute max_k |k| per head; a wave whose rows all have
    |q'| |k|_max within 90 (log2 units) of their first tile's row max runs against that bound as a static maximum (no row max per
    tile, never a rescale), the others keep the online maximum.  Both modes and the boundary between them against the oracle:
      A  ordinary data                                   -> static
      B  the bound 80-89 above the scores (one large-norm key orthogonal to every query, tuned) -> static, weights ~ 2^-85
      C  the same key 100x larger                         -> online everywhere (bound - max > 90)
      D  one query anti-aligned with every key            -> its wave online, the other waves static"""
    n_q, n_kv, H = 300, 8192, 2
    dev = torch.device(DEV, torch.cuda.current_device())
    # A
    q, k, v = U((n_q, H, 128), "sm.q", 2.0), U((n_kv, H, 128), "sm.k"), U((n_kv, H, 128), "sm.v")
    _check(ops, q, k, v, H)
    assert _modes(ops) == (True, False)
    # B / C: queries live in dims 0..63, one key has a large component in dims 64..127 only: orthogonal to every query (exact zeros)
    qh = q.clone()
    qh[:, :, 64:] = 0
    ks = U((n_kv, H, 128), "sm.ks", 0.3)
    qn = float(qh.float().norm(dim=-1).max())
    c = 128 ** -0.5 * 1.4426950408889634
    big = ks.clone()
    big[100, :, :64] = 0
    big[100, :, 64:] = 0
    big[100, :, 64] = 84.0 / (qn * c)           # |q'|_max |k|_max ~ 84: bound - row max (a few units) in (80, 90) for the longest query
    _check(ops, qh, big, v, H)
    assert _modes(ops)[0]
    huge = big.clone()
    huge[100, :, 64] = 100 * 84.0 / (qn * c)
    _check(ops, qh, huge, v, H)
    assert _modes(ops) == (False, True)
    # D: query 7 (wave 0 of workgroup 0) points away from every key
    kd = U((n_kv, H, 128), "sm.kd", 0.2)
    qd = q.clone()
    mean_dir = torch.ones(128) / 128 ** 0.5
    kd = (kd.float() + 3.0 * mean_dir).to(torch.bfloat16)          # every key has a common component of norm 3
    qd[7] = (-144.0 * mean_dir).to(torch.bfloat16)                 # scores of query 7 ~ -144 * 3 * 0.1275 = -55 (log2 units), bound ~ +69: gap > 90
    _check(ops, qd, kd, v, H)
    assert _modes(ops) == (True, True)
