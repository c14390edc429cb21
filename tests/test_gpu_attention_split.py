"""GPU: the KV-split (load-balance) path of hv_attn_fwd_bf16 - taken for shallow grids when a workspace is supplied - must
agree with the single-pass path and with the oracle, including a partial last tile in the second half and a forced rescale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

E = R.Prec(True)
DEV = "cuda"


@pytest.mark.parametrize("n_q,n_kv,H", [(600, 8192, 1), (300, 4133, 3), (1000, 12345, 2)])
def test_kv_split_equals_single_pass_and_oracle(n_q, n_kv, H):
    from hunyuanvideo_efficiency_amd import ops
    U = lambda shape, key: (syn.hashed_uniform(shape, key, 21) * math.sqrt(3.0)).to(torch.bfloat16)
    q, k, v = U((n_q, H * 128), "sp.q"), U((n_kv, H * 128), "sp.k"), U((n_kv, H * 128), "sp.v")
    k[n_kv - 50] = q[7] * 2.0          # a dominant key late in the second half: rescale + merge weights far from 1
    outs = []
    for split in (False, True):
        o = torch.empty(n_q, H * 128, dtype=torch.bfloat16, device=DEV)
        ops.attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), o, H, kv_split_workspace=split)
        outs.append(o.float().cpu())
    assert ops._attn_workspace(n_q, n_kv, H, torch.device(DEV)) is not None      # the split path was eligible
    torch.testing.assert_close(outs[1], outs[0], rtol=2 ** -7, atol=4e-3)
    ref = R.sdpa(q.float().reshape(1, n_q, H, 128), k.float().reshape(1, n_kv, H, 128), v.float().reshape(1, n_kv, H, 128), E)
    torch.testing.assert_close(outs[1], ref.reshape(n_q, H * 128), rtol=2 ** -7, atol=8e-3)


@pytest.mark.parametrize("n_q,chunks,H", [(300, (257, 130, 4096), 2), (777, (1000, 1000, 1000, 11), 1), (64, (64,), 3)])
def test_ring_partials_and_merge(n_q, chunks, H):
    """hv_attn_partial_bf16 over K/V chunks (1- and 2-slot forms, ragged chunk sizes, a chunk shorter than a tile) +
    hv_attn_merge_bf16 == attention over the concatenated keys (single-pass kernel and oracle)."""
    from hunyuanvideo_efficiency_amd import ops
    U = lambda shape, key: (syn.hashed_uniform(shape, key, 23) * math.sqrt(3.0)).to(torch.bfloat16)
    n_kv = sum(chunks)
    q, k, v = U((n_q, H * 128), "rg.q"), U((n_kv, H * 128), "rg.k"), U((n_kv, H * 128), "rg.v")
    k[n_kv - 3] = q[5] * 2.0            # dominant key in the LAST chunk: earlier partials get merge weights ~ 0
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    full = torch.empty(n_q, H * 128, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(qd, kd, vd, full, H, kv_split_workspace=False)
    for two_slot in (False, True):
        splits = [2 if (two_slot and c >= 128) else 1 for c in chunks]
        parts = ops.AttnPartials(sum(splits), n_q, H, torch.device(DEV))
        lo = 0
        for c, s in zip(chunks, splits):
            ops.attn_partial(qd, kd[lo:lo + c], vd[lo:lo + c], parts, H, s)
            lo += c
        assert parts.used == sum(splits)
        out = torch.full((n_q, H * 128 + 64), 7.0, dtype=torch.bfloat16, device=DEV)      # wider row stride; tail untouched
        ops.attn_merge(parts, out[:, :H * 128])
        assert float((out[:, H * 128:] - 7.0).abs().max()) == 0
        torch.testing.assert_close(out[:, :H * 128].float().cpu(), full.float().cpu(), rtol=2 ** -7, atol=4e-3)
        ref = R.sdpa(q.float().reshape(1, n_q, H, 128), k.float().reshape(1, n_kv, H, 128), v.float().reshape(1, n_kv, H, 128), E)
        torch.testing.assert_close(out[:, :H * 128].float().cpu(), ref.reshape(n_q, H * 128), rtol=2 ** -7, atol=8e-3)
    assert ops.attn_suggest_splits(118811, 14850 * 4, 6) in (1, 2)
    with pytest.raises(Exception):
        ops.attn_partial(qd, kd[:64], vd[:64], parts, H, 1)      # no free slot left -> loud failure
