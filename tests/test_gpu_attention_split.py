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
