"""CPU, gloo: hybrid Ulysses x Ring attention (xfuser `--ulysses-degree U --ring-degree R`).
Host logic under test: process-group construction, the K/V ring (batch_isend_irecv, double-buffered, joint keys only with the
local chunk), per-chunk partials + online-softmax merge bookkeeping, the Ulysses exchange inside its sub-group, and the
model-level token sharding over U*R ranks.  Kernels are the CPU doubles of tests/ (same partial format as the HIP entry points).
Property (reference tests/test_attention.py:107-109,172-174): output == unsharded attention over img|txt to bf16 rounding (TOL);
model forward sharded == unsharded."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The ring rounds P to bf16 against each chunk's own running max (the unsharded kernel: against the global running max), so
# outputs differ from the unsharded oracle by bf16 rounding of the result: 2 ulp (2^-7 relative) + 2e-3 absolute.
TOL = dict(rtol=2 ** -7, atol=2e-3)


def _worker(rank, world, port, U, R, results):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    try:
        from hunyuanvideo_efficiency_amd import synthetic as syn
        from hunyuanvideo_efficiency_amd.long_ctx_attention import UlyssesLongContextAttention
        from hunyuanvideo_efficiency_amd.inference import init_distributed, parallelize_transformer_module
        from tests.test_ulysses_gloo import CpuKernelDouble
        from oracle import dit_ref as Rf
        init_distributed(U, R, backend="gloo")
        ug, rg = UlyssesLongContextAttention._default_group, UlyssesLongContextAttention._default_ring_group
        assert dist.get_world_size(ug) == U and dist.get_world_size(rg) == R
        assert dist.get_rank(ug) == rank % U and dist.get_rank(rg) == rank // U
        E = Rf.Prec(True)
        H, n_txt = 4, 11
        s_loc = 80                                   # per-rank image tokens; U*s_loc >= 128 exercises the 2-slot path
        s_img = s_loc * world
        bf = lambda t: t.to(torch.bfloat16)
        q, k, v = (bf(syn.hashed_uniform((1, s_img + n_txt, H, 128), f"ring.{n}", 7) * 1.7) for n in "qkv")
        ref = Rf.sdpa(q.float(), k.float(), v.float(), E)
        sl = slice(rank * s_loc, (rank + 1) * s_loc)
        sp = UlyssesLongContextAttention(kernels=CpuKernelDouble)          # groups: the registered defaults
        out = sp(None, q[:, sl], k[:, sl], v[:, sl], joint_tensor_query=q[:, s_img:], joint_tensor_key=k[:, s_img:],
                 joint_tensor_value=v[:, s_img:], joint_strategy="rear")
        exp = torch.cat([ref[:, sl], ref[:, s_img:]], 1)
        torch.testing.assert_close(out.float(), exp, **TOL)
        # no joint tensors
        out = sp(None, q[:, sl], k[:, sl], v[:, sl])
        ref2 = Rf.sdpa(q[:, :s_img].float(), k[:, :s_img].float(), v[:, :s_img].float(), E)
        torch.testing.assert_close(out.float(), ref2[:, sl], **TOL)
        # overlapped API (begin / send / attend) as the blocks drive it
        d = H * 128
        rows = s_loc + n_txt
        qkv = torch.zeros(rows, 3 * d, dtype=torch.bfloat16)
        for i, t in enumerate((q, k, v)):
            qkv[:s_loc, i * d:(i + 1) * d] = t[0, sl].reshape(s_loc, d)
            qkv[s_loc:, i * d:(i + 1) * d] = t[0, s_img:].reshape(n_txt, d)
        cat = torch.zeros(rows, d + 64, dtype=torch.bfloat16)
        sp.begin(s_loc, n_txt, H, qkv.device)
        for i, nm in enumerate("qkv"):
            sp.send(nm, qkv[:, i * d:], 3 * d, qkv[s_loc:, i * d:], 3 * d)
        sp.attend(cat, d + 64)
        torch.testing.assert_close(cat[:, :d].float(), exp.reshape(-1, d), **TOL)
        assert float(cat[:, d:].abs().max()) == 0
        # heads not divisible by the ULYSSES degree -> loud error (the ring degree does not constrain heads)
        if U > 1:
            with pytest.raises(ValueError):
                sp(None, q[:, sl, :U + 1], k[:, sl, :U + 1], v[:, sl, :U + 1])

        # ---- model level: tokens sharded over all U*R ranks
        from tests import cpu_kernel_doubles as D
        from tests.test_model_sp_gloo import _build_cpu_model, _inputs
        D.install()
        cfg, model = _build_cpu_model()
        if cfg.heads_num % U == 0:
            thw = (3, 4 * world, 16)                 # (H/2) % world == 0 -> split along H
            x, t, kw = _inputs(cfg, thw)
            with torch.no_grad():
                base = model(x, t, **kw)["x"].clone()
                _, sp_model = _build_cpu_model()
                parallelize_transformer_module(sp_model, None, CpuKernelDouble)
                got = sp_model(x, t, **kw)["x"]
            err = float((got.float() - base.float()).abs().max() / base.float().abs().max())
            assert err < 1e-2, err
        results[rank] = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        results[rank] = "FAIL: " + traceback.format_exc()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("U,R", [(1, 2), (2, 2), (1, 3)])
def test_hybrid_ulysses_ring_gloo(U, R):
    world = U * R
    port = 29300 + 10 * U + R + (os.getpid() % 150)
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, U, R, results), nprocs=world, join=True)
    assert all(results.get(r) == "ok" for r in range(world)), dict(results)
