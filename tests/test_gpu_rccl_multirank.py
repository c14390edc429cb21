"""GPU, REAL RCCL, one rank per GPU (needs >= 2 GPUs; skipped cleanly on a 1-GPU box): the sequence-parallel path with nothing
substituted - `all_to_all_single` (async, waited on the compute stream), `all_gather` / `all_gather_into_tensor`,
`batch_isend_irecv` ring hops on nccl sub-groups, the tile-parallel VAE all-gather - i.e. BASELINE.json configs 3 and 5 in
miniature.  Property (reference tests/test_attention.py:90-109 and hyvideo/inference.py:157-176): the sharded forward,
gathered, equals the unsharded forward on the same weights and inputs; the exchange-level op equals unsharded attention at the
reference's own bar, rtol = atol = 1e-3 (tests/test_attention.py:109).

Also: `python bench.py --gpus 2 --workload tiny` from a cold shell (no RANK in the environment) must launch its own ranks and
print one JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

N_GPU = torch.cuda.device_count()          # counting devices does not initialise HIP
needs2 = pytest.mark.skipif(N_GPU < 2, reason="real multi-rank RCCL needs >= 2 GPUs (RCCL refuses two ranks on one device)")


def _worker(rank, world, port, outdir, U, R):
    sys.path.insert(0, ROOT)
    res = "FAIL: no result"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        from hunyuanvideo_efficiency_amd import builders, synthetic as syn
        from hunyuanvideo_efficiency_amd.inference import init_distributed, parallelize_transformer_module
        from hunyuanvideo_efficiency_amd.long_ctx_attention import UlyssesLongContextAttention
        from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
        from hunyuanvideo_efficiency_amd import ops
        dev = init_distributed(U, R, backend="nccl")       # cuda:LOCAL_RANK, nccl (= RCCL) world + Ulysses / ring sub-groups
        # ---- exchange level: the reference's own property (tests/test_attention.py:90-109)
        H, n_txt, s_loc = 4, 11, 640
        s_img = s_loc * world
        q, k, v = (syn.hashed_uniform((1, s_img + n_txt, H, 128), f"rccl.{n}", 5, dev).to(torch.bfloat16) * 1.7 for n in "qkv")
        full = torch.empty(s_img + n_txt, H * 128, dtype=torch.bfloat16, device=dev)
        ops.attn_fwd(q[0].reshape(-1, H * 128), k[0].reshape(-1, H * 128), v[0].reshape(-1, H * 128), full, H)
        sl = slice(rank * s_loc, (rank + 1) * s_loc)
        sp = UlyssesLongContextAttention()
        out = sp(None, q[:, sl], k[:, sl], v[:, sl], joint_tensor_query=q[:, s_img:], joint_tensor_key=k[:, s_img:],
                 joint_tensor_value=v[:, s_img:], joint_strategy="rear")
        exp = torch.cat([full[sl], full[s_img:]], 0).reshape(1, s_loc + n_txt, H, 128)
        torch.testing.assert_close(out.float(), exp.float(), rtol=1e-3, atol=1e-3)      # /root/reference/tests/test_attention.py:109
        # ---- model level: sharded forward == unsharded forward (twice: buffers and async handles are reused across steps)
        cfg = syn.DiTConfig(hidden_size=512, heads_num=4, mm_double_blocks_depth=1, mm_single_blocks_depth=2)
        base_model = builders.build_model(cfg, dev)
        # the default exchange (one all_to_all_single per tensor, copy3d pack) and the opt-in forms (HV_SP_*: segmented output
        # exchange as a2a staging / point-to-point pairs, scatter-packed q/k) - the first >= 2-GPU run exercises all of them
        UlyssesLongContextAttention.MIN_SEG_ROWS = 16
        sp_models = []
        for env in (dict(HV_SP_NSEG="1", HV_SP_OUT_EXCHANGE="a2a", HV_SP_SCATTER_PACK="0"),
                    dict(HV_SP_NSEG="2", HV_SP_OUT_EXCHANGE="a2a", HV_SP_SCATTER_PACK="1"),
                    dict(HV_SP_NSEG="2", HV_SP_OUT_EXCHANGE="p2p", HV_SP_SCATTER_PACK="1")):
            os.environ.update(env)
            m = builders.build_model(cfg, dev)
            parallelize_transformer_module(m, None)
            sp_models.append(m)
        for (thw, txt_len, n_valid), sp_model in [(c, m) for c in (((5, 16, 8 * world), 32, 11), ((3, 8 * world + 2, 8 * world), 32, 32))
                                                  for m in sp_models]:
            x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, thw, txt_len, n_valid, seed=1)
            T, Hh, W = thw
            cos, sin = get_nd_rotary_pos_embed(cfg.rope_dim_list, [T, Hh // 2, W // 2], theta=256, use_real=True)
            kw = dict(text_states=ts.to(torch.bfloat16).to(dev), text_mask=tm.to(dev), text_states_2=ts2.to(dev),
                      freqs_cos=cos.to(dev), freqs_sin=sin.to(dev), guidance=torch.tensor([6016.0], device=dev), return_dict=True)
            t = torch.tensor([997.093], device=dev)
            with torch.no_grad():
                base = base_model(x.to(dev), t, **kw)["x"].float()
                for _ in range(2):
                    got = sp_model(x.to(dev), t, **kw)["x"].float()
            torch.cuda.synchronize()
            err = float((got - base).abs().max() / base.abs().max())
            assert got.shape == base.shape and err < 1e-2, (U, R, thw, err)
        # every rank must hold the same gathered output (the scheduler step is replicated)
        chk = got.double().sum().reshape(1)
        lst = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(bool(a == lst[0]) for a in lst), lst
        # ---- tile-parallel VAE decode over the same group == single-rank tiled decode (bit-identical)
        if R == 1:
            from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
            vae = AutoencoderKLCausal3D(block_out_channels=(64, 64, 128, 128), device=dev)
            with torch.no_grad():
                for kk, p in vae.state_dict().items():
                    p.copy_(syn.synth_param("vae." + kk, tuple(p.shape), 0, dev).to(p.dtype))
            vae.enable_tiling()
            z = syn.hashed_uniform((1, 16, 5, 40, 40), "rccl.z", 0, dev) * 1.7
            with torch.no_grad():
                one = vae.decode(z, return_dict=False)[0]
                vae.enable_tile_parallel()
                par = vae.decode(z, return_dict=False)[0]
            torch.cuda.synchronize()
            assert torch.equal(one, par)
        res = "ok"
    except Exception:  # noqa: BLE001
        import traceback
        res = "FAIL: " + traceback.format_exc()
    finally:
        with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
            f.write(res)
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass


@needs2
@pytest.mark.parametrize("U,R", [(2, 1), (1, 2)] + ([(4, 1), (2, 2)] if N_GPU >= 4 else []))
def test_sequence_parallel_forward_real_rccl(U, R, tmp_path):
    world = U * R
    port = 29700 + (os.getpid() % 40) + 4 * U + R
    # ranks come from the fork server started in conftest.py (never from this process once it has touched the GPU)
    mp.start_processes(_worker, args=(world, port, str(tmp_path), U, R), nprocs=world, join=True, start_method="forkserver")
    results = {r: open(tmp_path / f"rank{r}.txt").read() for r in range(world)}
    assert all(v == "ok" for v in results.values()), results


def _bench_launcher_helper(_idx, outdir):
    """Runs in a fork-server child that has never touched the GPU: IT may start bench.py by fork+exec (the pytest process, which
    has initialised HIP by the time this test runs, may not - tests/conftest.py:14-17)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    with open(os.path.join(outdir, "bench.json"), "w") as f:
        json.dump({"returncode": p.returncode, "stdout": p.stdout, "stderr": p.stderr[-3000:]}, f)


@needs2
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a cold shell: the parent never touches the GPU, starts 2 ranks, rank 0 prints the line.
    The cold shell is a fork-server worker (started in conftest.py before any test touched the GPU), not this process."""
    mp.start_processes(_bench_launcher_helper, args=(str(tmp_path),), nprocs=1, join=True, start_method="forkserver")
    r = json.load(open(tmp_path / "bench.json"))
    assert r["returncode"] == 0, r["stderr"]
    line = [ln for ln in r["stdout"].splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and "ulysses2" in out["config"]["parallelism"]


def test_bench_launcher_never_imports_torch_in_the_parent():
    """CPU-checkable part of the launcher contract: with --gpus > 1 and no RANK, bench.main() hands over to launch_ranks()
    before `import torch` (a parent that initialised HIP must never start rank processes by exec)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("launch_ranks(a.gpus)") < body.index("import torch")
