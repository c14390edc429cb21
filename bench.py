#!/usr/bin/env python3
"""Benchmark of the HunyuanVideo denoise hot path on MI355X (BASELINE.json metric).

One "step" = one full denoise step: HYVideoDiffusionTransformer.forward (20 double + 40 single blocks,
d=3072, 24 heads) + FlowMatchDiscreteScheduler.step on synthetic inputs already resident in HBM,
random-init (deterministic hash) bf16 weights.  Workload at N=1: 720x1280x129f (latent 16x33x90x160,
S = 118,800 image + 256 text tokens), the configuration BASELINE.json's metric is quoted on.
N>1: the same single video sharded over the token axis (Ulysses sequence parallelism over RCCL all-to-all),
i.e. strong scaling; `value` is whole-job denoise-steps/s.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant kernel = flash
attention, MFMA-bound, measured with HIP events on the launch stream inside the timed region) and
`cpu_baseline` (the oracle, i.e. the CPU restatement of the reference path, timed on this box's host cores on
a bounded sample and FLOP-scaled - a reported baseline, not a target).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (latent T, H, W)
    "720p129f": (33, 90, 160),
    "544p65f": (17, 68, 120),
    "720p257f": (65, 90, 160),
    "tiny": (5, 16, 16),
}
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def step_flops(s_img, s_txt, d=3072, n_double=20, n_single=40):
    s = s_img + s_txt
    return (n_double + n_single) * (24 * d * d * s + 4 * s * s * d) + 256 * s_img * d


def cpu_baseline(f_step, seconds_budget=30.0):
    """the oracle (CPU port of the reference path, fp32 eager, all host threads) timed on
    (i) BASELINE.json configs[0] exactly (tiny DiT, one denoise step), (ii) one double + one single block at full width d=3072,
    S=4,096+256 tokens - the leg `value` is FLOP-scaled from (stated as an extrapolation) - and (iii) one reduced VAE decoder
    tile.  Bounded: about 10-30 s of CPU work on the GPU box's host cores.
    The oracle outputs of leg (ii) are not thrown away: the same two blocks run on the GPU (HIP kernels, bf16) on the same
    bf16-rounded weights and inputs and the distance is reported as `block_check` and asserted (oracle = checker, never the thing
    measured as `value`)."""
    import torch
    from hunyuanvideo_efficiency_amd import synthetic as syn
    from oracle import dit_ref as R
    from oracle import vae_ref as VR
    cores = torch.get_num_threads()
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"
    legs = {}
    with torch.no_grad():
        # (i) config 1: tiny DiT (d=256, 1+1 blocks), 16x16x5 latent, one denoise step
        tcfg = syn.tiny_config()
        tsd = {k: syn.synth_param(k, shp, 0, "cpu") for k, shp in syn.dit_param_shapes(tcfg).items()}
        x, ts, tm, ts2 = syn.synth_dit_inputs(tcfg, (5, 16, 16), 32, 11, seed=0)
        rc, rs = R.rope_tables([5, 8, 8], tcfg.rope_dim_list, 256.0)
        sig = R.flow_sigmas(1, 7.0)
        g = torch.tensor([6016.0])
        t0 = time.perf_counter()
        v = R.dit_forward(tsd, tcfg, x, R.flow_timesteps(sig)[0:1], ts, tm, ts2, rc, rs, g, R.FP32)
        R.euler_step(x, v, sig, 0)
        legs["config1_tiny_step_s"] = time.perf_counter() - t0
        # (ii) one double + one single block, d=3072, S=4096+256
        cfg = syn.DiTConfig(mm_double_blocks_depth=1, mm_single_blocks_depth=1)
        d, s_img, s_txt = cfg.hidden_size, 4096, 256
        from tests.oracle_checks import fullwidth_block_inputs, fullwidth_blocks
        sd = {}
        for k, shp in syn.dit_param_shapes(cfg).items():
            if k.startswith("double_blocks.0.") or k.startswith("single_blocks.0."):
                sd[k] = syn.synth_param(k, shp, 0, gen_dev).to(torch.bfloat16).float().cpu()      # the values the GPU model holds
        img, txt, vec = fullwidth_block_inputs(cfg, s_img, s_txt)
        cos, sin = R.rope_tables([4, 32, 32], cfg.rope_dim_list, 256.0)
        cu = torch.tensor([0, s_img + 11, s_img + s_txt], dtype=torch.int32)
        f_sample = 2 * (24 * d * d * (s_img + s_txt) + 4 * (s_img + s_txt) ** 2 * d)
        reps, t0 = 0, time.perf_counter()
        while True:
            io, to = R.double_block(sd, "double_blocks.0.", img, txt, vec, cu, cos, sin, cfg.heads_num, R.FP32)
            so = R.single_block(sd, "single_blocks.0.", torch.cat([img, txt], 1), vec, s_txt, cu, cos, sin, cfg.heads_num, R.FP32)
            reps += 1
            el = time.perf_counter() - t0
            if el > seconds_budget * 0.5 or reps >= 4:
                break
        sec_per_sample = el / reps
        cpu_flops = f_sample / sec_per_sample
        del sd
        # the same blocks on the GPU against the fp32 oracle outputs just timed: bf16 drift bound (the tight, bf16-emulated
        # comparison is tests/test_gpu_fullsize_c5.py::test_blocks_shipped_width_vs_oracle)
        block_check = None
        if torch.cuda.is_available():
            r = fullwidth_blocks("cuda", s_img=s_img, s_txt=s_txt, oracle_out=(io, to, so))
            errs = {k: float((g - o).abs().max() / o.abs().max()) for k, (g, o) in r.items()}
            block_check = {"what": "GPU double+single block (d=3072, 24 heads, S=4096+256, bf16) vs the fp32 oracle outputs of the "
                                   "timed CPU sample: max |diff| / max |ref|", **errs, "tol": 3e-2}
            assert max(errs.values()) < 3e-2, f"GPU blocks deviate from the oracle: {errs}"
        del img, txt, io, to, so
        # (iii) one reduced VAE decoder tile: channels (32,64,128,128), latent 5x16x16 -> [3,17,128,128]
        boc = (32, 64, 128, 128)
        vsd = syn.synth_vae_state_dict(boc, seed=0)
        z = syn.hashed_uniform((1, 16, 5, 16, 16), "cpu.z", 0) * 1.7
        t0 = time.perf_counter()
        VR.decode_tile(vsd, z, boc, VR.FP32)
        legs["vae_reduced_tile_s"] = time.perf_counter() - t0
    return {"value": cpu_flops / f_step, "unit": "denoise-steps/s (FLOP-scaled extrapolation from the sample)",
            "cores": cores, "kind": "port",
            "sample": f"oracle fp32: 1 double + 1 single block, d=3072, S={s_img}+{s_txt}, {reps} reps, "
                      f"{sec_per_sample:.2f} s each = {cpu_flops / 1e12:.3f} TFLOP/s; config 1 (tiny DiT, 1 step) "
                      f"{legs['config1_tiny_step_s']:.3f} s"
                      f"; reduced VAE decoder tile (32,64,128,128) 5x16x16 latent {legs['vae_reduced_tile_s']:.2f} s",
            "block_check": block_check, **legs}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` from a cold shell (no RANK in the environment): start N rank processes, one per GPU, as fresh
    children of THIS process - which never imports torch or touches HIP, so nothing that initialised the GPU ever execs - with
    the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the reference's recipe is
    `torchrun --nproc_per_node=8 sample_video.py ...`, scripts/run_sample_video_multigpu.sh:35).  Rank 0 inherits stdout and
    prints the one JSON line; returns non-zero if any rank fails (the others are then terminated by exact PID)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            code = pr.poll()
            if code is None:
                continue
            pending.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:      # one rank died: the rest would hang in a collective
                    other.terminate()
        time.sleep(0.2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="720p129f", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sp", action="store_true", help="N=1 only: run the sequence-parallel code path on a 1-rank RCCL group "
                    "(pack/unpack + self all-to-all + chunked QKV GEMMs) to price its overhead")
    ap.add_argument("--ring-degree", type=int, default=1, help="N > 1: hybrid Ulysses x Ring with this ring degree (ulysses = N / ring); "
                    "default 1 = pure Ulysses")
    ap.add_argument("--use-fp8", action="store_true", help="BASELINE.json configs[3]: FP8 (e4m3) weight storage for the block linears, "
                    "dequantised per call into the bf16 MFMA GEMM (the reference's weight-only semantics)")
    ap.add_argument("--fp8-mfma", action="store_true", help="with --use-fp8: run the block linears on the CDNA4 FP8 matrix cores "
                    "(v_mfma_scale_f32_16x16x128_f8f6f4; activations quantised per token to e4m3) instead of dequantising the weights "
                    "to bf16 per call - an opt-in approximation beyond the reference's weight-only FP8 (tests/test_gpu_fp8_mfma.py)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (544x960x65f, 720p x 257f, FP8-MFMA) that a "
                    "default single-GPU 720p129f run appends after its timed leg")
    ap.add_argument("--secondary-budget", type=float, default=150.0, help="wall-clock seconds the secondary workloads may take in total")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-by-`value`) VAE tiled decode of the same video")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # cold shell: become the launcher (before torch is imported; this process never touches the GPU)
        raise SystemExit(launch_ranks(a.gpus))

    # Only the ONE JSON line may reach stdout: RCCL prints a version banner to stdout when its first communicator is created
    # (and libraries may log there too), so this process keeps a private handle on the real stdout for the result line and
    # points fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.force_sp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from hunyuanvideo_efficiency_amd import _lib, ops, synthetic as syn
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    _lib.load()

    tiny = a.workload == "tiny"
    cfg = syn.tiny_config() if tiny else syn.DiTConfig()
    T, H, W = WORKLOADS[a.workload]
    s_img, s_txt = T * (H // 2) * (W // 2), 256
    model = build_model(cfg, dev, seed=0)
    if a.use_fp8:
        from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear
        convert_fp8_linear(model, None, torch.bfloat16)
        if a.fp8_mfma:
            from hunyuanvideo_efficiency_amd.modules.fp8_optimization import enable_fp8_mfma
            enable_fp8_mfma(model)
    elif a.fp8_mfma:
        raise SystemExit("--fp8-mfma needs --use-fp8 (FP8 weights)")
    if world > 1 or a.force_sp:
        from hunyuanvideo_efficiency_amd.inference import parallelize_transformer_module
        if a.ring_degree > 1:
            from hunyuanvideo_efficiency_amd.inference import set_sequence_parallel_groups
            assert world % a.ring_degree == 0
            set_sequence_parallel_groups(world // a.ring_degree, a.ring_degree)
            parallelize_transformer_module(model, None)
        else:
            parallelize_transformer_module(model, dist.group.WORLD)
    guidance = (torch.tensor([6.0], dtype=torch.float32, device=dev).to(torch.bfloat16) * 1000.0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(workload, steps, warmup):
        """W untimed + K timed denoise steps of `workload` on the model as it stands; returns (seconds for the K steps on this rank,
        [(ms, flop)] of the main-segment attention launches inside the timed region, final latents)."""
        lt, lh, lw = WORKLOADS[workload]
        x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, (lt, lh, lw), s_txt, 11, seed=42, device=dev)
        ts = ts.to(torch.bfloat16)
        cos, sin = get_nd_rotary_pos_embed(cfg.rope_dim_list, [lt, lh // 2, lw // 2], theta=256, use_real=True, device=dev)
        n_total = warmup + steps
        sched = FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
        sched.set_timesteps(max(n_total, 50), device=dev)
        lat = x.clone()

        def one_step(i, lat):
            t = sched.timesteps[i]
            v = model(lat, t.repeat(1), text_states=ts, text_mask=tm, text_states_2=ts2, freqs_cos=cos, freqs_sin=sin,
                      guidance=guidance, return_dict=True)["x"]
            return sched.step(v, t, lat, return_dict=False)[0]

        with torch.no_grad():
            for i in range(warmup):
                lat = one_step(i, lat)
            barrier()
            ops.PROFILE_ATTN = []          # HIP events around every main-segment attention launch, on the launch stream
            t0 = time.perf_counter()
            for i in range(warmup, n_total):
                lat = one_step(i, lat)
            barrier()
            el = time.perf_counter() - t0
        prof, ops.PROFILE_ATTN = ops.PROFILE_ATTN, None
        att = [(e0.elapsed_time(e1), 4.0 * nq * nkv * 128 * nh) for e0, e1, nq, nkv, nh in prof if nkv > 1024 or workload == "tiny"]
        assert bool(torch.isfinite(lat).all()), f"non-finite latents ({workload})"
        return el, att, lat

    elapsed, att, lat = run_leg(a.workload, a.steps, a.warmup)
    att_ms, att_flop = [m for m, _ in att], [f for _, f in att]

    # Secondary workloads (BASELINE.json configs 2, 5 and 4) on the same model, AFTER the timed headline leg and never part of
    # `value`: a few steps each inside a stated wall-clock budget (what does not fit is skipped and says so).
    secondary = []
    if world == 1 and a.workload == "720p129f" and not a.no_secondary and not a.use_fp8 and not a.force_sp:
        budget_s, t_sec0 = a.secondary_budget, time.perf_counter()
        ms129 = elapsed / a.steps * 1e3

        def leg_entry(workload, steps, warmup, dtype, note=None):
            el, at, _ = run_leg(workload, steps, warmup)
            lt, lh, lw = WORKLOADS[workload]
            si = lt * (lh // 2) * (lw // 2)
            fs = step_flops(si, s_txt, cfg.hidden_size, cfg.mm_double_blocks_depth, cfg.mm_single_blocks_depth)
            ms = el / steps * 1e3
            am = sum(m for m, _ in at) / max(len(at), 1)
            af = sum(f for _, f in at) / max(len(at), 1)
            tf = af / (am * 1e-3) / 1e12 if at else 0.0
            e = {"workload": workload, "steps": steps, "warmup": warmup, "ms_per_step": ms, "dtype": dtype,
                 "step_pflop": fs / 1e15, "step_mfma_frac": fs / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
                 "roofline": {"kernel": "attn_fwd_kernel (main segment)", "bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS,
                              "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "avg_launch_ms": am, "launches": len(at)}}
            if note:
                e["note"] = note
            return e

        # estimated cost of each leg from the headline leg's measured step time (FLOP-scaled), checked against what is left
        plan = [("544p65f", 3, 1, "bf16", 4 * ms129 * 1.375 / 12.07 / 1e3 + 2, None),
                ("720p257f", 1, 0, "bf16", ms129 * 43.64 / 12.07 / 1e3 + 3,
                 "one timed step, no separate warm-up step (kernels and weights are warm from the previous legs; the S = 234,256 "
                 "workspace is allocated before the clock starts)")]
        for wl, st, wu, dt, est, note in plan:
            left = budget_s - (time.perf_counter() - t_sec0)
            if est > left:
                secondary.append({"workload": wl, "skipped": f"estimated {est:.0f} s > {left:.0f} s left of the {budget_s:.0f} s budget"})
                continue
            if wl == "720p257f":
                model._workspace(65 * 45 * 80, s_txt, dev)
            secondary.append(leg_entry(wl, st, wu, dt, note))
        model._ws = None
        est = 3 * ms129 / 1e3 + 15
        left = budget_s - (time.perf_counter() - t_sec0)
        if est > left:
            secondary.append({"workload": "720p129f fp8-mfma", "skipped": f"estimated {est:.0f} s > {left:.0f} s left of the {budget_s:.0f} s budget"})
        else:
            from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear, enable_fp8_mfma
            convert_fp8_linear(model, None, torch.bfloat16)
            enable_fp8_mfma(model)
            e = leg_entry("720p129f", 2, 1, "bf16+fp8(e4m3) linears",
                          "BASELINE.json configs[3] on the CDNA4 FP8 matrix cores (--use-fp8 --fp8-mfma): e4m3 weights, activations "
                          "quantised per token, bf16 attention.  Tested tolerance (tests/test_gpu_fp8_mfma.py): GEMM == the "
                          "quantisation-aware oracle to 1 bf16 ulp (rtol 2^-7, atol 2e-2), quantisers bit-exact, tiny-model output "
                          "within 3e-2 of range of that oracle and within 1e-1 of range of the reference's weight-only FP8 semantics")
            e["workload"] = "720p129f fp8-mfma"
            secondary.append(e)
        model._ws = None
        torch.cuda.empty_cache()
    # end-to-end leg (reported beside `value`, never part of it): VAE tiled decode of this video's latents on rank 0's GPU
    vae_s = None
    if not tiny and not a.no_vae:
        # N > 1: the 84 tiles are shared out over the ranks (AutoencoderKLCausal3D.enable_tile_parallel), all ranks take part
        from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
        vae = AutoencoderKLCausal3D(device=dev)
        with torch.no_grad():
            for k, p in vae.state_dict().items():
                p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))
        vae.enable_tiling()
        z = syn.hashed_uniform((1, 16, T, H, W), "vae.z", 0, dev) * 1.7
        vae.decode(z[:, :, :5, :32, :32], return_dict=False)           # warm-up (weight re-layout, kernel load)
        if world > 1:
            vae.enable_tile_parallel()
        barrier()
        t0 = time.perf_counter()
        img = vae.decode(z, return_dict=False)[0]
        barrier()
        vae_s = time.perf_counter() - t0
        assert bool(torch.isfinite(img).all())
        del vae, img, z
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # dominant kernel: flash attention (main segment launches only: n_kv > 1024) - att_ms / att_flop from the headline leg
    if rank == 0:
        f_step = step_flops(s_img, s_txt, cfg.hidden_size, cfg.mm_double_blocks_depth, cfg.mm_single_blocks_depth)
        ms_per_step = elapsed / a.steps * 1e3
        avg_ms = sum(att_ms) / max(len(att_ms), 1)
        avg_flop = sum(att_flop) / max(len(att_flop), 1)
        achieved = avg_flop / (avg_ms * 1e-3) / 1e12 if att_ms else 0.0
        # HBM-side bytes per launch: NOT collected by this run - cited from the committed rocprofv3 PMC passes of the same kernel
        # and shape (separate --pmc passes; FETCH_SIZE x2 gfx950 correction), newest round first; `traffic_source` names the file
        traffic, traffic_note, traffic_source = None, None, None
        if world == 1 and a.workload == "720p129f":
            for rnd in ("r03", "r02", "r01"):
                tf = os.path.join(ROOT, "profiles", rnd, "attn_traffic.json")
                if os.path.exists(tf):
                    tj = json.load(open(tf))
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_source = f"profiles/{rnd}/attn_traffic.json"
                    traffic_note = (f"cited, not live: bytes/launch of kernel {tj.get('kernel', 'attn_fwd_kernel_v2')} from {traffic_source} "
                                    f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes); algorithmic "
                                    f"{tj['algorithmic_bytes_per_launch']:.3e} B; the counter includes Infinity-Cache hits "
                                    "(per-XCD K/V re-streams)")
                    break
        out = {
            "metric": ("denoise-steps/sec (" + {"720p129f": "720x1280x129f", "544p65f": "544x960x65f", "720p257f": "720x1280x257f"}.get(a.workload, a.workload)
                       + ", HunyuanVideo DiT 20+40 blocks, " + (("fp8 e4m3 MFMA linears (per-token activation scales), bf16 attention, fp8 e4m3 weights)" if a.fp8_mfma else "bf16 compute, fp8 e4m3 weights)") if a.use_fp8 else "bf16)")) if not tiny else "denoise-steps/sec (tiny)",
            "value": a.steps / elapsed, "unit": "denoise-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16+fp8(e4m3) linears" if a.fp8_mfma else "bf16", "data": "synthetic (hash-generated latents/text embeddings, random-init weights)",
            "config": {"workload": f"{a.workload}: latent 16x{T}x{H}x{W}, S_img={s_img}, S_txt={s_txt} (11 valid), "
                                   f"d={cfg.hidden_size}, heads={cfg.heads_num}, {cfg.mm_double_blocks_depth}+{cfg.mm_single_blocks_depth} blocks",
                       "parallelism": ("single GPU" + (" (SP code path forced on a 1-rank group)" if a.force_sp else "")) if world == 1 else (f"ulysses{world} (token-axis shard, RCCL all-to-all)" if a.ring_degree == 1 else f"ulysses{world // a.ring_degree} x ring{a.ring_degree} (RCCL all-to-all + point-to-point K/V ring)")},
            "sec_per_video_50steps_denoise_only": 50 * ms_per_step / 1e3,
            "vae_tiled_decode_s": vae_s,
            "sec_per_video_50steps_plus_vae_decode": (50 * ms_per_step / 1e3 + vae_s) if vae_s is not None else None,
            "step_pflop": f_step / 1e15,
            "step_mfma_frac": f_step / (ms_per_step * 1e-3) / (world * PEAK_BF16_TFLOPS * 1e12),
            "roofline": {"kernel": "attn_fwd_kernel (hv_attn_fwd_bf16, main segment)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note,
                         "launches": len(att_ms), "avg_launch_ms": avg_ms, "flop_per_launch": avg_flop,
                         "peak_note": "peak = dense bf16 MFMA at full clock, as the contract says.  Cited, not live (profiles/r03/mfma_power_roof.txt, "
                                      "attn_power_clock.txt): under the 1400 W cap a BARE v_mfma_f32_32x32x16_bf16 stream sustains 1780-1850 TFLOP/s on "
                                      "random bf16 operands (2470 on zeros); this kernel's own instruction stream runs at 2256 TFLOP/s on all-zero inputs"},
        }
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(f_step)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1 or a.force_sp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
