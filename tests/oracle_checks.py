"""Checker: one tiny denoise step on the GPU against the oracle - used by __graft_entry__.smoke() and tests/test_gpu_model.py.
Test infrastructure (this file lives under tests/ because it imports oracle/; the package itself never does)."""
from __future__ import annotations

import torch

from hunyuanvideo_efficiency_amd import synthetic as syn
from hunyuanvideo_efficiency_amd.builders import build_model


def tiny_step_vs_oracle(device="cuda:0", latent_thw=(5, 16, 16), txt_len=32, n_valid=11) -> float:
    from oracle import dit_ref as R
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    cfg = syn.tiny_config()
    model = build_model(cfg, device)
    x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, latent_thw, txt_len, n_valid, seed=0)
    T, H, W = latent_thw
    cos, sin = get_nd_rotary_pos_embed(cfg.rope_dim_list, [T, H // 2, W // 2], theta=256, use_real=True)
    sched = FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
    sched.set_timesteps(3, device=device)
    t = sched.timesteps[0]
    g = (torch.tensor([6.0], dtype=torch.float32, device=device).to(torch.bfloat16) * 1000.0)
    lat = x.to(device)
    with torch.no_grad():
        v = model(lat, t.repeat(1), text_states=ts.to(device), text_mask=tm.to(device), text_states_2=ts2.to(device),
                  freqs_cos=cos.to(device), freqs_sin=sin.to(device), guidance=g, return_dict=True)["x"]
        new = sched.step(v, t, lat, return_dict=False)[0]
    torch.cuda.synchronize()
    # oracle, bf16-emulated contract, same weights (bf16-rounded) and inputs
    E = R.Prec(True)
    sd = {k: p.float().cpu() for k, p in model.state_dict().items()}
    rc, rs = R.rope_tables([T, H // 2, W // 2], cfg.rope_dim_list, 256.0)
    sig = R.flow_sigmas(3, 7.0)
    tsb = E.r(ts)  # text states enter the GPU path as bf16
    v_ref = R.dit_forward(sd, cfg, x, R.flow_timesteps(sig)[0:1], tsb, tm, ts2, rc, rs, g.float().cpu(), E)
    new_ref = R.euler_step(x, v_ref, sig, 0)
    err_v = float((v.float().cpu() - v_ref).abs().max() / v_ref.abs().max())
    err_x = float((new.float().cpu() - new_ref).abs().max() / new_ref.abs().max())
    if not (err_v < 3e-2 and err_x < 3e-2):
        raise AssertionError(f"tiny denoise step differs from the oracle: noise_pred {err_v:.3e}, latents {err_x:.3e}")
    return max(err_v, err_x)


def fullwidth_blocks(device="cuda:0", prec=None, s_img=4096, s_txt=256, n_valid=11, grid=(4, 32, 32), oracle_out=None):
    """One MMDoubleStreamBlock + one MMSingleStreamBlock at the SHIPPED width (d = 3072, 24 heads, mlp 12288) on S = s_img + s_txt
    tokens: the GPU blocks through their reference call surfaces and the oracle (precision `prec`, default the bf16-emulated
    contract) on the same bf16-rounded weights and inputs.  Returns {name: (gpu fp32 cpu tensor, oracle tensor)} for
    double_img / double_txt / single.  `oracle_out`: (io, to, so) already computed by the caller with the SAME weights and inputs
    (bench.py's cpu_baseline leg times exactly that computation) - then only the GPU side runs here."""
    from oracle import dit_ref as R
    prec = prec or R.Prec(True)
    cfg = syn.DiTConfig(mm_double_blocks_depth=1, mm_single_blocks_depth=1)
    model = build_model(cfg, device)
    img, txt, vec = fullwidth_block_inputs(cfg, s_img, s_txt)
    cos, sin = R.rope_tables(list(grid), cfg.rope_dim_list, 256.0)
    cu = torch.tensor([0, s_img + n_valid, s_img + s_txt], dtype=torch.int32)
    bf = lambda t: t.to(device).to(torch.bfloat16)
    S = s_img + s_txt
    with torch.no_grad():
        io, to = model.double_blocks[0](bf(img), bf(txt), bf(vec), cu, cu, S, S, (cos.to(device), sin.to(device)))
        io, to = io.float().cpu(), to.float().cpu()
        so = model.single_blocks[0](bf(torch.cat([img, txt], 1)), bf(vec), s_txt, cu, cu, S, S, (cos.to(device), sin.to(device)))
        so = so.float().cpu()
    torch.cuda.synchronize()
    if oracle_out is None:
        sd = {k: p.float().cpu() for k, p in model.state_dict().items()
              if k.startswith("double_blocks.0.") or k.startswith("single_blocks.0.")}
        rio, rto = R.double_block(sd, "double_blocks.0.", img, txt, vec, cu, cos, sin, cfg.heads_num, prec)
        rso = R.single_block(sd, "single_blocks.0.", torch.cat([img, txt], 1), vec, s_txt, cu, cos, sin, cfg.heads_num, prec)
    else:
        rio, rto, rso = oracle_out
    del model
    return {"double_img": (io, rio), "double_txt": (to, rto), "single": (so, rso)}


def fullwidth_block_inputs(cfg, s_img=4096, s_txt=256):
    """bf16-representable inputs of fullwidth_blocks (and of bench.py's cpu_baseline sample): residual streams of range ~ +-1.7"""
    r = lambda t: t.to(torch.bfloat16).float()
    d = cfg.hidden_size
    img = r(syn.hashed_uniform((1, s_img, d), "cpu.img", 0) * 1.7)
    txt = r(syn.hashed_uniform((1, s_txt, d), "cpu.txt", 0) * 1.7)
    vec = r(syn.hashed_uniform((1, d), "cpu.vec", 0) * 0.5)
    return img, txt, vec
