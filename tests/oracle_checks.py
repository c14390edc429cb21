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
