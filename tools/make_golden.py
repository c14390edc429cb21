#!/usr/bin/env python3
"""Generate tests/golden/dit_*.npz by IMPORTING the reference (/root/reference) in the build
container and running it on CPU in fp32 (no autocast).  The reference never travels to the GPU
box; only these small input/output vectors do.

How the import works: the reference's hyvideo.modules needs only base
classes from `diffusers` (ModelMixin / ConfigMixin / register_to_config), which is not installed
here; three in-memory modules supply bare base classes before the import.  Two CUDA-only spots of
the imported module object are substituted: get_cu_seqlens (device="cuda" literal,
attenion.py:48) and attention(mode="flash") (flash-attn is absent; emulated with per-segment
SDPA over cu_seqlens).  Every other arithmetic operation is executed by reference code.

Weights/inputs come from hunyuanvideo_efficiency_amd.synthetic (a pure function of key names), and
are loaded into the reference model with load_state_dict(strict=True), which pins the key/shape
table.  Run:  python tools/make_golden.py
"""
import argparse
import contextlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402


def install_shims():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    class ModelMixin(nn.Module):
        pass

    class ConfigMixin:
        pass

    def register_to_config(fn):
        import functools
        import inspect

        @functools.wraps(fn)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(fn)
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
            self.config = types.SimpleNamespace(**cfg)
            return fn(self, *a, **kw)
        return wrapper

    class BaseOutput(dict):
        pass

    class SchedulerMixin:
        pass

    d = types.ModuleType("diffusers")
    dm = types.ModuleType("diffusers.models")
    dm.ModelMixin = ModelMixin
    dc = types.ModuleType("diffusers.configuration_utils")
    dc.ConfigMixin = ConfigMixin
    dc.register_to_config = register_to_config
    du = types.ModuleType("diffusers.utils")
    du.BaseOutput = BaseOutput
    du.logging = types.SimpleNamespace(get_logger=lambda name: types.SimpleNamespace(
        warning=lambda *a, **k: None, info=lambda *a, **k: None))
    ds = types.ModuleType("diffusers.schedulers")
    dsu = types.ModuleType("diffusers.schedulers.scheduling_utils")
    dsu.SchedulerMixin = SchedulerMixin
    for name, mod in [("diffusers", d), ("diffusers.models", dm), ("diffusers.configuration_utils", dc),
                      ("diffusers.utils", du), ("diffusers.schedulers", ds),
                      ("diffusers.schedulers.scheduling_utils", dsu)]:
        sys.modules[name] = mod


@contextlib.contextmanager
def zeros_without_cuda():
    """get_cu_seqlens (attenion.py:48) hard-codes device="cuda"; drop that kwarg for the call."""
    orig = torch.zeros

    def z(*a, **kw):
        kw.pop("device", None)
        return orig(*a, **kw)
    torch.zeros = z
    try:
        yield
    finally:
        torch.zeros = orig


def patch_cuda_only(models_mod, att_mod):
    orig_attention = att_mod.attention
    orig_cu = att_mod.get_cu_seqlens

    def get_cu_seqlens_cpu(text_mask, img_len):
        with zeros_without_cuda():
            return orig_cu(text_mask, img_len)

    def attention_cpu(q, k, v, mode="flash", drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None,
                      cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, batch_size=1):
        if mode != "flash":
            return orig_attention(q, k, v, mode=mode, drop_rate=drop_rate, attn_mask=attn_mask, causal=causal)
        b, s, h, d = q.shape
        qf, kf, vf = (t.reshape(b * s, h, d) for t in (q, k, v))
        out = torch.zeros_like(qf)
        for i in range(cu_seqlens_q.numel() - 1):
            lo, hi = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
            if hi > lo:
                o = torch.nn.functional.scaled_dot_product_attention(
                    qf[lo:hi].transpose(0, 1)[None], kf[lo:hi].transpose(0, 1)[None], vf[lo:hi].transpose(0, 1)[None])
                out[lo:hi] = o[0].transpose(0, 1)
        return out.reshape(b, s, h * d)

    models_mod.get_cu_seqlens = get_cu_seqlens_cpu
    models_mod.attention = attention_cpu
    return get_cu_seqlens_cpu


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.parse_args()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    install_shims()
    import hyvideo.modules.models as M
    import hyvideo.modules.attenion as A
    from hyvideo.modules.norm_layers import RMSNorm
    from hyvideo.modules.posemb_layers import apply_rotary_emb, get_nd_rotary_pos_embed, rotate_half
    from hyvideo.modules.embed_layers import timestep_embedding
    from hyvideo.modules.modulate_layers import modulate, apply_gate
    from hyvideo.modules.mlp_layers import MLP
    from hyvideo.modules.activation_layers import get_activation_layer
    from hyvideo.modules import fp8_optimization as F8
    get_cu = patch_cuda_only(M, A)

    U = syn.hashed_uniform
    s3 = 3.0 ** 0.5

    # ---------------- (i) leaves
    x = U((1, 40, 2, 128), "g.rms.x", 1) * s3
    w = 1 + 0.1 * U((128,), "g.rms.w", 1)
    rn = RMSNorm(128)
    rn.weight.copy_(w)
    save("dit_rmsnorm", x=x, w=w, y=rn(x))

    cos, sin = get_nd_rotary_pos_embed([16, 56, 56], [5, 8, 8], theta=256, use_real=True, theta_rescale_factor=1)
    q = U((1, 320, 1, 128), "g.rope.q", 1) * s3
    k = U((1, 320, 1, 128), "g.rope.k", 1) * s3
    qo, ko = apply_rotary_emb(q, k, (cos, sin), head_first=False)
    save("dit_rope", sizes=np.array([5, 8, 8]), cos=cos, sin=sin, q=q, k=k, qo=qo, ko=ko,
         rot8=rotate_half(torch.arange(8.0).reshape(1, 1, 1, 8)))
    cos2, sin2 = get_nd_rotary_pos_embed([16, 56, 56], [3, 6, 10], theta=256, use_real=True, theta_rescale_factor=1)
    save("dit_rope_tables2", sizes=np.array([3, 6, 10]), cos=cos2, sin=sin2)

    xm = U((1, 24, 256), "g.mod.x", 1) * s3
    sh, sc, gt = (0.3 * U((1, 256), f"g.mod.{n}", 1) for n in ("shift", "scale", "gate"))
    ln = nn.LayerNorm(256, elementwise_affine=False, eps=1e-6)
    save("dit_modulate", x=xm, shift=sh, scale=sc, gate=gt, ln_mod=modulate(ln(xm), shift=sh, scale=sc),
         gated=xm + apply_gate(ln(xm), gate=gt))

    tt = torch.tensor([1000.0, 997.0930, 500.25, 0.5, 6016.0])
    save("dit_timestep_embedding", t=tt, emb=timestep_embedding(tt, 256))

    mlp = MLP(256, 1024, act_layer=get_activation_layer("gelu_tanh"), bias=True)
    msd = {k_: syn.synth_param("g.mlp." + k_, tuple(v.shape), 1) for k_, v in mlp.state_dict().items()}
    mlp.load_state_dict(msd, strict=True)
    save("dit_mlp", x=xm, y=mlp(xm))  # weights: synth_param("g.mlp."+key, shape, seed=1)

    tm = torch.zeros(2, 32, dtype=torch.int64)
    tm[0, :11] = 1
    tm[1, :32] = 1
    save("dit_cu_seqlens", text_mask=tm, img_len=np.array(320), cu=get_cu(tm, 320))

    # ---------------- scheduler (reference class, loaded by file path: the package __init__ pulls the pipeline)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_sched", os.path.join(REF, "hyvideo/diffusion/schedulers/scheduling_flow_match_discrete.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    sched_out = {}
    for n in (30, 50):
        sch = S.FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
        sch.set_timesteps(n, device="cpu", n_tokens=1234)
        sched_out[f"sigmas{n}"] = sch.sigmas
        sched_out[f"timesteps{n}"] = sch.timesteps
    sch = S.FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
    sch.set_timesteps(50, device="cpu")
    lat = U((1, 16, 2, 4, 4), "g.sched.lat", 1).to(torch.float16)  # fp16 latents as prepare_latents makes them
    seq = [lat.float()]
    cur = lat
    for i in range(3):
        v = (U((1, 16, 2, 4, 4), f"g.sched.v{i}", 1)).to(torch.bfloat16)
        cur = sch.step(v, sch.timesteps[i], cur, return_dict=False)[0]
        seq.append(cur)
        sched_out[f"v{i}"] = v.float()
    sched_out["traj"] = torch.stack(seq)
    save("dit_scheduler", **sched_out)

    # ---------------- fp8 weight path (fp8_optimization.py:7-80)
    lin = nn.Linear(64, 64, bias=True)
    lw = syn.synth_param("g.fp8.weight", (64, 64), 1)
    lb = syn.synth_param("g.fp8.bias", (64,), 1)
    lin.weight.copy_(lw)
    lin.bias.copy_(lb)
    xin = U((3, 5, 64), "g.fp8.x", 1) * s3
    maxval = F8.get_fp_maxval()
    scale = torch.max(torch.abs(lw.flatten())) / maxval
    qdq, scale_b, _ = F8.fp8_tensor_quant(lw, scale)
    w8 = qdq.to(torch.float8_e4m3fn)
    lin.original_forward = lin.forward
    y_fly = F8.fp8_linear_forward(lin, torch.float32, xin)            # weight quantised on the fly
    # (the stored-e4m3fn branch calls cls.weight.sum() on an fp8 tensor, fp8_optimization.py:69, which
    #  torch-CPU does not implement - an ordinary NotImplementedError; its arithmetic is the same
    #  fp8_activation_dequant + F.linear exercised by the on-the-fly branch.)
    save("dit_fp8", w=lw, b=lb, x=xin, maxval=maxval, scale=scale, w8_bits=w8.view(torch.uint8),
         w_dequant=F8.fp8_activation_dequant(w8, scale, torch.float32), y_fly=y_fly)

    # ---------------- (ii)+(iii) tiny transformer (BASELINE.json configs[0])
    cfg = syn.tiny_config()
    args = types.SimpleNamespace(text_states_dim=cfg.text_states_dim, text_states_dim_2=cfg.text_states_dim_2)
    model = M.HYVideoDiffusionTransformer(
        args, in_channels=16, out_channels=16, hidden_size=cfg.hidden_size, heads_num=cfg.heads_num,
        mm_double_blocks_depth=1, mm_single_blocks_depth=1, rope_dim_list=[16, 56, 56], guidance_embed=True,
        dtype=torch.float32)
    sd = syn.synth_dit_state_dict(cfg, seed=0)
    missing = model.load_state_dict(sd, strict=True)
    print("reference load_state_dict(strict=True):", missing)
    model.eval()
    T, H, W = 5, 16, 16
    txt_len, n_valid = 32, 11
    xl, ts, tmask, ts2 = syn.synth_dit_inputs(cfg, (T, H, W), txt_len, n_valid, seed=0)
    cos, sin = get_nd_rotary_pos_embed([16, 56, 56], [T, H // 2, W // 2], theta=256, use_real=True,
                                       theta_rescale_factor=1)
    tstep = torch.tensor([997.0930], dtype=torch.float32)
    guid = torch.tensor([6016.0], dtype=torch.float32)

    taps = {}

    def hook(name):
        def f(mod, inp, out):
            taps[name] = out
        return f
    model.time_in.register_forward_hook(hook("time_in"))
    model.txt_in.register_forward_hook(hook("txt0"))
    model.img_in.register_forward_hook(hook("img0"))
    model.double_blocks[0].register_forward_hook(hook("double0"))
    model.single_blocks[0].register_forward_hook(hook("single0"))
    model.final_layer.register_forward_hook(hook("final"))
    out = model(xl, tstep, text_states=ts, text_mask=tmask, text_states_2=ts2, freqs_cos=cos, freqs_sin=sin,
                guidance=guid, return_dict=True)["x"]
    save("dit_tiny_forward", x=xl, t=tstep, text_states=ts, text_mask=tmask, text_states_2=ts2, guidance=guid,
         out=out, img0=taps["img0"], txt0=taps["txt0"], img_d=taps["double0"][0], txt_d=taps["double0"][1],
         x_s=taps["single0"], final=taps["final"], seed=np.array(0), latent_thw=np.array([T, H, W]))

    # standalone blocks with explicit inputs (n_valid < txt_len)
    vec = 0.5 * U((1, 256), "g.blk.vec", 1)
    img = U((1, 320, 256), "g.blk.img", 1) * s3
    txt = U((1, txt_len, 256), "g.blk.txt", 1) * s3
    cu = get_cu(tmask, 320)
    io, to = model.double_blocks[0](img, txt, vec, cu, cu, 320 + txt_len, 320 + txt_len, (cos, sin))
    xs = torch.cat([img, txt], 1)
    so = model.single_blocks[0](xs, vec, txt_len, cu, cu, 320 + txt_len, 320 + txt_len, (cos, sin))
    save("dit_blocks", vec=vec, img=img, txt=txt, cu=cu, img_out=io, txt_out=to, single_out=so)

    # one full denoise step + 2 more (loop order of pipeline_hunyuan_video.py:961-1023)
    sch = S.FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
    sch.set_timesteps(3, device="cpu")
    lat = xl.clone()
    preds = []
    for i, t in enumerate(sch.timesteps):
        t_expand = t.repeat(lat.shape[0])
        npred = model(lat, t_expand, text_states=ts, text_mask=tmask, text_states_2=ts2, freqs_cos=cos,
                      freqs_sin=sin, guidance=guid, return_dict=True)["x"]
        preds.append(npred)
        lat = sch.step(npred, t, lat, return_dict=False)[0]
    save("dit_tiny_denoise3", preds=torch.stack(preds), final_latents=lat, timesteps=sch.timesteps, sigmas=sch.sigmas)


if __name__ == "__main__":
    main()
