"""GPU: reference-layout checkpoints loaded through hunyuanvideo_efficiency_amd.checkpoint drive the HIP path (SURVEY.md 8f row
2; hyvideo/inference.py:279-354 for the DiT, fp8_optimization.py:85-100 for `<ckpt>_map.pt`, vae/__init__.py:94-102 for the
VAE).  No real checkpoint exists in this environment (ckpts/ in the reference holds no weights), so the files are written here
in the reference's layouts from the deterministic synthetic weights; what is checked is that a model filled ONLY from disk
computes exactly what the directly-built model computes, that the FP8 checkpoint (e4m3 weights + per-layer scales in the map
file) reproduces the oracle's weight-only semantics, and that the text-refiner prefix cache does not change a single bit over
consecutive steps and prompts."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import checkpoint as ck, synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

DEV = "cuda"
E = R.Prec(True)


def _inputs(cfg, thw=(5, 16, 16), n_valid=11, seed=4):
    from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    x, ts, tm, ts2 = syn.synth_dit_inputs(cfg, thw, 32, n_valid, seed=seed)
    T, H, W = thw
    cos, sin = get_nd_rotary_pos_embed(cfg.rope_dim_list, [T, H // 2, W // 2], theta=256, use_real=True)
    kw = dict(text_states=ts.to(torch.bfloat16).to(DEV), text_mask=tm.to(DEV), text_states_2=ts2.to(DEV), freqs_cos=cos.to(DEV),
              freqs_sin=sin.to(DEV), guidance=torch.tensor([6016.0], device=DEV), return_dict=True)
    return x.to(DEV), kw, (x, ts, tm, ts2, cos, sin)


def _empty_model(cfg):
    """the module tree with uninitialised parameters - everything it computes with must come from the checkpoint"""
    from hunyuanvideo_efficiency_amd.modules.models import HYVideoDiffusionTransformer
    args = types.SimpleNamespace(text_states_dim=cfg.text_states_dim, text_states_dim_2=cfg.text_states_dim_2)
    m = HYVideoDiffusionTransformer(args, in_channels=cfg.in_channels, out_channels=cfg.out_channels, hidden_size=cfg.hidden_size,
                                    heads_num=cfg.heads_num, mm_double_blocks_depth=cfg.mm_double_blocks_depth,
                                    mm_single_blocks_depth=cfg.mm_single_blocks_depth, rope_dim_list=cfg.rope_dim_list,
                                    guidance_embed=cfg.guidance_embed, dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(float("nan"))
    return m.eval()


def test_dit_checkpoint_drives_the_hip_path(tmp_path):
    from hunyuanvideo_efficiency_amd.builders import build_model
    cfg = syn.DiTConfig(hidden_size=512, heads_num=4, mm_double_blocks_depth=1, mm_single_blocks_depth=2)
    ref_model = build_model(cfg, DEV, seed=3)
    sd = {k: v.detach().cpu() for k, v in ref_model.state_dict().items()}
    x, kw, _ = _inputs(cfg)
    t = torch.tensor([997.093], device=DEV)
    with torch.no_grad():
        want = ref_model(x, t, **kw)["x"].clone()
    # deepspeed-style directory (inference.py:300-318): <dir>/mp_rank_00_model_states.pt with the state dict under "module"
    d = tmp_path / "t2v_720p" / "transformers"
    d.mkdir(parents=True)
    torch.save({"module": sd, "ema": {k: torch.zeros_like(v) for k, v in sd.items()}}, d / "mp_rank_00_model_states.pt")
    model = _empty_model(cfg)
    ck.load_state_dict(types.SimpleNamespace(dit_weight=str(d), load_key="module", model_resolution="720p"), model)
    with torch.no_grad():
        got = model(x, t, **kw)["x"]
    assert torch.equal(got, want)
    # text-refiner prefix cache: second step with the SAME text tensors (hit) and a new prompt (miss) stay bit-identical to a fresh model
    with torch.no_grad():
        t2 = torch.tensor([950.0], device=DEV)
        again = model(x, t2, **kw)["x"].clone()
        assert getattr(kw["text_states"], "_hv_txt_cache", None) is not None and "emb" in kw["text_states"]._hv_txt_cache["rows"][0]
        fresh = build_model(cfg, DEV, seed=3)
        assert torch.equal(again, fresh(x, t2, **kw)["x"])
        x3, kw3, _ = _inputs(cfg, n_valid=5, seed=9)
        assert torch.equal(model(x3, t2, **kw3)["x"], fresh(x3, t2, **kw3)["x"])


def test_fp8_checkpoint_with_scale_map(tmp_path):
    """An FP8 checkpoint as the reference ships it: e4m3fn weights for the block Linears inside the .pt and `<name>_map.pt`
    holding each layer's scale (fp8_optimization.py:85-100).  convert_fp8_linear(model, path) + load -> forward == the oracle with
    the dequantised weights (weight-only FP8, the reference's arithmetic); enable_fp8_mfma on top stays within the FP8-MFMA bound."""
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear, enable_fp8_mfma, quantize_weight
    from hunyuanvideo_efficiency_amd.modules.layers import ParamLinear
    cfg = syn.DiTConfig(hidden_size=512, heads_num=4, mm_double_blocks_depth=1, mm_single_blocks_depth=2)
    src = build_model(cfg, DEV, seed=6)
    sd, fmap, sd_deq = {}, {}, {}
    for name, mod in src.named_modules():
        if isinstance(mod, ParamLinear) and ("double_blocks" in name or "single_blocks" in name):
            w8, scale = quantize_weight(mod.weight.data)
            sd[name + ".weight"] = w8.cpu()
            fmap[name] = scale.reshape(()).float().cpu()
    for k, v in src.state_dict().items():
        sd.setdefault(k, v.detach().cpu())
    f = tmp_path / "hunyuan_video_720_fp8.pt"
    torch.save({"module": sd}, f)
    torch.save(fmap, ck.fp8_map_path(f))
    model = _empty_model(cfg)
    assert convert_fp8_linear(model, str(f), torch.bfloat16) == len(fmap)
    ck.load_state_dict(types.SimpleNamespace(dit_weight=str(f), load_key="module"), model)
    x, kw, (xc, ts, tm, ts2, cos, sin) = _inputs(cfg)
    t = torch.tensor([997.093], device=DEV)
    with torch.no_grad():
        out = model(x, t, **kw)["x"].float().cpu()
    for k, v in sd.items():
        sd_deq[k] = v.float()
    for name, scale in fmap.items():
        sd_deq[name + ".weight"] = (sd[name + ".weight"].to(torch.bfloat16) * scale.to(torch.bfloat16)).float()
    ref = R.dit_forward(sd_deq, cfg, xc, t.cpu(), E.r(ts), tm, ts2, cos, sin, torch.tensor([6016.0]), E)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(out, ref) < 3e-2, rel(out, ref)
    assert enable_fp8_mfma(model) == 3
    with torch.no_grad():
        out8 = model(x, t, **kw)["x"].float().cpu()
    assert rel(out8, ref) < 1e-1, rel(out8, ref)


def test_vae_checkpoint_drives_the_hip_path(tmp_path):
    import json
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D, load_vae
    boc = (32, 64, 128, 128)
    sd = {k: v.to(torch.float16) for k, v in syn.synth_vae_state_dict(boc, seed=0).items()}
    d = tmp_path / "hunyuan-video-t2v-720p" / "vae"
    d.mkdir(parents=True)
    torch.save({"state_dict": {"vae." + k: v for k, v in sd.items()}}, d / "pytorch_model.pt")       # both wrappers at once
    json.dump({"block_out_channels": list(boc), "latent_channels": 16, "in_channels": 3, "out_channels": 3, "layers_per_block": 2,
               "sample_size": 256, "sample_tsize": 64, "scaling_factor": 0.476986, "time_compression_ratio": 4,
               "spatial_compression_ratio": 8, "mid_block_add_attention": True, "norm_num_groups": 32, "act_fn": "silu"},
              open(d / "config.json", "w"))
    vae, _, s_ratio, t_ratio = load_vae("884-16c-hy", "fp16", vae_path=str(d), device=DEV)
    assert (s_ratio, t_ratio) == (8, 4)
    direct = AutoencoderKLCausal3D(block_out_channels=boc, device=DEV)
    direct.load_state_dict(sd, strict=True)
    z = (syn.hashed_uniform((1, 16, 3, 12, 12), "ck.z", 0) * 1.7).to(DEV)
    assert torch.equal(vae.decode(z, return_dict=False)[0], direct.decode(z, return_dict=False)[0])
