"""CPU: the reference's on-disk checkpoint formats round-trip into the modules.  The files are written by
this test (torch.save of plain tensor dicts = what the reference's checkpoints contain); they are read back only through
torch.load(weights_only=True) / safetensors.  Path-resolution rules follow hyvideo/inference.py:279-354, the VAE wrapper rules
hyvideo/vae/__init__.py:94-102, the FP8 map fp8_optimization.py:85-100."""
import json
import types

import pytest
import torch

from hunyuanvideo_efficiency_amd import checkpoint as ck
from hunyuanvideo_efficiency_amd import synthetic as syn


def _tiny_model():
    from hunyuanvideo_efficiency_amd.modules.models import HYVideoDiffusionTransformer
    cfg = syn.tiny_config()
    args = types.SimpleNamespace(text_states_dim=cfg.text_states_dim, text_states_dim_2=cfg.text_states_dim_2)
    m = HYVideoDiffusionTransformer(args, in_channels=16, out_channels=16, hidden_size=cfg.hidden_size, heads_num=cfg.heads_num,
                                    mm_double_blocks_depth=1, mm_single_blocks_depth=1, guidance_embed=True, dtype=torch.bfloat16)
    sd = {k: v.to(torch.bfloat16) for k, v in syn.synth_dit_state_dict(cfg, seed=5).items()}
    return m, sd


def _same(model, sd):
    got = model.state_dict()
    return set(got) == set(sd) and all(torch.equal(got[k].float(), sd[k].float()) for k in sd)


def test_dit_file_wrapped_bare_and_directory_layouts(tmp_path):
    model, sd = _tiny_model()
    # (1) single file, deepspeed-style wrapper, load_key "module"
    f = tmp_path / "mp_rank_00_model_states.pt"
    torch.save({"module": sd, "ema": {k: v * 0 for k, v in sd.items()}}, f)
    a = types.SimpleNamespace(dit_weight=str(f), load_key="module", model_resolution="540p")
    assert _same(ck.load_state_dict(a, model), sd)
    a.load_key = "ema"
    ck.load_state_dict(a, model)
    assert float(model.state_dict()["img_in.proj.weight"].abs().sum()) == 0.0
    a.load_key = "nope"
    with pytest.raises(KeyError):
        ck.load_state_dict(a, model)
    # (2) single bare file
    g = tmp_path / "bare.pt"
    torch.save(sd, g)
    assert _same(ck.load_state_dict(types.SimpleNamespace(dit_weight=str(g), load_key="module"), model), sd)
    # (3) directory with *_model_states.pt
    d = tmp_path / "dsdir"
    d.mkdir()
    torch.save({"module": sd}, d / "mp_rank_00_model_states.pt")
    assert ck.resolve_dit_weight(str(d))[1] is False
    assert _same(ck.load_state_dict(types.SimpleNamespace(dit_weight=str(d), load_key="module"), model), sd)
    # (4) directory with pytorch_model_<key>.pt (bare)
    e = tmp_path / "baredir"
    e.mkdir()
    torch.save(sd, e / "pytorch_model_module.pt")
    p, bare = ck.resolve_dit_weight(str(e), load_key="module")
    assert bare is True and p.name == "pytorch_model_module.pt"
    assert _same(ck.load_state_dict(types.SimpleNamespace(dit_weight=str(e), load_key="module"), model), sd)
    # (5) pretrained_model_path / t2v_<resolution> when --dit-weight is not given
    r = tmp_path / "ckpts" / "t2v_720p"
    r.mkdir(parents=True)
    torch.save({"module": sd}, r / "x_model_states.pt")
    a = types.SimpleNamespace(dit_weight=None, load_key="module", model_resolution="720p")
    assert _same(ck.load_state_dict(a, model, tmp_path / "ckpts"), sd)
    # errors
    with pytest.raises(ValueError):
        ck.resolve_dit_weight(str(tmp_path / "missing"))
    h = tmp_path / "empty"
    h.mkdir()
    with pytest.raises(ValueError):
        ck.resolve_dit_weight(str(h))
    (h / "weights.pt").write_bytes(b"")
    with pytest.raises(ValueError):
        ck.resolve_dit_weight(str(h))
    # strict=True: a missing key must raise
    bad = dict(sd)
    bad.pop("img_in.proj.weight")
    torch.save(bad, g)
    with pytest.raises(RuntimeError):
        ck.load_state_dict(types.SimpleNamespace(dit_weight=str(g), load_key="module"), model)


def test_safetensors_and_refused_pickle(tmp_path):
    from safetensors.torch import save_file
    model, sd = _tiny_model()
    f = tmp_path / "model.safetensors"
    save_file({k: v.contiguous() for k, v in sd.items()}, str(f))
    assert _same(ck.load_state_dict(types.SimpleNamespace(dit_weight=str(f), load_key="module"), model), sd)
    # a pickle with a non-tensor object is refused, never executed
    import pickle

    class Boom:
        def __reduce__(self):
            return (print, ("executed!",))
    g = tmp_path / "evil.pt"
    with open(g, "wb") as fh:
        pickle.dump({"module": Boom()}, fh)
    with pytest.raises(ck.CheckpointFormatError):
        ck.read_tensors(g)


def test_vae_checkpoint_wrappers_and_config(tmp_path):
    from hunyuanvideo_efficiency_amd.vae import load_vae
    boc = (32, 64, 128, 128)
    sd = {k: v.to(torch.float16) for k, v in syn.synth_vae_state_dict(boc, seed=2).items()}
    # a real checkpoint carries the encode half too: load_vae then builds it (with_encoder=None -> detected from the keys)
    extra = {k: v.to(torch.float16) for k, v in syn.synth_vae_state_dict(boc, seed=2, encoder=True).items() if k not in sd}
    for wrap in (lambda s: s, lambda s: {"state_dict": s}, lambda s: {"state_dict": {"vae." + k: v for k, v in s.items()}}):
        d = tmp_path / f"vae{id(wrap)}"
        d.mkdir()
        torch.save(wrap({**sd, **extra}), d / "pytorch_model.pt")
        json.dump({"block_out_channels": list(boc), "sample_size": 128, "sample_tsize": 16, "scaling_factor": 0.5,
                   "_class_name": "AutoencoderKLCausal3D"}, open(d / "config.json", "w"))
        vae, path, sr, tr = load_vae("884-16c-hy", "fp16", vae_path=str(d), device="cpu")
        assert (sr, tr) == (8, 4) and vae.config.scaling_factor == 0.5 and vae.config.block_out_channels == boc
        assert (vae.tile_latent_min_size, vae.tile_latent_min_tsize) == (16, 4)
        got = vae.state_dict()
        assert vae.with_encoder and set(got) == set(sd) | set(extra) and all(torch.equal(got[k], {**sd, **extra}[k]) for k in got)
        # decode-only build of the same checkpoint: the encode half's keys are ignored
        vae2 = load_vae("884-16c-hy", "fp16", vae_path=str(d), device="cpu", with_encoder=False)[0]
        assert set(vae2.state_dict()) == set(sd)
    with pytest.raises(ValueError):
        ck.read_vae_checkpoint(tmp_path / "nowhere")


def test_fp8_map_file(tmp_path):
    from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear
    from hunyuanvideo_efficiency_amd.modules.layers import ParamLinear
    model, sd = _tiny_model()
    model.load_state_dict(sd, strict=True)
    keys = [k for k, m in model.named_modules() if isinstance(m, ParamLinear) and ("double_blocks" in k or "single_blocks" in k)]
    f = tmp_path / "dit_fp8.pt"
    torch.save(sd, f)
    assert ck.fp8_map_path(f).endswith("dit_fp8_map.pt")
    with pytest.raises(ValueError):
        convert_fp8_linear(model, str(f), torch.bfloat16)          # map file missing (fp8_optimization.py:87-90)
    torch.save({k: torch.tensor(0.01 * (i + 1)) for i, k in enumerate(keys)}, ck.fp8_map_path(f))
    n = convert_fp8_linear(model, str(f), torch.bfloat16)
    assert n == len(keys) == 13
    mods = dict(model.named_modules())
    for i, k in enumerate(keys):
        assert mods[k].weight.dtype == torch.float8_e4m3fn
        assert abs(float(mods[k].fp8_scale) - 0.01 * (i + 1)) < 1e-3 * (i + 1)


def test_vae_tile_parallel_plan_720p():
    """Host logic of the tile-parallel VAE decode: the 84 tiles of a 720p x 129f latent are each assigned to exactly
    one rank, identically on every rank, with a balanced load (longest-processing-time-first over the tile sizes)."""
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
    vae = AutoencoderKLCausal3D(block_out_channels=(32, 64, 128, 128), device="cpu")
    vae.enable_tiling()
    z4 = torch.zeros(16, 33, 90, 160)
    views = list(vae._tile_views(z4))
    assert len(views) == 84
    costs = [v.shape[1] * v.shape[2] * v.shape[3] for v in views]
    for world in (2, 3, 4, 8):
        plan = vae._assign_tiles(costs, world)
        assert sorted(k for p in plan for k in p) == list(range(84))
        loads = [sum(costs[k] for k in p) for p in plan]
        assert max(loads) <= 1.08 * (sum(costs) / world), (world, loads)
        assert plan == vae._assign_tiles(costs, world)
