"""GPU parity of the whole hot path through the reference's top call surface (row a16): HunyuanVideoPipeline.__call__
= set_timesteps -> [transformer -> scheduler.step] x n -> / scaling_factor -> tiled VAE decode -> (x/2+0.5).clamp(0,1),
against the oracle chain denoise_loop (bf16-emulated) -> decode (fp16-emulated) -> postprocess on the same inputs
(pipeline_hunyuan_video.py:955-1092).  Tolerance: 3 bf16 denoise steps (per-step bar 2e-2 of the velocity range, pinned in
test_gpu_model.py) feed a ~30-layer fp16 decoder, so single rounding flips are amplified: max error <= 4e-2 of the [0,1]
output range, MEAN error <= 2e-3 (an indexing / ordering / blend mistake moves the mean by orders of magnitude)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as RD  # noqa: E402
from oracle import vae_ref as RV  # noqa: E402

DEV = "cuda:0"


def test_pipeline_denoise_and_decode_vs_oracle():
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.diffusion.pipelines import HunyuanVideoPipeline
    from hunyuanvideo_efficiency_amd.inference import get_rotary_pos_embed

    cfg = syn.tiny_config()
    model = build_model(cfg, DEV, seed=0)
    boc = (32, 64, 128, 128)
    # tile geometry small enough that the 21-frame x 192 x 160 video needs temporal AND spatial tiling
    vae = AutoencoderKLCausal3D(block_out_channels=boc, sample_size=128, sample_tsize=16, device=DEV)
    vsd = syn.synth_vae_state_dict(boc, seed=0)
    vae.load_state_dict({k: v.to(torch.float16) for k, v in vsd.items()}, strict=True)
    sched = FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler")
    pipe = HunyuanVideoPipeline(vae, model, sched, types.SimpleNamespace())

    frames, height, width, n_steps = 21, 192, 160, 3
    lt, lh, lw = (frames - 1) // 4 + 1, height // 8, width // 8
    x0, ts, tm, ts2 = syn.synth_dit_inputs(cfg, (lt, lh, lw), 32, 11, seed=3)
    freqs = get_rotary_pos_embed(model, frames, height, width, "884-16c-hy", 256, device=DEV)
    out = pipe(ts.to(torch.float16).to(DEV), tm.to(DEV), ts2.to(torch.float16).to(DEV), height, width, frames,
               num_inference_steps=n_steps, guidance_scale=1.0, embedded_guidance_scale=6.0, latents=x0.clone(), freqs_cis=freqs,
               vae_ver="884-16c-hy", enable_tiling=True, n_tokens=freqs[0].shape[0])
    video = out.videos
    assert video.shape == (1, 3, frames, height, width) and video.dtype == torch.float32 and video.device.type == "cpu"
    assert float(video.min()) >= 0.0 and float(video.max()) <= 1.0

    # ---- oracle chain on the same inputs
    E = RD.Prec(True)
    sd = {k: p.float().cpu() for k, p in model.state_dict().items()}
    cos, sin = RD.rope_tables([lt, lh // 2, lw // 2], cfg.rope_dim_list, 256.0)
    assert torch.allclose(cos, freqs[0].float().cpu(), atol=1e-6)
    # prompt embeddings arrive in fp16 (prompt_embeds.dtype) exactly as the pipeline passes them
    lat, _ = RD.denoise_loop(sd, cfg, x0.clone(), n_steps, ts.to(torch.float16).float(), tm, ts2.to(torch.float16).float(),
                             cos, sin, 6.0, 7.0, E)
    # the GPU latents after the loop, via the output_type="latent" surface - which, like the reference
    # (pipeline_hunyuan_video.py:1088-1092), returns them post-scaled: (x / 2 + 0.5).clamp(0, 1)
    lat_gpu = pipe(ts.to(torch.float16).to(DEV), tm.to(DEV), ts2.to(torch.float16).to(DEV), height, width, frames,
                   num_inference_steps=n_steps, embedded_guidance_scale=6.0, latents=x0.clone(), freqs_cis=freqs,
                   output_type="latent", n_tokens=freqs[0].shape[0]).videos
    lat_post = (lat / 2 + 0.5).clamp(0, 1)
    assert float(lat_post.min()) == 0.0 and float(lat_post.max()) == 1.0 and 0.2 < float(((lat_post > 0) & (lat_post < 1)).float().mean())
    err_lat = float((lat_gpu - lat_post).abs().max())
    assert err_lat < 2e-2, err_lat

    EV = RV.Prec(True)
    sd16 = {k: v.to(torch.float16).float() for k, v in vsd.items()}
    tp = RV.TileParams(sample_size=128, sample_tsize=16, n_blocks=4)
    assert (vae.tile_latent_min_tsize, vae.tile_latent_min_size) == (tp.tile_latent_min_tsize, tp.tile_latent_min_size)
    z = lat / vae.config.scaling_factor
    ref = RV.postprocess(RV.decode(sd16, z, boc, tp, EV, tiling=True), EV)
    assert ref.shape == video.shape
    diff = (video - ref).abs()
    assert float(diff.max()) < 4e-2, float(diff.max())
    assert float(diff.mean()) < 2e-3, float(diff.mean())
    # decode of the ORACLE's latents on the GPU isolates the VAE leg (no denoise drift): tighter
    vae.enable_tiling()
    y = vae.decode(z.to(DEV), return_dict=False)[0]
    from hunyuanvideo_efficiency_amd import vae_ops
    img = vae_ops.postprocess(y.contiguous()).cpu().float()
    assert float((img - ref).abs().max()) < 2e-2
    assert float((img - ref).abs().mean()) < 1e-3


def test_pipeline_prompt_surface_with_text_encoders():
    """`pipe(prompt=...)` (the reference's keyword surface): tiny random LLM + CLIP text encoders on the GPU feed the DiT; the result
    must equal the run that is handed the same embeddings explicitly."""
    from tests.test_text_encoder_cpu import _llm, _toy_tokenizer
    from transformers import CLIPTextConfig, CLIPTextModel
    from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.diffusion.pipelines import HunyuanVideoPipeline
    from hunyuanvideo_efficiency_amd.inference import get_rotary_pos_embed
    cfg = syn.DiTConfig(hidden_size=256, heads_num=2, mm_double_blocks_depth=1, mm_single_blocks_depth=1, text_states_dim=64,
                        text_states_dim_2=64)      # the GEMM contracts over multiples of 64 (real model: 4096 / 768)
    model = build_model(cfg, DEV, seed=0)
    boc = (32, 64, 128, 128)
    vae = AutoencoderKLCausal3D(block_out_channels=boc, device=DEV)
    vae.load_state_dict({k: v.to(torch.float16) for k, v in syn.synth_vae_state_dict(boc, seed=0).items()}, strict=True)
    tpl = {"template": "describe the video : {}", "crop_start": 4}
    te = TextEncoder("llm", max_length=16 + 4, text_encoder_precision="fp16", prompt_template=tpl, prompt_template_video=tpl,
                     hidden_state_skip_layer=2, model=_llm(64), tokenizer=_toy_tokenizer(64), device=DEV)
    torch.manual_seed(1)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=32, hidden_size=64, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                                        max_position_embeddings=16, projection_dim=64, pad_token_id=0, bos_token_id=1, eos_token_id=2))
    te2 = TextEncoder("clipL", max_length=10, text_encoder_precision="fp16", model=clip, tokenizer=_toy_tokenizer(16), device=DEV)
    pipe = HunyuanVideoPipeline(vae, model, FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler"), types.SimpleNamespace(),
                                text_encoder=te, text_encoder_2=te2)
    frames, height, width = 5, 64, 64
    freqs = get_rotary_pos_embed(model, frames, height, width, "884-16c-hy", 256, device=DEV)
    x0 = syn.hashed_uniform((1, 16, 2, 8, 8), "prompt.lat", 2) * 1.7
    kw = dict(height=height, width=width, video_length=frames, num_inference_steps=2, embedded_guidance_scale=6.0, freqs_cis=freqs,
              enable_tiling=False, n_tokens=freqs[0].shape[0])
    prompt = "a cat walks on grass [EOS]"
    v1 = pipe(prompt=prompt, data_type="video", latents=x0.clone(), **kw).videos
    emb, _, mask, _ = pipe.encode_prompt(prompt, DEV, data_type="video")
    emb2 = pipe.encode_prompt(prompt, DEV, text_encoder=te2, data_type="video")[0]
    assert emb.shape == (1, 16, 64) and emb.dtype == torch.float16 and int(mask.sum()) == 6 and emb2.shape == (1, 64)
    v2 = pipe(emb, mask, emb2, latents=x0.clone(), **kw).videos
    assert v1.shape == (1, 3, frames, height, width) and torch.equal(v1, v2)
    with pytest.raises(ValueError):
        HunyuanVideoPipeline(vae, model, pipe.scheduler)(prompt="x", **kw)


def test_pipeline_classifier_free_guidance_batch_vs_oracle():
    """The CFG branch of the reference pipeline (pipeline_hunyuan_video.py:966-1019: batch [uncond | cond] through the transformer,
    uncond + scale (cond - uncond), optional guidance_rescale :56-71) on a non-distilled tiny model (guidance_embed=False, what
    "HYVideo-T/2" is): the batch-2 forward (two passes through one workspace) and the combination against the oracle run per
    branch in the bf16-emulated contract.  Different prompts AND different valid-token counts in the two branches."""
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.diffusion.pipelines import HunyuanVideoPipeline
    from hunyuanvideo_efficiency_amd.diffusion.pipelines.pipeline_hunyuan_video import rescale_noise_cfg
    from hunyuanvideo_efficiency_amd.inference import get_rotary_pos_embed
    cfg = syn.tiny_config()
    cfg.guidance_embed = False
    model = build_model(cfg, DEV, seed=0)
    pipe = HunyuanVideoPipeline(None, model, FlowMatchDiscreteScheduler(shift=7.0, reverse=True, solver="euler"), types.SimpleNamespace())
    frames, height, width, n_steps, scale, resc = 17, 128, 128, 2, 3.5, 0.3
    lt, lh, lw = (frames - 1) // 4 + 1, height // 8, width // 8
    x0, ts, tm, ts2 = syn.synth_dit_inputs(cfg, (lt, lh, lw), 32, 11, seed=5)
    _, nts, ntm, nts2 = syn.synth_dit_inputs(cfg, (lt, lh, lw), 32, 4, seed=6)          # the "negative prompt": other states, 4 valid tokens
    freqs = get_rotary_pos_embed(model, frames, height, width, "884-16c-hy", 256, device=DEV)
    f16 = lambda t: t.to(torch.float16).to(DEV)
    kw = dict(height=height, width=width, video_length=frames, num_inference_steps=n_steps, embedded_guidance_scale=None, freqs_cis=freqs,
              output_type="latent", n_tokens=freqs[0].shape[0])
    got = pipe(f16(ts), tm.to(DEV), f16(ts2), guidance_scale=scale, guidance_rescale=resc, latents=x0.clone(),
               negative_prompt_embeds=f16(nts), negative_prompt_mask=ntm.to(DEV), negative_prompt_embeds_2=f16(nts2), **kw).videos
    # oracle: per step, the two branches separately, combined in the pipeline's dtype (bf16 model outputs)
    E = RD.Prec(True)
    sd = {k: p.float().cpu() for k, p in model.state_dict().items()}
    cos, sin = RD.rope_tables([lt, lh // 2, lw // 2], cfg.rope_dim_list, 256.0)
    sig = RD.flow_sigmas(n_steps, 7.0)
    tsteps = RD.flow_timesteps(sig)
    lat = x0.clone().float()
    h = lambda t: t.to(torch.float16).float()
    for i in range(n_steps):
        vu = RD.dit_forward(sd, cfg, lat, tsteps[i:i + 1], h(nts), ntm, h(nts2), cos, sin, None, E).to(torch.bfloat16)
        vc = RD.dit_forward(sd, cfg, lat, tsteps[i:i + 1], h(ts), tm, h(ts2), cos, sin, None, E).to(torch.bfloat16)
        v = vu + scale * (vc - vu)
        v = rescale_noise_cfg(v, vc, guidance_rescale=resc)
        lat = RD.euler_step(lat, v.float(), sig, i)
    ref = (lat / 2 + 0.5).clamp(0, 1)
    assert 0.2 < float(((ref > 0) & (ref < 1)).float().mean())
    err = float((got - ref).abs().max())
    # 3.5 v_cond - 2.5 v_uncond amplifies the per-branch bf16 drift (bar 2e-2 of range per forward, test_gpu_model.py; ~1e-2 measured) by
    # up to 6x, the (x / 2 + 0.5) post-scale halves it: max bar 6e-2; the MEAN shows ordering / indexing mistakes and stays tight
    assert err < 6e-2, err
    assert float((got - ref).abs().mean()) < 4e-3, float((got - ref).abs().mean())
    # and the batch-2 forward itself == the two single forwards, bit for bit (same kernels, same workspace)
    t = torch.tensor([tsteps[0]], device=DEV)
    both = model(torch.cat([x0, x0]).to(DEV), t.repeat(2), text_states=torch.cat([f16(nts), f16(ts)]), text_mask=torch.cat([ntm, tm]).to(DEV),
                 text_states_2=torch.cat([f16(nts2), f16(ts2)]), freqs_cos=freqs[0], freqs_sin=freqs[1], guidance=None)["x"]
    one_u = model(x0.to(DEV), t, text_states=f16(nts), text_mask=ntm.to(DEV), text_states_2=f16(nts2), freqs_cos=freqs[0], freqs_sin=freqs[1])["x"]
    one_c = model(x0.to(DEV), t, text_states=f16(ts), text_mask=tm.to(DEV), text_states_2=f16(ts2), freqs_cos=freqs[0], freqs_sin=freqs[1])["x"]
    assert torch.equal(both[0:1], one_u) and torch.equal(both[1:2], one_c)
