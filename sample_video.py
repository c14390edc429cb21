#!/usr/bin/env python3
"""sample_video.py with the reference's flag names (sample_video.py:12-55, hyvideo/config.py) for the hot path on
MI355X.  There are no checkpoints or text encoders in this environment, so weights are random-init (deterministic
hash) and prompt embeddings are synthetic; with real weights, load state dicts into the same modules
(state-dict key names are the reference's).  torchrun --nproc_per_node=8 sample_video.py --ulysses-degree 8 ...
shards the token axis exactly like the reference."""
import argparse
import os
import time

import torch


def parse_args():
    p = argparse.ArgumentParser(description="HunyuanVideo denoise + decode on MI355X kernels")
    p.add_argument("--model", default="HYVideo-T/2-cfgdistill")
    p.add_argument("--precision", default="bf16", choices=["bf16"])
    p.add_argument("--rope-theta", type=int, default=256)
    p.add_argument("--vae", default="884-16c-hy")
    p.add_argument("--vae-precision", default="fp16", choices=["fp16"])
    p.add_argument("--vae-tiling", action="store_true", default=True)
    p.add_argument("--flow-shift", type=float, default=7.0)
    p.add_argument("--flow-reverse", action="store_true", help="If reverse, learning/sampling from t=1 -> t=0 (config.py:193-196; the "
                   "reference's launch scripts always pass it)")
    p.add_argument("--flow-solver", default="euler")
    p.add_argument("--infer-steps", type=int, default=50)
    p.add_argument("--video-size", type=int, nargs="+", default=[720, 1280])
    p.add_argument("--video-length", type=int, default=129)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--cfg-scale", type=float, default=1.0)
    p.add_argument("--embedded-cfg-scale", type=float, default=6.0)
    p.add_argument("--use-fp8", action="store_true")
    p.add_argument("--ulysses-degree", type=int, default=1)
    p.add_argument("--ring-degree", type=int, default=1)
    p.add_argument("--text-len", type=int, default=256)
    p.add_argument("--dit-weight", default=None, help="reference-format DiT checkpoint: file or directory (inference.py:279-354); "
                                                      "default: random-init weights")
    p.add_argument("--load-key", default="module", help="key of the state dict inside a wrapped checkpoint (module | ema)")
    p.add_argument("--model-base", default=None, help="root holding t2v_<resolution>/ when --dit-weight is a bare name")
    p.add_argument("--model-resolution", default="540p")
    p.add_argument("--vae-path", default=None, help="directory with the reference's VAE config.json + pytorch_model.pt")
    p.add_argument("--prompt", default=None, help="text prompt; needs --text-encoder-path (+ --text-encoder-2-path). Default: synthetic "
                                                   "prompt embeddings (no text-encoder checkpoints exist in this environment)")
    p.add_argument("--neg-prompt", default=None, help="negative prompt of the classifier-free-guidance branch (--cfg-scale > 1; default \"\")")
    p.add_argument("--text-encoder-path", default=None, help="HF directory of the LLM text encoder (+ tokenizer)")
    p.add_argument("--text-encoder-2-path", default=None, help="HF directory of CLIP-L (+ tokenizer)")
    p.add_argument("--text-encoder-precision", default="fp16")
    p.add_argument("--text-len-2", type=int, default=77)
    p.add_argument("--prompt-template", default="dit-llm-encode")
    p.add_argument("--prompt-template-video", default="dit-llm-encode-video")
    p.add_argument("--hidden-state-skip-layer", type=int, default=2)
    p.add_argument("--apply-final-norm", action="store_true")
    p.add_argument("--tiny", action="store_true", help="tiny DiT (d=256, 1+1 blocks) and reduced VAE: plumbing check")
    p.add_argument("--save-path", default="./results")
    return p.parse_args()


def main():
    a = parse_args()
    from hunyuanvideo_efficiency_amd import synthetic as syn
    from hunyuanvideo_efficiency_amd.inference import init_distributed, parallelize_transformer, get_rotary_pos_embed
    from hunyuanvideo_efficiency_amd.builders import build_model
    from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
    from hunyuanvideo_efficiency_amd.diffusion.schedulers import FlowMatchDiscreteScheduler
    from hunyuanvideo_efficiency_amd.diffusion.pipelines import HunyuanVideoPipeline
    from hunyuanvideo_efficiency_amd.modules.fp8_optimization import convert_fp8_linear
    dev = init_distributed(a.ulysses_degree, a.ring_degree)
    rank = int(os.environ.get("RANK", "0"))
    h, w = (a.video_size * 2)[:2]
    if h % 16 or w % 16 or (a.video_length - 1) % 4:
        raise ValueError("height/width must be multiples of 16 and (video_length - 1) a multiple of 4 (inference.py:571-586)")
    cfg = syn.tiny_config() if a.tiny else syn.DiTConfig()
    # "HYVideo-T/2-cfgdistill" (the shipped model: guidance embedded, --cfg-scale 1) or "HYVideo-T/2" (no guidance embedding: classifier-free
    # guidance with --cfg-scale > 1, the batch [uncond | cond] of pipeline_hunyuan_video.py:966-1019)
    from hunyuanvideo_efficiency_amd.modules.models import HUNYUAN_VIDEO_CONFIG
    if a.model not in HUNYUAN_VIDEO_CONFIG:
        raise ValueError(f"--model must be one of {list(HUNYUAN_VIDEO_CONFIG)}")
    cfg.guidance_embed = bool(HUNYUAN_VIDEO_CONFIG[a.model].get("guidance_embed", False))
    model = build_model(cfg, dev, seed=0)
    if a.dit_weight or a.model_base:
        # real weights: same order as the reference (inference.py:199-202): fp8 conversion (scales from <ckpt>_map.pt), then load.
        # The checkpoint path is resolved FIRST so that convert_fp8_linear reads the map that belongs to the weights being loaded
        # (quantising the random-init weights with their own scales and then copying real fp8 weights over them would be wrong).
        from hunyuanvideo_efficiency_amd import checkpoint
        ckpt = checkpoint.resolve_dit_path(a, a.model_base)
        if a.use_fp8:
            convert_fp8_linear(model, str(ckpt), torch.bfloat16)
        checkpoint.load_state_dict(a, model, a.model_base)
    elif a.use_fp8:
        convert_fp8_linear(model, None, torch.bfloat16)
    if a.vae_path:
        from hunyuanvideo_efficiency_amd.vae import load_vae
        vae = load_vae(a.vae, a.vae_precision, vae_path=a.vae_path, device=dev)[0]
    else:
        boc = (64, 64, 128, 128) if a.tiny else syn.VAE_BLOCK_OUT_CHANNELS
        vae = AutoencoderKLCausal3D(block_out_channels=boc, device=dev)
        with torch.no_grad():
            for k, p in vae.state_dict().items():
                p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))
    sched = FlowMatchDiscreteScheduler(shift=a.flow_shift, reverse=a.flow_reverse, solver=a.flow_solver)
    text_encoder = text_encoder_2 = None
    if a.prompt is not None:
        # inference.py:216-265: max_length = text_len + crop_start of the video template
        if a.text_encoder_path is None:
            raise ValueError("--prompt needs --text-encoder-path (a Hugging Face model directory)")
        from hunyuanvideo_efficiency_amd.constants import PROMPT_TEMPLATE
        from hunyuanvideo_efficiency_amd.text_encoder import TextEncoder
        tv = PROMPT_TEMPLATE[a.prompt_template_video]
        text_encoder = TextEncoder("llm", a.text_len + tv.get("crop_start", 0), a.text_encoder_precision, a.text_encoder_path,
                                   tokenizer_type="llm", prompt_template=PROMPT_TEMPLATE[a.prompt_template], prompt_template_video=tv,
                                   hidden_state_skip_layer=a.hidden_state_skip_layer, apply_final_norm=a.apply_final_norm, device=dev)
        if a.text_encoder_2_path is not None:
            text_encoder_2 = TextEncoder("clipL", a.text_len_2, a.text_encoder_precision, a.text_encoder_2_path, tokenizer_type="clipL",
                                         device=dev)
    pipe = HunyuanVideoPipeline(vae, model, sched, a, text_encoder=text_encoder, text_encoder_2=text_encoder_2)
    if a.ulysses_degree > 1 or a.ring_degree > 1:      # inference.py:157,408
        parallelize_transformer(pipe)
    lt = (a.video_length - 1) // 4 + 1
    _, ts, tm, ts2 = syn.synth_dit_inputs(cfg, (lt, h // 8, w // 8), a.text_len, 11, seed=a.seed, device=dev)
    freqs = get_rotary_pos_embed(model, a.video_length, h, w, a.vae, a.rope_theta, device=dev)
    gen = torch.Generator(device=dev).manual_seed(a.seed)
    torch.cuda.synchronize()
    t0 = time.time()
    def progress(i, t, latents):       # the reference shows a tqdm bar (pipeline_hunyuan_video.py:955-961); one line every 5 steps here
        if rank == 0:
            print(f"  step {i + 1}/{a.infer_steps}  t={float(t):.1f}  elapsed {time.time() - t0:.1f} s", flush=True)
    common = dict(callback=progress, callback_steps=5, height=h, width=w, video_length=a.video_length, num_inference_steps=a.infer_steps, guidance_scale=a.cfg_scale,
                  embedded_guidance_scale=a.embedded_cfg_scale if cfg.guidance_embed else None, generator=gen, freqs_cis=freqs, vae_ver=a.vae,
                  enable_tiling=a.vae_tiling, n_tokens=freqs[0].shape[0])
    if a.prompt is not None:
        out = pipe(prompt=a.prompt, negative_prompt=getattr(a, "neg_prompt", None),
                   prompt_embeds_2=None if text_encoder_2 is not None else ts2.to(torch.float16), data_type="video", **common)
    else:
        neg = {}
        if a.cfg_scale > 1.0:      # synthetic "negative prompt" embeddings (no text encoder in this run)
            _, nts, ntm, nts2 = syn.synth_dit_inputs(cfg, (lt, h // 8, w // 8), a.text_len, 4, seed=a.seed + 1, device=dev)
            neg = dict(negative_prompt_embeds=nts.to(torch.float16), negative_prompt_mask=ntm, negative_prompt_embeds_2=nts2.to(torch.float16))
        out = pipe(ts.to(torch.float16), tm, ts2.to(torch.float16), **neg, **common)
    dt = time.time() - t0
    if rank == 0:
        v = out.videos
        print(f"Success, time: {dt:.2f} s; video tensor {tuple(v.shape)} {v.dtype} range [{float(v.min()):.3f}, {float(v.max()):.3f}]")
        os.makedirs(a.save_path, exist_ok=True)
        torch.save(v[:, :, :1].clone(), os.path.join(a.save_path, "first_frame.pt"))
        from hunyuanvideo_efficiency_amd.utils.file_utils import save_videos_grid
        print("Sample save to:", save_videos_grid(v, os.path.join(a.save_path, f"seed{a.seed}.mp4"), fps=24))


if __name__ == "__main__":
    main()
