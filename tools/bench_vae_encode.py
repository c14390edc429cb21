"""Encode-side timings on one MI355X with the shipped channel widths (128,256,512,512), synthetic weights:
(a) tiled encode of a 720x1280x129f video (the i2v / training-data direction of the hot path's VAE),
(b) the fork's infer.py case: untiled auto-encoder forward of a 240p clip, with and without a t_ops config."""
import sys, time, json, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import synthetic as syn
from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D

dev = "cuda"
boc = syn.VAE_BLOCK_OUT_CHANNELS
vae = AutoencoderKLCausal3D(block_out_channels=boc, device=dev, with_encoder=True)
with torch.no_grad():
    for k, p in vae.state_dict().items():
        p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))

def timed(fn, n=1):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out

# conv FLOPs of the encoder per input voxel are dominated by the full-resolution 128-channel layers
x = (syn.hashed_uniform((1, 3, 129, 720, 1280), "enc.video", 0, dev)).to(torch.float16)
vae.enable_tiling()
dt, post = timed(lambda: vae.encode(x).latent_dist)
print(f"tiled encode 720x1280x129f: {dt:.2f} s -> moments {tuple(post.parameters.shape)}", flush=True)
del x, post
vae.disable_tiling()
clip = (syn.hashed_uniform((1, 3, 65, 240, 416), "enc.clip", 0, dev)).to(torch.float16)
dt, out = timed(lambda: vae(clip, return_dict=False)[0], n=2)
print(f"untiled auto-encoder forward 240x416x65f: {dt*1e3:.0f} ms -> {tuple(out.shape)}", flush=True)
cfg = {"encoder": {"down_blocks": [{"block_index": 0, "pool_t_kernel": 3, "pool_t_stride": 2, "enable_t_pool_before_block": [True, False],
                                    "enable_t_pool_after_block": [False, False], "downsample_stride": [1, 2, 2]},
                                   {"block_index": 1, "pool_t_kernel": 3, "pool_t_stride": 2, "enable_t_pool_before_block": [False, False],
                                    "enable_t_pool_after_block": [False, False], "downsample_stride": [1, 2, 2]}]},
       "decoder": {"up_blocks": [{"block_index": 3, "enable_t_interp_before_block": [False, False, False],
                                  "enable_t_interp_after_block": [False, False, True], "interp_t_scale_factor": 2}]}}
vae.apply_t_ops_config(cfg)
dt, out = timed(lambda: vae(clip, return_dict=False)[0], n=2)
print(f"same with t_ops (temporal pool before encoder block 0, interp after the last decoder block): {dt*1e3:.0f} ms -> {tuple(out.shape)}", flush=True)
