#!/usr/bin/env python3
"""Generate tests/golden/vae_enc_*.npz by executing the reference's VAE ENCODE code and the fork's t_ops hooks on CPU (fp32).

Same loading recipe as tools/make_golden_vae.py (the two reference files loaded by path into a synthetic package, in-memory stubs
for the absent third-party imports, `Attention` = our restatement of diffusers' deprecated attn block - parity unpinned for that
piece).  The tiled-encode methods of AutoencoderKLCausal3D are compiled from the class body and run bound to a stand-in object.
The t_ops configuration is applied with the reference's own `apply_t_ops_config*` methods (the loop of vae/__init__.py:15-63 is
restated here because that module imports the diffusers mixin stack).
Run: python tools/make_golden_vae_enc.py"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
import make_golden_vae as G  # noqa: E402

REF = G.REF


def tiled_encode_methods(dist_cls):
    import ast
    import typing
    path = os.path.join(REF, "hyvideo", "vae", "autoencoder_kl_causal_3d.py")
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AutoencoderKLCausal3D")
    wanted = {"blend_v", "blend_h", "blend_t", "spatial_tiled_encode", "temporal_tiled_encode"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for f in fns:
        f.decorator_list = []
    ns = {"torch": torch, "DiagonalGaussianDistribution": dist_cls, "AutoencoderKLOutput": lambda latent_dist=None, **k: types.SimpleNamespace(latent_dist=latent_dist)}
    ns.update({k: getattr(typing, k) for k in ("Union", "Optional", "Tuple", "Dict")})
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return {k: ns[k] for k in wanted}


T_OPS = {
    "encoder": {
        "down_blocks": [
            {"block_index": 0, "pool_t_kernel": 3, "pool_t_stride": 2, "enable_t_pool_before_block": [False, True],
             "enable_t_pool_after_block": [False, False], "downsample_stride": [1, 2, 2]},
            {"block_index": 1, "pool_t_kernel": 3, "pool_t_stride": 2, "enable_t_pool_before_block": [False, False],
             "enable_t_pool_after_block": [False, False], "downsample_stride": [1, 2, 2]},      # time stride 2 -> 1
            {"block_index": 2, "pool_t_kernel": 2, "pool_t_stride": 2, "enable_t_pool_before_block": [False, False],
             "enable_t_pool_after_block": [True, False], "downsample_stride": [1, 2, 2]},
            {"block_index": 3, "pool_t_kernel": 3, "pool_t_stride": 2, "enable_t_pool_before_block": [False, False],
             "enable_t_pool_after_block": [False, False], "downsample_stride": [1, 1, 1]},
        ],
        "mid_block": {"pool_t_kernel": 3, "pool_t_stride": 1, "enable_t_pool_before_block": [False, True],
                      "enable_t_pool_after_block": [False, False]},
    },
    "decoder": {
        "up_blocks": [
            {"block_index": 0, "enable_t_interp_before_block": [False, False, False], "enable_t_interp_after_block": [False, False, False],
             "interp_t_scale_factor": 2, "interp_mode": "nearest"},
            {"block_index": 1, "enable_t_interp_before_block": [False, True, False], "enable_t_interp_after_block": [False, False, False],
             "interp_t_scale_factor": 2, "interp_mode": "nearest"},
            {"block_index": 2, "enable_t_interp_before_block": [False, False, False], "enable_t_interp_after_block": [False, False, True],
             "interp_t_scale_factor": 2, "interp_mode": "nearest"},
            {"block_index": 3, "enable_t_interp_before_block": [False, False, False], "enable_t_interp_after_block": [False, False, False],
             "interp_t_scale_factor": 2, "interp_mode": "nearest"},
        ],
        "mid_block": {"enable_t_pool_before_block": [False, False], "enable_t_pool_after_block": [False, False]},
    },
}


def apply_t_ops(enc, dec, cfg):
    """the loop of vae/__init__.py:15-63 over the reference's own per-block methods"""
    for bc in cfg["encoder"]["down_blocks"]:
        enc.down_blocks[bc["block_index"]].apply_t_ops_config(bc)
    enc.mid_block.apply_t_ops_config_midblock(cfg["encoder"]["mid_block"])
    for bc in cfg["decoder"]["up_blocks"]:
        dec.up_blocks[bc["block_index"]].apply_t_ops_config(bc)
    dec.mid_block.apply_t_ops_config_midblock(cfg["decoder"]["mid_block"])


def main():
    torch.set_grad_enabled(False)
    m = G.install_stubs()
    B3, V = m["unet_causal_3d_blocks"], m["vae"]
    U = syn.hashed_uniform
    s3 = 3.0 ** 0.5

    # ---- leaf: strided causal conv (DownsampleCausal3D), all stride patterns the encoder / t_ops use, odd extents
    w, b = syn.synth_param("ge.conv.weight", (16, 8, 3, 3, 3), 1), syn.synth_param("ge.conv.bias", (16,), 1)
    x = U((1, 8, 5, 7, 6), "ge.conv.x", 1) * s3
    outs = {}
    for st in ((2, 2, 2), (1, 2, 2), (2, 1, 1), (1, 1, 1)):
        ds = B3.DownsampleCausal3D(8, use_conv=True, out_channels=16, stride=st, name="op")
        ds.conv.conv.weight.copy_(w), ds.conv.conv.bias.copy_(b)
        outs["y" + "".join(map(str, st))] = ds(x)
    G.save("vae_enc_downsample", x=x, w=w, b=b, **outs)

    # ---- reduced-channel encoder (32,64,128,128): video [1,3,9,32,32] -> moments [1,32,3,4,4]
    boc = (32, 64, 128, 128)
    enc = V.EncoderCausal3D(in_channels=3, out_channels=16, down_block_types=("DownEncoderBlockCausal3D",) * 4,
                            block_out_channels=boc, layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True,
                            time_compression_ratio=4, spatial_compression_ratio=8, mid_block_add_attention=True)
    dec = V.DecoderCausal3D(in_channels=16, out_channels=3, up_block_types=("UpDecoderBlockCausal3D",) * 4,
                            block_out_channels=boc, layers_per_block=2, norm_num_groups=32, act_fn="silu",
                            time_compression_ratio=4, spatial_compression_ratio=8, mid_block_add_attention=True)
    sd = syn.synth_vae_state_dict(boc, seed=0, encoder=True)
    print("reference EncoderCausal3D.load_state_dict(strict=True):",
          enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True))
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=True)
    enc.eval(), dec.eval()
    qc, pq = nn.Conv3d(32, 32, kernel_size=1), nn.Conv3d(16, 16, kernel_size=1)
    qc.weight.copy_(sd["quant_conv.weight"]), qc.bias.copy_(sd["quant_conv.bias"])
    pq.weight.copy_(sd["post_quant_conv.weight"]), pq.bias.copy_(sd["post_quant_conv.bias"])
    xv = U((1, 3, 9, 32, 32), "ge.video", 1)
    moments = qc(enc(xv))
    dist = V.DiagonalGaussianDistribution(moments)
    G.save("vae_enc_tile", x=xv, moments=moments, mean=dist.mode(), std=dist.std, kl=dist.kl(), block_out_channels=np.array(boc),
           recon=dec(pq(dist.mode())))

    # ---- tiled encode through the reference's own tiling code: sample tile 16 px / 8+1 frames, latent tile 2 px / 2 frames
    Tm = tiled_encode_methods(V.DiagonalGaussianDistribution)
    ae = types.SimpleNamespace(encoder=enc, quant_conv=qc, use_spatial_tiling=True, use_temporal_tiling=True,
                               tile_sample_min_tsize=8, tile_latent_min_tsize=2, tile_sample_min_size=32, tile_latent_min_size=4,
                               tile_overlap_factor=0.25)
    for n, f in Tm.items():
        setattr(ae, n, types.MethodType(f, ae))
    xt = U((1, 3, 13, 56, 40), "ge.video.tiled", 1)       # 2 temporal x (3 x 2) spatial tiles, ragged edges
    mt = ae.temporal_tiled_encode(xt, return_dict=True).latent_dist.parameters
    ms = ae.spatial_tiled_encode(xt[:, :, :5], return_moments=True)
    G.save("vae_enc_tiled", x=xt, moments=mt, moments_spatial_only=ms, tile=np.array([8, 2, 32, 4]))

    # ---- the fork's t_ops on both halves (temporal pools, downsample stride override, nearest temporal interpolation)
    apply_t_ops(enc, dec, T_OPS)
    xo = U((1, 3, 17, 16, 16), "ge.video.tops", 1)
    mo = qc(enc(xo))
    zo = V.DiagonalGaussianDistribution(mo).mode()
    ro = dec(pq(zo))
    print("t_ops: video", tuple(xo.shape), "-> moments", tuple(mo.shape), "-> recon", tuple(ro.shape))
    import json
    G.save("vae_enc_tops", x=xo, moments=mo, recon=ro, t_ops_json=np.frombuffer(json.dumps(T_OPS).encode(), dtype=np.uint8))


if __name__ == "__main__":
    main()
