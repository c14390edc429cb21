#!/usr/bin/env python3
"""Generate tests/golden/vae_*.npz by executing the reference's VAE decode code on CPU (fp32).

hyvideo/vae/unet_causal_3d_blocks.py and hyvideo/vae/vae.py are loaded BY FILE PATH into a synthetic package (the
package __init__ pulls autoencoder_kl_causal_3d.py and the full diffusers mixin stack, which is not installed).  In-memory
stubs supply what those two files import from absent third-party packages: loguru.logger (no-op),
diffusers.utils.{logging, BaseOutput, is_torch_version}, diffusers.utils.torch_utils.randn_tensor,
diffusers.models.activations.get_activation (silu/swish -> nn.SiLU), class placeholders SpatialNorm / AdaGroupNorm /
RMSNorm (unused at norm_type="group"), and `Attention`: OUR restatement of diffusers 0.31's deprecated-attn-block path
(SURVEY.md 8c item 1 - the mid-block attention arithmetic is third-party and absent, so it is "parity unpinned").
The tiling/blending methods of AutoencoderKLCausal3D are executed bound to a small stand-in object
(they touch only self.tile_*, self.post_quant_conv, self.decoder, self.use_spatial_tiling, self.blend_*).
Run: python tools/make_golden_vae.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402


class AttentionRestated(nn.Module):
    """diffusers.models.attention_processor.Attention, _from_deprecated_attn_block path, restated (not reference code)."""

    def __init__(self, query_dim, heads=1, dim_head=64, rescale_output_factor=1.0, eps=1e-5, norm_num_groups=None,
                 spatial_norm_dim=None, residual_connection=False, bias=False, upcast_softmax=False,
                 _from_deprecated_attn_block=False, **kw):
        super().__init__()
        assert heads == 1 and residual_connection and _from_deprecated_attn_block
        inner = heads * dim_head
        self.scale = dim_head ** -0.5
        self.rescale = rescale_output_factor
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, temb=None, attention_mask=None):
        residual = hidden_states
        h = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], attn_mask=attention_mask[:, None])[:, 0]
        return (self.to_out[0](a) + residual) / self.rescale


def install_stubs():
    sys.dont_write_bytecode = True
    noop = lambda *a, **k: None
    lg = types.ModuleType("loguru")
    lg.logger = types.SimpleNamespace(info=noop, warning=noop, debug=noop, error=noop)
    mods = {"loguru": lg}

    def mk(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        mods[name] = m
        return m
    mk("diffusers")
    mk("diffusers.utils", logging=types.SimpleNamespace(get_logger=lambda n: types.SimpleNamespace(warning=noop, info=noop)),
       BaseOutput=type("BaseOutput", (dict,), {}), is_torch_version=lambda op, v: True)
    mk("diffusers.utils.torch_utils", randn_tensor=lambda shape, generator=None, device=None, dtype=None: torch.randn(shape))
    mk("diffusers.models")
    mk("diffusers.models.activations", get_activation=lambda n: {"silu": nn.SiLU(), "swish": nn.SiLU()}[n])
    mk("diffusers.models.attention_processor", SpatialNorm=type("SpatialNorm", (nn.Module,), {}), Attention=AttentionRestated)
    mk("diffusers.models.normalization", AdaGroupNorm=type("AdaGroupNorm", (nn.Module,), {}),
       RMSNorm=type("RMSNorm", (nn.Module,), {}))
    mk("diffusers.models.modeling_utils", ModelMixin=nn.Module)
    sys.modules.update(mods)
    pkg = types.ModuleType("refvae")
    pkg.__path__ = [os.path.join(REF, "hyvideo", "vae")]
    sys.modules["refvae"] = pkg
    out = {}
    for name in ("unet_causal_3d_blocks", "vae"):
        spec = importlib.util.spec_from_file_location(f"refvae.{name}", os.path.join(REF, "hyvideo", "vae", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refvae.{name}"] = m
        spec.loader.exec_module(m)
        out[name] = m
    return out


def tiling_methods():
    """Extract the reference's tiling/blending method objects from the source of autoencoder_kl_causal_3d.py WITHOUT
    importing the module (its top-level imports need the diffusers mixin stack): the class body is compiled with
    placeholder bases and only the plain functions are taken."""
    path = os.path.join(REF, "hyvideo", "vae", "autoencoder_kl_causal_3d.py")
    src = open(path).read()
    import ast
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AutoencoderKLCausal3D")
    wanted = {"blend_v", "blend_h", "blend_t", "spatial_tiled_decode", "temporal_tiled_decode", "_decode"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for f in fns:
        f.decorator_list = []
    mod = ast.Module(body=fns, type_ignores=[])
    ns = {"torch": torch, "Union": object, "DecoderOutput": lambda sample: types.SimpleNamespace(sample=sample)}
    import typing
    ns.update({k: getattr(typing, k) for k in ("Union", "Optional", "Tuple", "Dict")})
    exec(compile(mod, path, "exec"), ns)
    return {k: ns[k] for k in wanted}


def save(name, **arrs):
    conv = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


def main():
    torch.set_grad_enabled(False)
    m = install_stubs()
    B3, V = m["unet_causal_3d_blocks"], m["vae"]
    U = syn.hashed_uniform
    s3 = 3.0 ** 0.5

    # ---- leaves
    conv = B3.CausalConv3d(8, 16, kernel_size=3)
    w, b = syn.synth_param("gv.conv.weight", (16, 8, 3, 3, 3), 1), syn.synth_param("gv.conv.bias", (16,), 1)
    conv.conv.weight.copy_(w), conv.conv.bias.copy_(b)
    x = U((1, 8, 3, 5, 6), "gv.conv.x", 1) * s3
    save("vae_causal_conv", x=x, w=w, b=b, y=conv(x))

    up = B3.UpsampleCausal3D(4, use_conv=False, upsample_factor=(2, 2, 2))
    up1 = B3.UpsampleCausal3D(4, use_conv=False, upsample_factor=(1, 2, 2))
    xu = U((1, 4, 3, 3, 2), "gv.up.x", 1)
    save("vae_upsample", x=xu, y222=up(xu), y122=up1(xu), y222_t1=up(xu[:, :, :1]))

    rb = B3.ResnetBlockCausal3D(in_channels=32, out_channels=64, temb_channels=None, groups=32, eps=1e-6,
                                non_linearity="silu", output_scale_factor=1.0)
    rsd = {k: syn.synth_param("gv.res." + k, tuple(v.shape), 1) for k, v in rb.state_dict().items()}
    rb.load_state_dict(rsd, strict=True)
    xr = U((1, 32, 3, 6, 5), "gv.res.x", 1) * s3
    save("vae_resnet", x=xr, y=rb(xr, None))

    mask = B3.prepare_causal_attention_mask(3, 4, torch.float32, "cpu", batch_size=None)
    save("vae_causal_mask", mask=mask)

    # ---- reduced-channel decoder tile (32,64,128,128), tile [1,16,3,4,4] -> [1,3,9,32,32]
    boc = (32, 64, 128, 128)
    dec = V.DecoderCausal3D(in_channels=16, out_channels=3, up_block_types=("UpDecoderBlockCausal3D",) * 4,
                            block_out_channels=boc, layers_per_block=2, norm_num_groups=32, act_fn="silu",
                            time_compression_ratio=4, spatial_compression_ratio=8, mid_block_add_attention=True)
    sd = syn.synth_vae_state_dict(boc, seed=0)
    dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    print("reference DecoderCausal3D.load_state_dict(strict=True):", dec.load_state_dict(dsd, strict=True))
    dec.eval()
    pq = nn.Conv3d(16, 16, kernel_size=1)
    pq.weight.copy_(sd["post_quant_conv.weight"]), pq.bias.copy_(sd["post_quant_conv.bias"])
    z = U((1, 16, 3, 4, 4), "gv.z", 1) * s3
    save("vae_decoder_tile", z=z, y=dec(pq(z)), block_out_channels=np.array(boc))

    # ---- blends + tiled decode on a toy latent through the reference's own tiling code (tile 4 lat px / 2+1 lat frames)
    T = tiling_methods()
    ae = types.SimpleNamespace(post_quant_conv=pq, decoder=dec, use_spatial_tiling=True, use_temporal_tiling=True,
                               tile_sample_min_tsize=8, tile_latent_min_tsize=2, tile_sample_min_size=32,
                               tile_latent_min_size=4, tile_overlap_factor=0.25)
    for n in ("blend_v", "blend_h", "blend_t", "spatial_tiled_decode", "temporal_tiled_decode"):
        setattr(ae, n, types.MethodType(T[n], ae))
    a, bb = U((1, 3, 4, 6, 5), "gv.bl.a", 1), U((1, 3, 4, 6, 5), "gv.bl.b", 1)
    save("vae_blend", a=a, b=bb, v=ae.blend_v(a.clone(), bb.clone(), 4), h=ae.blend_h(a.clone(), bb.clone(), 3),
         t=ae.blend_t(a.clone(), bb.clone(), 2))
    zt = U((1, 16, 4, 7, 6), "gv.zt", 1) * s3      # 2 temporal x (2 x 2) spatial tiles, ragged edges
    y = types.MethodType(T["_decode"], ae)(zt, return_dict=True).sample
    ys = ae.spatial_tiled_decode(zt[:, :, :2], return_dict=True).sample
    save("vae_tiled_decode", z=zt, y=y, y_spatial_only=ys, tile=np.array([8, 2, 32, 4]))


if __name__ == "__main__":
    main()
