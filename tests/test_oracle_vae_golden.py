"""CPU: the VAE-decode oracle (oracle/vae_ref.py) against golden vectors produced by executing the reference's
hyvideo/vae code (tools/make_golden_vae.py).  fp32 vs fp32."""
import torch

from hunyuanvideo_efficiency_amd import synthetic as syn
from oracle import vae_ref as V

P = V.FP32


def close(a, b, rtol=2e-5, atol=2e-5):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def test_causal_conv(golden):
    g = golden("vae_causal_conv")
    close(V.causal_conv3d(g["x"], g["w"], g["b"], P), g["y"])
    # causality: output frame t must not depend on input frames > t
    x2 = g["x"].clone()
    x2[:, :, 2] += 1.0
    y2 = V.causal_conv3d(x2, g["w"], g["b"], P)
    assert torch.equal(y2[:, :, :2], V.causal_conv3d(g["x"], g["w"], g["b"], P)[:, :, :2])


def test_upsample(golden):
    g = golden("vae_upsample")
    assert torch.equal(V.upsample_causal(g["x"], (2, 2, 2)), g["y222"])
    assert torch.equal(V.upsample_causal(g["x"], (1, 2, 2)), g["y122"])
    assert torch.equal(V.upsample_causal(g["x"][:, :, :1], (2, 2, 2)), g["y222_t1"])
    assert g["y222"].shape[2] == 2 * g["x"].shape[2] - 1


def test_resnet_block(golden):
    g = golden("vae_resnet")
    names = ["norm1.weight", "norm1.bias", "conv1.conv.weight", "conv1.conv.bias", "norm2.weight", "norm2.bias",
             "conv2.conv.weight", "conv2.conv.bias", "conv_shortcut.conv.weight", "conv_shortcut.conv.bias"]
    shapes = {"norm1.weight": (32,), "norm1.bias": (32,), "conv1.conv.weight": (64, 32, 3, 3, 3), "conv1.conv.bias": (64,),
              "norm2.weight": (64,), "norm2.bias": (64,), "conv2.conv.weight": (64, 64, 3, 3, 3), "conv2.conv.bias": (64,),
              "conv_shortcut.conv.weight": (64, 32, 1, 1, 1), "conv_shortcut.conv.bias": (64,)}
    sd = {"r." + k: syn.synth_param("gv.res." + k, shapes[k], 1) for k in names}
    close(V.resnet_block(sd, "r.", g["x"], P), g["y"], rtol=1e-4, atol=1e-4)


def test_causal_mask(golden):
    g = golden("vae_causal_mask")
    m = V.causal_frame_mask(3, 4)
    assert torch.equal(m, g["mask"] == 0) and torch.equal(~m, torch.isinf(g["mask"]))


def test_decoder_tile(golden):
    g = golden("vae_decoder_tile")
    boc = tuple(g["block_out_channels"].tolist())
    sd = syn.synth_vae_state_dict(boc, seed=0)
    y = V.decode_tile(sd, g["z"], boc, P)
    assert y.shape == g["y"].shape == (1, 3, 9, 32, 32)
    close(y, g["y"], rtol=2e-4, atol=2e-4)


def test_blends(golden):
    g = golden("vae_blend")
    close(V._blend(g["a"].clone(), g["b"].clone(), 4, 3, P), g["v"], atol=1e-6)
    close(V._blend(g["a"].clone(), g["b"].clone(), 3, 4, P), g["h"], atol=1e-6)
    close(V._blend(g["a"].clone(), g["b"].clone(), 2, 2, P), g["t"], atol=1e-6)


def test_tiled_decode(golden):
    g = golden("vae_tiled_decode")
    boc = (32, 64, 128, 128)
    sd = syn.synth_vae_state_dict(boc, seed=0)
    ts, tl, ss, sl = g["tile"].tolist()
    tp = V.TileParams(sample_size=ss, sample_tsize=ts, n_blocks=4)
    assert (tp.tile_latent_min_tsize, tp.tile_latent_min_size) == (tl, sl)
    y = V.decode(sd, g["z"], boc, tp, P, tiling=True)
    assert y.shape == g["y"].shape
    close(y, g["y"], rtol=3e-4, atol=3e-4)
    ys = V.spatial_tiled_decode(sd, g["z"][:, :, :2], boc, tp, P)
    close(ys, g["y_spatial_only"], rtol=3e-4, atol=3e-4)


def test_postprocess():
    x = torch.tensor([-3.0, -1.0, 0.0, 0.5, 1.0, 2.0])
    assert V.postprocess(x, P).tolist() == [0.0, 0.0, 0.5, 0.75, 1.0, 1.0]
