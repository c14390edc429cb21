"""CPU: the encode-side oracle (oracle/vae_enc_ref.py) against golden vectors produced by executing the reference's
hyvideo/vae code (tools/make_golden_vae_enc.py): strided causal conv, EncoderCausal3D + quant_conv + posterior, the tiled
encodes with blending, and the fork's t_ops hooks on encoder and decoder.  fp32 oracle vs fp32 reference: 1e-4."""
import json

import torch

from hunyuanvideo_efficiency_amd import synthetic as syn
from oracle import vae_enc_ref as E
from oracle import vae_ref as R


def close(a, b, tol=1e-4):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max() / b.abs().max())
    assert err < tol, err


def test_strided_causal_conv(golden):
    g = golden("vae_enc_downsample")
    for st in ((2, 2, 2), (1, 2, 2), (2, 1, 1), (1, 1, 1)):
        close(E.causal_conv3d_strided(g["x"], g["w"], g["b"], st, R.FP32), g["y" + "".join(map(str, st))], 1e-5)


def test_encoder_tile_posterior_and_recon(golden):
    g = golden("vae_enc_tile")
    boc = tuple(g["block_out_channels"].tolist())
    sd = syn.synth_vae_state_dict(boc, seed=0, encoder=True)
    assert set(syn.vae_encoder_param_shapes(boc)) <= set(sd)
    m = E.encode_tile(sd, g["x"], boc, R.FP32)
    close(m, g["moments"])
    mean, logvar, std = E.posterior(m)
    close(mean, g["mean"]), close(std, g["std"])
    close(E.posterior_kl(m), g["kl"])
    close(E.vae_forward(sd, g["x"], boc, R.FP32), g["recon"])
    noise = syn.hashed_uniform(tuple(mean.shape), "noise", 0)
    assert torch.equal(E.posterior_sample(m, noise), mean + std * noise)
    # fp16-emulated contract stays within fp16 drift of the fp32 reference
    close(E.encode_tile(sd, g["x"], boc, R.Prec(True)), g["moments"], 2e-2)


def test_tiled_encode(golden):
    g = golden("vae_enc_tiled")
    boc = (32, 64, 128, 128)
    sd = syn.synth_vae_state_dict(boc, seed=0, encoder=True)
    ts, tl, ss, sl = g["tile"].tolist()
    tp = R.TileParams(sample_size=ss, sample_tsize=ts, n_blocks=4)
    assert (tp.tile_latent_min_tsize, tp.tile_latent_min_size) == (tl, sl)
    close(E.encode(sd, g["x"], boc, tp, R.FP32, tiling=True), g["moments"])
    close(E.spatial_tiled_encode(sd, g["x"][:, :, :5], boc, tp, R.FP32), g["moments_spatial_only"])


def test_t_ops_encoder_and_decoder(golden):
    g = golden("vae_enc_tops")
    t_ops = json.loads(bytes(g["t_ops_json"].numpy().tobytes()).decode())
    boc = (32, 64, 128, 128)
    sd = syn.synth_vae_state_dict(boc, seed=0, encoder=True)
    m = E.encode_tile(sd, g["x"], boc, R.FP32, t_ops)
    close(m, g["moments"])
    close(E.vae_forward(sd, g["x"], boc, R.FP32, t_ops), g["recon"])
    # leaf semantics of the two temporal ops
    x = syn.hashed_uniform((1, 2, 5, 2, 2), "tp", 0)
    p = E.t_pool(x, 3, 2, R.FP32)
    assert p.shape[2] == 3
    assert torch.allclose(p[:, :, 0], x[:, :, 0]) and torch.allclose(p[:, :, 1], (x[:, :, 0] + x[:, :, 1] + x[:, :, 2]) / 3, atol=1e-6)
    u = E.t_interp(x, 2)
    assert u.shape[2] == 10 and torch.equal(u[:, :, 3], x[:, :, 1])
