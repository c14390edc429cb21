"""Decode ONE full-size VAE tile (17x32x32 latent -> 65x256x256) twice; for rocprofv3 --kernel-trace."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import synthetic as syn
from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
dev = 'cuda'
vae = AutoencoderKLCausal3D(device=dev)
with torch.no_grad():
    for k, p in vae.state_dict().items():
        p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))
z = syn.hashed_uniform((16, 17, 32, 32), "vae.z", 0, dev) * 1.7
for _ in range(2):
    vae._decode_tile(z)
torch.cuda.synchronize()
