"""Same-box A/B of the tiled decode on 1 / 2 / 3 HIP streams (decode_streams)."""
import sys, time, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import synthetic as syn
from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
dev = 'cuda'
vae = AutoencoderKLCausal3D(device=dev)
with torch.no_grad():
    for k, p in vae.state_dict().items():
        p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))
vae.enable_tiling()
z = syn.hashed_uniform((1, 16, 33, 90, 160), "vae.z", 0, dev) * 1.7
ref = None
for n in (1, 2, 1, 2, 3):
    vae.decode_streams = n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if ref is None: ref = y
    print(f"decode_streams={n}: {dt:.3f} s  identical={bool(torch.equal(y, ref))}  peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
