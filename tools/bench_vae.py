"""VAE tiled decode benchmark at full topology (128,256,512,512)."""
import sys, time, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import synthetic as syn
from hunyuanvideo_efficiency_amd.vae import AutoencoderKLCausal3D
dev = 'cuda'
T, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (33, 90, 160)))
vae = AutoencoderKLCausal3D(device=dev)
with torch.no_grad():
    for k, p in vae.state_dict().items():
        p.copy_(syn.synth_param("vae." + k, tuple(p.shape), 0, dev).to(p.dtype))
vae.enable_tiling()
import os
if os.environ.get("VAE_STREAMS"):          # tools/trace_vae.sh: one stream, so that kernel durations in the trace are not co-run times
    vae.decode_streams = int(os.environ["VAE_STREAMS"])
z = syn.hashed_uniform((1, 16, T, H, W), "vae.z", 0, dev) * 1.7
# one tile first (timing of a single decoder call)
torch.cuda.synchronize(); t0 = time.perf_counter()
buf, t, h, w = vae._decode_tile(z[0, :, :17, :32, :32].float())
torch.cuda.synchronize(); print(f"first tile (incl. weight prep): {time.perf_counter()-t0:.3f} s -> {t}x{h}x{w}", flush=True)
t0 = time.perf_counter()
buf, t, h, w = vae._decode_tile(z[0, :, :17, :32, :32].float())
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one full tile 17x32x32: {dt*1e3:.1f} ms  (~73 TFLOP conv -> {73/dt/1e3:.2f} PFLOP/s)", flush=True)
t0 = time.perf_counter()
y = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"tiled decode latent {T}x{H}x{W} -> {tuple(y.shape)}: {dt:.2f} s; finite={bool(torch.isfinite(y).all())}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
