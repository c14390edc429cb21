"""Per-rank attention cost of pure Ulysses-8 vs hybrid Ulysses-4 x Ring-2 / Ulysses-2 x Ring-4 at 720p x 129f on ONE card
(no communication: the K/V chunks are local slices) - what the per-chunk partials + merge cost over the single-pass kernel."""
import sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

S_IMG, N_TXT, H = 118800, 11, 24
dev = torch.device('cuda')
for U, R in ((8, 1), (4, 2), (2, 4), (1, 8)):
    hp = H // U
    s_u = S_IMG // R
    n_q = s_u + N_TXT
    q = torch.randn(n_q, hp * 128, device=dev).to(torch.bfloat16)
    kv = [(torch.randn(s_u + (N_TXT if t == 0 else 0), hp * 128, device=dev).to(torch.bfloat16),
           torch.randn(s_u + (N_TXT if t == 0 else 0), hp * 128, device=dev).to(torch.bfloat16)) for t in range(R)]
    out = torch.empty(n_q, hp * 128, dtype=torch.bfloat16, device=dev)
    if R == 1:
        ms = timeit(lambda: ops.attn_fwd(q, kv[0][0], kv[0][1], out, hp))
    else:
        sp = [ops.attn_suggest_splits(n_q, k.shape[0], hp) for k, _ in kv]
        parts = ops.AttnPartials(sum(sp), n_q, hp, dev)
        def run():
            parts.used = 0
            for (k, v), s in zip(kv, sp):
                ops.attn_partial(q, k, v, parts, hp, s)
            ops.attn_merge(parts, out)
        ms = timeit(run)
    flop = 4.0 * n_q * (S_IMG + N_TXT) * 128 * hp
    print(f"ulysses{U} x ring{R}: {ms:.2f} ms per rank  {flop / ms / 1e9:.0f} TFLOP/s", flush=True)
