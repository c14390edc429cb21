"""One process, a handful of launches of ONE hot kernel at its production shape - the target of the rocprofv3 PMC passes of
tools/profile_r03.sh (the program goes directly after `--`; bench.py itself cannot be profiled with --pmc).
usage: profile_r03_targets.py linear1|linear2|conv128|subpixel256|attn
Prints one line `ALGO <json>` with the algorithmic bytes / flops per launch so the parser does not restate shapes."""
import json, math, sys, torch
sys.path.insert(0, '.')
from hunyuanvideo_efficiency_amd import ops, synthetic as syn
which = sys.argv[1]
dev, BF16, F16 = 'cuda', torch.bfloat16, torch.float16
S, D = 119056, 3072
def rnd(shape, dtype, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)
if which == "linear1":      # single-stream block: [S,3072] x [21504,3072]^T -> qkv [S,9216] | GELU -> cat[:,3072:]  (models.py:339-341)
    x, w, b = rnd((S, D), BF16), rnd((7 * D, D), BF16, 0.02), rnd((7 * D,), BF16)
    qkv, cat = torch.empty(S, 3 * D, dtype=BF16, device=dev), torch.empty(S, 5 * D, dtype=BF16, device=dev)
    run = lambda: ops.gemm(x, w, b, out=qkv, n_split=3 * D, out1=cat[:, D:], act1=ops.ACT_GELU_TANH)
    algo = dict(kernel="gemm8_kernel", flop=2.0 * S * 7 * D * D, bytes=2.0 * (S * D + 7 * D * D + S * 7 * D), shape=f"M={S} N={7*D} K={D}")
elif which == "linear2":    # [S,15360] x [3072,15360]^T + gate*y + residual, in place (models.py:392-393)
    cat, w, b, g = rnd((S, 5 * D), BF16), rnd((D, 5 * D), BF16, 0.01), rnd((D,), BF16), rnd((D,), BF16, 0.5)
    x = rnd((S, D), BF16)
    run = lambda: ops.gemm(cat, w, b, out=x, gate=g, res=x)
    algo = dict(kernel="gemm8_kernel", flop=2.0 * S * D * 5 * D, bytes=2.0 * (S * 5 * D + D * 5 * D + 2 * S * D), shape=f"M={S} N={D} K={5*D}")
elif which in ("conv128", "subpixel256"):
    from hunyuanvideo_efficiency_amd import vae_ops as V
    T, H, W = 65, 256, 256
    if which == "conv128":  # up_blocks.3 resnet conv: 128 -> 128 @ 65x256x256 + residual + epilogue GroupNorm statistics
        c = 128
        x, wt, b = rnd((T * H * W, c), F16), rnd((c, 27, c), F16, 1 / math.sqrt(27 * c)), rnd((c,), F16, 0.1)
        res = rnd((T * H * W, c), F16)
        run = lambda: V.conv3d_causal(x, wt, b, T, H, W, c, c, res=res, gn_stats=True)
        M = T * H * W
        algo = dict(kernel="conv128s_kernel", flop=2.0 * 27 * c * c * M, bytes=2.0 * (3 * M * c + 27 * c * c), shape=f"128->128 @ {T}x{H}x{W}")
    else:                   # up_blocks.2 upsampler in sub-pixel form: 256 -> 256, source 65x128x128 -> 65x256x256
        c, sT, sH, sW = 256, 65, 128, 128
        x = rnd((sT * sH * sW, c), F16)
        w5, b = rnd((c, c, 3, 3, 3), F16, 1 / math.sqrt(27 * c)), rnd((c,), F16, 0.1)
        w_sub, table, ntap = V.subpixel_weights(w5, False, "fast")
        run = lambda: V.conv3d_upsampled_subpixel(x, w_sub, table, ntap, b, sT, sH, sW, c, c, False, gn_stats=True)
        M = T * H * W
        algo = dict(kernel="gemm8_kernel", flop=2.0 * ntap * c * c * M, bytes=2.0 * (sT * sH * sW * c + M * c + 4 * ntap * c * c),
                    shape=f"sub-pixel 256->256, {sT}x{sH}x{sW} -> {T}x{H}x{W}, {ntap} taps")
elif which == "attn":
    n = 118811
    qkv = rnd((n, 3 * D), BF16)
    cat = torch.empty(n, 5 * D, dtype=BF16, device=dev)
    run = lambda: ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], cat[:, :D], 24)
    algo = dict(kernel="attn_fwd_kernel", flop=4.0 * n * n * 128 * 24, bytes=2.0 * 4 * n * D, shape=f"n_q=n_kv={n}, 24 heads")
else:
    raise SystemExit(f"unknown target {which}")
for _ in range(3):
    run()
torch.cuda.synchronize()
print("ALGO " + json.dumps(algo))
