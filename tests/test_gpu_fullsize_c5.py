"""GPU parity at the size of BASELINE.json config 5 (720x1280x257f: S = 234,000 image + 256 text tokens = 234,256 rows) and, for
the row-wise kernels, also at config 3's S = 119,056 - by SAMPLED rows against the oracle (oracle/dit_ref.py, bf16-emulated
contract).  S = 234,256 is where the fused [S, 9216] bf16 buffer (4.32 GB) crosses 2^32 bytes and the [S, 15360] concat buffer
(7.2 GB) crosses 2^32 and 3 * 2^31: sampled rows sit on both sides of every 2^31-byte mark of the buffers a kernel touches, so a
32-bit byte offset anywhere in a kernel's addressing shows up as a wrong row.  Also: one double + one single block at the SHIPPED
width (d = 3072, 24 heads, S = 4,096 + 256) against oracle.dit_ref.double_block / single_block.
Reference shapes: /root/reference/tests/test_attention.py:48-49,72-75, hyvideo/modules/models.py:339-341,392-393."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from hunyuanvideo_efficiency_amd import synthetic as syn  # noqa: E402
from oracle import dit_ref as R  # noqa: E402

E = R.Prec(True)
DEV = "cuda"
BF16 = torch.bfloat16
D, H = 3072, 24
S_TXT, N_VALID = 256, 11
SIZES = {"c3": (33, 45, 80), "c5": (65, 45, 80)}      # latent token grid (T, H/2, W/2): 118,800 / 234,000 image tokens


@pytest.fixture(scope="module")
def ops():
    from hunyuanvideo_efficiency_amd import ops as _ops, _lib
    _lib.torch_ops()
    return _ops


def _u(shape, key, scale=1.0, dtype=BF16):
    return (syn.hashed_uniform(shape, key, 23, DEV) * (scale * math.sqrt(3.0))).to(dtype)


def _rows_around_marks(n_rows, row_bytes, count, key):
    """row indices: both ends, tile borders, `count` hashed rows, and the two rows on either side of every multiple of 2^31 bytes"""
    rows = {0, 1, 255, 256, n_rows - 257, n_rows - 256, n_rows - 65, n_rows - 64, n_rows - 2, n_rows - 1}
    for rb in (row_bytes if isinstance(row_bytes, (list, tuple)) else [row_bytes]):
        m = 1
        while m * (1 << 31) < n_rows * rb:
            r = (m * (1 << 31)) // rb
            rows.update(x for x in (r - 1, r, r + 1) if 0 <= x < n_rows)
            m += 1
    h = (syn.hashed_uniform((count,), key, 3) * 0.5 + 0.5).clamp(0, 0.999999)
    rows.update(int(v * n_rows) for v in h.tolist())
    return torch.tensor(sorted(rows))


def _attn_rows_ref(q_rows, k, v):
    return R.sdpa(q_rows[None, :, None, :], k[None, :, None, :], v[None, :, None, :], E)[0, :, 0]


def test_attention_config5_sampled_rows(ops):
    """hv_attn_fwd_bf16 at n_q = n_kv = 234,011 x 24 heads, q|k|v strided out of a real [234,256, 9216] fused buffer into the
    [234,256, 15360] concat buffer; sampled query rows (incl. both sides of the 2^31 / 2^32 / 3*2^31-byte marks of either buffer)
    x heads {0, 23} against a CPU softmax over all 234,011 keys.  Queries scaled so the softmax is peaked (outputs O(1))."""
    s_img = math.prod(SIZES["c5"])
    S, cu1 = s_img + S_TXT, s_img + N_VALID
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    assert qkv.numel() * 2 > (1 << 32)
    for c, (nm, sc) in enumerate((("q", 4.0), ("k", 1.0), ("v", 1.0))):
        for lo in range(0, S, 65536):       # generated in slabs: hashed_uniform materialises int64 index tensors
            hi = min(S, lo + 65536)
            qkv[lo:hi, c * D:(c + 1) * D] = _u((hi - lo, D), f"c5.{nm}.{lo}", sc)
    cat = torch.zeros(S, 5 * D, dtype=BF16, device=DEV)
    ops.attn_fwd(qkv[:cu1, :D], qkv[:cu1, D:2 * D], qkv[:cu1, 2 * D:], cat[:cu1, :D], H)
    torch.cuda.synchronize()
    assert float(cat[:, D:].abs().max()) == 0 and float(cat[cu1:].abs().max()) == 0, "attention wrote outside its rows / columns"
    rows = _rows_around_marks(cu1, [3 * D * 2, 5 * D * 2], 32, "c5.rows")
    rd = rows.to(DEV)
    for h in (0, 23):
        c = slice(h * 128, (h + 1) * 128)
        k = qkv[:cu1, D + h * 128:D + (h + 1) * 128].float().cpu()
        v = qkv[:cu1, 2 * D + h * 128:2 * D + (h + 1) * 128].float().cpu()
        ref = _attn_rows_ref(qkv[rd, c].float().cpu(), k, v)
        got = cat[rd, c].float().cpu()
        assert float(ref.abs().max()) > 0.3
        torch.testing.assert_close(got, ref, rtol=2 ** -7, atol=8e-3)


def test_gemm_config5_sampled_rows(ops):
    """hv_gemm_bf16 linear1 / linear2 of the single-stream block at M = 234,256 (models.py:339-341,392-393 of the reference): the
    split-column epilogue into the 4.32 GB qkv buffer and the 7.2 GB concat buffer, then K = 15,360 with gate + residual in place."""
    S = math.prod(SIZES["c5"]) + S_TXT
    x = torch.empty(S, D, dtype=BF16, device=DEV)
    for lo in range(0, S, 65536):
        hi = min(S, lo + 65536)
        x[lo:hi] = _u((hi - lo, D), f"g5.x.{lo}")
    w1, b1 = _u((7 * D, D), "g5.w1", 1 / math.sqrt(D)), _u((7 * D,), "g5.b1", 0.1)
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    cat = torch.zeros(S, 5 * D, dtype=BF16, device=DEV)
    ops.gemm(x, w1, b1, out=qkv, n_split=3 * D, out1=cat[:, D:], act1=ops.ACT_GELU_TANH)
    torch.cuda.synchronize()
    rows = _rows_around_marks(S, [D * 2, 3 * D * 2, 5 * D * 2], 96, "g5.rows")
    rd = rows.to(DEV)
    y = E.r(x[rd].float().cpu() @ w1.float().cpu().T + b1.float().cpu())
    torch.testing.assert_close(qkv[rd].float().cpu(), y[:, :3 * D], rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(cat[rd, D:].float().cpu(), R.gelu_tanh(y[:, 3 * D:], E), rtol=2 ** -7, atol=2e-2)
    assert float(cat[:, :D].abs().max()) == 0
    for lo in range(0, S, 65536):
        hi = min(S, lo + 65536)
        cat[lo:hi, :D] = _u((hi - lo, D), f"g5.attn.{lo}")
    w2, b2 = _u((D, 5 * D), "g5.w2", 1 / math.sqrt(5 * D)), _u((D,), "g5.b2", 0.1)
    gate = _u((D,), "g5.gate", 0.5)
    a_rows, x_rows = cat[rd].float().cpu(), x[rd].float().cpu()
    ops.gemm(cat, w2, b2, out=x, gate=gate, res=x)
    torch.cuda.synchronize()
    y2 = E.r(a_rows @ w2.float().cpu().T + b2.float().cpu())
    ref = R.gate_residual(x_rows[None], y2[None], gate.float().cpu()[None], E)[0]
    torch.testing.assert_close(x[rd].float().cpu(), ref, rtol=2 ** -7, atol=2e-2)


@pytest.mark.parametrize("size", ["c3", "c5"])
def test_ln_modulate_production_rows(ops, size):
    """hv_ln_modulate_bf16 over the whole residual stream [S, 3072] (one launch, as the single-stream block issues it)."""
    S = math.prod(SIZES[size]) + S_TXT
    x = torch.empty(S, D, dtype=BF16, device=DEV)
    for lo in range(0, S, 65536):
        hi = min(S, lo + 65536)
        x[lo:hi] = _u((hi - lo, D), f"ln.{size}.{lo}", 2.0)
    sh, sc = _u((D,), "ln.sh", 0.3), _u((D,), "ln.sc", 0.3)
    out = torch.full((S, D), 77.0, dtype=BF16, device=DEV)
    ops.ln_modulate(x, sh, sc, out=out)
    rows = _rows_around_marks(S, D * 2, 200, "ln.rows." + size)
    rd = rows.to(DEV)
    ref = E.r(R.ln_modulate(x[rd].float().cpu()[None], sh.float().cpu()[None], sc.float().cpu()[None], E))[0]
    torch.testing.assert_close(out[rd].float().cpu(), ref, rtol=2 ** -7, atol=2e-2)
    # every row was written exactly once: no row keeps the fill value, and a second launch reproduces the first
    assert not bool((out == 77.0).all(dim=1).any())
    again = ops.ln_modulate(x, sh, sc)
    assert torch.equal(again, out)


@pytest.mark.parametrize("size", ["c3", "c5"])
def test_qknorm_rope_production_rows(ops, size):
    """hv_qknorm_rope_bf16 in place on the fused [S, 9216] rows with the REAL RoPE tables of the latent grid (RoPE on the image rows
    only), and hv_qknorm_rope_scatter_bf16 of a [S_loc, 3072] q chunk into the Ulysses-8 send layout == the in-place result."""
    from hunyuanvideo_efficiency_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    grid = SIZES[size]
    s_img = math.prod(grid)
    S = s_img + S_TXT
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    for lo in range(0, S, 32768):
        hi = min(S, lo + 32768)
        qkv[lo:hi] = _u((hi - lo, 3 * D), f"qk.{size}.{lo}", 1.5)
    qw, kw = (1 + _u((128,), "qk.qw", 0.1).float()).to(BF16), (1 + _u((128,), "qk.kw", 0.1).float()).to(BF16)
    cos, sin = get_nd_rotary_pos_embed([16, 56, 56], list(grid), theta=256, use_real=True, device=DEV)
    cos, sin = cos.float().contiguous(), sin.float().contiguous()
    rows = _rows_around_marks(S, 3 * D * 2, 160, "qk.rows." + size)
    rows = torch.unique(torch.cat([rows, torch.tensor([s_img - 1, s_img, s_img + 1])]))     # the RoPE / no-RoPE border
    rd = rows.to(DEV)
    before = qkv[rd].float().cpu()
    q_chunk_src = qkv[:s_img // 8, :D].clone()        # a rank's local q chunk under Ulysses-8, before normalisation
    ops.qknorm_rope_(qkv, qw, kw, cos, sin, s_img, H, D)
    torch.cuda.synchronize()
    q, k, v = before.reshape(1, len(rows), 3, H, 128).unbind(2)
    qn, kn = R.rms_norm(q, qw.float().cpu(), E), R.rms_norm(k, kw.float().cpu(), E)
    is_img = rows < s_img
    ci = rows.clamp(max=s_img - 1).to(DEV)
    c_r, s_r = cos[ci].cpu(), sin[ci].cpu()

    def rope_rows(t):       # per-row tables: apply_rope broadcasts [n, 128] tables over heads
        return torch.where(is_img[None, :, None, None], R.apply_rope(t, c_r, s_r, E), t)

    ref = torch.stack([rope_rows(qn), rope_rows(kn), v], 2).reshape(len(rows), 3 * D)
    torch.testing.assert_close(qkv[rd].float().cpu(), ref, rtol=2 ** -7, atol=2e-2)
    assert torch.equal(qkv[rd, 2 * D:].float().cpu(), before[:, 2 * D:])       # v untouched
    # scatter form at the per-rank chunk shape: 24 q heads of the chunk as 2 x 12 with the same gain, 8 head blocks of 3 heads
    n_loc = s_img // 8
    buf = torch.full((8, n_loc, 3 * 128), 7.0, dtype=BF16, device=DEV)
    ops.qknorm_rope_(q_chunk_src, qw, qw, cos, sin, n_loc, H // 2, (H // 2) * 128, out=buf.permute(1, 0, 2))
    want = qkv[:n_loc, :D].reshape(n_loc, 8, 3 * 128).permute(1, 0, 2)
    assert torch.equal(buf, want)


def test_copy3d_patchify_unpatchify_euler_production(ops):
    """The bytes-moving kernels at config 5 size against torch indexing on the GPU (bit-exact): hv_copy3d_bf16 as the Ulysses-8
    pack of a v chunk and as a > 2^32-byte strided copy out of the fused buffer; hv_patchify_f32_bf16 / hv_unpatchify_bf16 on the
    [16, 65, 90, 160] latent; hv_euler_step_f32 on it."""
    T, Hh, Ww = 65, 90, 160
    s_img = T * (Hh // 2) * (Ww // 2)
    S = s_img + S_TXT
    qkv = torch.empty(S, 3 * D, dtype=BF16, device=DEV)
    for lo in range(0, S, 32768):
        hi = min(S, lo + 32768)
        qkv[lo:hi] = _u((hi - lo, 3 * D), f"cp.{lo}")
    # (a) whole v column block [S, 3072] out of the fused rows (source offsets up to 4.32 GB) as 8 head blocks [8][S][384]
    dst = torch.empty(8, S, 384, dtype=BF16, device=DEV)
    ops.copy3d(qkv[:, 2 * D:], dst, 8, S, 384, 384, 3 * D, S * 384, 384)
    assert torch.equal(dst, qkv[:, 2 * D:].reshape(S, 8, 384).permute(1, 0, 2))
    del dst
    # (b) and back into a wider row (the unpack direction): [8][S][384] -> cat[:, :3072]
    src = qkv[:, :D].reshape(S, 8, 384).permute(1, 0, 2).contiguous()
    cat = torch.zeros(S, 5 * D, dtype=BF16, device=DEV)
    ops.copy3d(src, cat, 8, S, 384, S * 384, 384, 384, 5 * D)
    assert torch.equal(cat[:, :D], qkv[:, :D]) and float(cat[:, D:].abs().max()) == 0
    del src, cat, qkv
    # patchify / unpatchify / euler on the full latent
    x = syn.hashed_uniform((16, T, Hh, Ww), "pu5.x", 0, DEV) * 1.7
    A = ops.patchify(x)
    ref = x.reshape(16, T, Hh // 2, 2, Ww // 2, 2).permute(1, 2, 4, 0, 3, 5).reshape(s_img, 64).to(BF16)
    assert torch.equal(A, ref)
    back = ops.unpatchify(A, 16, T, Hh, Ww)
    assert torch.equal(back, x.to(BF16))
    v = _u((1, 16, T, Hh, Ww), "pu5.v")
    s0 = x[None].clone()
    sig = torch.tensor([0.5, 0.4877])
    got = ops.euler_step_(s0.clone(), v, float(sig[1] - sig[0]))
    ref = R.euler_step(s0.cpu(), v.float().cpu(), sig, 0)
    torch.testing.assert_close(got.cpu(), ref, rtol=0, atol=1e-6)


def test_blocks_shipped_width_vs_oracle():
    """One MMDoubleStreamBlock + one MMSingleStreamBlock at d = 3072, 24 heads, mlp 12288, S = 4,096 + 256 (11 valid text tokens)
    through their reference call surfaces vs oracle.dit_ref.double_block / single_block in the bf16-emulated contract - the
    tolerances of test_gpu_model.py::test_block_call_surfaces_vs_oracle (2 bf16 ulps of the residual stream's range)."""
    from tests.oracle_checks import fullwidth_blocks
    r = fullwidth_blocks(DEV, prec=E)
    for name in ("double_img", "double_txt", "single"):
        got, ref = r[name]
        torch.testing.assert_close(got, ref, rtol=2 ** -6, atol=6e-2)
        assert float(ref.abs().max()) > 1.0
