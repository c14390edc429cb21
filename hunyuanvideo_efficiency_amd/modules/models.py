"""Host-side mirror of hyvideo/modules/models.py (reference) for the MI355X kernels.

Same class names, constructor keywords, forward signatures and state-dict key names as the
reference (HYVideoDiffusionTransformer models.py:396-581, MMDoubleStreamBlock :21-252,
MMSingleStreamBlock :255-393), so `load_state_dict(strict=True)` accepts reference checkpoints and
`parallelize_transformer` can install `hybrid_seq_parallel_attn` on the blocks.  The modules only HOLD
parameters; every forward is a fixed sequence of C-ABI kernel launches (ops.py) on preallocated
workspaces - there is no torch arithmetic and no CPU path.

Data layout in HBM (one sample, S = S_img + S_txt tokens, d = hidden):
  x      [S, d]      bf16  residual stream; img = rows [0,S_img), txt = rows [S_img,S)  (no torch.cat)
  xmod   [S, d]      bf16  LayerNorm+modulate output (GEMM A operand)
  qkv    [S, 3d]     bf16  fused q|k|v rows; q,k are normalised/rotated in place; attention reads strided
  cat    [S, d+4d]   bf16  [attn | gelu(mlp)] of the single block = A operand of linear2; the double block
                           uses cols [0,d) for attn and cols [d,5d) as the MLP hidden buffer
Numeric contract (production autocast-bf16 semantics): see oracle/dit_ref.py Prec(True).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from .attenion import get_cu_seqlens, n_valid_text, segment_attention_, pad_segment_attention_
from .token_refiner import SingleTokenRefiner
from .layers import ParamLinear, ParamNormWeight, ModulateDiT, MLP, MLPEmbedder, TimestepEmbedder, PatchEmbed, FinalLayer

BF16 = torch.bfloat16


class _Workspace:
    """Preallocated activation buffers for one (S_img, S_txt, d, mlp) problem size."""

    def __init__(self, s_img: int, s_txt: int, d: int, mlp: int, device):
        s = s_img + s_txt
        self.key = (s_img, s_txt, d, mlp, str(device))
        self.x = torch.empty(s, d, dtype=BF16, device=device)
        self.xmod = torch.empty(s, d, dtype=BF16, device=device)
        self.qkv = torch.empty(s, 3 * d, dtype=BF16, device=device)
        self.cat = torch.empty(s, d + mlp, dtype=BF16, device=device)
        self.patches = None
        self.final = None
        # FP8-MFMA path (allocated on first use): per-token quantised A operands + their row scales
        self.xq = self.xs = self.catq = self.cats = None

    def fp8_buffers(self):
        if self.xq is None:
            s, d = self.x.shape
            dev = self.x.device
            self.xq = torch.empty(s, d, dtype=torch.float8_e4m3fn, device=dev)
            self.xs = torch.empty(s, dtype=torch.float32, device=dev)
            self.catq = torch.empty(s, self.cat.shape[1], dtype=torch.float8_e4m3fn, device=dev)
            self.cats = torch.empty(s, dtype=torch.float32, device=dev)
        return self


def _use_fp8_mfma(block, *layers) -> bool:
    """FP8 MFMA GEMMs for this block: opted in (fp8_optimization.enable_fp8_mfma), every Linear involved stores e4m3 weights and
    has a reduction length hv_gemm_fp8 takes (K % 128 == 0, K >= 384: three 128-byte K-tiles); otherwise the block keeps the
    reference's weight-only semantics (per-call dequantisation + bf16 MFMA)."""
    return bool(getattr(block, "fp8_mfma", False)) and all(
        l.weight.dtype == torch.float8_e4m3fn and l.in_features >= 384 and l.in_features % 128 == 0 for l in layers)


def _ln(ws: _Workspace, lo: int, hi: int, shift, scale, fp8: bool):
    """K1 on rows [lo, hi): bf16 into ws.xmod, or (FP8-MFMA path) e4m3 + row scales into ws.xq / ws.xs."""
    if fp8:
        ws.fp8_buffers()
        ops.ln_modulate_fp8(ws.x[lo:hi], shift, scale, out_q=ws.xq[lo:hi], out_scale=ws.xs[lo:hi])
    else:
        ops.ln_modulate(ws.x[lo:hi], shift, scale, out=ws.xmod[lo:hi])


def _gemm(ws: _Workspace, layer, src: str, lo: int, hi: int, fp8: bool, wrows: Optional[slice] = None, **epi):
    """One block Linear on rows [lo, hi).  `src` names the A operand: "xmod" (the LayerNorm output of _ln), "attn" = cat[:, :d],
    "hid" = cat[:, d:], "cat" = the whole [attn | mlp] row.  bf16 path: hv_gemm_bf16 on layer.w() (for an FP8 layer the weight
    is dequantised per call - the reference's semantics).  FP8-MFMA path: the stored e4m3 weight goes to the matrix cores as is and
    the A operand is quantised per token (LayerNorm outputs already are; the others are quantised here)."""
    d = ws.x.shape[1]
    cols = {"xmod": None, "attn": slice(0, d), "hid": slice(d, None), "cat": slice(None)}[src]
    b = layer.bias if wrows is None or layer.bias is None else layer.bias[wrows]
    if not fp8:
        a = ws.xmod[lo:hi] if src == "xmod" else ws.cat[lo:hi, cols]
        w = layer.w()
        return ops.gemm(a, w if wrows is None else w[wrows], b, **epi)
    ws.fp8_buffers()
    if src == "xmod":
        aq, asc = ws.xq[lo:hi], ws.xs[lo:hi]
    else:
        aq, asc = ops.quant_rows_fp8(ws.cat[lo:hi, cols], out_q=ws.catq[lo:hi, cols], out_scale=ws.cats[lo:hi])
    w8 = layer.weight if wrows is None else layer.weight[wrows]
    return ops.gemm_fp8(aq, asc, w8, layer.fp8_scale, b, **epi)


def _sp_chunked_qkv(sp, ws: _Workspace, layer, fp8, q_norm_w, k_norm_w, cos, sin, s_img: int, cu1: int, n_rope: int, H: int, d: int):
    """Sequence-parallel form of "QKV GEMM -> RMSNorm/RoPE -> all-to-all" for the local image rows [0, s_img): q, k and v are
    produced by three column-chunk MFMA GEMMs (rows [c*d, (c+1)*d) of the fused weight) and each chunk is handed to the
    Ulysses object as soon as it is normalised, so its RCCL all-to-all overlaps the next chunk's GEMM.  The text rows
    [s_img, cu1) of ws.qkv (joint tensors, replicated on every rank) must already be final."""
    ld = ws.qkv.stride(0)
    sp.begin(s_img, cu1 - s_img, H, ws.qkv.device)
    for c, which, norm_w in ((0, "q", q_norm_w), (1, "k", k_norm_w), (2, "v", None)):
        # where the chunk is produced: the Ulysses object's own operand rows when it has nothing to exchange (one rank), else the
        # fused-QKV workspace, from which send() packs it per peer
        dst = sp.chunk_dst(which) if hasattr(sp, "chunk_dst") else None
        chunk = dst if dst is not None else ws.qkv[:s_img, c * d:(c + 1) * d]
        _gemm(ws, layer, "xmod", 0, s_img, fp8, wrows=slice(c * d, (c + 1) * d), out=chunk)
        joint = ws.qkv[s_img:cu1, c * d:(c + 1) * d]
        if norm_w is not None:
            # the kernel normalises 2 x (H/2) head vectors: with both halves given the same gain that is this chunk's H heads.
            # P > 1: it stores them straight into the exchange's send layout (head blocks per peer) - the pack copy is its store
            packed = sp.pack_dst(which) if dst is None and hasattr(sp, "pack_dst") else None
            ops.qknorm_rope_(chunk, norm_w, norm_w, cos, sin, n_rope, H // 2, (H // 2) * 128, out=packed)
            if packed is not None:
                sp.send_packed(which, joint, ld)
                continue
        sp.send(which, chunk, chunk.stride(0), joint, ld)


class MMDoubleStreamBlock(nn.Module):
    """Parameter holder + kernel sequence of the reference MMDoubleStreamBlock (models.py:21-252)."""

    def __init__(self, hidden_size: int, heads_num: int, mlp_width_ratio: float, mlp_act_type: str = "gelu_tanh",
                 qk_norm: bool = True, qk_norm_type: str = "rms", qkv_bias: bool = False,
                 dtype: Optional[torch.dtype] = None, device: Optional[torch.device] = None):
        super().__init__()
        if mlp_act_type != "gelu_tanh" or not qk_norm or qk_norm_type != "rms":
            raise NotImplementedError("only gelu_tanh MLP and RMS qk-norm have kernels (the shipped configuration)")
        fk = {"device": device, "dtype": dtype}
        self.deterministic = False
        self.heads_num = heads_num
        head_dim = hidden_size // heads_num
        mlp_hidden = int(hidden_size * mlp_width_ratio)
        for s in ("img", "txt"):
            setattr(self, f"{s}_mod", ModulateDiT(hidden_size, 6, **fk))
            setattr(self, f"{s}_attn_qkv", ParamLinear(hidden_size, hidden_size * 3, qkv_bias, **fk))
            setattr(self, f"{s}_attn_q_norm", ParamNormWeight(head_dim, **fk))
            setattr(self, f"{s}_attn_k_norm", ParamNormWeight(head_dim, **fk))
            setattr(self, f"{s}_attn_proj", ParamLinear(hidden_size, hidden_size, qkv_bias, **fk))
            setattr(self, f"{s}_mlp", MLP(hidden_size, mlp_hidden, **fk))
        self.hybrid_seq_parallel_attn = None

    def enable_deterministic(self):
        self.deterministic = True

    def disable_deterministic(self):
        self.deterministic = False

    def run(self, ws: _Workspace, s_img: int, s_txt: int, vec: torch.Tensor, cu1: int,
            cos: Optional[torch.Tensor], sin: Optional[torch.Tensor]):
        d = ws.x.shape[1]
        H = self.heads_num
        # txt first: under sequence parallelism its (replicated) q/k/v rows are the joint tensors of the exchange
        streams = (("txt", s_img, s_img + s_txt, 0), ("img", 0, s_img, s_img if cos is not None else 0))
        sp = self.hybrid_seq_parallel_attn
        overlap = sp is not None and hasattr(sp, "send") and H % 2 == 0
        mods = {}
        fp8 = {s: _use_fp8_mfma(self, getattr(self, f"{s}_attn_qkv"), getattr(self, f"{s}_attn_proj"), getattr(self, f"{s}_mlp").fc1,
                                getattr(self, f"{s}_mlp").fc2) for s in ("img", "txt")}
        for s, lo, hi, n_rope in streams:
            mod = getattr(self, f"{s}_mod")
            m = ops.linear_smallm(vec, mod.linear.w(), mod.linear.bias, silu_in=True)  # [1, 6d]
            mods[s] = [m[0, i * d:(i + 1) * d] for i in range(6)]   # shift1, scale1, gate1, shift2, scale2, gate2
            sh1, sc1 = mods[s][0], mods[s][1]
            _ln(ws, lo, hi, sh1, sc1, fp8[s])
            qkv_l = getattr(self, f"{s}_attn_qkv")
            qw, kw = getattr(self, f"{s}_attn_q_norm").weight, getattr(self, f"{s}_attn_k_norm").weight
            if overlap and s == "img":
                # q / k / v as three column-chunk GEMMs; each chunk's all-to-all travels while the next chunk's GEMM runs
                _sp_chunked_qkv(sp, ws, qkv_l, fp8[s], qw, kw, cos, sin, s_img, cu1, n_rope, H, d)
            else:
                _gemm(ws, qkv_l, "xmod", lo, hi, fp8[s], out=ws.qkv[lo:hi])
                ops.qknorm_rope_(ws.qkv[lo:hi], qw, kw, cos, sin, n_rope, H, d)
        segs = None
        if overlap:
            # the output exchange travels in row segments: segment i+1 is on the wire while the out-projection of segment i runs
            segs = sp.attend_async(ws.cat, ws.cat.stride(0))
            pad_segment_attention_(ws.qkv, ws.cat, cu1, H, d)
        else:
            segment_attention_(sp, ws.qkv, ws.cat, s_img, cu1, H, d)
        streams = (streams[1], streams[0])
        for s, lo, hi, _ in streams:
            _, _, g1, sh2, sc2, g2 = mods[s]
            proj, mlp = getattr(self, f"{s}_attn_proj"), getattr(self, f"{s}_mlp")
            x = ws.x[lo:hi]
            if s == "img" and segs is not None:
                for r0, r1, finish in segs:
                    finish()
                    r1 = min(r1, hi)
                    _gemm(ws, proj, "attn", r0, r1, fp8[s], out=ws.x[r0:r1], gate=g1, res=ws.x[r0:r1])
            else:
                _gemm(ws, proj, "attn", lo, hi, fp8[s], out=x, gate=g1, res=x)
            _ln(ws, lo, hi, sh2, sc2, fp8[s])
            _gemm(ws, mlp.fc1, "xmod", lo, hi, fp8[s], out=ws.cat[lo:hi, d:], act=ops.ACT_GELU_TANH)
            _gemm(ws, mlp.fc2, "hid", lo, hi, fp8[s], out=x, gate=g2, res=x)

    def forward(self, img, txt, vec, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis: tuple = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference call surface (models.py:132-142): img [1,S_img,d], txt [1,S_txt,d], vec [1,d]."""
        assert img.shape[0] == 1, "batch 1 (cfg-distilled model, SURVEY.md top)"
        s_img, s_txt, d = img.shape[1], txt.shape[1], img.shape[2]
        ws = _Workspace(s_img, s_txt, d, self.img_mlp.fc1.weight.shape[0], img.device)
        ws.x[:s_img].copy_(img[0])
        ws.x[s_img:].copy_(txt[0])
        cu1 = int(cu_seqlens_q[1])
        cos, sin = freqs_cis if freqs_cis is not None else (None, None)
        self.run(ws, s_img, s_txt, vec.to(BF16).contiguous(), cu1, cos, sin)
        return ws.x[None, :s_img], ws.x[None, s_img:]


class MMSingleStreamBlock(nn.Module):
    """Parameter holder + kernel sequence of the reference MMSingleStreamBlock (models.py:255-393)."""

    def __init__(self, hidden_size: int, heads_num: int, mlp_width_ratio: float = 4.0, mlp_act_type: str = "gelu_tanh",
                 qk_norm: bool = True, qk_norm_type: str = "rms", qk_scale: float = None,
                 dtype: Optional[torch.dtype] = None, device: Optional[torch.device] = None):
        super().__init__()
        if mlp_act_type != "gelu_tanh" or not qk_norm or qk_norm_type != "rms" or qk_scale is not None:
            raise NotImplementedError("only gelu_tanh MLP, RMS qk-norm and the default scale have kernels")
        fk = {"device": device, "dtype": dtype}
        self.deterministic = False
        self.hidden_size = hidden_size
        self.heads_num = heads_num
        head_dim = hidden_size // heads_num
        self.mlp_hidden_dim = int(hidden_size * mlp_width_ratio)
        self.scale = head_dim ** -0.5
        self.linear1 = ParamLinear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim, True, **fk)
        self.linear2 = ParamLinear(hidden_size + self.mlp_hidden_dim, hidden_size, True, **fk)
        self.q_norm = ParamNormWeight(head_dim, **fk)
        self.k_norm = ParamNormWeight(head_dim, **fk)
        self.modulation = ModulateDiT(hidden_size, 3, **fk)
        self.hybrid_seq_parallel_attn = None

    def enable_deterministic(self):
        self.deterministic = True

    def disable_deterministic(self):
        self.deterministic = False

    def run(self, ws: _Workspace, s_img: int, s_txt: int, vec: torch.Tensor, cu1: int,
            cos: Optional[torch.Tensor], sin: Optional[torch.Tensor]):
        d, H = self.hidden_size, self.heads_num
        m = ops.linear_smallm(vec, self.modulation.linear.w(), self.modulation.linear.bias, silu_in=True)
        shift, scale, gate = m[0, :d], m[0, d:2 * d], m[0, 2 * d:]
        fp8 = _use_fp8_mfma(self, self.linear1, self.linear2)
        s_all = ws.x.shape[0]
        _ln(ws, 0, s_all, shift, scale, fp8)
        sp = self.hybrid_seq_parallel_attn
        n_rope = s_img if cos is not None else 0
        if sp is not None and hasattr(sp, "send") and H % 2 == 0:
            # sequence parallel: q | k | v | mlp as four column-chunk GEMMs of linear1; the all-to-all of each attention chunk
            # runs on RCCL's stream under the following chunk's MFMA GEMM (the 12288-wide mlp chunk covers the v exchange)
            # text rows of all three chunks first (they are the joint tensors every send needs)
            _gemm(ws, self.linear1, "xmod", s_img, s_all, fp8, wrows=slice(0, 3 * d), out=ws.qkv[s_img:])
            ops.qknorm_rope_(ws.qkv[s_img:], self.q_norm.weight, self.k_norm.weight, None, None, 0, H, d)
            _sp_chunked_qkv(sp, ws, self.linear1, fp8, self.q_norm.weight, self.k_norm.weight, cos, sin, s_img, cu1, n_rope, H, d)
            _gemm(ws, self.linear1, "xmod", 0, s_all, fp8, wrows=slice(3 * d, None), out=ws.cat[:, d:], act=ops.ACT_GELU_TANH)
            segs = sp.attend_async(ws.cat, ws.cat.stride(0))
            pad_segment_attention_(ws.qkv, ws.cat, cu1, H, d)
            # linear2 by row segments as their attention columns arrive (the last one takes the text rows, valid and padding)
            for i, (r0, r1, finish) in enumerate(segs):
                finish()
                r1 = s_all if i == len(segs) - 1 else r1
                _gemm(ws, self.linear2, "cat", r0, r1, fp8, out=ws.x[r0:r1], gate=gate, res=ws.x[r0:r1])
            return
        else:
            # linear1: cols [0,3d) -> qkv ; cols [3d, 3d+mlp) -> GELU-tanh -> cat[:, d:]   (models.py:339-341,392)
            _gemm(ws, self.linear1, "xmod", 0, s_all, fp8, out=ws.qkv, n_split=3 * d, out1=ws.cat[:, d:], act1=ops.ACT_GELU_TANH)
            ops.qknorm_rope_(ws.qkv, self.q_norm.weight, self.k_norm.weight, cos, sin, n_rope, H, d)
            segment_attention_(sp, ws.qkv, ws.cat, s_img, cu1, H, d)
        _gemm(ws, self.linear2, "cat", 0, s_all, fp8, out=ws.x, gate=gate, res=ws.x)

    def forward(self, x, vec, txt_len, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis: Tuple[torch.Tensor, torch.Tensor] = None) -> torch.Tensor:
        """Reference call surface (models.py:326-336): x [1,S,d], vec [1,d]."""
        assert x.shape[0] == 1
        s, d = x.shape[1], x.shape[2]
        ws = _Workspace(s - txt_len, txt_len, d, self.mlp_hidden_dim, x.device)
        ws.x.copy_(x[0])
        cos, sin = freqs_cis if freqs_cis is not None else (None, None)
        self.run(ws, s - txt_len, txt_len, vec.to(BF16).contiguous(), int(cu_seqlens_q[1]), cos, sin)
        return ws.x[None]


class _Config(dict):
    __getattr__ = dict.__getitem__


class HYVideoDiffusionTransformer(nn.Module):
    """HunyuanVideo transformer backbone on MI355X kernels; constructor and forward as the reference
    (models.py:448-470,595-606).  `.config` carries the registered constructor arguments that the
    pipeline reads (pipeline_hunyuan_video.py:927: transformer.config.in_channels)."""

    def __init__(self, args: Any, patch_size: list = [1, 2, 2], in_channels: int = 4, out_channels: int = None,
                 hidden_size: int = 3072, heads_num: int = 24, mlp_width_ratio: float = 4.0, mlp_act_type: str = "gelu_tanh",
                 mm_double_blocks_depth: int = 20, mm_single_blocks_depth: int = 40, rope_dim_list: List[int] = [16, 56, 56],
                 qkv_bias: bool = True, qk_norm: bool = True, qk_norm_type: str = "rms", guidance_embed: bool = False,
                 text_projection: str = "single_refiner", use_attention_mask: bool = True,
                 dtype: Optional[torch.dtype] = None, device: Optional[torch.device] = None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.config = _Config(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
                              hidden_size=hidden_size, heads_num=heads_num, mlp_width_ratio=mlp_width_ratio,
                              mlp_act_type=mlp_act_type, mm_double_blocks_depth=mm_double_blocks_depth,
                              mm_single_blocks_depth=mm_single_blocks_depth, rope_dim_list=rope_dim_list,
                              qkv_bias=qkv_bias, qk_norm=qk_norm, qk_norm_type=qk_norm_type,
                              guidance_embed=guidance_embed, text_projection=text_projection,
                              use_attention_mask=use_attention_mask)
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.unpatchify_channels = self.out_channels
        self.guidance_embed = guidance_embed
        self.rope_dim_list = rope_dim_list
        self.use_attention_mask = use_attention_mask
        self.text_projection = text_projection
        self.text_states_dim = args.text_states_dim
        self.text_states_dim_2 = args.text_states_dim_2
        if hidden_size % heads_num != 0:
            raise ValueError(f"Hidden size {hidden_size} must be divisible by heads_num {heads_num}")
        pe_dim = hidden_size // heads_num
        if sum(rope_dim_list) != pe_dim:
            raise ValueError(f"Got {rope_dim_list} but expected positional dim {pe_dim}")
        if pe_dim != 128:
            raise NotImplementedError("kernels are built for head_dim 128 (HunyuanVideo: 3072/24)")
        if list(patch_size) != [1, 2, 2]:
            raise NotImplementedError("kernels are built for patch_size [1,2,2]")
        if text_projection != "single_refiner":
            raise NotImplementedError(f"Unsupported text_projection: {text_projection}")
        self.hidden_size = hidden_size
        self.heads_num = heads_num
        self.mlp_hidden = int(hidden_size * mlp_width_ratio)

        self.img_in = PatchEmbed(self.patch_size, self.in_channels, self.hidden_size, **fk)
        self.txt_in = SingleTokenRefiner(self.text_states_dim, hidden_size, heads_num, depth=2, **fk)
        self.time_in = TimestepEmbedder(self.hidden_size, **fk)
        self.vector_in = MLPEmbedder(self.text_states_dim_2, self.hidden_size, **fk)
        self.guidance_in = TimestepEmbedder(self.hidden_size, **fk) if guidance_embed else None
        self.double_blocks = nn.ModuleList([
            MMDoubleStreamBlock(self.hidden_size, self.heads_num, mlp_width_ratio=mlp_width_ratio, mlp_act_type=mlp_act_type,
                                qk_norm=qk_norm, qk_norm_type=qk_norm_type, qkv_bias=qkv_bias, **fk)
            for _ in range(mm_double_blocks_depth)])
        self.single_blocks = nn.ModuleList([
            MMSingleStreamBlock(self.hidden_size, self.heads_num, mlp_width_ratio=mlp_width_ratio, mlp_act_type=mlp_act_type,
                                qk_norm=qk_norm, qk_norm_type=qk_norm_type, **fk)
            for _ in range(mm_single_blocks_depth)])
        self.final_layer = FinalLayer(self.hidden_size, self.patch_size, self.out_channels, **fk)
        self._ws: Optional[_Workspace] = None

    @property
    def dtype(self):
        return self.img_in.proj.weight.dtype   # (block Linears may hold float8_e4m3fn after convert_fp8_linear)

    def enable_deterministic(self):
        for b in list(self.double_blocks) + list(self.single_blocks):
            b.enable_deterministic()

    def disable_deterministic(self):
        for b in list(self.double_blocks) + list(self.single_blocks):
            b.disable_deterministic()

    def _workspace(self, s_img: int, s_txt: int, device) -> _Workspace:
        key = (s_img, s_txt, self.hidden_size, self.mlp_hidden, str(device))
        if self._ws is None or self._ws.key != key:
            self._ws = None
            self._ws = _Workspace(s_img, s_txt, self.hidden_size, self.mlp_hidden, device)
        return self._ws

    def forward(self, x: torch.Tensor, t: torch.Tensor, text_states: torch.Tensor = None, text_mask: torch.Tensor = None,
                text_states_2: Optional[torch.Tensor] = None, freqs_cos: Optional[torch.Tensor] = None,
                freqs_sin: Optional[torch.Tensor] = None, guidance: torch.Tensor = None,
                return_dict: bool = True) -> Union[torch.Tensor, Dict[str, torch.Tensor]]:
        """models.py:595-695 of the reference.  Batch B > 1 (the pipeline's classifier-free-guidance batch [uncond | cond],
        pipeline_hunyuan_video.py:966-1019; the non-distilled "HYVideo-T/2") runs as B passes through the SAME activation workspace:
        the reference's batched forward is B independent samples (per-sample cu_seqlens attention, per-sample modulation), and at
        S = 119,056 one sample already fills the chip, so a batched launch would buy nothing and double the 8 GB workspace."""
        if self.dtype != BF16:
            raise TypeError(f"parameters must be bf16 (got {self.dtype}); build with dtype=torch.bfloat16 or call .to()")
        B = x.shape[0]
        if B == 1:
            out = self._forward_one(x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, 0)
        else:
            t = t.reshape(-1)
            if t.numel() == 1:
                t = t.expand(B)
            outs = []
            for b in range(B):
                g_b = None if guidance is None else guidance.reshape(-1)[b:b + 1] if guidance.numel() > 1 else guidance
                outs.append(self._forward_one(x[b:b + 1], t[b:b + 1], text_states, text_mask,
                                              None if text_states_2 is None else text_states_2[b:b + 1], freqs_cos, freqs_sin, g_b, b))
            out = torch.cat(outs, 0)
        if return_dict:
            return {"x": out}
        return out

    def _forward_one(self, x, t, text_states, text_mask, text_states_2, freqs_cos, freqs_sin, guidance, b: int) -> torch.Tensor:
        """One sample: x [1,C,T,H,W]; text_states / text_mask are the WHOLE batch tensors (the refiner cache hangs on them), row b is used."""
        dev = x.device
        _, _, ot, oh, ow = x.shape
        tt, th, tw = ot // self.patch_size[0], oh // self.patch_size[1], ow // self.patch_size[2]
        s_img, s_txt, d = tt * th * tw, text_states.shape[1], self.hidden_size
        ws = self._workspace(s_img, s_txt, dev)

        # ---- modulation vector (models.py:618-631)
        t32 = t.reshape(-1).to(torch.float32)
        vec = self.time_in.run(t32)
        vec = self.vector_in.run(text_states_2.to(BF16).reshape(1, -1), addend=vec)
        if self.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            vec = self.guidance_in.run(guidance.reshape(-1).to(torch.float32), addend=vec)

        # ---- embed image and text (models.py:634-642)
        self.img_in.run(x[0].to(torch.float32).contiguous(), ws, s_img)
        mask_all = text_mask if self.use_attention_mask else None
        # per-prompt cache of the refiner's timestep-independent prefix: stashed on the caller's text_states tensor OBJECT (the
        # pipeline passes the same tensor every step; never keyed on an address), one entry per batch row, and valid only for the
        # SAME mask object at the same _version and for unmodified refiner parameters (every parameter's _version enters the key:
        # an in-place update of any of them - optimizer step, load_state_dict - drops the cache; replacing a parameter's .data
        # without a version bump is not detectable and needs `del text_states._hv_txt_cache`: inference-only use)
        tc_all = getattr(text_states, "_hv_txt_cache", None)
        key = (text_states._version, None if mask_all is None else mask_all._version, id(self.txt_in),
               tuple(p._version for p in self.txt_in.parameters()))
        if tc_all is None or tc_all.get("key") != key or tc_all.get("mask") is not mask_all:
            tc_all = {"key": key, "mask": mask_all, "rows": {}}
            try:
                text_states._hv_txt_cache = tc_all
            except (AttributeError, RuntimeError):
                pass
        tc = tc_all["rows"].get(b)
        if tc is None:
            tc = tc_all["rows"][b] = {"text_bf16": text_states[b].to(BF16).contiguous(),
                                      "mask": None if mask_all is None else (mask_all if mask_all.shape[0] == 1 else mask_all[b:b + 1]),
                                      "mask_cu": None if text_mask is None else (text_mask if text_mask.shape[0] == 1 else text_mask[b:b + 1])}
        mask = tc["mask"]
        self.txt_in.run(tc["text_bf16"], t32, mask, out=ws.x[s_img:], cache=tc)

        # ---- cu_seqlens (models.py:648): segment 1 = img + valid text, segment 2 = padding text
        # (the row's mask view is kept in the cache entry: n_valid_text stashes its host-side count on that tensor object)
        if text_mask is not None and tc.get("mask_cu_src") is not text_mask:
            tc["mask_cu"] = text_mask if text_mask.shape[0] == 1 else text_mask[b:b + 1]
            tc["mask_cu_src"] = text_mask
        cu1 = s_img + (n_valid_text(tc["mask_cu"]) if text_mask is not None else s_txt)
        cos = freqs_cos.to(device=dev, dtype=torch.float32).contiguous() if freqs_cos is not None else None
        sin = freqs_sin.to(device=dev, dtype=torch.float32).contiguous() if freqs_sin is not None else None

        for block in self.double_blocks:
            block.run(ws, s_img, s_txt, vec, cu1, cos, sin)
        for block in self.single_blocks:
            block.run(ws, s_img, s_txt, vec, cu1, cos, sin)

        img = self.final_layer.run(ws, s_img, vec)                       # [S_img, 64]
        out = ops.unpatchify(img, self.unpatchify_channels, ot, oh, ow)  # [C,T,H,W] bf16
        return out[None]

    def params_count(self):
        counts = {
            "double": sum(sum(p.numel() for p in m.parameters())
                          for b in self.double_blocks
                          for m in (b.img_attn_qkv, b.img_attn_proj, b.img_mlp, b.txt_attn_qkv, b.txt_attn_proj, b.txt_mlp)),
            "single": sum(sum(p.numel() for p in m.parameters()) for b in self.single_blocks for m in (b.linear1, b.linear2)),
            "total": sum(p.numel() for p in self.parameters()),
        }
        counts["attn+mlp"] = counts["double"] + counts["single"]
        return counts


HUNYUAN_VIDEO_CONFIG = {
    "HYVideo-T/2": dict(mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=[16, 56, 56],
                        hidden_size=3072, heads_num=24, mlp_width_ratio=4),
    "HYVideo-T/2-cfgdistill": dict(mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=[16, 56, 56],
                                   hidden_size=3072, heads_num=24, mlp_width_ratio=4, guidance_embed=True),
}
