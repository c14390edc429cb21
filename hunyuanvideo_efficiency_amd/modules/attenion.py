"""Attention entry points with the reference's names and argument meaning (hyvideo/modules/attenion.py - the
module name keeps the reference's spelling so `from hyvideo.modules.attenion import attention` ports 1:1).

All modes run the gfx950 flash kernel (hv_attn_fwd_bf16); there is no torch SDPA/CPU path.  The varlen
semantics of mode="flash" (two segments per sample given by cu_seqlens, attenion.py:34-57,107-120) are
realised as one kernel launch per segment on strided views - no packing, no concatenation."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops

BF16 = torch.bfloat16
def n_valid_text(text_mask: torch.Tensor) -> int:
    """Number of valid text tokens of sample 0 (text_mask.sum(dim=1), attenion.py:45).  One host read per mask tensor
    OBJECT (the pipeline passes the same tensor every step): the count is stashed on the tensor itself together with its
    `_version`, never keyed on its address - a freed mask's address is recycled by the next mask, which must not inherit
    the previous prompt's count.  The mask must be a prefix mask, which is what right-padded tokenisation produces and
    what get_cu_seqlens itself assumes."""
    hit = getattr(text_mask, "_hv_n_valid", None)
    if hit is not None and hit[0] == text_mask._version:
        return hit[1]
    m = text_mask[0].to("cpu").to(torch.int64)
    n = int(m.sum())
    if not bool((m[:n] != 0).all()):
        raise NotImplementedError("text_mask must be a prefix mask (valid tokens first)")
    try:
        text_mask._hv_n_valid = (text_mask._version, n)
    except (AttributeError, RuntimeError):      # exotic tensor subclasses: just recompute next time
        pass
    return n


def get_cu_seqlens(text_mask: torch.Tensor, img_len: int) -> torch.Tensor:
    """attenion.py:34-57: int32 [2B+1] = per sample [.., i*max_len + img_len + n_valid_i, (i+1)*max_len].
    Returned on the mask's device (the reference hard-codes "cuda")."""
    batch_size = text_mask.shape[0]
    text_len = text_mask.sum(dim=1).to("cpu")
    max_len = text_mask.shape[1] + img_len
    cu = [0]
    for i in range(batch_size):
        cu.append(i * max_len + int(text_len[i]) + img_len)
        cu.append((i + 1) * max_len)
    return torch.tensor(cu, dtype=torch.int32, device=text_mask.device)


def _flat2d(t: torch.Tensor) -> torch.Tensor:
    """[S,H,D] (or [1,S,H,D]) -> [S, H*D] view; heads must be contiguous inside a token row."""
    if t.dim() == 4:
        assert t.shape[0] == 1
        t = t[0]
    s, h, d = t.shape
    assert t.stride(2) == 1 and t.stride(1) == d, "heads must be packed inside a token row"
    return t.as_strided((s, h * d), (t.stride(0), 1), t.storage_offset())


def attention(q, k, v, mode="flash", drop_rate=0, attn_mask=None, causal=False, cu_seqlens_q=None, cu_seqlens_kv=None,
              max_seqlen_q=None, max_seqlen_kv=None, batch_size=1):
    """Reference signature (attenion.py:60-95).  q,k,v: [b, s, a, d] bf16 on the GPU -> [b, s, a*d].
    mode "flash": per-segment attention over cu_seqlens; "torch"/"vanilla": one segment (mask-free, non-causal)."""
    if drop_rate != 0 or causal:
        raise NotImplementedError("inference path: dropout 0, non-causal")
    if attn_mask is not None:
        raise NotImplementedError("attn_mask: use cu_seqlens segments (the DiT path never passes a mask here)")
    b, s, a, d = q.shape
    if b != 1:
        raise NotImplementedError("batch size 1")
    out = torch.empty(b, s, a * d, dtype=BF16, device=q.device)
    q2, k2, v2 = _flat2d(q), _flat2d(k), _flat2d(v)
    if mode == "flash":
        bounds = [int(x) for x in cu_seqlens_q.tolist()]
        assert list(cu_seqlens_kv.tolist()) == bounds, "self-attention: q and kv segments coincide"
    elif mode in ("torch", "vanilla"):
        bounds = [0, s]
    else:
        raise NotImplementedError(f"Unsupported attention mode: {mode}")
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if hi > lo:
            ops.attn_fwd(q2[lo:hi], k2[lo:hi], v2[lo:hi], out[0, lo:hi], a)
    return out


def segment_attention_(hybrid_seq_parallel_attn, qkv: torch.Tensor, cat: torch.Tensor, s_img: int, cu1: int, heads: int, d: int):
    """Block-internal form: q|k|v live in the fused `qkv` rows [S, 3d]; output goes to cat[:, :d].
    Without SP: segment 1 = rows [0, cu1) (image + valid text), segment 2 = rows [cu1, S) (padding text)
    (attenion.py:34-57).  With SP the block's `hybrid_seq_parallel_attn` object runs the Ulysses exchange for
    the first segment exactly as parallel_attention does (attenion.py:159-212)."""
    s = qkv.shape[0]
    q, k, v, o = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], cat[:, :d]
    if hybrid_seq_parallel_attn is None:
        if cu1 > 0:
            ops.attn_fwd(q[:cu1], k[:cu1], v[:cu1], o[:cu1], heads)
    else:
        hybrid_seq_parallel_attn.run_fused(qkv, cat, s_img, cu1, heads, d)
    if s > cu1:
        ops.attn_fwd(q[cu1:], k[cu1:], v[cu1:], o[cu1:], heads)


def pad_segment_attention_(qkv: torch.Tensor, cat: torch.Tensor, cu1: int, heads: int, d: int):
    """attn2 of parallel_attention (attenion.py:181-207): the padding-text rows [cu1, S) attend among themselves, locally."""
    if qkv.shape[0] > cu1:
        ops.attn_fwd(qkv[cu1:, :d], qkv[cu1:, d:2 * d], qkv[cu1:, 2 * d:], cat[cu1:, :d], heads)


def parallel_attention(hybrid_seq_parallel_attn, q, k, v, img_q_len, img_kv_len, cu_seqlens_q, cu_seqlens_kv):
    """Reference signature (attenion.py:159-212): attn1 = SP attention of [img | valid text] with the text as
    the replicated joint tensor ("rear"), attn2 = plain attention over the padding text; concatenated."""
    cu1 = int(cu_seqlens_q[1])
    attn1 = hybrid_seq_parallel_attn(
        None, q[:, :img_q_len], k[:, :img_kv_len], v[:, :img_kv_len], dropout_p=0.0, causal=False,
        joint_tensor_query=q[:, img_q_len:cu1], joint_tensor_key=k[:, img_kv_len:cu1],
        joint_tensor_value=v[:, img_kv_len:cu1], joint_strategy="rear")
    b, s, a, d = q.shape
    out = torch.empty(b, s, a * d, dtype=BF16, device=q.device)
    out[:, :cu1] = attn1.reshape(b, cu1, a * d)
    if s > cu1:
        ops.attn_fwd(_flat2d(q)[cu1:], _flat2d(k)[cu1:], _flat2d(v)[cu1:], out[0, cu1:], a)
    return out
