"""Ulysses sequence-parallel attention over RCCL/xGMI - the object the reference installs on every block as
`hybrid_seq_parallel_attn` (xfuser.core.long_ctx_attention.xFuserLongContextAttention, a third-party class reached
from hyvideo/inference.py:80-83 and called at hyvideo/modules/attenion.py:169-180 and tests/test_attention.py:90-102).

Semantics (pinned by the reference's tests/test_attention.py:107-109: output == unsharded attention over img|txt):
every rank holds S_loc image tokens x all H heads plus the (replicated) joint text tokens.  One all-to-all per
tensor turns "my tokens, all heads" into "all tokens, my H/P heads"; flash attention runs over the full sequence
for those heads with the joint tokens appended at the rear; the inverse all-to-all returns each token's output to
its owner and a tiny all-gather returns the text rows of every head group.

MI355X mapping: the 8 GPUs form a full xGMI mesh, so all-to-all is the mesh-native collective (each pair has its own
link); messages are S_loc x (H/P*128) bf16 = 11.4 MB per peer at 720p/P=8, large enough to run at link rate.  The
exchange is `torch.distributed.all_to_all_single` on the "nccl" (= RCCL) backend; the token<->head re-layout on each
side is one HBM-bound strided-copy kernel (hv_copy3d_bf16) and the attention is hv_attn_fwd_bf16 reading the
received buffers in place (no concatenation)."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

BF16 = torch.bfloat16


class _HipKernels:
    """The product's kernel set (C ABI).  tests/ substitutes a CPU double to exercise the exchange logic under gloo."""

    @staticmethod
    def copy3d(src, dst, n_batch, rows, cols, src_bs, src_ld, dst_bs, dst_ld):
        from . import ops
        return ops.copy3d(src, dst, n_batch, rows, cols, src_bs, src_ld, dst_bs, dst_ld)

    @staticmethod
    def attn_fwd(q, k, v, out, heads):
        from . import ops
        return ops.attn_fwd(q, k, v, out, heads)

    # ring attention: per-chunk partials + merge
    @staticmethod
    def attn_partials(n_slots, n_q, heads, device):
        from . import ops
        return ops.AttnPartials(n_slots, n_q, heads, device)

    @staticmethod
    def attn_suggest_splits(n_q, n_kv, heads):
        from . import ops
        return ops.attn_suggest_splits(n_q, n_kv, heads)

    @staticmethod
    def attn_partial(q, k, v, parts, heads, splits):
        from . import ops
        return ops.attn_partial(q, k, v, parts, heads, splits)

    @staticmethod
    def attn_merge(parts, out):
        from . import ops
        return ops.attn_merge(parts, out)


class UlyssesLongContextAttention:
    """Constructible with no arguments like xFuserLongContextAttention(); uses the default (WORLD) group as the Ulysses
    group unless groups were registered with set_sequence_parallel_group(ulysses_group, ring_group).

    Hybrid Ulysses x Ring (xfuser `--ulysses-degree U --ring-degree R`, world = U*R; hyvideo/inference.py:157-176): the
    all-to-alls run inside the Ulysses group (U ranks trade tokens for heads), after which a rank holds the tokens of its
    Ulysses group for H/U heads; the K/V of the other R-1 Ulysses groups arrive over the ring group by point-to-point
    send/recv (xGMI links are point-to-point, so a ring step IS one link), one hop per step, while the attention of the chunk
    already here runs (hv_attn_partial_bf16); the per-chunk partials are folded by hv_attn_merge_bf16 (online softmax).
    The joint (text) keys ride with the local chunk only, the joint queries sit at the rear of every rank's Q."""

    _default_group = None
    _default_ring_group = None
    MIN_SEG_ROWS = 256       # attend_async: no segment of the output exchange smaller than one GEMM tile of rows (tests lower it)

    # Exchange variants for P > 1, selectable per object or by environment so that the first >= 2-GPU run can A/B them.  The
    # DEFAULTS are the forms that have the longest record (gloo, lazy-collective proxy, host-staged one-card runs): ONE
    # all_to_all_single for the whole output + one unpack, q/k packed by hv_copy3d.  The segmented / scatter-packed forms have
    # never run on RCCL (no multi-GPU box in this build's sessions) and stay opt-in until tests/test_gpu_rccl_multirank.py has
    # passed with them:
    #   HV_SP_NSEG=<n>            output exchange cut into n row segments overlapped with the out-projection (default 1)
    #   HV_SP_OUT_EXCHANGE=a2a|p2p  how a SEGMENT travels: "a2a" (default) = pack + all_to_all_single + unpack per segment;
    #                             "p2p" = one isend/irecv pair per peer, coalesced by batch_isend_irecv, no staging copies
    #   HV_SP_SCATTER_PACK=1      q/k chunks stored straight into the send layout by hv_qknorm_rope_scatter_bf16 (pack_dst)
    def __init__(self, group=None, kernels=None, ring_group=None, nseg: Optional[int] = None, out_exchange: Optional[str] = None,
                 scatter_pack: Optional[bool] = None):
        self.group = group if group is not None else UlyssesLongContextAttention._default_group
        self.ring_group = ring_group if ring_group is not None else UlyssesLongContextAttention._default_ring_group
        self.k = kernels or _HipKernels
        self._bufs = {}
        self._parts = None
        self.min_seg_rows = UlyssesLongContextAttention.MIN_SEG_ROWS
        self.nseg = int(os.environ.get("HV_SP_NSEG", "1")) if nseg is None else int(nseg)
        self.out_exchange = (os.environ.get("HV_SP_OUT_EXCHANGE", "a2a") if out_exchange is None else out_exchange).lower()
        if self.out_exchange not in ("a2a", "p2p"):
            raise ValueError(f"HV_SP_OUT_EXCHANGE must be a2a or p2p, got {self.out_exchange!r}")
        self.scatter_pack = (os.environ.get("HV_SP_SCATTER_PACK", "0") == "1") if scatter_pack is None else bool(scatter_pack)

    @classmethod
    def set_sequence_parallel_group(cls, group, ring_group=None):
        cls._default_group = group
        cls._default_ring_group = ring_group

    # ------------------------------------------------------------------ helpers
    def _world(self):
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def _ring(self):
        if self.ring_group is None:
            return 1, 0
        return dist.get_world_size(self.ring_group), dist.get_rank(self.ring_group)

    def _buf(self, name, shape, device):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.device != device:
            b = torch.empty(*shape, dtype=BF16, device=device)
            self._bufs[name] = b
        return b

    def _full_bufs(self, n_tot, w, device):
        """q / k / v / o over all tokens of this rank's Ulysses group (+ joint rows at the rear) for its head group; K and V share
        one allocation so a ring step ships them as two contiguous pieces."""
        kv = self._buf("kvf", (2, n_tot, w), device)
        self._bufs["kf"], self._bufs["vf"] = kv[0], kv[1]
        return {"qf": self._buf("qf", (n_tot, w), device), "kf": kv[0], "vf": kv[1], "of": self._buf("of", (n_tot, w), device)}

    def _attention(self, b, hp, s_u, n_j):
        """Attention of this rank's queries (s_u group tokens + n_j joint rows) over ALL keys: local chunk (+ joint keys), then,
        with a ring group, the chunks of the other Ulysses groups as they arrive."""
        R, rr = self._ring()
        qf, kf, vf, of = b["qf"], b["kf"], b["vf"], b["of"]
        if R == 1:
            self.k.attn_fwd(qf, kf, vf, of, hp)
            return
        n_q, w, dev = qf.shape[0], qf.shape[1], qf.device
        nxt = dist.get_global_rank(self.ring_group, (rr + 1) % R)
        prv = dist.get_global_rank(self.ring_group, (rr - 1) % R)
        rot = [self._buf("ring_a", (2, s_u, w), dev), self._buf("ring_b", (2, s_u, w), dev)]
        sp0 = self.k.attn_suggest_splits(n_q, s_u + n_j, hp)
        sp = self.k.attn_suggest_splits(n_q, s_u, hp)
        n_slots = sp0 + (R - 1) * sp
        p = self._parts
        if p is None or (p.n_slots, p.n_q, p.n_heads) != (n_slots, n_q, hp) or p.o.device != dev:
            p = self._parts = self.k.attn_partials(n_slots, n_q, hp, dev)
        p.used = 0
        cur_k, cur_v, n_cur = kf, vf, s_u + n_j
        for t in range(R):
            works = []
            if t + 1 < R:
                dst = rot[t % 2]
                works = dist.batch_isend_irecv([
                    dist.P2POp(dist.isend, cur_k[:s_u], nxt, self.ring_group), dist.P2POp(dist.isend, cur_v[:s_u], nxt, self.ring_group),
                    dist.P2POp(dist.irecv, dst[0], prv, self.ring_group), dist.P2POp(dist.irecv, dst[1], prv, self.ring_group)])
            # the chunk already here is attended while the next one crosses the link
            self.k.attn_partial(qf, cur_k[:n_cur], cur_v[:n_cur], p, hp, sp0 if t == 0 else sp)
            for wk in works:
                wk.wait()
            if t + 1 < R:
                cur_k, cur_v, n_cur = dst[0], dst[1], s_u
        self.k.attn_merge(p, of)

    def _core(self, q_src, k_src, v_src, ld_src, jq, jk, jv, ld_j, s_loc, n_j, heads, out, ld_out):
        """q_src/k_src/v_src: tensors whose data_ptr is (row 0, head 0) of the local image rows, row stride ld_src;
        jq/jk/jv: same for the n_j joint (text) rows, row stride ld_j; out: destination [s_loc + n_j rows, heads*128]
        with row stride ld_out."""
        P, rank = self._world()
        if heads % P != 0:
            raise ValueError(f"Ulysses degree {P} must divide the head count {heads} (put the remaining factor on --ring-degree)")
        hp = heads // P
        w = hp * 128
        dev = q_src.device
        s_img = P * s_loc
        n_tot = s_img + n_j
        full = self._full_bufs(n_tot, w, dev)
        send = self._buf("send", (P * s_loc, w), dev)
        for name, src, j in (("qf", q_src, jq), ("kf", k_src, jk), ("vf", v_src, jv)):
            # pack: send[p][r][:] = src[r][p*w : (p+1)*w]
            self.k.copy3d(src, send, P, s_loc, w, w, ld_src, s_loc * w, w)
            dist.all_to_all_single(full[name][:s_img], send, group=self.group)
            if n_j:
                # joint rows at the rear, my head group only: full[s_img + r][:] = j[r][rank*w : (rank+1)*w]
                self.k.copy3d(j[:, rank * w:], full[name][s_img:], 1, n_j, w, 0, ld_j, 0, w)
        self._attention(full, hp, s_img, n_j)
        recv = self._buf("recv", (P * s_loc, w), dev)
        dist.all_to_all_single(recv, full["of"][:s_img], group=self.group)
        # unpack: out[r][p*w + c] = recv[p][r][c]
        self.k.copy3d(recv, out, P, s_loc, w, s_loc * w, w, w, ld_out)
        if n_j:
            recv_t = self._buf("recv_t", (P * n_j, w), dev)
            dist.all_gather_into_tensor(recv_t, full["of"][s_img:].contiguous(), group=self.group)
            self.k.copy3d(recv_t, out[s_loc:], P, n_j, w, n_j * w, w, w, ld_out)

    # ------------------------------------------------------------------ overlapped form (exchange under the QKV GEMMs)
    # The block computes q, k, v with three column-chunk GEMMs and hands each chunk over as soon as it is ready:
    #   GEMM_q -> norm/rope -> send("q") | GEMM_k -> norm -> send("k") | GEMM_v -> send("v") | [single block: GEMM_mlp+GELU]
    # Each send packs on the compute stream and starts an ASYNC all-to-all (RCCL runs it on its own stream), so the
    # exchange of chunk i travels over xGMI while the MFMA GEMM of chunk i+1 runs; attend() waits for the three handles.
    def begin(self, s_loc: int, n_j: int, heads: int, device):
        P, rank = self._world()
        if heads % P != 0:
            raise ValueError(f"Ulysses degree {P} must divide the head count {heads} (put the remaining factor on --ring-degree)")
        w = (heads // P) * 128
        self._geo = (P, rank, s_loc, n_j, heads, w)
        n_tot = P * s_loc + n_j
        self._full_bufs(n_tot, w, device)
        self._works = []

    def chunk_dst(self, which: str) -> Optional[torch.Tensor]:
        """Where the caller may PRODUCE the local q / k / v chunk so that send() has nothing to move: with a Ulysses group of one
        rank there is no exchange, and the chunk's rows are simply rows [0, s_loc) of this rank's attention operand (a [s_loc, H*128]
        view, row stride H*128).  None when the chunk has to be packed per peer (P > 1)."""
        P, rank, s_loc, n_j, heads, w = self._geo
        if P != 1:
            return None
        return self._bufs[{"q": "qf", "k": "kf", "v": "vf"}[which]][:s_loc]

    def pack_dst(self, which: str) -> Optional[torch.Tensor]:
        """P > 1: the send buffer of the q / k / v exchange as a [s_loc, P, w] view (row r, peer p at send[p][r][:]) - a producer that
        can store by head block (ops.qknorm_rope_(..., out=)) writes the packed layout itself and calls send_packed(); None for P = 1."""
        P, rank, s_loc, n_j, heads, w = self._geo
        if P == 1 or not self.scatter_pack:
            return None
        send = self._buf("send_" + which, (P * s_loc, w), self._bufs["qf"].device)
        return send.view(P, s_loc, w).permute(1, 0, 2)

    def send_packed(self, which: str, joint: Optional[torch.Tensor], ld_j: int):
        """send() for a chunk already stored in pack_dst(which): starts the exchange, places the joint rows."""
        P, rank, s_loc, n_j, heads, w = self._geo
        name = {"q": "qf", "k": "kf", "v": "vf"}[which]
        full = self._bufs[name]
        send = self._bufs["send_" + which]
        self._works.append(dist.all_to_all_single(full[:P * s_loc], send, group=self.group, async_op=True))
        if n_j:
            self.k.copy3d(joint[:, rank * w:], full[P * s_loc:], 1, n_j, w, 0, ld_j, 0, w)

    def send(self, which: str, src: torch.Tensor, ld_src: int, joint: Optional[torch.Tensor], ld_j: int):
        """src: view whose data_ptr is (local image row 0, head 0) of the q / k / v chunk; joint: same for the valid text rows."""
        P, rank, s_loc, n_j, heads, w = self._geo
        name = {"q": "qf", "k": "kf", "v": "vf"}[which]
        full = self._bufs[name]
        if P == 1:
            # one rank: the "exchange" is the identity.  Produced in place (chunk_dst) -> nothing to do; otherwise one copy
            if src.data_ptr() != full.data_ptr():
                self.k.copy3d(src, full, 1, s_loc, w, 0, ld_src, 0, w)
        else:
            send = self._buf("send_" + which, (P * s_loc, w), src.device)      # one send buffer per tensor: exchanges overlap
            self.k.copy3d(src, send, P, s_loc, w, w, ld_src, s_loc * w, w)
            self._works.append(dist.all_to_all_single(full[:P * s_loc], send, group=self.group, async_op=True))
        if n_j:
            self.k.copy3d(joint[:, rank * w:], full[P * s_loc:], 1, n_j, w, 0, ld_j, 0, w)

    def attend(self, out: torch.Tensor, ld_out: int):
        """Synchronous form: attention + output exchange + unpack of every row into `out` before returning."""
        for _, _, finish in self.attend_async(out, ld_out, nseg=1):
            finish()

    def attend_async(self, out: torch.Tensor, ld_out: int, nseg: Optional[int] = None):
        """Attention over all tokens for this rank's heads, then the OUTPUT exchange cut into `nseg` row segments of the local
        image tokens so that it overlaps the out-projection GEMM that consumes it (north_star: "all-to-all ... overlapped with the
        per-head GEMMs"): every segment is its own asynchronous exchange (started at once, RCCL runs them on its stream in order);
        the caller walks the returned [(row_lo, row_hi, finish)] list - finish() makes the compute stream wait for THAT segment
        and unpacks it into out[row_lo:row_hi] - and launches the GEMM of those rows while the next segment is still on the wire.
        The last segment also covers the joint (text) rows [s_loc, s_loc + n_j).
        A segment of rows [r0, r1) is, per peer p, the contiguous slice of[p*s_loc + r0 : p*s_loc + r1] -> recv[p][r0:r1].
        out_exchange "p2p": one point-to-point pair per peer (`batch_isend_irecv`, coalesced by RCCL into one grouped launch = an
        all-to-all of that segment; every xGMI link carries exactly its pair's bytes), no staging.  "a2a": the per-peer slices
        are packed into one contiguous [P][r1-r0][w] staging buffer (hv_copy3d), exchanged by all_to_all_single and unpacked
        from its receive twin.  nseg None -> this object's setting (HV_SP_NSEG, default 1: the whole output in one
        all_to_all_single)."""
        P, rank, s_loc, n_j, heads, w = self._geo
        if nseg is None:
            nseg = self.nseg
        for wk in self._works:
            wk.wait()                       # the compute stream waits for the q/k/v exchanges (no host sync)
        self._works = []
        b = self._bufs
        s_img = P * s_loc
        if P == 1 and out.dim() == 2 and out.shape[0] >= s_loc + n_j and out.shape[1] >= w and out.stride(0) == ld_out:
            # one rank: every output row is already this rank's - attention writes straight into the caller's rows (no staging
            # buffer, no unpack); one segment, nothing to wait for
            self._attention({"qf": b["qf"], "kf": b["kf"], "vf": b["vf"], "of": out[:s_loc + n_j, :w]}, heads, s_img, n_j)
            return [(0, s_loc + n_j, lambda: None)]
        self._attention(b, heads // P, s_img, n_j)
        of = b["of"]
        recv = self._buf("recv", (s_img, w), out.device)
        nseg = max(1, min(nseg, s_loc // self.min_seg_rows))
        bounds = [((s_loc * i) // nseg) for i in range(nseg + 1)]
        seg_works = []
        for r0, r1 in zip(bounds[:-1], bounds[1:]):
            if P == 1:
                seg_works.append([])
                continue
            if r0 == 0 and r1 == s_loc:
                seg_works.append([dist.all_to_all_single(recv, of[:s_img], group=self.group, async_op=True)])
                continue
            if self.out_exchange == "a2a":
                # one staging pair per segment (distinct buffers: the segments' exchanges are all in flight at once)
                n = r1 - r0
                sseg = self._buf(f"seg_send_{len(seg_works)}", (P * n, w), out.device)
                rseg = self._buf(f"seg_recv_{len(seg_works)}", (P * n, w), out.device)
                self.k.copy3d(of[r0:], sseg, P, n, w, s_loc * w, w, n * w, w)
                seg_works.append([dist.all_to_all_single(rseg, sseg, group=self.group, async_op=True)])
                continue
            p2p = []
            for p in range(P):
                if p == rank:
                    continue
                peer = dist.get_global_rank(self.group, p) if self.group is not None else p
                p2p.append(dist.P2POp(dist.isend, of[p * s_loc + r0:p * s_loc + r1], peer, self.group))
                p2p.append(dist.P2POp(dist.irecv, recv[p * s_loc + r0:p * s_loc + r1], peer, self.group))
            seg_works.append(dist.batch_isend_irecv(p2p))
        txt_work = None
        if n_j:
            recv_t = self._buf("recv_t", (P * n_j, w), out.device)
            txt_work = dist.all_gather_into_tensor(recv_t, of[s_img:].contiguous(), group=self.group, async_op=True)

        def make_finish(r0, r1, works, last, idx):
            def finish():
                for wk in works:
                    wk.wait()
                if P > 1 and not (r0 == 0 and r1 == s_loc) and self.out_exchange == "a2a":
                    # unpack this segment's receive staging: out[r0 + r][p*w + c] = rseg[p][r][c]
                    n = r1 - r0
                    self.k.copy3d(b[f"seg_recv_{idx}"], out[r0:], P, n, w, n * w, w, w, ld_out)
                elif P == 1 or not (r0 == 0 and r1 == s_loc):
                    # this rank's own slice never crosses a link: it is unpacked straight from the attention output
                    self.k.copy3d(of[rank * s_loc + r0:], out[r0:, rank * w:], 1, r1 - r0, w, 0, w, 0, ld_out)
                    if P > 1:
                        for p in range(P):
                            if p != rank:
                                self.k.copy3d(recv[p * s_loc + r0:], out[r0:, p * w:], 1, r1 - r0, w, 0, w, 0, ld_out)
                else:
                    # unpack: out[r][p*w + c] = recv[p][r][c]
                    self.k.copy3d(recv, out, P, s_loc, w, s_loc * w, w, w, ld_out)
                if last and n_j:
                    txt_work.wait()
                    self.k.copy3d(recv_t, out[s_loc:], P, n_j, w, n_j * w, w, w, ld_out)
            return finish

        segs = []
        for i, ((r0, r1), works) in enumerate(zip(zip(bounds[:-1], bounds[1:]), seg_works)):
            last = i == len(seg_works) - 1
            segs.append((r0, (s_loc + n_j) if last else r1, make_finish(r0, r1, works, last, i)))
        return segs

    # ------------------------------------------------------------------ reference hook signature
    def __call__(self, attn, query, key, value, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                 alibi_slopes=None, deterministic=False, return_attn_probs=False, joint_tensor_query=None,
                 joint_tensor_key=None, joint_tensor_value=None, joint_strategy="none"):
        """query/key/value: [1, S_loc, H, D]; joint_*: [1, n_j, H, D] replicated on every rank -> [1, S_loc + n_j, H, D]."""
        if dropout_p != 0.0 or causal or softmax_scale is not None:
            raise NotImplementedError("inference path: dropout 0, non-causal, default 1/sqrt(D) scale")
        has_joint = joint_tensor_query is not None
        if has_joint and joint_strategy != "rear":
            raise NotImplementedError('joint tensors are appended at the rear (joint_strategy="rear")')
        b, s_loc, h, d = query.shape
        if b != 1 or d != 128:
            raise NotImplementedError("batch 1, head_dim 128")
        q2, k2, v2 = (_flat(t) for t in (query, key, value))
        n_j = joint_tensor_query.shape[1] if has_joint else 0
        out = torch.empty(1, s_loc + n_j, h, d, dtype=BF16, device=query.device)
        if has_joint and n_j:
            jq, jk, jv = (_flat(t) for t in (joint_tensor_query, joint_tensor_key, joint_tensor_value))
            ld_j = jq.stride(0)
        else:
            jq = jk = jv = None
            ld_j = 0
        assert q2.stride(0) == k2.stride(0) == v2.stride(0)
        self._core(q2, k2, v2, q2.stride(0), jq, jk, jv, ld_j, s_loc, n_j, h, out.view(s_loc + n_j, h * d), h * d)
        return out

    # ------------------------------------------------------------------ block-internal fast path
    def run_fused(self, qkv: torch.Tensor, cat: torch.Tensor, s_img_loc: int, cu1: int, heads: int, d: int):
        """q|k|v live in the fused rows of `qkv` [S_loc_total, 3d]; rows [0, s_img_loc) are local image tokens, rows
        [s_img_loc, cu1) the valid text tokens (joint); output goes to cat[:cu1, :d]."""
        n_j = cu1 - s_img_loc
        ld = qkv.stride(0)
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        self._core(q, k, v, ld, q[s_img_loc:], k[s_img_loc:], v[s_img_loc:], ld, s_img_loc, n_j, heads, cat, cat.stride(0))


def _flat(t: torch.Tensor) -> torch.Tensor:
    """[1,S,H,D] -> [S, H*D] view (heads packed inside the token row)."""
    assert t.dim() == 4 and t.shape[0] == 1 and t.stride(3) == 1 and t.stride(2) == t.shape[3]
    return t.as_strided((t.shape[1], t.shape[2] * t.shape[3]), (t.stride(1), 1), t.storage_offset())


# name used by the reference's import site (inference.py:80: `from xfuser.core.long_ctx_attention import ...`)
xFuserLongContextAttention = UlyssesLongContextAttention
