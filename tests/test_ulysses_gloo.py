"""CPU, world_size 2 (gloo): the N>1 host logic - Ulysses exchange indexing (pack -> all-to-all -> attention over
all tokens for H/P heads -> inverse all-to-all -> unpack, joint text at the rear) and the token-axis sharding /
output gather of parallelize_transformer.  The HIP kernels are replaced by CPU doubles defined HERE (test
infrastructure: torch as_strided copy + the oracle's attention); the property checked is the reference's own
(tests/test_attention.py:107-109,172-174): SP output == unsharded attention over img|txt, rtol = atol = 1e-3."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CpuKernelDouble:
    @staticmethod
    def copy3d(src, dst, n_batch, rows, cols, src_bs, src_ld, dst_bs, dst_ld):
        s = torch.as_strided(src, (n_batch, rows, cols), (src_bs, src_ld, 1), src.storage_offset())
        d = torch.as_strided(dst, (n_batch, rows, cols), (dst_bs, dst_ld, 1), dst.storage_offset())
        d.copy_(s)
        return dst

    @staticmethod
    def attn_fwd(q, k, v, out, heads):
        from oracle import dit_ref as R
        o = R.sdpa(q.float().reshape(1, q.shape[0], heads, 128), k.float().reshape(1, k.shape[0], heads, 128),
                   v.float().reshape(1, v.shape[0], heads, 128), R.Prec(True))
        out.copy_(o.reshape(q.shape[0], heads * 128).to(torch.bfloat16))
        return out


    # ---- ring attention doubles: the same partial format as hv_attn_partial_bf16 / hv_attn_merge_bf16 (log2-domain max)
    class _Parts:
        def __init__(self, n_slots, n_q, heads):
            self.n_slots, self.n_q, self.n_heads, self.used = n_slots, n_q, heads, 0
            self.o = torch.zeros(n_slots, n_q, heads, 128)
            self.ml = torch.zeros(n_slots, n_q, heads, 2)

    @staticmethod
    def attn_partials(n_slots, n_q, heads, device):
        return CpuKernelDouble._Parts(n_slots, n_q, heads)

    @staticmethod
    def attn_suggest_splits(n_q, n_kv, heads):
        return 2 if n_kv >= 128 else 1          # exercise both slot counts

    @staticmethod
    def attn_partial(q, k, v, parts, heads, splits):
        n_q, n_kv = q.shape[0], k.shape[0]
        assert parts.used + splits <= parts.n_slots and q.shape[1] == heads * 128 == k.shape[1] == v.shape[1]
        qh = q.float().reshape(n_q, heads, 128).transpose(0, 1)
        bounds = [0, n_kv] if splits == 1 else [0, ((n_kv + 63) // 64 + 1) // 2 * 64, n_kv]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            kh = k[lo:hi].float().reshape(hi - lo, heads, 128).transpose(0, 1)
            vh = v[lo:hi].float().reshape(hi - lo, heads, 128).transpose(0, 1)
            s = qh @ kh.transpose(1, 2) * (128 ** -0.5 * 1.4426950408889634)
            m = s.max(-1, keepdim=True).values
            pr = torch.exp2(s - m)
            parts.o[parts.used] = (pr.to(torch.bfloat16).float() @ vh).transpose(0, 1)
            parts.ml[parts.used, :, :, 0] = m[..., 0].transpose(0, 1)
            parts.ml[parts.used, :, :, 1] = pr.sum(-1).transpose(0, 1)
            parts.used += 1

    @staticmethod
    def attn_merge(parts, out):
        o, ml = parts.o[:parts.used], parts.ml[:parts.used]
        m = ml[..., 0].max(0).values
        wgt = torch.exp2(ml[..., 0] - m)
        acc = (o * wgt[..., None]).sum(0) / (ml[..., 1] * wgt).sum(0)[..., None]
        out[:, :parts.n_heads * 128] = acc.reshape(parts.n_q, -1).to(torch.bfloat16)
        return out


def _worker(rank, world, port, results):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hunyuanvideo_efficiency_amd import synthetic as syn
        from hunyuanvideo_efficiency_amd.long_ctx_attention import UlyssesLongContextAttention
        from hunyuanvideo_efficiency_amd.inference import parallelize_transformer_module
        from oracle import dit_ref as R
        E = R.Prec(True)
        H, s_img, n_txt = (8 if world == 8 else 4), 96, 11        # world 8 = the driver's largest scaling point (24 heads / 8)
        s_loc = s_img // world
        bf = lambda t: t.to(torch.bfloat16)
        q, k, v = (bf(syn.hashed_uniform((1, s_img + n_txt, H, 128), f"ul.{n}", 5) * 1.7) for n in "qkv")
        ref = R.sdpa(q.float(), k.float(), v.float(), E)                      # unsharded attention over img|txt
        sl = slice(rank * s_loc, (rank + 1) * s_loc)
        sp = UlyssesLongContextAttention(None, CpuKernelDouble)
        # (1) reference hook signature
        out = sp(None, q[:, sl], k[:, sl], v[:, sl], dropout_p=0.0, causal=False, joint_tensor_query=q[:, s_img:],
                 joint_tensor_key=k[:, s_img:], joint_tensor_value=v[:, s_img:], joint_strategy="rear")
        assert out.shape == (1, s_loc + n_txt, H, 128)
        exp = torch.cat([ref[:, sl], ref[:, s_img:]], 1)
        torch.testing.assert_close(out.float(), exp, rtol=1e-3, atol=1e-3)
        # (2) fused-buffer fast path used inside the blocks: rows = [local img | valid txt | pad txt]
        d = H * 128
        pad = 5
        rows = s_loc + n_txt + pad
        qkv = torch.zeros(rows, 3 * d, dtype=torch.bfloat16)
        for i, t in enumerate((q, k, v)):
            qkv[:s_loc, i * d:(i + 1) * d] = t[0, sl].reshape(s_loc, d)
            qkv[s_loc:s_loc + n_txt, i * d:(i + 1) * d] = t[0, s_img:].reshape(n_txt, d)
        cat = torch.zeros(rows, d + 64, dtype=torch.bfloat16)
        sp.run_fused(qkv, cat, s_loc, s_loc + n_txt, H, d)
        torch.testing.assert_close(cat[:s_loc + n_txt, :d].float(), exp.reshape(-1, d), rtol=1e-3, atol=1e-3)
        assert float(cat[s_loc + n_txt:].abs().max()) == 0 and float(cat[:, d:].abs().max()) == 0
        # (3) no joint tensors
        out = sp(None, q[:, sl], k[:, sl], v[:, sl])
        ref2 = R.sdpa(q[:, :s_img].float(), k[:, :s_img].float(), v[:, :s_img].float(), E)
        torch.testing.assert_close(out.float(), ref2[:, sl], rtol=1e-3, atol=1e-3)
        # (4) head count not divisible by the degree -> loud error
        with pytest.raises(ValueError):
            sp(None, q[:, sl, :3], k[:, sl, :3], v[:, sl, :3])

        # (5) parallelize_transformer: token-axis shard of x and the RoPE tables + output gather
        class FakeTransformer:
            double_blocks, single_blocks = [type("B", (), {})()], [type("B", (), {})()]

            def forward(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None,
                        freqs_sin=None, guidance=None, return_dict=True):
                # "velocity" that depends on the local latent and on the local RoPE rows, token by token
                T, Hh, Ww = x.shape[2], x.shape[3] // 2, x.shape[4] // 2
                f = (freqs_cos + 2 * freqs_sin).sum(-1).reshape(1, 1, T, Hh, Ww)
                f = f.repeat_interleave(2, 3).repeat_interleave(2, 4)
                return {"x": x * 3 + f}
        shapes = ((3, 16, 6), (3, 6, 16)) if world == 8 else ((3, 8, 6), (3, 6, 8))
        for (T, Hl, Wl) in shapes:       # (H/2) % P == 0 -> split H ; else split W
            x = syn.hashed_uniform((1, 16, T, Hl, Wl), "ul.x", 1)
            cos, sin = R.rope_tables([T, Hl // 2, Wl // 2], [16, 56, 56], 256.0)
            full = FakeTransformer().forward(x, None, freqs_cos=cos, freqs_sin=sin)["x"]
            ft = FakeTransformer()
            parallelize_transformer_module(ft, None, CpuKernelDouble)
            got = ft.forward(x, None, freqs_cos=cos, freqs_sin=sin)["x"]
            torch.testing.assert_close(got, full, rtol=0, atol=0)
            assert ft.double_blocks[0].hybrid_seq_parallel_attn is ft.single_blocks[0].hybrid_seq_parallel_attn
        with pytest.raises(ValueError):
            ft.forward(syn.hashed_uniform((1, 16, 3, 6, 6), "ul.y", 1), None, freqs_cos=cos, freqs_sin=sin)
        results[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        results[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ulysses_exchange_and_sharding_gloo(world):
    port = 29600 + world + (os.getpid() % 200)
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert all(results.get(r) == "ok" for r in range(world)), dict(results)
