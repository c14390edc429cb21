// Flash attention forward, bf16, head_dim 128, non-causal, one contiguous KV segment (K6 / K6' / K18-style
// callers pick the segment by pointer + length).  softmax(scale * Q K^T) V with fp32 statistics.
//
// gfx950 structure (8 waves x 32 query rows = 256 rows per workgroup, KV tile = 64 keys):
//   * S^T = K . Q^T  ("swapped" QK^T): v_mfma_f32_32x32x16_bf16 with the K tile as the A operand and the
//     wave's Q rows (held in registers for the whole kernel) as B.  The accumulator then has the QUERY on
//     the lane (col = lane&31) and 16 keys per lane in registers, so the softmax row max / row sum are
//     lane-local (+ one v_permlane32_swap between the two half-waves) - no LDS, no shuffles.
//   * O^T += V^T . P^T: the exponentiated accumulator registers, converted pairwise to bf16, ARE the B
//     operand of the next MFMA (k = key); V^T fragments come from the row-major V tile in LDS through
//     ds_read_b64_tr_b16 (hardware transpose).  O^T keeps the query on the lane too, so the online-softmax
//     rescale is a per-lane scalar multiply and is skipped (wave-uniformly) when no row max moved.
//   * K and V tiles are staged HBM -> VGPR -> LDS (loads issued one tile ahead, written after the
//     barrier), double buffered; K rows are XOR-swizzled by key for conflict-free ds_read_b128, V rows are
//     swizzled in 64-byte quarters for conflict-free transposed reads.
// Workgroups are ordered head-major so co-resident workgroups stream the same head's K/V (L2 / MALL reuse).
// Algorithmic work: 4 * n_q * n_kv * 128 flop per head.
#include "hv_common.hpp"
#include "../../include/hv_kernels.h"

namespace {

constexpr int D = 128;
constexpr int QROWS_WAVE = 32;
constexpr int NWAVES = 8;
constexpr int QTILE = QROWS_WAVE * NWAVES;  // 256
constexpr int KVT = 64;
constexpr int KV_TILE_BYTES = KVT * D * 2;       // 16 KiB
constexpr int BUF_BYTES = 2 * KV_TILE_BYTES;     // K + V
constexpr int ATT_LDS = 2 * BUF_BYTES;           // 64 KiB

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t sq, sk, sv, so;   // token strides (elements); head h lives at column h*128
    int n_q, n_kv, n_heads, n_qtiles;
    float scale_log2e;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ float half_swap_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    const int q0 = qt * QTILE + wave * QROWS_WAVE;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0+lr][16*ks + 8*lh .. +8]
    bf16x8 qf[8];
    {
        const int qrow = min(q0 + lr, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- staging: thread -> (key = i*32 + tid/16, chunk = tid%16) for i = 0,1 ; K and V
    const int skey = tid >> 4, sch = tid & 15;
    const bf16_t* kbase = a.k + head * D + sch * 8;
    const bf16_t* vbase = a.v + head * D + sch * 8;
    int k_lds[2], v_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = i * 32 + skey;
        k_lds[i] = key * 256 + ((sch ^ (key & 15)) << 4);
        v_lds[i] = KV_TILE_BYTES + key * 256 + ((sch ^ ((key & 3) << 2)) << 4);
    }
    u32x4 kreg[2], vreg[2];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(tile * KVT + i * 32 + skey, a.n_kv - 1);
            kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)key * a.sk);
            vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)key * a.sv);
        }
    };
    auto write_tile = [&](int buf) {
        char* b = smem + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(b + k_lds[i]) = kreg[i];
            *reinterpret_cast<u32x4*>(b + v_lds[i]) = vreg[i];
        }
    };

    // ---- read offsets
    // K (A operand): key = kb*32 + lr, chunk = 2*ks + lh  ->  key*256 + ((chunk ^ (key&15)) << 4)
    int kread[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + lr;
        kread[kb] = key * 256 + ((lh ^ (key & 15)) << 4);   // ks folded in by XOR of (2*ks)<<4 below
    }
    // V^T (A operand of O^T += V^T.P^T) via transposed reads: 16-lane group G = lane>>4, i = lane&15,
    // q = i>>2 (row of the 4x16 block), p = i&3.  Block rows: keys key0 + q, cols d0 + 4p.. ; d0 = db*32 + (G&1)*16.
    const int vq = (lane & 15) >> 2, vp = lane & 3, vG = lane >> 4;
    int vread;  // byte offset for (kb=0, s=0, half-block 0, db=0); others are added/XORed below
    {
        const int key = 4 * (vG >> 1) + vq;              // key0 = 4*lh (+ kb*32 + 16*s + 8*second)
        const int chunk16 = ((vG & 1) * 16 + 4 * vp) >> 3; // + db*4
        vread = KV_TILE_BYTES + key * 256 + ((chunk16 ^ ((key & 3) << 2)) << 4) + (vp & 1) * 8;
    }

    f32x16 oT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = a.scale_log2e;

    const int ntiles = (a.n_kv + KVT - 1) / KVT;
    load_tile(0);
    write_tile(0);
    __syncthreads();
    if (ntiles > 1) load_tile(1);

    for (int t = 0; t < ntiles; ++t) {
        const char* buf = smem + (t & 1) * BUF_BYTES;
        // ---------------- S^T = K . Q^T   (2 key blocks x 8 k-steps)
        f32x16 sT[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf + (kread[kb] ^ (ks << 5)));
                sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sT[kb], 0, 0, 0);
            }
        }
        // ---------------- tail mask (last tile only): key = t*64 + kb*32 + (r&3) + 8*(r>>2) + 4*lh
        if (t == ntiles - 1 && (a.n_kv & (KVT - 1))) {
            const int kbase_i = t * KVT + 4 * lh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_i + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= a.n_kv) sT[kb][r] = -INFINITY;
                }
        }
        // ---------------- online softmax (query = lane&31; this lane holds 32 of the tile's 64 keys)
        float mx = sT[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sT[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[1][r]);
        mx = half_swap_max(mx) * c;
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {   // wave-uniform: rescale only when some row's max moved
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[db][r] *= alpha;
            m_run = m_new;
        }
        bf16x8 pf[2][2];
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[kb][8 * s + j], c, -m_run));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[kb][8 * s + j + 1], c, -m_run));
                    ls += p0 + p1;
                    w[j >> 1] = pack_bf2(p0, p1);
                }
                pf[kb][s] = __builtin_bit_cast(bf16x8, w);   // (whole-vector cast: element-wise __bf16 inserts miscompile)
            }
        l_run += ls;
        // ---------------- O^T += V^T . P^T   (4 d-blocks x 4 k-steps of 16 keys)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int koff = (kb * 32 + 16 * s) * 256;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    // chunk16 += db*4 (bits 2-3 of the chunk index; the swizzle XORs the same bits) -> XOR (db<<6)
                    const char* p0 = buf + ((vread + koff) ^ (db << 6));
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], oT[db], 0, 0, 0);
                }
            }
        // ---------------- rotate buffers: tile t+1 (in registers since the last barrier) -> LDS
        if (t + 1 < ntiles) {
            write_tile((t + 1) & 1);
            __syncthreads();
            if (t + 2 < ntiles) load_tile(t + 2);
        }
    }

    // ---------------- epilogue: O[q][d] = O^T / l ; lane (q = lr, lh) holds d = db*32 + (r&3) + 8*(r>>2) + 4*lh
    const float l_tot = half_swap_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + lr;
    if (qrow < a.n_q) {
        bf16_t* op = a.o + (int64_t)qrow * a.so + head * D + 4 * lh;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 w;
                w[0] = pack_bf2(oT[db][4 * g] * inv, oT[db][4 * g + 1] * inv);
                w[1] = pack_bf2(oT[db][4 * g + 2] * inv, oT[db][4 * g + 3] * inv);
                *reinterpret_cast<u32x2*>(op + db * 32 + g * 8) = w;
            }
    }
}

}  // namespace

extern "C" int hv_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t stride_q, int64_t stride_k,
                                int64_t stride_v, int64_t stride_o, int n_q, int n_kv, int n_heads, int head_dim,
                                float scale, hipStream_t stream) {
    if (!q || !k || !v || !o || head_dim != D || n_heads <= 0 || n_q < 0 || n_kv < 0 || (stride_q & 7) || (stride_k & 7) ||
        (stride_v & 7) || (stride_o & 3))
        return HV_ERR_ARG;
    if (n_q == 0) return HV_OK;
    if (n_kv == 0) return HV_ERR_ARG;  // softmax over an empty set is undefined (flash-attn returns 0; callers never ask)
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o;
    a.sq = stride_q; a.sk = stride_k; a.sv = stride_v; a.so = stride_o;
    a.n_q = n_q; a.n_kv = n_kv; a.n_heads = n_heads;
    a.n_qtiles = (n_q + QTILE - 1) / QTILE;
    a.scale_log2e = scale * 1.4426950408889634f;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS) != hipSuccess)
            return HV_ERR_LAUNCH;
        attr_set = true;
    }
    attn_fwd_kernel<<<dim3((unsigned)(a.n_qtiles * n_heads)), dim3(512), ATT_LDS, stream>>>(a);
    return hv_check_launch();
}
