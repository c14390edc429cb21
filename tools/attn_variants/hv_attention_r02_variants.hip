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
//   * K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4; the bank swizzle is applied on the per-lane
//     SOURCE address so the LDS image stays lane-linear): K two tiles ahead, V one tile ahead, 2-deep rings;
//     K rows are XOR-swizzled by key for conflict-free ds_read_b128, V rows in 64-byte quarters for
//     conflict-free transposed reads (rocprofv3: SQ_LDS_BANK_CONFLICT = 0).
//   * software pipeline across KV tiles inside each wave: S(t+1) = K(t+1).Q^T is issued while the softmax of
//     tile t runs on the VALU, then O^T += V(t)^T.P(t)^T; one workgroup barrier per tile.
// Workgroups are ordered head-major so co-resident workgroups stream the same head's K/V (L2 / MALL reuse).
// Algorithmic work: 4 * n_q * n_kv * 128 flop per head.
#include "hv_common.hpp"   // -I hunyuanvideo_efficiency_amd/csrc (tools/attn_variants/Makefile)
#include "../../include/hv_kernels.h"
#include <type_traits>
#include <cstdlib>

namespace {

// timing experiments only (tools/…): -DHV_DBG_NOEXP replaces the main loop's v_exp_f32 by a full-rate multiply (wrong results)
#ifdef HV_DBG_NOEXP
#define HV_DBG_EXP(x) ((x) * 1.0001f)
#else
#define HV_DBG_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
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
    // KV split (load balance when the grid is only a few rounds of workgroups): blockIdx.y = split s handles keys
    // [s*split_keys, min(n_kv, (s+1)*split_keys)); partial O (unnormalised, fp32) and (m, l) go to the workspace
    int n_splits, split_keys;
    float* part_o;      // [n_splits][n_q][n_heads][128]
    float* part_ml;     // [n_splits][n_q][n_heads][2]
    int partial;        // 1: always leave the unnormalised partial (ring attention merges K/V chunks later), even unsplit
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

// =====================================================================================================================
// v2: software-pipelined across KV tiles inside each wave so that MFMA and VALU work of DIFFERENT tiles are adjacent in
// the instruction stream (the two co-resident waves of a SIMD run the same program in lockstep behind one barrier per
// tile, so without this the matrix pipe idles during every softmax):
//   stage 1:  S(t+1) = K(t+1).Q^T  (16 MFMA)   ||  row max of S(t), rescale check, exp of key block 0 of S(t)
//   stage 2:  O^T += V(t)[kb0]^T.P(t)[kb0] (8 MFMA)  ||  exp of key block 1 of S(t)
//   stage 3:  O^T += V(t)[kb1]^T.P(t)[kb1] (8 MFMA)
// K and V tiles arrive by LDS-DMA (global_load_lds_dwordx4, swizzle applied on the per-lane SOURCE address; no staging
// VGPRs): K two tiles ahead, V one tile ahead, each into a 2-deep ring; one barrier per tile.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v2(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1], 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    if (a.n_splits > 1) {   // this workgroup's key range
        const int kv0 = blockIdx.y * a.split_keys;
        a.k += (int64_t)kv0 * a.sk;
        a.v += (int64_t)kv0 * a.sv;
        a.n_kv = min(a.n_kv - kv0, a.split_keys);
    }
    const int q0 = qt * QTILE + wave * QROWS_WAVE;
    constexpr int KOFF = 0, VOFF = 2 * KV_TILE_BYTES;

    bf16x8 qf[8];
    {
        const int qrow = min(q0 + lr, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // ---- DMA addressing: wave w owns keys [8w, 8w+8) of a tile; piece i (0,1): key = 8w + 4i + (lane>>4), LDS chunk pos = lane&15
    // Source address = wave-uniform tile base (scalar registers, advanced by scalar adds) + a per-lane byte offset that never
    // changes: no vector integer work per tile.  Only a partial last tile takes the slow path that clamps rows to n_kv - 1.
    const int dkey0 = 8 * wave + (lane >> 4), dcp = lane & 15;
    const char* kbase = reinterpret_cast<const char*>(a.k + head * D);
    const char* vbase = reinterpret_cast<const char*>(a.v + head * D);
    uint32_t koff[2], voff[2], koff_last[2], voff_last[2];
    const int last_tile = (a.n_kv + KVT - 1) / KVT - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = dkey0 + 4 * i;
        const int kx = ((dcp ^ (key & 15)) << 4), vx = ((dcp ^ ((key & 3) << 2)) << 4);
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + kx);
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + vx);
        const int keyl = min(key, a.n_kv - 1 - last_tile * KVT);      // rows past the end re-read the last valid row
        koff_last[i] = (uint32_t)(keyl * (int)a.sk * 2 + kx);
        voff_last[i] = (uint32_t)(keyl * (int)a.sv * 2 + vx);
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 2048;   // 8 keys x 256 B; scalar: the DMA destination goes through M0
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    auto dma_k = [&](int tile, int buf) {
        const char* tb = kbase + tile * k_tile_bytes;
        const bool lastt = tile == last_tile;      // wave-uniform select, no branch
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(tb + (lastt ? koff_last[i] : koff[i])),
                                             (lds_void_ptr)(smem + KOFF + buf * KV_TILE_BYTES + wave_lds + i * 1024), 16, 0, 0);
    };
    auto dma_v = [&](int tile, int buf) {
        const char* tb = vbase + tile * v_tile_bytes;
        const bool lastt = tile == last_tile;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(tb + (lastt ? voff_last[i] : voff[i])),
                                             (lds_void_ptr)(smem + VOFF + buf * KV_TILE_BYTES + wave_lds + i * 1024), 16, 0, 0);
    };

    int kread[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + lr;
        kread[kb] = key * 256 + ((lh ^ (key & 15)) << 4);
    }
    const int vq = (lane & 15) >> 2, vp = lane & 3, vG = lane >> 4;
    int vread;
    {
        const int key = 4 * (vG >> 1) + vq;
        const int chunk16 = ((vG & 1) * 16 + 4 * vp) >> 3;
        vread = key * 256 + ((chunk16 ^ ((key & 3) << 2)) << 4) + (vp & 1) * 8;
    }

    f32x16 oT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = a.scale_log2e;
    const int ntiles = (a.n_kv + KVT - 1) / KVT;

    auto qk = [&](f32x16 (&S)[2], int buf) {
        const char* kb_ = smem + KOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + (kread[kb] ^ (ks << 5)));
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[kb], 0, 0, 0);
            }
        }
    };
    auto exp_block = [&](const f32x16& S, bf16x8 (&pf)[2], float& ls) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[8 * s + j], c, -m_run));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[8 * s + j + 1], c, -m_run));
                ls += p0 + p1;
                w[j >> 1] = pack_bf2(p0, p1);
            }
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto pv_block = [&](const bf16x8 (&pf)[2], int kb, int buf) {
        const char* vb_ = smem + VOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (kb * 32 + 16 * s) * 256;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* p0 = vb_ + ((vread + koff) ^ (db << 6));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oT[db], 0, 0, 0);
            }
        }
    };
    // row max of S(t) (+ tail mask on the last tile) and the rare rescale; ends in a wave-uniform branch
    auto softmax_head = [&](f32x16 (&Sc)[2], int t, bool last) {
        if (last && (a.n_kv & (KVT - 1))) {
            const int kbase_i = t * KVT + 4 * lh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_i + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= a.n_kv) Sc[kb][r] = -INFINITY;
                }
        }
        float mx = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[1][r]);
        mx = half_swap_max(mx) * c;
        const float m_new = fmaxf(m_run, mx);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[db][r] *= alpha;
            m_run = m_new;
        }
    };
    // steady-state iteration (branch-free after the rescale check, so the scheduler can interleave MFMA and VALU):
    // consumes Sc = S(t), produces Sn = S(t+1)
    auto body_main = [&](f32x16 (&Sc)[2], f32x16 (&Sn)[2], int t) {
        if (t + 2 < ntiles) dma_k(t + 2, t & 1);
        dma_v(t + 1, (t + 1) & 1);
        softmax_head(Sc, t, false);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        qk(Sn, (t + 1) & 1);            // stage 1: 16 MFMA  ||  exp of key block 0
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & 1);         // stage 2: 8 MFMA   ||  exp of key block 1
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & 1);         // stage 3: 8 MFMA
        l_run += ls;
        __syncthreads();
    };
    auto body_last = [&](f32x16 (&Sc)[2], int t) {
        softmax_head(Sc, t, true);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & 1);
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & 1);
        l_run += ls;
    };

    f32x16 sA[2], sB[2];
    dma_k(0, 0);
    dma_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    __syncthreads();
    qk(sA, 0);
    __syncthreads();   // every wave finished reading K[0] before tile 2 is DMA'd over it
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        body_main(sA, sB, t);
        body_main(sB, sA, t + 1);
    }
    if (t + 1 < ntiles) {
        body_main(sA, sB, t);
        body_last(sB, t + 1);
    } else {
        body_last(sA, t);
    }

    const float l_tot = half_swap_sum(l_run);
    if (a.n_splits > 1 || a.partial) {   // partial result: O^T unnormalised (fp32) + (m, l); merged by attn_combine_kernel
        const int qrow_p = q0 + lr;
        if (qrow_p < a.n_q) {
            const int64_t rowi = ((int64_t)blockIdx.y * a.n_q + qrow_p) * a.n_heads + head;
            float* po = a.part_o + rowi * D + 4 * lh;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(po + db * 32 + g * 8) =
                        make_float4(oT[db][4 * g], oT[db][4 * g + 1], oT[db][4 * g + 2], oT[db][4 * g + 3]);
            if (lh == 0) {
                a.part_ml[rowi * 2] = m_run;
                a.part_ml[rowi * 2 + 1] = l_tot;
            }
        }
        return;
    }
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


// =====================================================================================================================
// v3 = v2 with the softmax's scale/subtract pass moved into the matrix pipe and the row max moved under the P.V MFMAs:
//   * Q is pre-multiplied by scale*log2(e) once (bf16), so S = K.Q'^T is already in the exp2 domain;
//   * the running max is DEFERRED (threshold 2^8) and enters as the C operand of each S chain (16 registers holding -m), so
//     P = exp2(S') needs no VALU work besides the exp itself;
//   * the row max of S'(t+1) is taken in stage 3 of iteration t (where the VALU used to idle under 8 MFMAs), leaving only a
//     wave-uniform compare at the head of the next iteration.
// VALU issue per wave and tile: 32 exp2 + 32 adds + 16 cvt_pk + 16 max3 (~550 cycles) instead of ~700 with the fma pass and an
// unoverlapped max in front; the matrix pipe (1024 cycles per wave and tile, two waves per SIMD) is what every saved slot feeds.
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v3(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1], 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    if (a.n_splits > 1) {   // this workgroup's key range
        const int kv0 = blockIdx.y * a.split_keys;
        a.k += (int64_t)kv0 * a.sk;
        a.v += (int64_t)kv0 * a.sv;
        a.n_kv = min(a.n_kv - kv0, a.split_keys);
    }
    const int q0 = qt * QTILE + wave * QROWS_WAVE;
    constexpr int KOFF = 0, VOFF = 2 * KV_TILE_BYTES;

    bf16x8 qf[8];
    {
        const int qrow = min(q0 + lr, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // Q' = bf16(Q * scale * log2(e)): the scores leave the MFMA already in the exp2 domain (one rounding of q, 2^-9 relative)
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + ks * 16);
            u32x4 sc;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = pack_bf2(bf2f_lo(raw[j]) * a.scale_log2e, bf2f_hi(raw[j]) * a.scale_log2e);
            qf[ks] = __builtin_bit_cast(bf16x8, sc);
        }
    }

    // ---- DMA addressing: wave w owns keys [8w, 8w+8) of a tile; piece i (0,1): key = 8w + 4i + (lane>>4), LDS chunk pos = lane&15
    // Source address = wave-uniform tile base (scalar registers, advanced by scalar adds) + a per-lane byte offset that never
    // changes: no vector integer work per tile.  Only a partial last tile takes the slow path that clamps rows to n_kv - 1.
    const int dkey0 = 8 * wave + (lane >> 4), dcp = lane & 15;
    const char* kbase = reinterpret_cast<const char*>(a.k + head * D);
    const char* vbase = reinterpret_cast<const char*>(a.v + head * D);
    uint32_t koff[2], voff[2], koff_last[2], voff_last[2];
    const int last_tile = (a.n_kv + KVT - 1) / KVT - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = dkey0 + 4 * i;
        const int kx = ((dcp ^ (key & 15)) << 4), vx = ((dcp ^ ((key & 3) << 2)) << 4);
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + kx);
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + vx);
        const int keyl = min(key, a.n_kv - 1 - last_tile * KVT);      // rows past the end re-read the last valid row
        koff_last[i] = (uint32_t)(keyl * (int)a.sk * 2 + kx);
        voff_last[i] = (uint32_t)(keyl * (int)a.sv * 2 + vx);
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 2048;   // 8 keys x 256 B; scalar: the DMA destination goes through M0
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    auto dma_k = [&](int tile, int buf) {
        const char* tb = kbase + tile * k_tile_bytes;
        const bool lastt = tile == last_tile;      // wave-uniform select, no branch
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(tb + (lastt ? koff_last[i] : koff[i])),
                                             (lds_void_ptr)(smem + KOFF + buf * KV_TILE_BYTES + wave_lds + i * 1024), 16, 0, 0);
    };
    auto dma_v = [&](int tile, int buf) {
        const char* tb = vbase + tile * v_tile_bytes;
        const bool lastt = tile == last_tile;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(tb + (lastt ? voff_last[i] : voff[i])),
                                             (lds_void_ptr)(smem + VOFF + buf * KV_TILE_BYTES + wave_lds + i * 1024), 16, 0, 0);
    };

    int kread[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + lr;
        kread[kb] = key * 256 + ((lh ^ (key & 15)) << 4);
    }
    const int vq = (lane & 15) >> 2, vp = lane & 3, vG = lane >> 4;
    int vread;
    {
        const int key = 4 * (vG >> 1) + vq;
        const int chunk16 = ((vG & 1) * 16 + 4 * vp) >> 3;
        vread = key * 256 + ((chunk16 ^ ((key & 3) << 2)) << 4) + (vp & 1) * 8;
    }

    f32x16 oT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.f;
    // m_run: the row max the running sums are scaled by (log2 domain).  It is DEFERRED: it only moves when a tile's scores exceed
    // it by more than THR, so P <= 2^THR instead of <= 1 (bf16 and fp32 keep their relative precision at any scale).  -m_run sits
    // in all 16 registers of `negm`, the C operand of the first MFMA of every S chain: S' = K.Q'^T - m_run comes out of the matrix
    // pipe ready for exp2 - no scale/subtract pass on the VALU at all.
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm;
    constexpr float THR = 8.0f;
    const int ntiles = (a.n_kv + KVT - 1) / KVT;

    auto qk = [&](f32x16 (&S)[2], int buf, const f32x16& c0) {
        const char* kb_ = smem + KOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + (kread[kb] ^ (ks << 5)));
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? c0 : S[kb], 0, 0, 0);
            }
        }
    };
    auto exp_block = [&](const f32x16& S, bf16x8 (&pf)[2], float& ls) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float p0 = __builtin_amdgcn_exp2f(S[8 * s + j]);
                const float p1 = __builtin_amdgcn_exp2f(S[8 * s + j + 1]);
                ls += p0;      // ONE serial chain of plain v_add_f32: (p0 + p1) pairs would be SLP-packed into v_pk_add_f32, which
                ls += p1;      // beside MFMAs costs more than it saves
                w[j >> 1] = pack_bf2(p0, p1);
            }
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto pv_block = [&](const bf16x8 (&pf)[2], int kb, int buf) {
        const char* vb_ = smem + VOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (kb * 32 + 16 * s) * 256;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* p0 = vb_ + ((vread + koff) ^ (db << 6));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oT[db], 0, 0, 0);
            }
        }
    };
    // tail mask (last tile) + row max of a score tile (relative to m_run); VALU only, placed under the P.V MFMAs of the tile before
    auto tile_max = [&](f32x16 (&Sc)[2], int t, bool last) -> float {
        if (last && (a.n_kv & (KVT - 1))) {
            const int kbase_i = t * KVT + 4 * lh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_i + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= a.n_kv) Sc[kb][r] = -INFINITY;
                }
        }
        float mx = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[1][r]);
        return half_swap_max(mx);
    };
    // rare: some row of this tile exceeds the running max by more than THR -> move every row's max to its true value, rescale
    // the sums, the scores of THIS tile (already relative to the old max) and the C operand of the next S chain
    auto raise_max = [&](f32x16 (&Sc)[2], float mx) {
        const float d = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
        m_run += d;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Sc[kb][r] -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
        asm volatile("" : "+v"(negm));      // 16 live registers, not 16 re-materialised moves per tile
    };
    // steady-state iteration: consumes Sc = S'(t) and its row max mx_c, produces Sn = S'(t+1) and mx_n
    auto body_main = [&](f32x16 (&Sc)[2], f32x16 (&Sn)[2], int t, float mx_c, float& mx_n) {
        if (t + 2 < ntiles) dma_k(t + 2, t & 1);
        dma_v(t + 1, (t + 1) & 1);
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        qk(Sn, (t + 1) & 1, negm);      // stage 1: 16 MFMA  ||  exp of key block 0
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & 1);         // stage 2: 8 MFMA   ||  exp of key block 1
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & 1);         // stage 3: 8 MFMA   ||  row max of S'(t+1)
        mx_n = tile_max(Sn, t + 1, t + 2 == ntiles);
        l_run += ls;
        __syncthreads();
    };
    auto body_last = [&](f32x16 (&Sc)[2], int t, float mx_c) {
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & 1);
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & 1);
        l_run += ls;
    };

    f32x16 sA[2], sB[2];
    dma_k(0, 0);
    dma_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    __syncthreads();
    {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        qk(sA, 0, zero);
    }
    // tile 0 fixes the initial max exactly: m_run = rowmax(S(0)), S'(0) = S(0) - m_run (the only explicit subtraction)
    m_run = tile_max(sA, 0, ntiles == 1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[kb][r] -= m_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -m_run;
    asm volatile("" : "+v"(negm));
    __syncthreads();   // every wave finished reading K[0] before tile 2 is DMA'd over it
    int t = 0;
    float mxA = 0.f, mxB = 0.f;
    for (; t + 2 < ntiles; t += 2) {
        body_main(sA, sB, t, mxA, mxB);
        body_main(sB, sA, t + 1, mxB, mxA);
    }
    if (t + 1 < ntiles) {
        body_main(sA, sB, t, mxA, mxB);
        body_last(sB, t + 1, mxB);
    } else {
        body_last(sA, t, mxA);
    }

    const float l_tot = half_swap_sum(l_run);
    if (a.n_splits > 1 || a.partial) {   // partial result: O^T unnormalised (fp32) + (m, l); merged by attn_combine_kernel
        const int qrow_p = q0 + lr;
        if (qrow_p < a.n_q) {
            const int64_t rowi = ((int64_t)blockIdx.y * a.n_q + qrow_p) * a.n_heads + head;
            float* po = a.part_o + rowi * D + 4 * lh;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(po + db * 32 + g * 8) =
                        make_float4(oT[db][4 * g], oT[db][4 * g + 1], oT[db][4 * g + 2], oT[db][4 * g + 3]);
            if (lh == 0) {
                a.part_ml[rowi * 2] = m_run;
                a.part_ml[rowi * 2 + 1] = l_tot;
            }
        }
        return;
    }
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


// =====================================================================================================================
// v4 = v3's arithmetic with (a) the iteration body laid out MFMA gap by MFMA gap in source order (see body_main) and (b) the K/V
// LDS-DMA issued as bounds-checked buffer loads (no clamped offsets for the ragged last tile).
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v4(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1], 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    if (a.n_splits > 1) {   // this workgroup's key range
        const int kv0 = blockIdx.y * a.split_keys;
        a.k += (int64_t)kv0 * a.sk;
        a.v += (int64_t)kv0 * a.sv;
        a.n_kv = min(a.n_kv - kv0, a.split_keys);
    }
    const int q0 = qt * QTILE + wave * QROWS_WAVE;
    constexpr int KOFF = 0, VOFF = 2 * KV_TILE_BYTES;

    bf16x8 qf[8];
    {
        const int qrow = min(q0 + lr, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // Q' = bf16(Q * scale * log2(e)): the scores leave the MFMA already in the exp2 domain (one rounding of q, 2^-9 relative)
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + ks * 16);
            u32x4 sc;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = pack_bf2(bf2f_lo(raw[j]) * a.scale_log2e, bf2f_hi(raw[j]) * a.scale_log2e);
            qf[ks] = __builtin_bit_cast(bf16x8, sc);
        }
    }

    // ---- DMA addressing: wave w owns keys [8w, 8w+8) of a tile; piece i (0,1): key = 8w + 4i + (lane>>4), LDS chunk pos = lane&15.
    // LDS-DMA by buffer_load ... lds: the descriptor (4 SGPRs, rebuilt per tile by scalar arithmetic) starts at the tile and covers
    // exactly its valid rows, the per-lane byte offset never changes.  Keys past n_kv in a ragged last tile fall outside the
    // descriptor: the hardware range check returns zeros for them - no clamped copies of the offsets (4 VGPRs), no selects; those
    // keys are masked to -inf after the S MFMAs as before.
    const char* kbase = reinterpret_cast<const char*>(a.k + head * D);
    const char* vbase = reinterpret_cast<const char*>(a.v + head * D);
    uint32_t koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = 8 * wave + (lane >> 4) + 4 * i, dcp = lane & 15;
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + ((dcp ^ (key & 15)) << 4));
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + ((dcp ^ ((key & 3) << 2)) << 4));
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 2048;   // 8 keys x 256 B; scalar: the DMA destination goes through M0
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    auto dma_tile = [&](const char* base, int64_t tile_bytes, int row_bytes, const uint32_t (&off)[2], int tile, int lds_base) {
        const int rows = min(a.n_kv - tile * KVT, KVT);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + tile * tile_bytes), 0, rows * row_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + lds_base + wave_lds + i * 1024), 16, off[i], 0, 0, 0);
    };
    auto dma_k = [&](int tile, int buf) { dma_tile(kbase, k_tile_bytes, (int)a.sk * 2, koff, tile, KOFF + buf * KV_TILE_BYTES); };
    auto dma_v = [&](int tile, int buf) { dma_tile(vbase, v_tile_bytes, (int)a.sv * 2, voff, tile, VOFF + buf * KV_TILE_BYTES); };

    int kread[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + lr;
        kread[kb] = key * 256 + ((lh ^ (key & 15)) << 4);
    }
    const int vq = (lane & 15) >> 2, vp = lane & 3, vG = lane >> 4;
    int vread;
    {
        const int key = 4 * (vG >> 1) + vq;
        const int chunk16 = ((vG & 1) * 16 + 4 * vp) >> 3;
        vread = key * 256 + ((chunk16 ^ ((key & 3) << 2)) << 4) + (vp & 1) * 8;
    }

    f32x16 oT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.f;
    // m_run: the row max the running sums are scaled by (log2 domain).  It is DEFERRED: it only moves when a tile's scores exceed
    // it by more than THR, so P <= 2^THR instead of <= 1 (bf16 and fp32 keep their relative precision at any scale).  -m_run sits
    // in all 16 registers of `negm`, the C operand of the first MFMA of every S chain: S' = K.Q'^T - m_run comes out of the matrix
    // pipe ready for exp2 - no scale/subtract pass on the VALU at all.
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm;
    constexpr float THR = 8.0f;
    const int ntiles = (a.n_kv + KVT - 1) / KVT;

    auto qk = [&](f32x16 (&S)[2], int buf, const f32x16& c0) {
        const char* kb_ = smem + KOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + (kread[kb] ^ (ks << 5)));
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? c0 : S[kb], 0, 0, 0);
            }
        }
    };
    auto exp_block = [&](const f32x16& S, bf16x8 (&pf)[2], float& ls) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float p0 = __builtin_amdgcn_exp2f(S[8 * s + j]);
                const float p1 = __builtin_amdgcn_exp2f(S[8 * s + j + 1]);
                ls += p0;      // ONE serial chain of plain v_add_f32: (p0 + p1) pairs would be SLP-packed into v_pk_add_f32, which
                ls += p1;      // beside MFMAs costs more than it saves
                w[j >> 1] = pack_bf2(p0, p1);
            }
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto pv_block = [&](const bf16x8 (&pf)[2], int kb, int buf) {
        const char* vb_ = smem + VOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (kb * 32 + 16 * s) * 256;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* p0 = vb_ + ((vread + koff) ^ (db << 6));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oT[db], 0, 0, 0);
            }
        }
    };
    // tail mask (last tile) + row max of a score tile (relative to m_run); VALU only, placed under the P.V MFMAs of the tile before
    auto tile_max = [&](f32x16 (&Sc)[2], int t, bool last) -> float {
        if (last && (a.n_kv & (KVT - 1))) {
            const int kbase_i = t * KVT + 4 * lh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_i + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= a.n_kv) Sc[kb][r] = -INFINITY;
                }
        }
        float mx = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[1][r]);
        return half_swap_max(mx);
    };
    // rare: some row of this tile exceeds the running max by more than THR -> move every row's max to its true value, rescale
    // the sums, the scores of THIS tile (already relative to the old max) and the C operand of the next S chain
    auto raise_max = [&](f32x16 (&Sc)[2], float mx) {
        const float d = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
        m_run += d;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Sc[kb][r] -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
        asm volatile("" : "+v"(negm));      // 16 live registers, not 16 re-materialised moves per tile
    };
    // steady-state iteration: consumes Sc = S'(t) and its row max mx_c, produces Sn = S'(t+1) and mx_n.
    // The 32 MFMAs of a tile are laid out as 32 "gaps" in SOURCE order, each closed by a scheduling fence, so that every MFMA is
    // followed by its share of the tile's VALU and LDS work (guide: <= 5 single-issue fillers per 32-cycle MFMA, at most one or two
    // of them an 8-cycle v_exp_f32) instead of the compiler's front-loaded order (all exps under the first seven MFMAs):
    //   gaps  0-15  S'(t+1) chain (K fragments 2 gaps ahead) | exp/pack of P block 0, element g; of block 1, element g/2 (even g)
    //   gaps 16-23  O += V.P block 0 (V fragments 2 ahead)    | exp/pack of P block 1, elements 8..15
    //   gaps 24-31  O += V.P block 1                          | row max of S'(t+1), 4 values per gap
    // NEXT_LAST: tile t+1 may be the (possibly ragged) last one - its masked row max is taken after the gaps instead.
    auto body_main = [&](f32x16 (&Sc)[2], f32x16 (&Sn)[2], int t, float mx_c, float& mx_n, auto next_last_c) {
        constexpr bool NEXT_LAST = decltype(next_last_c)::value;
        if (t + 2 < ntiles) dma_k(t + 2, t & 1);
        dma_v(t + 1, (t + 1) & 1);
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        const char* kb_ = smem + KOFF + ((t + 1) & 1) * KV_TILE_BYTES;
        const char* vb_ = smem + VOFF + (t & 1) * KV_TILE_BYTES;
        auto kload = [&](int i) { return *reinterpret_cast<const bf16x8*>(kb_ + (kread[i >> 3] ^ ((i & 7) << 5))); };
        auto vload = [&](int j) {      // j = (kb*2 + s)*4 + db
            const int koff_ = ((j >> 3) * 32 + 16 * ((j >> 2) & 1)) * 256;
            const char* p0 = vb_ + ((vread + koff_) ^ ((j & 3) << 6));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        bf16x8 fr[3];                  // ONE 3-slot fragment ring for the K and V operands: the load of gap g+2 goes to slot (g+2) % 3
        u32x4 w0[2], w1[2];
        float P0[16], P1[16];          // this tile's exponentials (compile-time indices: registers, each live for about one gap)
        float ls = 0.f, mxa = -INFINITY, mxb = -INFINITY;
        // row sum: one pinned v_add_f32 per value, issued ONE GAP AFTER its v_exp_f32 (a pure `ls += p` chain is emitted as 32
        // dependent adds after the last gap with all 32 values held live; pinned in the exp's own gap an asm add could sit in the
        // trans->VALU forwarding slot, which the compiler only pads for instructions it can see)
        auto acc = [&](float p) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls) : "v"(p)); };
        fr[0] = kload(0);
        fr[1] = kload(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g < 16) {
                if (g + 2 < 16) fr[(g + 2) % 3] = kload(g + 2);
                else fr[(g + 2) % 3] = vload(g - 14);
                if (g >= 1) {
                    acc(P0[g - 1]);
                    if (!((g - 1) & 1)) acc(P1[(g - 1) >> 1]);
                }
                Sn[g >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % 3], qf[g & 7], (g & 7) == 0 ? negm : Sn[g >> 3], 0, 0, 0);
                P0[g] = __builtin_amdgcn_exp2f(Sc[0][g]);
                if (g & 1) w0[g >> 3][(g & 7) >> 1] = pack_bf2(P0[g - 1], P0[g]);
                if (!(g & 1)) {
                    const int e = g >> 1;
                    P1[e] = __builtin_amdgcn_exp2f(Sc[1][e]);
                    if (e & 1) w1[0][(e & 7) >> 1] = pack_bf2(P1[e - 1], P1[e]);
                }
            } else {
                const int j = g - 16;
                if (j + 2 < 16) fr[(g + 2) % 3] = vload(j + 2);
                if (j == 0) acc(P0[15]);
                if (j >= 1 && j <= 8) acc(P1[7 + j]);
                const int s_ = (j >> 2) & 1, db = j & 3;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, j < 8 ? w0[s_] : w1[s_]);
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % 3], pf, oT[db], 0, 0, 0);
                if (j < 8) {
                    const int e = 8 + j;
                    P1[e] = __builtin_amdgcn_exp2f(Sc[1][e]);
                    if (e & 1) w1[1][(e & 7) >> 1] = pack_bf2(P1[e - 1], P1[e]);
                } else if (!NEXT_LAST) {
                    const int q4 = (j - 8) * 4;            // values q4 .. q4+3 of the 32 scores per lane
                    // v_max3_f32 by hand, volatile: (a) fmaxf() on MFMA results costs two canonicalising v_max each, (b) a pure
                    // expression is emitted next to its only consumer after the last gap; the fences pin a volatile asm to ITS gap.
                    // Operands: S'(t+1) registers whose MFMA chains retired >= 8 gaps (> 256 cycles) ago - no MFMA->VALU hazard.
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mxa) : "v"(Sn[q4 >> 4][q4 & 15]), "v"(Sn[q4 >> 4][(q4 & 15) + 1]));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mxb) : "v"(Sn[q4 >> 4][(q4 & 15) + 2]), "v"(Sn[q4 >> 4][(q4 & 15) + 3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NEXT_LAST) mx_n = tile_max(Sn, t + 1, t + 2 == ntiles);
        else mx_n = half_swap_max(fmaxf(mxa, mxb));
        l_run += ls;
        __syncthreads();
    };
    auto body_last = [&](f32x16 (&Sc)[2], int t, float mx_c) {
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & 1);
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & 1);
        l_run += ls;
    };

    f32x16 sA[2], sB[2];
    dma_k(0, 0);
    dma_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    __syncthreads();
    {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        qk(sA, 0, zero);
    }
    // tile 0 fixes the initial max exactly: m_run = rowmax(S(0)), S'(0) = S(0) - m_run (the only explicit subtraction)
    m_run = tile_max(sA, 0, ntiles == 1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[kb][r] -= m_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -m_run;
    asm volatile("" : "+v"(negm));
    __syncthreads();   // every wave finished reading K[0] before tile 2 is DMA'd over it
    int t = 0;
    float mxA = 0.f, mxB = 0.f;
    // inside the loop neither call may PRODUCE the last tile (t+1 and t+2 <= ntiles-2): no mask code there, the row max rides
    // under the P.V MFMAs; the 1-3 tiles left over take the masked variant for the call that produces the last tile
    for (; t + 3 < ntiles; t += 2) {
        body_main(sA, sB, t, mxA, mxB, std::false_type{});
        body_main(sB, sA, t + 1, mxB, mxA, std::false_type{});
    }
    // the 0-2 iterations left before the final tile: ONE more instance of the body (row max after the gaps, tail mask when the
    // tile it produces is the last), scores handed back through a register copy instead of a second unrolled name swap
    for (; t + 1 < ntiles; ++t) {
        body_main(sA, sB, t, mxA, mxB, std::true_type{});
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sA[kb] = sB[kb];
        mxA = mxB;
    }
    body_last(sA, t, mxA);
    // pin the accumulators under the full EXEC mask: the stores below sit in per-lane `row < n_q` regions and the compiler sinks
    // pure instructions towards their uses (see hv_gemm.hip: the last MFMAs must not end up inside a divergent region)
#pragma unroll
    for (int db = 0; db < 4; ++db) asm volatile("" : : "v"(oT[db]));

    const float l_tot = half_swap_sum(l_run);
    if (a.n_splits > 1 || a.partial) {   // partial result: O^T unnormalised (fp32) + (m, l); merged by attn_combine_kernel
        const int qrow_p = q0 + lr;
        if (qrow_p < a.n_q) {
            const int64_t rowi = ((int64_t)blockIdx.y * a.n_q + qrow_p) * a.n_heads + head;
            float* po = a.part_o + rowi * D + 4 * lh;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(po + db * 32 + g * 8) =
                        make_float4(oT[db][4 * g], oT[db][4 * g + 1], oT[db][4 * g + 2], oT[db][4 * g + 3]);
            if (lh == 0) {
                a.part_ml[rowi * 2] = m_run;
                a.part_ml[rowi * 2 + 1] = l_tot;
            }
        }
        return;
    }
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


// =====================================================================================================================
// v5 = v4 with the per-tile barrier moved two MFMA gaps before the end of the iteration (see gap 30 in body_main): the next
// tile's LDS-DMA issue and its first K fragment reads ride under the last two MFMAs.
// NW = waves per workgroup: 8 (256 query rows) or 4 (128 query rows, two workgroups per CU: the same two waves per SIMD, but each
// workgroup has its own barrier, so the two waves of a SIMD no longer reach every per-tile barrier together)
// (the body is a device function template and the two kernels plain __global__ wrappers: as a __global__ TEMPLATE in this file the
// host stubs of attn_fwd_kernel_v5<4/8> stayed undefined symbols - hipcc 7.2 - although gemm8_kernel's instantiate fine)
// PD = how many MFMA gaps ahead of its MFMA a K/V fragment is read from LDS (ring of RING = 2*PD register slots), see body_main
// ALT (NW = 8): the two waves of a SIMD (w and w + 4) take turns issuing the tile's DMA - in the unrolled pair of iterations the waves
// of group t & 1 issue all eight pieces of their key rows (both 32-key halves), the other group none - so that in every tile one
// wave of each SIMD keeps feeding the matrix pipe while its sibling is busy with descriptor arithmetic and DMA issue
// S2: four-deep K and V tile rings (128 KiB) and ONE barrier per TWO tiles: the odd iteration of the unrolled pair synchronises and
// issues the DMAs of two tiles of each tensor (K(t+4), K(t+5), V(t+3), V(t+4): their slots were released by this or the previous
// barrier), the even iteration has no barrier and no DMA at all; see body_main for the read / landing argument
template <int NW, bool DOT2 = false, int PD = 2, bool ALT = false, bool S2 = false>
__device__ __forceinline__ void attn_v5_body(AttnArgs a) {
    static_assert(!ALT || NW == 8, "ALT pairs waves w and w + 4");
    static_assert(!(ALT && S2), "one experiment at a time");
    constexpr int NBUF = S2 ? 4 : 2;
    constexpr int RING = 2 * PD;
    constexpr int KEYS_W = KVT / NW, NP = KEYS_W / 4;       // keys a wave stages per tile, 1-KiB DMA pieces (4 keys) per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1], 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    if (a.n_splits > 1) {   // this workgroup's key range
        const int kv0 = blockIdx.y * a.split_keys;
        a.k += (int64_t)kv0 * a.sk;
        a.v += (int64_t)kv0 * a.sv;
        a.n_kv = min(a.n_kv - kv0, a.split_keys);
    }
    const int q0 = qt * (QROWS_WAVE * NW) + wave * QROWS_WAVE;
    constexpr int KOFF = 0, VOFF = NBUF * KV_TILE_BYTES;

    bf16x8 qf[8];
    {
        const int qrow = min(q0 + lr, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // Q' = bf16(Q * scale * log2(e)): the scores leave the MFMA already in the exp2 domain (one rounding of q, 2^-9 relative)
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + ks * 16);
            u32x4 sc;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = pack_bf2(bf2f_lo(raw[j]) * a.scale_log2e, bf2f_hi(raw[j]) * a.scale_log2e);
            qf[ks] = __builtin_bit_cast(bf16x8, sc);
        }
    }

    // ---- DMA addressing: wave w owns keys [KEYS_W*w, KEYS_W*(w+1)) of a tile; piece i: key = KEYS_W*w + 4i + (lane>>4), LDS chunk pos = lane&15.
    // LDS-DMA by buffer_load ... lds: the descriptor (4 SGPRs, rebuilt per tile by scalar arithmetic) starts at the tile and covers
    // exactly its valid rows, the per-lane byte offset never changes.  Keys past n_kv in a ragged last tile fall outside the
    // descriptor: the hardware range check returns zeros for them - no clamped copies of the offsets (4 VGPRs), no selects; those
    // keys are masked to -inf after the S MFMAs as before.
    const char* kbase = reinterpret_cast<const char*>(a.k + head * D);
    const char* vbase = reinterpret_cast<const char*>(a.v + head * D);
    uint32_t koff[NP], voff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int key = (ALT ? 8 * (wave & 3) : KEYS_W * wave) + (lane >> 4) + 4 * i, dcp = lane & 15;     // ALT: position inside a 32-key half
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + ((dcp ^ (key & 15)) << 4));
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + ((dcp ^ ((key & 3) << 2)) << 4));
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int dgrp = wave_u >> 2;                                            // ALT: which of a SIMD's two waves; its own half of the keys
    const int wave_lds = (ALT ? (wave_u & 3) : wave_u) * (KEYS_W * 256);     // KEYS_W keys x 256 B; scalar: the DMA destination goes through M0
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    // half: ALT only - which 32-key half of the tile (scalar offset of the buffer load; the descriptor still covers the whole tile, so
    // the range check on voffset + soffset zero-fills keys past n_kv); -1 = both halves (the double-duty turn)
    auto dma_tile = [&](const char* base, int64_t tile_bytes, int row_bytes, const uint32_t (&off)[NP], int tile, int lds_base, int half) {
        const int rows = min(a.n_kv - tile * KVT, KVT);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + tile * tile_bytes), 0, rows * row_bytes, 0x00020000);
        if constexpr (ALT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (half >= 0 && half != h) continue;
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + lds_base + h * 8192 + wave_lds + i * 1024), 16, off[i],
                                                             h * 32 * row_bytes, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + lds_base + wave_lds + i * 1024), 16, off[i], 0, 0, 0);
        }
    };
    auto dma_k = [&](int tile, int buf, int half = -2) {
        dma_tile(kbase, k_tile_bytes, (int)a.sk * 2, koff, tile, KOFF + buf * KV_TILE_BYTES, half == -2 ? dgrp : half);
    };
    auto dma_v = [&](int tile, int buf, int half = -2) {
        dma_tile(vbase, v_tile_bytes, (int)a.sv * 2, voff, tile, VOFF + buf * KV_TILE_BYTES, half == -2 ? dgrp : half);
    };

    int kread[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kb * 32 + lr;
        kread[kb] = key * 256 + ((lh ^ (key & 15)) << 4);
    }
    const int vq = (lane & 15) >> 2, vp = lane & 3, vG = lane >> 4;
    int vread;
    {
        const int key = 4 * (vG >> 1) + vq;
        const int chunk16 = ((vG & 1) * 16 + 4 * vp) >> 3;
        vread = key * 256 + ((chunk16 ^ ((key & 3) << 2)) << 4) + (vp & 1) * 8;
    }

    f32x16 oT[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.f;
    // m_run: the row max the running sums are scaled by (log2 domain).  It is DEFERRED: it only moves when a tile's scores exceed
    // it by more than THR, so P <= 2^THR instead of <= 1 (bf16 and fp32 keep their relative precision at any scale).  -m_run sits
    // in all 16 registers of `negm`, the C operand of the first MFMA of every S chain: S' = K.Q'^T - m_run comes out of the matrix
    // pipe ready for exp2 - no scale/subtract pass on the VALU at all.
    float m_run = 0.f, l_run = 0.f;
    f32x16 negm;
    bf16x8 fr[RING];   // K/V fragment ring (gap g uses slot g % RING, loaded PD gaps ahead; 32 gaps per tile keep the slots aligned)
    constexpr float THR = 8.0f;
    const int ntiles = (a.n_kv + KVT - 1) / KVT;

    auto qk = [&](f32x16 (&S)[2], int buf, const f32x16& c0) {
        const char* kb_ = smem + KOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + (kread[kb] ^ (ks << 5)));
                S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? c0 : S[kb], 0, 0, 0);
            }
        }
    };
    auto exp_block = [&](const f32x16& S, bf16x8 (&pf)[2], float& ls) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float p0 = __builtin_amdgcn_exp2f(S[8 * s + j]);
                const float p1 = __builtin_amdgcn_exp2f(S[8 * s + j + 1]);
                ls += p0;      // ONE serial chain of plain v_add_f32: (p0 + p1) pairs would be SLP-packed into v_pk_add_f32, which
                ls += p1;      // beside MFMAs costs more than it saves
                w[j >> 1] = pack_bf2(p0, p1);
            }
            pf[s] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto pv_block = [&](const bf16x8 (&pf)[2], int kb, int buf) {
        const char* vb_ = smem + VOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (kb * 32 + 16 * s) * 256;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* p0 = vb_ + ((vread + koff) ^ (db << 6));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], oT[db], 0, 0, 0);
            }
        }
    };
    // tail mask (last tile) + row max of a score tile (relative to m_run); VALU only, placed under the P.V MFMAs of the tile before
    auto tile_max = [&](f32x16 (&Sc)[2], int t, bool last) -> float {
        if (last && (a.n_kv & (KVT - 1))) {
            const int kbase_i = t * KVT + 4 * lh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase_i + kb * 32 + (r & 3) + 8 * (r >> 2);
                    if (key >= a.n_kv) Sc[kb][r] = -INFINITY;
                }
        }
        float mx = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[1][r]);
        return half_swap_max(mx);
    };
    // rare: some row of this tile exceeds the running max by more than THR -> move every row's max to its true value, rescale
    // the sums, the scores of THIS tile (already relative to the old max) and the C operand of the next S chain
    auto raise_max = [&](f32x16 (&Sc)[2], float mx) {
        const float d = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
        m_run += d;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Sc[kb][r] -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
        asm volatile("" : "+v"(negm));      // 16 live registers, not 16 re-materialised moves per tile
    };
    // steady-state iteration: consumes Sc = S'(t) and its row max mx_c, produces Sn = S'(t+1) and mx_n.
    // The 32 MFMAs of a tile are laid out as 32 "gaps" in SOURCE order, each closed by a scheduling fence, so that every MFMA is
    // followed by its share of the tile's VALU and LDS work (guide: <= 5 single-issue fillers per 32-cycle MFMA, at most one or two
    // of them an 8-cycle v_exp_f32) instead of the compiler's front-loaded order (all exps under the first seven MFMAs):
    //   gaps  0-15  S'(t+1) chain (K fragments 2 gaps ahead) | exp/pack of P block 0, element g; of block 1, element g/2 (even g)
    //   gaps 16-23  O += V.P block 0 (V fragments 2 ahead)    | exp/pack of P block 1, elements 8..15
    //   gaps 24-31  O += V.P block 1                          | row max of S'(t+1), 4 values per gap
    // NEXT_LAST: tile t+1 may be the (possibly ragged) last one - its masked row max is taken after the gaps instead.
    auto body_main = [&](f32x16 (&Sc)[2], f32x16 (&Sn)[2], int t, float mx_c, float& mx_n, auto next_last_c, auto par_c) {
        constexpr bool NEXT_LAST = decltype(next_last_c)::value;
        constexpr int PAR = decltype(par_c)::value;      // ALT: the wave group whose turn it is to issue this iteration's DMA (-1: everyone its own half)
        // on entry: fr[0], fr[1] hold K(t+1) fragments 0 and 1; the DMAs of K(t+2) and V(t+1) are in flight
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        const char* kb_ = smem + KOFF + ((t + 1) & (NBUF - 1)) * KV_TILE_BYTES;
        const char* vb_ = smem + VOFF + (t & (NBUF - 1)) * KV_TILE_BYTES;
        auto kload = [&](int i) { return *reinterpret_cast<const bf16x8*>(kb_ + (kread[i >> 3] ^ ((i & 7) << 5))); };
        auto vload = [&](int j) {      // j = (kb*2 + s)*4 + db
            const int koff_ = ((j >> 3) * 32 + 16 * ((j >> 2) & 1)) * 256;
            const char* p0 = vb_ + ((vread + koff_) ^ ((j & 3) << 6));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 8 * 256));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const char* kn_ = smem + KOFF + ((t + 2) & (NBUF - 1)) * KV_TILE_BYTES;      // K(t+2) (two buffers: where K(t) was)
        auto knext = [&](int i) { return *reinterpret_cast<const bf16x8*>(kn_ + (kread[i >> 3] ^ ((i & 7) << 5))); };
        u32x4 w0[2], w1[2];
        float P0[16], P1[16];          // this tile's exponentials (compile-time indices: registers, each live for about one gap)
        float ls = 0.f, mxa = -INFINITY, mxb = -INFINITY;
        // row sum: one pinned v_add_f32 per value, issued ONE GAP AFTER its v_exp_f32 (a pure `ls += p` chain is emitted as 32
        // dependent adds after the last gap with all 32 values held live; pinned in the exp's own gap an asm add could sit in the
        // trans->VALU forwarding slot, which the compiler only pads for instructions it can see)
        auto acc = [&](float p) {
            if constexpr (!DOT2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls) : "v"(p));
        };
        // DOT2: the row sum is taken from the PACKED bf16 pairs the P.V MFMAs consume - one v_dot2_f32_bf16 with (1, 1) per pair
        // instead of two v_add_f32 (l = sum of the rounded weights that multiply V), pinned one gap after the pack
        auto acc2 = [&](uint32_t w) {
            if constexpr (DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(ls) : "v"(w), "v"(0x3F803F80u));
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g < 16) {
#ifndef HV_DBG_NOLDS
#ifndef HV_DBG_NOK
                if (g + PD < 16) fr[(g + PD) % RING] = kload(g + PD);
#endif
#ifndef HV_DBG_NOV
                if (g + PD >= 16) fr[(g + PD) % RING] = vload(g + PD - 16);
#endif
#endif
                if (g >= 1) {
                    acc(P0[g - 1]);
                    if (!((g - 1) & 1)) acc(P1[(g - 1) >> 1]);
                    if (!(g & 1)) acc2(w0[(g - 1) >> 3][((g - 1) & 7) >> 1]);            // packed at gap g-1 (odd)
                    if (((g - 1) & 3) == 2) acc2(w1[0][(((g - 1) >> 1) & 7) >> 1]);      // e = (g-1)/2 odd: packed at gap g-1
                }
                Sn[g >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % RING], qf[g & 7], (g & 7) == 0 ? negm : Sn[g >> 3], 0, 0, 0);
                P0[g] = HV_DBG_EXP(Sc[0][g]);
                if (g & 1) w0[g >> 3][(g & 7) >> 1] = pack_bf2(P0[g - 1], P0[g]);
                if (!(g & 1)) {
                    const int e = g >> 1;
                    P1[e] = HV_DBG_EXP(Sc[1][e]);
                    if (e & 1) w1[0][(e & 7) >> 1] = pack_bf2(P1[e - 1], P1[e]);
                }
            } else {
                const int j = g - 16;
                if (g == 32 - PD) {
                    // the tile's one barrier sits HERE, two MFMAs before the end: every LDS read of this iteration has been issued
                    // (the last V fragments at gap 29), the DMAs started a whole iteration ago have long landed, so the next
                    // iteration's DMAs and its first two K fragment reads go under the last two MFMAs instead of in front of an
                    // idle matrix pipe after the barrier (8 waves x ~150 cycles of LDS latency + ~30 scalar instructions per tile).
                    // vmcnt(0): __syncthreads() alone compiles to lgkmcnt(0) + s_barrier here - the compiler does not count LDS-DMA
                    // as something a workgroup fence waits for; the DMAs this retires were issued a whole iteration ago (free)
#ifndef HV_DBG_NOBAR
                    if constexpr (S2) {
                        // S2, t odd (compile-time in the unrolled pair, run-time in the tail iterations): K(t+4) over K(t) (last read
                        // in iteration t-1), K(t+5) over K(t+1) (read in this iteration's S phase), V(t+3) over V(t-1), V(t+4) over
                        // V(t) (read up to gap 29) - all released by this barrier; what the next two iterations read (K(t+2), K(t+3),
                        // V(t+1), V(t+2), and K(t+3)/K(t+4) fragments at their ends) was issued at the previous barrier or this one
                        // and is retired by this barrier's / the next one's vmcnt(0)
                        if (PAR == 1 || (PAR < 0 && (t & 1))) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            if (t + 4 < ntiles) dma_k(t + 4, (t + 4) & 3);
                            if (t + 5 < ntiles) dma_k(t + 5, (t + 5) & 3);
                            if (t + 3 < ntiles) dma_v(t + 3, (t + 3) & 3);
                            if (t + 4 < ntiles) dma_v(t + 4, (t + 4) & 3);
                        }
                    } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (!ALT || PAR < 0) {
                        if (t + 3 < ntiles) dma_k(t + 3, (t + 1) & 1);
                        if (t + 2 < ntiles) dma_v(t + 2, t & 1);
                    } else if (dgrp == PAR) {
                        if (t + 3 < ntiles) dma_k(t + 3, (t + 1) & 1, -1);
                        if (t + 2 < ntiles) dma_v(t + 2, t & 1, -1);
                    }
                    }
#endif
                }
#ifndef HV_DBG_NOLDS
#ifndef HV_DBG_NOV
                if (j + PD < 16) fr[(g + PD) % RING] = vload(j + PD);
#endif
#ifndef HV_DBG_NOK
                if (j + PD >= 16 && t + 2 < ntiles) fr[(g + PD) % RING] = knext(g + PD - 32);      // the last PD gaps -> K(t+2) fragments 0 .. PD-1
#endif
#endif
                if (j == 0) acc(P0[15]);
                if (j >= 1 && j <= 8) acc(P1[7 + j]);
                if (j == 0) acc2(w0[1][3]);                                               // packed at gap 15
                if (j >= 2 && j <= 8 && !(j & 1)) acc2(w1[1][((8 + j - 1) & 7) >> 1]);    // e = 8 + (j-1) odd: packed at gap g-1
                const int s_ = (j >> 2) & 1, db = j & 3;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, j < 8 ? w0[s_] : w1[s_]);
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % RING], pf, oT[db], 0, 0, 0);
                if (j < 8) {
                    const int e = 8 + j;
                    P1[e] = HV_DBG_EXP(Sc[1][e]);
                    if (e & 1) w1[1][(e & 7) >> 1] = pack_bf2(P1[e - 1], P1[e]);
                } else if (!NEXT_LAST) {
                    const int q4 = (j - 8) * 4;            // values q4 .. q4+3 of the 32 scores per lane
                    // v_max3_f32 by hand, volatile: (a) fmaxf() on MFMA results costs two canonicalising v_max each, (b) a pure
                    // expression is emitted next to its only consumer after the last gap; the fences pin a volatile asm to ITS gap.
                    // Operands: S'(t+1) registers whose MFMA chains retired >= 8 gaps (> 256 cycles) ago - no MFMA->VALU hazard.
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mxa) : "v"(Sn[q4 >> 4][q4 & 15]), "v"(Sn[q4 >> 4][(q4 & 15) + 1]));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mxb) : "v"(Sn[q4 >> 4][(q4 & 15) + 2]), "v"(Sn[q4 >> 4][(q4 & 15) + 3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NEXT_LAST) mx_n = tile_max(Sn, t + 1, t + 2 == ntiles);
        else mx_n = half_swap_max(fmaxf(mxa, mxb));
        l_run += ls;
    };
    auto body_last = [&](f32x16 (&Sc)[2], int t, float mx_c) {
        if (__any(mx_c > THR)) raise_max(Sc, mx_c);
        bf16x8 p0[2], p1[2];
        float ls = 0.f;
        exp_block(Sc[0], p0, ls);
        pv_block(p0, 0, t & (NBUF - 1));
        exp_block(Sc[1], p1, ls);
        pv_block(p1, 1, t & (NBUF - 1));
        l_run += ls;
    };

    f32x16 sA[2], sB[2];
    dma_k(0, 0);
    dma_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(0) / V(0) must have landed: the barrier alone does not wait for LDS-DMA
    __syncthreads();
    {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        qk(sA, 0, zero);
    }
    // tile 0 fixes the initial max exactly: m_run = rowmax(S(0)), S'(0) = S(0) - m_run (the only explicit subtraction)
    m_run = tile_max(sA, 0, ntiles == 1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sA[kb][r] -= m_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -m_run;
    asm volatile("" : "+v"(negm));
    __syncthreads();   // every wave finished reading K[0] before tile 2 is DMA'd over it
    // entry state of the first iteration: DMAs of K(2) and V(1) in flight, K(1) fragments 0 and 1 in fr[0], fr[1]
    if constexpr (S2) {
        // everything the first pair of iterations reads before the first in-loop barrier (iteration 1, gap 30) must have landed
        // before the loop starts: K(2..4), V(1..3) now, then one more drain + barrier
        if (ntiles > 2) dma_k(2, 2);
        if (ntiles > 3) dma_k(3, 3);
        if (ntiles > 4) dma_k(4, 0);
        if (ntiles > 1) dma_v(1, 1);
        if (ntiles > 2) dma_v(2, 2);
        if (ntiles > 3) dma_v(3, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else if (ntiles > 2) dma_k(2, 0);
    if (ntiles > 1) {
        if constexpr (!S2) dma_v(1, 1);
        const char* k1_ = smem + KOFF + KV_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PD; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(k1_ + (kread[0] ^ (i << 5)));
    }
    int t = 0;
    float mxA = 0.f, mxB = 0.f;
    // inside the loop neither call may PRODUCE the last tile (t+1 and t+2 <= ntiles-2): no mask code there, the row max rides
    // under the P.V MFMAs; the 1-3 tiles left over take the masked variant for the call that produces the last tile
    for (; t + 3 < ntiles; t += 2) {
        body_main(sA, sB, t, mxA, mxB, std::false_type{}, std::integral_constant<int, 0>{});      // t even
        body_main(sB, sA, t + 1, mxB, mxA, std::false_type{}, std::integral_constant<int, 1>{});
    }
    // the 0-2 iterations left before the final tile: ONE more instance of the body (row max after the gaps, tail mask when the
    // tile it produces is the last), scores handed back through a register copy instead of a second unrolled name swap
    for (; t + 1 < ntiles; ++t) {
        body_main(sA, sB, t, mxA, mxB, std::true_type{}, std::integral_constant<int, -1>{});
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sA[kb] = sB[kb];
        mxA = mxB;
    }
    body_last(sA, t, mxA);
    // pin the accumulators under the full EXEC mask: the stores below sit in per-lane `row < n_q` regions and the compiler sinks
    // pure instructions towards their uses (see hv_gemm.hip: the last MFMAs must not end up inside a divergent region)
#pragma unroll
    for (int db = 0; db < 4; ++db) asm volatile("" : : "v"(oT[db]));

    const float l_tot = half_swap_sum(l_run);
    if (a.n_splits > 1 || a.partial) {   // partial result: O^T unnormalised (fp32) + (m, l); merged by attn_combine_kernel
        const int qrow_p = q0 + lr;
        if (qrow_p < a.n_q) {
            const int64_t rowi = ((int64_t)blockIdx.y * a.n_q + qrow_p) * a.n_heads + head;
            float* po = a.part_o + rowi * D + 4 * lh;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(po + db * 32 + g * 8) =
                        make_float4(oT[db][4 * g], oT[db][4 * g + 1], oT[db][4 * g + 2], oT[db][4 * g + 3]);
            if (lh == 0) {
                a.part_ml[rowi * 2] = m_run;
                a.part_ml[rowi * 2 + 1] = l_tot;
            }
        }
        return;
    }
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

// =====================================================================================================================
// v8: the v5 program on v_mfma_f32_16x16x32_bf16.  Why: the chip holds its clock down under this kernel (~1.7 GHz), and the clock it
// holds depends on the MFMA shape - a 16x16x32 loop delivers ~1.12-1.15x the FLOP/s of a 32x32x16 loop at equal cycles per FLOP
// (MI355X_MICROARCH.md "DVFS give-back" item 7; the GEMMs of this library already use that shape).  Same algorithm, tile sizes, LDS
// images, DMA and register budget; what changes is which lane owns what:
//   S^T block (16 keys x 16 queries) = K frag (lane: key l16, k chunk g) x Q'^T frag (lane: query l16, k chunk g), l16 = lane & 15,
//   g = lane >> 4: a lane holds keys 4g + r (r = 0..3) of key block kb for query (qb, l16) - 8 blocks (4 kb x 2 qb) per wave and KV tile,
//   4 k-steps each; a K fragment feeds the two query blocks.  Row max / row sum: lane-local over 16 values per query block, then the
//   four lanes l16 + 16g by v_permlane16_swap + v_permlane32_swap (no LDS).
//   O^T block (16 dims x 16 queries) += V^T frag x P^T frag: the B operand's 8 k-slots of lane (query l16, g) are its own
//   exponentials of key blocks 2kp and 2kp + 1 (keys 16(2kp) + 4g + r, 16(2kp+1) + 4g + r) - straight from the accumulators through
//   cvt_pk, no lane exchange; the A operand takes the same keys of dim l16 by two ds_read_b64_tr_b16 (one per key block).  8 dim blocks
//   x 2 query blocks x 2 key-block pairs = 32 MFMAs, a V fragment feeds the two query blocks.
// 64 MFMA gaps of 16 cycles per tile: gaps 0-31 S'(t+1) || 24 of the tile's 32 exponentials, gaps 32-47 O += V.P (pair 0) || the other 8,
// gaps 48-63 O += V.P (pair 1) || row max of S'(t+1); K/V fragments through a 4-slot ring two fragment steps (4 gaps) ahead; barrier
// at gap 60, next tile's DMA and first two K fragments under the last four MFMAs.
// compile-time loop: the body sees its index as a constant expression (a `#pragma unroll` loop of 64 large iterations was left
// rolled - an inner loop with runtime indices into P[], pw[], Sc[]: 1,387 scratch instructions, 50 TFLOP/s)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// v8's schedule of the 32 exponentials of a tile over its 64 MFMA gaps: three of every four gaps of the S phase (0..23), every
// second gap of the first P.V phase (24..31); -1 = none
__host__ __device__ constexpr int exp_of_gap(int gp) {
    return gp < 0 ? -1 : gp < 32 ? ((gp & 3) != 3 ? gp - (gp >> 2) : -1) : (gp < 48 && !(gp & 1)) ? 24 + ((gp - 32) >> 1) : -1;
}

__device__ __forceinline__ float quad_rows_max(float v) {       // over the lanes l16 + 16 g, g = 0..3
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

#ifndef HV_DBG_NOP
#define HV_DBG_NOP ""
#endif
#ifndef HV_DBG_NOP2
#define HV_DBG_NOP2 ""
#endif
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v8(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K0 | K1 | V0 | V1], 16 KiB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int head = blockIdx.x / a.n_qtiles;
    const int qt = blockIdx.x % a.n_qtiles;
    if (a.n_splits > 1) {   // this workgroup's key range
        const int kv0 = blockIdx.y * a.split_keys;
        a.k += (int64_t)kv0 * a.sk;
        a.v += (int64_t)kv0 * a.sv;
        a.n_kv = min(a.n_kv - kv0, a.split_keys);
    }
    const int q0 = qt * QTILE + wave * QROWS_WAVE;
    constexpr int KOFF = 0, VOFF = 2 * KV_TILE_BYTES;

    bf16x8 qf[2][4];      // Q' = bf16(Q * scale * log2 e) fragments: [query block][k-step]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = min(q0 + qb * 16 + l16, a.n_q - 1);
        const bf16_t* qp = a.q + (int64_t)qrow * a.sq + head * D + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp + ks * 32);
            u32x4 sc;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = pack_bf2(bf2f_lo(raw[j]) * a.scale_log2e, bf2f_hi(raw[j]) * a.scale_log2e);
            qf[qb][ks] = __builtin_bit_cast(bf16x8, sc);
        }
    }

    // ---- DMA: as v5 (wave w stages keys [8w, 8w+8) of a tile, two 1-KiB pieces per tensor, descriptor sized to the valid rows)
    const char* kbase = reinterpret_cast<const char*>(a.k + head * D);
    const char* vbase = reinterpret_cast<const char*>(a.v + head * D);
    uint32_t koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = 8 * wave + (lane >> 4) + 4 * i, dcp = lane & 15;
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + ((dcp ^ (key & 15)) << 4));
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + ((dcp ^ ((key & 3) << 2)) << 4));
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 2048;
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    auto dma_tile = [&](const char* base, int64_t tile_bytes, int row_bytes, const uint32_t (&off)[2], int tile, int lds_base) {
        const int rows = min(a.n_kv - tile * KVT, KVT);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + tile * tile_bytes), 0, rows * row_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(smem + lds_base + wave_lds + i * 1024), 16, off[i], 0, 0, 0);
    };
    auto dma_k = [&](int tile, int buf) { dma_tile(kbase, k_tile_bytes, (int)a.sk * 2, koff, tile, KOFF + buf * KV_TILE_BYTES); };
    auto dma_v = [&](int tile, int buf) { dma_tile(vbase, v_tile_bytes, (int)a.sv * 2, voff, tile, VOFF + buf * KV_TILE_BYTES); };

    // ---- fragment read addresses.  K: key kb*16 + l16, 16-B chunk ks*4 + g at LDS position chunk ^ (key & 15);
    // V^T (tr read, a 16-lane row = 4 keys x 16 dims): key kb*16 + 4g + vq, dims db*16 + 4vp .. +3, chunk position ^ ((key & 3) << 2)
    const int kread = l16 * 256 + ((g ^ l16) << 4);                                   // + kb * 4096, ^ (ks << 6)
    const int vq = l16 >> 2, vp = l16 & 3;
    const int vread = (4 * g + vq) * 256 + (((vp >> 1) ^ (vq << 2)) << 4) + (vp & 1) * 8;   // + kb * 4096, ^ (db << 5)

    f32x4 oT[2][8];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 8; ++db) oT[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    f32x4 negm[2];       // -m_run of the lane's query in each block: the C operand of every S chain (deferred max, see v3)
    bf16x8 fr[4];        // K/V fragment ring: fragment step s (two gaps) uses slot s % 4, loaded two steps ahead
    constexpr float THR = 8.0f;
    const int ntiles = (a.n_kv + KVT - 1) / KVT;

    auto vfrag = [&](const char* vb_, int kp, int db) {
        const char* p0 = vb_ + kp * 8192 + (vread ^ (db << 5));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 4096));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto qk_plain = [&](f32x4 (&S)[2][4], int buf, const f32x4 (&c0)[2]) {
        const char* kb_ = smem + KOFF + buf * KV_TILE_BYTES;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_ + kb * 4096 + (kread ^ (ks << 6)));
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    S[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], ks == 0 ? c0[qb] : S[qb][kb], 0, 0, 0);
            }
    };
    // tail mask (last tile) + row max of a score tile (relative to m_run) per query block
    auto tile_max = [&](f32x4 (&Sc)[2][4], int t, bool last, float (&mx)[2]) {
        if (last && (a.n_kv & (KVT - 1))) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * KVT + kb * 16 + 4 * g + r >= a.n_kv) Sc[qb][kb][r] = -INFINITY;
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float m = Sc[qb][0][0];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, Sc[qb][kb][r]);
            mx[qb] = quad_rows_max(m);
        }
    };
    // rare: some row exceeds the running max by more than THR -> move every row's max, rescale sums, this tile's scores, the C operand
    auto raise_max = [&](f32x4 (&Sc)[2][4], const float (&mx)[2]) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float d = fmaxf(mx[qb], 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_run[qb] *= alpha;
            m_run[qb] += d;
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) oT[qb][db][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) Sc[qb][kb][r] -= d;
#pragma unroll
            for (int r = 0; r < 4; ++r) negm[qb][r] = -m_run[qb];
        }
        asm volatile("" : "+v"(negm[0]), "+v"(negm[1]));
    };
    // the tile's exponentials in the order the P.V phases need them: v = 0..15 key-block pair 0, 16..31 pair 1;
    // within a pair: (qb, kb & 1, r) = (v >> 3 & 1, v >> 2 & 1, v & 3)
    // steady-state iteration: consumes Sc = S'(t) and its row maxima, produces Sn = S'(t+1) and its maxima
    auto body_main = [&](f32x4 (&Sc)[2][4], f32x4 (&Sn)[2][4], int t, const float (&mx_c)[2], float (&mx_n)[2], auto next_last_c) {
        constexpr bool NEXT_LAST = decltype(next_last_c)::value;
        // on entry: fr[0], fr[1] hold K(t+1) fragments 0 and 1; the DMAs of K(t+2) and V(t+1) are in flight
        if (__any(mx_c[0] > THR || mx_c[1] > THR)) raise_max(Sc, mx_c);
        const char* kb_ = smem + KOFF + ((t + 1) & 1) * KV_TILE_BYTES;
        const char* vb_ = smem + VOFF + (t & 1) * KV_TILE_BYTES;
        const char* kn_ = smem + KOFF + (t & 1) * KV_TILE_BYTES;      // K(t+2) will land where K(t) was
        auto kload = [&](const char* base, int f) {                     // f = kb * 4 + ks
            return *reinterpret_cast<const bf16x8*>(base + (f >> 2) * 4096 + (kread ^ ((f & 3) << 6)));
        };
        u32x4 pw[2][2];                 // packed P: [query block][key-block pair], the B operand of the P.V MFMAs
        float P[32];
        float ls0 = 0.f, ls1 = 0.f, mx0 = -INFINITY, mx1 = -INFINITY;
        auto acc = [&](int v) {          // row sum: one pinned v_add_f32 per value, one gap after its v_exp_f32 (see v4)
            if ((v >> 3) & 1) asm volatile(HV_DBG_NOP "v_add_f32 %0, %0, %1" : "+v"(ls1) : "v"(P[v]));
            else asm volatile(HV_DBG_NOP "v_add_f32 %0, %0, %1" : "+v"(ls0) : "v"(P[v]));
        };
        auto expv = [&](int v) {
            const int kp = v >> 4, qb = (v >> 3) & 1, kbl = (v >> 2) & 1, r = v & 3;
            P[v] = __builtin_amdgcn_exp2f(Sc[qb][2 * kp + kbl][r]);
            if (r & 1) pw[qb][kp][kbl * 2 + (r >> 1)] = pack_bf2(P[v - 1], P[v]);
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 64>([&](auto gpc) {
            constexpr int gp = decltype(gpc)::value;
            constexpr int s_ = gp >> 1;                     // fragment step of this gap
            if constexpr (!(gp & 1)) {                      // even gap: fetch the fragment of step s_ + 2
                constexpr int n = s_ + 2;
                if constexpr (gp == 60) {
                    // the tile's one barrier: every LDS read of this iteration has been issued (the last V fragment at gap 58); the
                    // next iteration's DMAs and its first two K fragments ride under the last four MFMAs
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // see v5: __syncthreads() does not wait for LDS-DMA
                    __syncthreads();
                    if (t + 3 < ntiles) dma_k(t + 3, (t + 1) & 1);
                    if (t + 2 < ntiles) dma_v(t + 2, t & 1);
                }
                if constexpr (n < 16) fr[n % 4] = kload(kb_, n);
                else if constexpr (n < 32) fr[n % 4] = vfrag(vb_, (n - 16) >> 3, (n - 16) & 7);
                else if (t + 2 < ntiles) fr[n % 4] = kload(kn_, n - 32);
            }
            if constexpr (gp < 32) {
                // ---- S'(t+1): K fragment s_ = kb*4 + ks feeds query blocks 0 (even gap) and 1 (odd gap)
                constexpr int kb = s_ >> 2, ks = s_ & 3, qb = gp & 1;
                constexpr int ve = gp - (gp >> 2);              // exponential of this gap (gaps with (gp & 3) == 3 carry none): 0..23
                Sn[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s_ % 4], qf[qb][ks], ks == 0 ? negm[qb] : Sn[qb][kb], 0, 0, 0);
                // row-sum add of the exponential taken TWO gaps ago, behind this gap's MFMA: the compiler pads the trans -> VALU
                // forwarding hazard only for instructions it can see, and one 16-cycle MFMA between a v_exp_f32 and an inline-asm
                // v_add_f32 of its result is not enough (run-to-run different sums: measured) - two MFMAs and the gap's fillers are
                if constexpr (exp_of_gap(gp - 2) >= 0) acc(exp_of_gap(gp - 2));
                if constexpr ((gp & 3) != 3) expv(ve);
            } else {
                constexpr int j = gp - 32;
                constexpr int kp = j >> 4, db = (j >> 1) & 7, qb = j & 1;
                oT[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[s_ % 4], __builtin_bit_cast(bf16x8, pw[qb][kp]), oT[qb][db], 0, 0, 0);
                if constexpr (exp_of_gap(gp - 2) >= 0) acc(exp_of_gap(gp - 2));
                if constexpr (j < 16 && !(j & 1)) expv(24 + (j >> 1));
                if constexpr (j >= 16 && !NEXT_LAST) {
                    // row max of S'(t+1): one v_max3 per gap, two values of one query block (its chains retired >= 16 gaps ago)
                    constexpr int i = j - 16, qm = i & 1, pr = i >> 1;       // pr = 0..7 -> values 2pr, 2pr+1 of the 16 per block
                    constexpr int kbm = pr >> 1, rm = (pr & 1) * 2;
                    if constexpr (qm) asm volatile(HV_DBG_NOP2 "v_max3_f32 %0, %0, %1, %2" : "+v"(mx1) : "v"(Sn[1][kbm][rm]), "v"(Sn[1][kbm][rm + 1]));
                    else asm volatile(HV_DBG_NOP2 "v_max3_f32 %0, %0, %1, %2" : "+v"(mx0) : "v"(Sn[0][kbm][rm]), "v"(Sn[0][kbm][rm + 1]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (NEXT_LAST) tile_max(Sn, t + 1, t + 2 == ntiles, mx_n);
        else {
            mx_n[0] = quad_rows_max(mx0);
            mx_n[1] = quad_rows_max(mx1);
        }
        l_run[0] += ls0;
        l_run[1] += ls1;
    };
    auto body_last = [&](f32x4 (&Sc)[2][4], int t, const float (&mx_c)[2]) {
        if (__any(mx_c[0] > THR || mx_c[1] > THR)) raise_max(Sc, mx_c);
        const char* vb_ = smem + VOFF + (t & 1) * KV_TILE_BYTES;
        u32x4 pw[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(Sc[qb][kb][r]);
                    const float p1 = __builtin_amdgcn_exp2f(Sc[qb][kb][r + 1]);
                    ls += p0;
                    ls += p1;
                    pw[qb][kb >> 1][(kb & 1) * 2 + (r >> 1)] = pack_bf2(p0, p1);
                }
            l_run[qb] += ls;
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const bf16x8 vf = vfrag(vb_, kp, db);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    oT[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pw[qb][kp]), oT[qb][db], 0, 0, 0);
            }
    };

    f32x4 sA[2][4], sB[2][4];
    dma_k(0, 0);
    dma_v(0, 0);
    if (ntiles > 1) dma_k(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(0) / V(0) must have landed: the barrier alone does not wait for LDS-DMA
    __syncthreads();
    {
        const f32x4 zero[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        qk_plain(sA, 0, zero);
    }
    // tile 0 fixes the initial max exactly: m_run = rowmax(S(0)), S'(0) = S(0) - m_run (the only explicit subtraction)
    tile_max(sA, 0, ntiles == 1, m_run);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sA[qb][kb][r] -= m_run[qb];
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[qb][r] = -m_run[qb];
    }
    asm volatile("" : "+v"(negm[0]), "+v"(negm[1]));
    __syncthreads();   // every wave finished reading K[0] before tile 2 is DMA'd over it
    if (ntiles > 2) dma_k(2, 0);
    if (ntiles > 1) {
        dma_v(1, 1);
        const char* k1_ = smem + KOFF + KV_TILE_BYTES;
        fr[0] = *reinterpret_cast<const bf16x8*>(k1_ + kread);
        fr[1] = *reinterpret_cast<const bf16x8*>(k1_ + (kread ^ (1 << 6)));
    }
    int t = 0;
    float mxA[2] = {0.f, 0.f}, mxB[2] = {0.f, 0.f};
    for (; t + 3 < ntiles; t += 2) {
        body_main(sA, sB, t, mxA, mxB, std::false_type{});
        body_main(sB, sA, t + 1, mxB, mxA, std::false_type{});
    }
    for (; t + 1 < ntiles; ++t) {
        body_main(sA, sB, t, mxA, mxB, std::true_type{});
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) sA[qb][kb] = sB[qb][kb];
        mxA[0] = mxB[0];
        mxA[1] = mxB[1];
    }
    body_last(sA, t, mxA);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 8; ++db) asm volatile("" : : "v"(oT[qb][db]));

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = quad_rows_sum(l_run[qb]);
        const int qrow = q0 + qb * 16 + l16;
        if (a.n_splits > 1 || a.partial) {   // partial result: O^T unnormalised (fp32) + (m, l); merged by attn_combine_kernel
            if (qrow < a.n_q) {
                const int64_t rowi = ((int64_t)blockIdx.y * a.n_q + qrow) * a.n_heads + head;
                float* po = a.part_o + rowi * D + 4 * g;
#pragma unroll
                for (int db = 0; db < 8; ++db)
                    *reinterpret_cast<float4*>(po + db * 16) = make_float4(oT[qb][db][0], oT[qb][db][1], oT[qb][db][2], oT[qb][db][3]);
                if (g == 0) {
                    a.part_ml[rowi * 2] = m_run[qb];
                    a.part_ml[rowi * 2 + 1] = l_tot;
                }
            }
            continue;
        }
        const float inv = 1.0f / l_tot;
        if (qrow < a.n_q) {
            bf16_t* op = a.o + (int64_t)qrow * a.so + head * D + 4 * g;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                u32x2 w;
                w[0] = pack_bf2(oT[qb][db][0] * inv, oT[qb][db][1] * inv);
                w[1] = pack_bf2(oT[qb][db][2] * inv, oT[qb][db][3] * inv);
                *reinterpret_cast<u32x2*>(op + db * 16) = w;
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v5(AttnArgs a) { attn_v5_body<8>(a); }
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_v5w4(AttnArgs a) { attn_v5_body<4>(a); }
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v5d(AttnArgs a) { attn_v5_body<8, true>(a); }
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v9(AttnArgs a) { attn_v5_body<8, false, 4>(a); }
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v5a(AttnArgs a) { attn_v5_body<8, false, 2, true>(a); }
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v5s(AttnArgs a) { attn_v5_body<8, false, 2, false, true>(a); }


// merge the KV-split partials: O = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m),  m = max_s m_s  (log2 domain)
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                            bf16_t* __restrict__ o, int64_t so, int n_q, int n_heads, int n_splits) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (row, head, 4 dims)
    const int64_t total = (int64_t)n_q * n_heads * 32;
    if (idx >= total) return;
    const int d4 = (int)(idx & 31);
    const int64_t rh = idx >> 5;            // row * n_heads + head
    const int head = (int)(rh % n_heads);
    const int64_t row = rh / n_heads;
    float m = -INFINITY;
    for (int s = 0; s < n_splits; ++s) m = fmaxf(m, part_ml[(((int64_t)s * n_q + row) * n_heads + head) * 2]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float l = 0.f;
    for (int s = 0; s < n_splits; ++s) {
        const int64_t ri = ((int64_t)s * n_q + row) * n_heads + head;
        const float w = __builtin_amdgcn_exp2f(part_ml[ri * 2] - m);
        const float4 v = *reinterpret_cast<const float4*>(part_o + ri * D + d4 * 4);
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
        l += part_ml[ri * 2 + 1] * w;
    }
    const float inv = 1.0f / l;
    u32x2 w2;
    w2[0] = pack_bf2(acc.x * inv, acc.y * inv);
    w2[1] = pack_bf2(acc.z * inv, acc.w * inv);
    *reinterpret_cast<u32x2*>(o + row * so + head * D + d4 * 4) = w2;
}

HvPerDeviceOnce g_attn_lds_once, g_attn3_lds_once, g_attn4_lds_once, g_attn5_lds_once;

// HV_ATTN_V2=1 / HV_ATTN_VER=2|3: keep a previous kernel (same-box A/B, tests/test_gpu_attention_v3.py); read per call
inline int attn_ver() {
    const char* e2 = std::getenv("HV_ATTN_V2");
    if (e2 && e2[0] == '1') return 2;
    const char* e = std::getenv("HV_ATTN_VER");
    if (e && e[0] == '1' && e[1] == '0') return 10;
    if (e && e[0] == '1' && e[1] == '1') return 11;
    return e && e[0] >= '2' && e[0] <= '9' ? e[0] - '0' : 5;
}

int attn_launch(const AttnArgs& a, dim3 grid, hipStream_t stream) {
    const int ver = attn_ver();
    if (ver == 2) {
        if (hv_set_max_lds(g_attn_lds_once, (const void*)attn_fwd_kernel_v2, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v2<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 5) {
        if (hv_set_max_lds(g_attn5_lds_once, (const void*)attn_fwd_kernel_v5, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v5<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 11) {     // v5 with four-deep K/V rings and one barrier per two tiles
        static HvPerDeviceOnce once11;
        if (hv_set_max_lds(once11, (const void*)attn_fwd_kernel_v5s, 8 * KV_TILE_BYTES) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v5s<<<grid, dim3(512), 8 * KV_TILE_BYTES, stream>>>(a);
    } else if (ver == 10) {     // v5 with the DMA issue taken in turns by the two waves of a SIMD
        static HvPerDeviceOnce once10;
        if (hv_set_max_lds(once10, (const void*)attn_fwd_kernel_v5a, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v5a<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 9) {      // v5 with the K/V fragments read four MFMA gaps ahead (8-slot ring)
        static HvPerDeviceOnce once9;
        if (hv_set_max_lds(once9, (const void*)attn_fwd_kernel_v9, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v9<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 8) {      // the v5 program on v_mfma_f32_16x16x32_bf16
        static HvPerDeviceOnce once8;
        if (hv_set_max_lds(once8, (const void*)attn_fwd_kernel_v8, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v8<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 7) {      // row sums by v_dot2_f32_bf16 on the packed pairs
        static HvPerDeviceOnce once7;
        if (hv_set_max_lds(once7, (const void*)attn_fwd_kernel_v5d, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v5d<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else if (ver == 6) {      // the same kernel as 4-wave workgroups of 128 query rows, two per CU
        static HvPerDeviceOnce once6;
        if (hv_set_max_lds(once6, (const void*)attn_fwd_kernel_v5w4, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        AttnArgs a4 = a;
        a4.n_qtiles = (a.n_q + 127) / 128;
        attn_fwd_kernel_v5w4<<<dim3((unsigned)(a4.n_qtiles * a.n_heads), grid.y), dim3(256), ATT_LDS, stream>>>(a4);
    } else if (ver == 4) {
        if (hv_set_max_lds(g_attn4_lds_once, (const void*)attn_fwd_kernel_v4, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v4<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    } else {
        if (hv_set_max_lds(g_attn3_lds_once, (const void*)attn_fwd_kernel_v3, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
        attn_fwd_kernel_v3<<<grid, dim3(512), ATT_LDS, stream>>>(a);
    }
    return HV_OK;
}

}  // namespace

extern "C" int64_t hv_attn_workspace_bytes(int n_q, int n_kv, int n_heads) {
    // enough for a 2-way KV split (used only when the grid is a few workgroup rounds deep; see hv_attn_fwd_bf16)
    (void)n_kv;
    return (int64_t)2 * n_q * n_heads * (128 + 2) * (int64_t)sizeof(float);
}

extern "C" int hv_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, int64_t stride_q, int64_t stride_k,
                                int64_t stride_v, int64_t stride_o, int n_q, int n_kv, int n_heads, int head_dim,
                                float scale, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
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
    // Load balance: workgroups are equal-cost items on 256 CUs; with only a few rounds (e.g. 3 heads per rank under Ulysses-8:
    // 1395 items = 5.45 rounds -> 6) a partially filled last round costs a whole item.  Splitting the key range in two makes
    // the items half as long (2790 items = 10.9 -> 11 half-rounds = 5.5): taken when it shortens the makespan by > 3 %.
    a.n_splits = 1; a.split_keys = n_kv; a.part_o = nullptr; a.part_ml = nullptr; a.partial = 0;
    {
        const int64_t nwg = (int64_t)a.n_qtiles * n_heads;
        const int ntile = (n_kv + KVT - 1) / KVT;
        const double r1 = (double)((nwg + 255) / 256), r2 = 0.5 * (double)((2 * nwg + 255) / 256);
        if (workspace && ntile >= 64 && r2 < 0.97 * r1 && workspace_bytes >= hv_attn_workspace_bytes(n_q, n_kv, n_heads)) {
            a.n_splits = 2;
            a.split_keys = ((ntile + 1) / 2) * KVT;
            a.part_o = (float*)workspace;
            a.part_ml = a.part_o + (int64_t)2 * n_q * n_heads * D;
        }
    }
    if (attn_launch(a, dim3((unsigned)(a.n_qtiles * n_heads), (unsigned)a.n_splits), stream) != HV_OK) return HV_ERR_LAUNCH;
    if (a.n_splits > 1) {
        const int64_t total = (int64_t)n_q * n_heads * 32;
        attn_combine_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(a.part_o, a.part_ml, a.o, a.so, n_q, n_heads,
                                                                                           a.n_splits);
    }
    return hv_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Ring attention building blocks: attention of the same queries against one K/V CHUNK at a time, each chunk leaving an
// unnormalised partial (O fp32, running max m in the log2 domain, denominator l) in a slot; hv_attn_merge_bf16 folds all
// slots into the normalised bf16 output.  Slot layout: part_o [n_slots][n_q][n_heads][128] fp32, part_ml [n_slots][n_q][n_heads][2].
extern "C" int hv_attn_partial_bf16(const void* q, const void* k, const void* v, int64_t stride_q, int64_t stride_k, int64_t stride_v,
                                    int n_q, int n_kv, int n_heads, int head_dim, float scale, void* part_o, void* part_ml,
                                    int n_slots, int slot, int splits, hipStream_t stream) {
    if (!q || !k || !v || !part_o || !part_ml || head_dim != D || n_heads <= 0 || n_q < 0 || n_kv <= 0 || (stride_q & 7) ||
        (stride_k & 7) || (stride_v & 7) || (splits != 1 && splits != 2) || slot < 0 || slot + splits > n_slots)
        return HV_ERR_ARG;
    if (n_q == 0) return HV_OK;
    const int ntile = (n_kv + KVT - 1) / KVT;
    if (splits == 2 && ntile < 2) return HV_ERR_ARG;
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = nullptr;
    a.sq = stride_q; a.sk = stride_k; a.sv = stride_v; a.so = 0;
    a.n_q = n_q; a.n_kv = n_kv; a.n_heads = n_heads;
    a.n_qtiles = (n_q + QTILE - 1) / QTILE;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.n_splits = splits;
    a.split_keys = splits == 2 ? ((ntile + 1) / 2) * KVT : n_kv;
    a.part_o = (float*)part_o + (int64_t)slot * n_q * n_heads * D;
    a.part_ml = (float*)part_ml + (int64_t)slot * n_q * n_heads * 2;
    a.partial = 1;
    if (attn_launch(a, dim3((unsigned)(a.n_qtiles * n_heads), (unsigned)splits), stream) != HV_OK) return HV_ERR_LAUNCH;
    return hv_check_launch();
}

extern "C" int hv_attn_merge_bf16(const void* part_o, const void* part_ml, void* o, int64_t stride_o, int n_q, int n_heads, int n_slots,
                                  hipStream_t stream) {
    if (!part_o || !part_ml || !o || n_heads <= 0 || n_q < 0 || n_slots <= 0 || (stride_o & 3)) return HV_ERR_ARG;
    if (n_q == 0) return HV_OK;
    const int64_t total = (int64_t)n_q * n_heads * 32;
    attn_combine_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>((const float*)part_o, (const float*)part_ml,
                                                                                       (bf16_t*)o, stride_o, n_q, n_heads, n_slots);
    return hv_check_launch();
}

// 1 or 2: whether halving the key range shortens the makespan of the (n_q/256 x n_heads)-workgroup grid on 256 CUs by > 3 %
extern "C" int hv_attn_suggest_splits(int n_q, int n_kv, int n_heads) {
    const int64_t nwg = (int64_t)((n_q + QTILE - 1) / QTILE) * n_heads;
    const int ntile = (n_kv + KVT - 1) / KVT;
    const double r1 = (double)((nwg + 255) / 256), r2 = 0.5 * (double)((2 * nwg + 255) / 256);
    return (ntile >= 64 && r2 < 0.97 * r1) ? 2 : 1;
}
