// Flash attention forward, bf16, head_dim 128, non-causal, one contiguous KV segment (K6 / K6' / K18-style
// callers pick the segment by pointer + length).  softmax(scale * Q K^T) V with fp32 statistics.
// Replaces /root/reference/hyvideo/modules/attenion.py:107-120 (flash_attn_varlen_func per cu_seqlens segment).
//
// gfx950 structure (8 waves x 32 query rows = 256 rows per workgroup, KV tile = 64 keys) - ONE kernel ships; the superseded
// generations and the rejected experiments of DESIGN.md section 7 live in tools/attn_variants/ (built only for same-box A/Bs):
//   * S^T = K . Q'^T  ("swapped" QK^T): v_mfma_f32_32x32x16_bf16 with the K tile as the A operand and the wave's Q rows
//     (pre-multiplied by scale*log2e, held in registers for the whole kernel) as B.  The accumulator then has the QUERY on
//     the lane (col = lane&31) and 16 keys per lane in registers, so the softmax row max / row sum are lane-local
//     (+ one v_permlane32_swap between the two half-waves) - no LDS, no shuffles.
//   * the running max is DEFERRED (threshold 2^8) and enters as the C operand of each S chain: P = exp2(S') needs no
//     scale/subtract pass.
//   * O^T += V^T . P^T: the exponentiated accumulator registers, converted pairwise to bf16, ARE the B operand of the next
//     MFMA (k = key); V^T fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16.
//   * K and V tiles arrive by LDS-DMA (buffer_load ... lds; the bank swizzle is applied on the per-lane SOURCE offset so
//     the LDS image stays lane-linear; the descriptor covers exactly the tile's valid rows, so a ragged last tile reads
//     zeros): K two tiles ahead, V one tile ahead, 2-deep rings; SQ_LDS_BANK_CONFLICT = 0.
//   * the 32 MFMAs of a tile are laid out gap by gap in source order (see body_main); one workgroup barrier per tile, two
//     gaps before the end of the iteration.
// Workgroups are ordered head-major so co-resident workgroups stream the same head's K/V (L2 / MALL reuse).
// Algorithmic work: 4 * n_q * n_kv * 128 flop per head.
#include "hv_attention.hpp"

using namespace hv_attn;

namespace {

constexpr int QROWS_WAVE = 32;
constexpr int NWAVES = 8;
constexpr int QTILE = QROWS_WAVE * NWAVES;  // 256
constexpr int BUF_BYTES = 2 * KV_TILE_BYTES;     // K + V
constexpr int ATT_LDS = 2 * BUF_BYTES;           // 64 KiB

// =====================================================================================================================
// v5 = v4 with the per-tile barrier moved two MFMA gaps before the end of the iteration (see gap 30 in body_main): the next
// tile's LDS-DMA issue and its first K fragment reads ride under the last two MFMAs.
__device__ __forceinline__ void attn_v5_body(AttnArgs a) {
    constexpr int NW = NWAVES;
    constexpr int NBUF = 2;              // K and V tile rings
    constexpr int PD = 2, RING = 2 * PD; // K/V fragments are read from LDS PD MFMA gaps ahead of their MFMA (ring of RING register slots)
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
        const int key = KEYS_W * wave + (lane >> 4) + 4 * i, dcp = lane & 15;
        koff[i] = (uint32_t)(key * (int)a.sk * 2 + ((dcp ^ (key & 15)) << 4));
        voff[i] = (uint32_t)(key * (int)a.sv * 2 + ((dcp ^ ((key & 3) << 2)) << 4));
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wave_lds = wave_u * (KEYS_W * 256);     // KEYS_W keys x 256 B; scalar: the DMA destination goes through M0
    const int64_t k_tile_bytes = (int64_t)KVT * a.sk * 2, v_tile_bytes = (int64_t)KVT * a.sv * 2;
    auto dma_tile = [&](const char* base, int64_t tile_bytes, int row_bytes, const uint32_t (&off)[NP], int tile, int lds_base) {
        const int rows = min(a.n_kv - tile * KVT, KVT);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + tile * tile_bytes), 0, rows * row_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NP; ++i)
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
    auto body_main = [&](f32x16 (&Sc)[2], f32x16 (&Sn)[2], int t, float mx_c, float& mx_n, auto next_last_c) {
        constexpr bool NEXT_LAST = decltype(next_last_c)::value;
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
        auto acc = [&](float p) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls) : "v"(p)); };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (g < 16) {
                if (g + PD < 16) fr[(g + PD) % RING] = kload(g + PD);
                if (g + PD >= 16) fr[(g + PD) % RING] = vload(g + PD - 16);
                if (g >= 1) {
                    acc(P0[g - 1]);
                    if (!((g - 1) & 1)) acc(P1[(g - 1) >> 1]);
                }
                Sn[g >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % RING], qf[g & 7], (g & 7) == 0 ? negm : Sn[g >> 3], 0, 0, 0);
                P0[g] = __builtin_amdgcn_exp2f(Sc[0][g]);
                if (g & 1) w0[g >> 3][(g & 7) >> 1] = pack_bf2(P0[g - 1], P0[g]);
                if (!(g & 1)) {
                    const int e = g >> 1;
                    P1[e] = __builtin_amdgcn_exp2f(Sc[1][e]);
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
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (t + 3 < ntiles) dma_k(t + 3, (t + 1) & 1);
                    if (t + 2 < ntiles) dma_v(t + 2, t & 1);
                }
                if (j + PD < 16) fr[(g + PD) % RING] = vload(j + PD);
                if (j + PD >= 16 && t + 2 < ntiles) fr[(g + PD) % RING] = knext(g + PD - 32);      // the last PD gaps -> K(t+2) fragments 0 .. PD-1
                if (j == 0) acc(P0[15]);
                if (j >= 1 && j <= 8) acc(P1[7 + j]);
                const int s_ = (j >> 2) & 1, db = j & 3;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, j < 8 ? w0[s_] : w1[s_]);
                oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % RING], pf, oT[db], 0, 0, 0);
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
    if (ntiles > 2) dma_k(2, 0);
    if (ntiles > 1) {
        dma_v(1, 1);
        const char* k1_ = smem + KOFF + KV_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < PD; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(k1_ + (kread[0] ^ (i << 5)));
    }
    int t = 0;
    float mxA = 0.f, mxB = 0.f;
    // inside the loop neither call may PRODUCE the last tile (t+1 and t+2 <= ntiles-2): no mask code there, the row max rides
    // under the P.V MFMAs; the 1-3 tiles left over take the masked variant for the call that produces the last tile
    for (; t + 3 < ntiles; t += 2) {
        body_main(sA, sB, t, mxA, mxB, std::false_type{});      // t even
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

__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_v5(AttnArgs a) { attn_v5_body(a); }


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

HvPerDeviceOnce g_attn_lds_once;

// Which kernel the C ABI launches is a BUILD decision (no run-time selection in the product library): 0 = the 8-wave x 32-row
// kernel of this file, 1 = the 4-wave x 64-row kernel of hv_attention_w4.hip (same AttnArgs, same 256-row workgroup tiles).
#ifndef HV_ATTN_USE_W4
#define HV_ATTN_USE_W4 0
#endif

int attn_launch(const AttnArgs& a, dim3 grid, hipStream_t stream) {
    if (HV_ATTN_USE_W4) return launch_w4(a, grid, stream);
    if (hv_set_max_lds(g_attn_lds_once, (const void*)attn_fwd_kernel_v5, ATT_LDS) != HV_OK) return HV_ERR_LAUNCH;
    attn_fwd_kernel_v5<<<grid, dim3(512), ATT_LDS, stream>>>(a);
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
