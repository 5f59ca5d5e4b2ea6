// Lean bf16 MFMA GEMM for the UNet's token-major projections (gfx950):  C[M,N] = X[M,K] . W[N,K]^T  + epilogue.
//
// Same tile, LDS image, swizzle and MFMA schedule as gemm_conv.hip's gemm_kernel (128 x 160 or 64 x 160, four waves, two
// persistent workgroups per CU, 2-stage LDS ring filled by 16-byte LDS-DMA), re-built around ONE measurement of round 3
// (profiles/round3_notes.md, "Where the GEMM group's time goes"): at K = 320 ... 1600 a work item's K loop is ~330
// instructions, while the general kernel spent ~630 instructions on the next item's decode + per-lane source pointers and
// ~1000 on its epilogue (64-bit address arithmetic per load / store, bounds checks, zero-page selects) -- and with two
// waves per SIMD the vector issue port, not the matrix pipe, was what the K = 320 GEMMs ran at.  Here:
//   * every global access is a BUFFER instruction: wave-uniform resource (SGPRs) + a per-lane 32-bit offset that is
//     computed ONCE per kernel + a scalar offset per item / K tile / row block + an immediate.  No per-item vector
//     address arithmetic at all; rows beyond M read zeros and their stores are dropped by the hardware's range check;
//   * the item decode is scalar and incremental (no division per item);
//   * full N tiles only (N % 160 == 0: every UNet projection), one store schedule, the bias / c1 / c2 vectors through
//     LDS by LDS-DMA under the first K tile, the residual and LayerNorm partial loads under the LAST K tile;
//   * everything the general kernel's std epilogue offers to the UNet plan: second K segment (virtual concat), bias +
//     bias2, residual, LayerNorm fold (consumer), LayerNorm row partials and GroupNorm block statistics (producer),
//     head-major K / V stores -- bit-identical results (same accumulation order, same rounding points).
// Shapes it does not take (N tails, split-K, per-sample weights, fp8, GEGLU / softmax epilogues, K < 128) stay on
// gemm_kernel; sd_launch_gemm decides (gemm_conv.hip).  Reference call site: src/models.py:227-235.
#include "common.h"
#include "kernels.h"
#include <mutex>
#include <stdlib.h>

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct LeanArgs {
    const void* X; const void* X2; const void* W; const void* R; void* C; void* KV;
    const float* vec0; const float* vec1;      // LN fold: c1, c2;  else: bias, bias2 (either may be null)
    const float* ln_rs; float* rowstats; float* stats;
    unsigned x_bytes, x2_bytes, w_bytes, r_bytes, c_bytes, kv_bytes;
    int M, N, K, K1;
    int ldxb, ldx2b, ldwb, ldrb, ldcb;         // row strides in BYTES
    int tiles_m, tiles_n;
    float ln_eps, inv_k;
    int hm_tpc;                                // head-major K / V: 160-column tiles per C (0 = off)
    int hm_tok_shift, hm_heads, hm_samples;    // tokens per sample = 1 << shift
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
// 16 B per lane global -> LDS; LDS destination = wave-uniform base + lane * 16; source = rsrc base + voff + soff
__device__ __forceinline__ void dma16(rsrc_t r, void* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds_wave_base), 16, voff, soff, 0, 0);
}

// LN = true: the LayerNorm-fold consumer (q|k|v with optional head-major K / V): vec0 = c1, vec1 = c2, no residual, one K
// segment.  LN = false: bias (+ bias2) (+ residual) (+ second K segment), optional row partials / block statistics.
// Two instantiations instead of one kernel with every feature live: the union ran out of scalar registers (11 buffer
// resources = 44 SGPRs) and spilled them into vector lanes.  Buffer resources only for the streams that need the range
// check or carry most of the traffic (X, X2, W, C / KV, R); the small side tensors use scalar-base global accesses.
// NP = number of LayerNorm partials per row (2, 4, 8, 16), 0 = the LN = false kernel.  GEGLU: the 256 x 256 tile on 8 waves
// (one workgroup per CU: half the LDS-fill bytes per flop of 128 x 128), weight rows packed [16 value | 16 gate] per 32
// columns, output N / 2 columns of value * gelu(gate); LayerNorm fold or plain bias, nothing else.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NP, bool GEGLU>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_lean_kernel(const LeanArgs p) {
    constexpr bool LN = NP > 0;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int XI = BM / 8, WI = BN / 8;
    static_assert(XI % NW == 0 && WI % NW == 0, "pieces split evenly over the waves");
    constexpr int XPW = XI / NW, WPW = WI / NW;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(GEGLU ? TN % 4 == 0 : TN % 2 == 1, "store schedule: GEGLU pairs of output tiles; std TN / 2 paired + one 8-byte store");
    constexpr bool FRAG_DB = TM * TN * 4 + (TM + TN) * 8 <= 200;     // accumulators + two fragment sets fit 256 registers
    constexpr int TNH = GEGLU ? TN / 2 : TN;                         // bias / c1 / c2 vectors held at a time

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lnbuf = smem + 2 * STAGE_BYTES;                 // [waves][64 rows] (mean, rstd)
    char* vecs = lnbuf + NW * 512;                        // [2][BN] fp32: vec0 | vec1 of this tile's columns

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lrow = lane & 15, lq = lane >> 4;

    // ---- persistent work order: XCD-aware (blocks with equal blockIdx % 8 walk consecutive items, n fastest) -----------
    const int nblk = gridDim.x;
    int work;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nwork = p.tiles_m * p.tiles_n;
    if (work >= nwork) return;
    int tile_m = work / p.tiles_n, tile_n = work - tile_m * p.tiles_n;      // the only divisions of the kernel
    const int dm = nblk / p.tiles_n, dn = nblk - dm * p.tiles_n;

    const rsrc_t rX = make_rsrc(p.X, p.x_bytes), rW = make_rsrc(p.W, p.w_bytes), rC = make_rsrc(p.C, p.c_bytes);
    const rsrc_t rX2 = make_rsrc(LN ? p.X : p.X2, LN ? 0 : p.x2_bytes), rR = make_rsrc(LN ? p.X : p.R, LN ? 0 : p.r_bytes);
    const rsrc_t rKV = make_rsrc(LN ? p.KV : p.C, LN ? p.kv_bytes : 0);

    // ---- per-lane offsets, computed once ------------------------------------------------------------------------------
    const int lrow8 = lane >> 3, gch = (lane & 7) ^ lrow8;        // LDS-DMA piece: 8 rows x 128 B, chunk swizzle on the source
    const int vX = lrow8 * p.ldxb + gch * 16, vX2 = lrow8 * p.ldx2b + gch * 16, vW = lrow8 * p.ldwb + gch * 16;
    const int rowl = wm * WTM + lrow;                             // this lane's row inside the tile (+ 16 b)
    const int vR = rowl * p.ldrb + (wn * WTN + lq * 4) * 2;       // residual: 4 columns (8 B) per 16 x 16 tile
    const int vCw = rowl * p.ldcb + (wn * WTN + (lq & 1) * 16 + (lq >> 1) * 8) * 2;    // paired tiles: 8 columns (16 B)
    const int vCn = rowl * p.ldcb + (wn * WTN + lq * 4) * 2;                           // last (odd) tile: 4 columns
    const int vCg = rowl * p.ldcb + (wn * WTN / 2 + (lq & 1) * 16 + (lq >> 1) * 8) * 2;  // GEGLU: 8 of the N / 2 output columns
    // head-major K / V: column cl of the tile -> (head cl / 40, channel cl % 40); an 8-column group never straddles a head
    int vH[TN / 2 + 1] = {};
#pragma unroll
    for (int j = 0; LN && !GEGLU && j <= TN / 2; ++j) {
        const int cl = wn * WTN + (j < TN / 2 ? j * 32 + (lq & 1) * 16 + (lq >> 1) * 8 : (TN - 1) * 16 + lq * 4);
        const int hl = (cl * 205) >> 13;
        vH[j] = ((hl << p.hm_tok_shift) + rowl) * 80 + (cl - hl * 40) * 2;
    }
    const int vLn = (wm * WTM + lane) * 8;                        // LayerNorm partials: lane l owns row l of its wave's 64
    const int vRs = rowl * 8;
    const int vSt = (wn * WTN + lq * 4) * 8;
    const int vVec = (wn * WTN + lane * 4) * 4;                   // lanes < WTN / 4 of the waves wm == 0

    // bf16 fragments: k-step ks reads chunk (4 ks + lq) ^ (row & 7)
    const int swz[2] = {((lq) ^ (lane & 7)) << 4, ((4 + lq) ^ (lane & 7)) << 4};
    const int xoff = (wm * WTM + lrow) * 128;
    const int woff = BM * 128 + (wn * WTN + lrow) * 128;

    const int KT = p.K >> 6;
    int m0 = tile_m * BM, n0 = tile_n * BN;

    auto stage = [&](int kt, int buf) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + BM * 128;
        const int k0 = kt * 64;
        if (LN || k0 < p.K1) {
            const int so = (m0 + wave * 8) * p.ldxb + k0 * 2;
#pragma unroll
            for (int i = 0; i < XPW; ++i) dma16(rX, xs + (wave + i * NW) * 1024, vX, so + i * (NW * 8) * p.ldxb);
        } else {
            const int so = (m0 + wave * 8) * p.ldx2b + (k0 - p.K1) * 2;
#pragma unroll
            for (int i = 0; i < XPW; ++i) dma16(rX2, xs + (wave + i * NW) * 1024, vX2, so + i * (NW * 8) * p.ldx2b);
        }
        const int sw = (n0 + wave * 8) * p.ldwb + k0 * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma16(rW, ws + (wave + i * NW) * 1024, vW, sw + i * (NW * 8) * p.ldwb);
    };

    f32x4 acc[TN][TM];
    bf16x8 xf0[TM], wf0[TN], xf1[FRAG_DB ? TM : 1], wf1[FRAG_DB ? TN : 1];
    auto load_frags = [&](const char* sb, int ks, bf16x8* xf, bf16x8* wf) {
#pragma unroll
        for (int t = 0; t < TM; ++t) xf[t] = *(const bf16x8*)(sb + xoff + t * 2048 + swz[ks]);
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[t] = *(const bf16x8*)(sb + woff + t * 2048 + swz[ks]);
    };
    auto mfmas = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    };

    // paired 16-byte stores + one 8-byte store per row block (always: full tiles) -- the counted wait below relies on it
    constexpr int EPI_STORES = GEGLU ? (TN / 4) * TM : (TN / 2 + 1) * TM;
    int g = 0;                     // running K-tile count: LDS buffer parity
    bool stores_pending = false;
    stage(0, 0);
    while (true) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- K tiles 0 .. KT - 2 ---------------------------------------------------------------------------------------
        for (int kt = 0; kt < KT - 1; ++kt) {
            // vmcnt counts stores too and retires in order: the first K tile of an item was issued BEFORE the previous
            // item's stores, so a counted wait retires the DMA and leaves those stores in flight under this tile's MFMAs
            if (kt == 0 && stores_pending) wait_vmcnt<EPI_STORES>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sb = smem + ((g + kt) & 1) * STAGE_BYTES;
            load_frags(sb, 0, xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            stage(kt + 1, (g + kt + 1) & 1);
            if (kt == 0 && wm == 0 && lane < WTN / 4) {           // bias / c1 / c2 of this wave's 80 columns -> LDS
                glds16((const char*)p.vec0 + (long)n0 * 4 + (unsigned)vVec, vecs + wn * WTN * 4);
                glds16((const char*)p.vec1 + (long)n0 * 4 + (unsigned)vVec, vecs + BN * 4 + wn * WTN * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FRAG_DB) {
                load_frags(sb, 1, xf1, wf1);
                mfmas(xf0, wf0);
                mfmas(xf1, wf1);
            } else {                 // register budget: one fragment set, reloaded between the k-steps
                mfmas(xf0, wf0);
                load_frags(sb, 1, xf0, wf0);
                mfmas(xf0, wf0);
            }
        }
        // ---- last K tile: the epilogue's loads are issued under its MFMAs (no LDS-DMA is in flight here) ---------------
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        static_assert(!LN || WTM == 64, "LayerNorm fold: lane l owns row l of its wave's 64 rows");
        u32x2 ep[LN ? NP : (GEGLU ? 1 : TN * TM)];        // the row's LayerNorm partials, or the residual's 8-byte pieces
        {
            const char* sb = smem + ((g + KT - 1) & 1) * STAGE_BYTES;
            load_frags(sb, 0, xf0, wf0);
            if (FRAG_DB) load_frags(sb, 1, xf1, wf1);
            if (LN) {
                const char* src = (const char*)p.ln_rs + (long)m0 * 8;
#pragma unroll
                for (int i = 0; i < NP; ++i) ep[i] = *(const u32x2*)((src + (long)i * p.M * 8) + (unsigned)vLn);
            } else if (!GEGLU && p.R) {
                const int so = m0 * p.ldrb + n0 * 2;
#pragma unroll
                for (int b = 0; b < TM; ++b)
#pragma unroll
                    for (int a = 0; a < TN; ++a)
                        ep[a * TM + b] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rR, vR + a * 32, so + b * 16 * p.ldrb, 0));
            }
            __builtin_amdgcn_sched_barrier(0);      // (the epilogue's arithmetic -- and its wait for these loads -- stays below the MFMAs)
            mfmas(xf0, wf0);
            if (FRAG_DB) {
                mfmas(xf1, wf1);
            } else {
                load_frags(sb, 1, xf0, wf0);
                mfmas(xf0, wf0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        g += KT;

        // ---- epilogue, phase A: fold everything that was loaded into the fp32 accumulators ------------------------------
        f32x4 v0[TNH], v1[TNH];
        auto vec_load = [&](int a0) {
#pragma unroll
            for (int a = 0; a < TNH; ++a) {
                v0[a] = *(const f32x4*)(vecs + (wn * WTN + (a0 + a) * 16 + lq * 4) * 4);
                v1[a] = *(const f32x4*)(vecs + BN * 4 + (wn * WTN + (a0 + a) * 16 + lq * 4) * 4);
            }
        };
        vec_load(0);
        if (LN) {
            // LayerNorm fold: rstd * (acc - mean * c1[n]) + c2[n]
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                // (bit_cast of the WHOLE vector: with this hipcc a bit_cast of element [1] of an ext-vector reads element [0])
                const f32x2_t e = __builtin_bit_cast(f32x2_t, ep[i]);
                s += e[0];
                q += e[1];
            }
            *(f32x2_t*)(lnbuf + wave * 512 + lane * 8) = ln_mean_rstd(s, q, p.inv_k, p.ln_eps);   // row = lane; lanes need rows 16 b + lrow
            // (The reads below take OTHER lanes' (mean, rstd) from this wave's slots; LDS operations of one wave execute in
            // order, no wait is needed between the write and the reads.  Round 4 kept an explicit s_waitcnt lgkmcnt(0) here
            // because the first launch of the 8-wave GEGLU instantiation differed in a few dozen elements without it: that
            // was the packed-FMA erratum of common.h::ln_fold -- the wait only moved the timing; with the scalar fold 0 of
            // 1400 launches deviate without it, profiles/round5_notes.md.)
            f32x2_t mr[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) mr[b] = *(const f32x2_t*)(lnbuf + wave * 512 + (b * 16 + lrow) * 8);
#pragma unroll
            for (int a0 = 0; a0 < TN; a0 += TNH) {
                if (a0 > 0) vec_load(a0);
#pragma unroll
                for (int a = 0; a < TNH; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) acc[a0 + a][b] = ln_fold(acc[a0 + a][b], v0[a], mr[b][0], mr[b][1], v1[a]);
            }
        } else {
            if (!GEGLU && p.R) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const u32x2 r = ep[a * TM + b];
                        acc[a][b][0] += bflo(r[0]); acc[a][b][1] += bfhi(r[0]);
                        acc[a][b][2] += bflo(r[1]); acc[a][b][3] += bfhi(r[1]);
                    }
            }
#pragma unroll
            for (int a0 = 0; a0 < TN; a0 += TNH) {
                if (a0 > 0) vec_load(a0);
#pragma unroll
                for (int a = 0; a < TNH; ++a) {
                    const f32x4 bv = v0[a] + v1[a];               // (an absent vector is the launcher's zero vector)
#pragma unroll
                    for (int b = 0; b < TM; ++b) acc[a0 + a][b] += bv;
                }
            }
        }

        // ---- next item: scalar decode, first K tile in flight before this item's stores ---------------------------------
        const int em0 = m0, en0 = n0, etile_n = tile_n;
        work += nblk;
        const bool more = work < nwork;
        if (more) {
            tile_n += dn; tile_m += dm;
            if (tile_n >= p.tiles_n) { tile_n -= p.tiles_n; tile_m += 1; }
            m0 = tile_m * BM; n0 = tile_n * BN;
            // (every wave has read vecs and its lnbuf slots: the LDS reads above are waited for by their uses; the stage
            // buffer g & 1 was last read two K tiles ago, before the last barrier)
            stage(0, g & 1);
            stores_pending = true;
        }

        // ---- epilogue, phase B: statistics, convert, store ---------------------------------------------------------------
        if (!LN && !GEGLU && WTM == 64 && p.stats) {
            // GroupNorm block statistics of the bf16-rounded outputs: [M / 64][N][2], this wave's 64 rows = one block
            char* dst = (char*)p.stats + ((long)((em0 + wm * WTM) >> 6) * p.N + en0) * 8 + (unsigned)vSt;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                float sv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const unsigned p0 = pack2bf(acc[a][b][0], acc[a][b][1]), p1 = pack2bf(acc[a][b][2], acc[a][b][3]);
                    const float v[4] = {bflo(p0), bfhi(p0), bflo(p1), bfhi(p1)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sv[j] += v[j]; qv[j] = __builtin_fmaf(v[j], v[j], qv[j]); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sv[j] = row_shr_add<8>(row_shr_add<4>(row_shr_add<2>(row_shr_add<1>(sv[j]))));
                    qv[j] = row_shr_add<8>(row_shr_add<4>(row_shr_add<2>(row_shr_add<1>(qv[j]))));
                }
                if (lrow == 15) {
                    *(f32x4*)(dst + a * 128) = f32x4{sv[0], qv[0], sv[1], qv[1]};
                    *(f32x4*)(dst + a * 128 + 16) = f32x4{sv[2], qv[2], sv[3], qv[3]};
                }
            }
        }
        if (!LN && !GEGLU && p.rowstats) {
            // LayerNorm partials of the bf16-rounded outputs: [2 tiles_n][M][2], per row the (sum, sum of squares) of this
            // N-wave's 80 columns
            char* dst = (char*)p.rowstats + ((long)(etile_n * WAVES_N + wn) * p.M + em0) * 8 + (unsigned)vRs;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const unsigned p0 = pack2bf(acc[a][b][0], acc[a][b][1]), p1 = pack2bf(acc[a][b][2], acc[a][b][3]);
                    const float v[4] = {bflo(p0), bfhi(p0), bflo(p1), bfhi(p1)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s += v[j]; q = __builtin_fmaf(v[j], v[j], q); }
                }
                s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (lq == 0) *(f32x2_t*)(dst + b * 16 * 8) = f32x2_t{s, q};
            }
        }
        if constexpr (GEGLU) {
            // value * gelu(gate) of the accumulator pairs (a, a + 1); two output tiles (four accumulator tiles) are paired by
            // v_permlane16_swap into 16-byte stores: lane group lq then holds output columns 16 (lq & 1) + 8 (lq >> 1) .. + 7
            auto geglu_tile = [&](int a, int b) -> u32x2 {
                const f32x4 va = acc[a][b], vg = acc[a + 1][b];
                const f32x2_t lo = geglu_pair(f32x2_t{va[0], va[1]}, f32x2_t{vg[0], vg[1]});
                const f32x2_t hi = geglu_pair(f32x2_t{va[2], va[3]}, f32x2_t{vg[2], vg[3]});
                return u32x2{pack2bf(lo[0], lo[1]), pack2bf(hi[0], hi[1])};
            };
            const int soC = em0 * p.ldcb + en0;                    // (n0 / 2 output columns x 2 bytes)
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int vo = vCg + (soC + b * 16 * p.ldcb);      // STORE_OFFSET_RULE (see the std stores below)
#pragma unroll
                for (int a = 0; a < TN; a += 4) {
                    const u32x2 ox = geglu_tile(a, b), oy = geglu_tile(a + 2, b);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(ox[0], oy[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(ox[1], oy[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rC, vo + a * 16, 0, 0);
                }
            }
        } else {
        // v_permlane16_swap pairs two adjacent 16 x 16 tiles so that a lane owns 8 consecutive columns (16 B) of its row:
            // after the swap lane group lq holds  0: tile a cols 0-7, 1: tile a+1 cols 0-7, 2: tile a cols 8-15, 3: tile a+1 cols 8-15
            u32x4 ow[TM][TN / 2];
            u32x2 on[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
#pragma unroll
                for (int a = 0; a + 1 < TN; a += 2) {
                    const f32x4 vx = acc[a][b], vy = acc[a + 1][b];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[0], vx[1]), pack2bf(vy[0], vy[1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[2], vx[3]), pack2bf(vy[2], vy[3]), false, false);
                    ow[b][a / 2] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
                const f32x4 v = acc[TN - 1][b];
                on[b] = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            }
            if (LN && p.hm_tpc > 0 && etile_n >= p.hm_tpc) {          // tile-uniform: a K or V tile of the q|k|v projection
                // KV[which][sample][head][token][40]: which = K or V, head0 = first of this tile's four heads
                const int which = etile_n >= 2 * p.hm_tpc ? 1 : 0;
                const int head0 = (etile_n - (which + 1) * p.hm_tpc) * 4;
                const int smp = em0 >> p.hm_tok_shift, tok0 = em0 & ((1 << p.hm_tok_shift) - 1);
                const int soC = ((((which * p.hm_samples + smp) * p.hm_heads + head0) << p.hm_tok_shift) + tok0) * 80;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int so = soC + b * 16 * 80;
#pragma unroll
                    for (int j = 0; j < TN / 2; ++j) __builtin_amdgcn_raw_buffer_store_b128(ow[b][j], rKV, vH[j] + so, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(on[b], rKV, vH[TN / 2] + so, 0, 0);
                }
            } else {
                // STORE_OFFSET_RULE: the scalar part of a store's address is ADDED INTO THE VECTOR OFFSET (one v_add per row
                // block), never passed as the instruction's SGPR soffset.  hipcc's hazard recognizer assumes that a 16-byte
                // buffer store WITH a register soffset needs no wait state before a VALU instruction overwrites its data
                // registers (GCNHazardRecognizer::createsVALUHazard); on gfx950 it does: the GEGLU epilogue of this file,
                // where the next pair's v_med3 lands in the store's data registers one instruction later, stored the NEW
                // value in lanes 12-15 of every row (round 4, found by the bit-identity test against gemm_kernel).
                // Without soffset the compiler inserts the wait state itself.
                const int soC = em0 * p.ldcb + en0 * 2;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int so = soC + b * 16 * p.ldcb;
#pragma unroll
                    for (int j = 0; j < TN / 2; ++j) __builtin_amdgcn_raw_buffer_store_b128(ow[b][j], rC, vCw + so + j * 64, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(on[b], rC, vCn + so + (TN - 1) * 32, 0, 0);
                }
            }
        }
        if (!more) break;
    }
}

// zeros for an absent bias vector (the kernel always fetches two vectors of N floats).  One vector per device, created under
// a lock; the blocking copy back orders the memset (legacy stream) before ANY later launch, also one on a non-blocking
// stream -- the first GEMM of a model can be the fp8 calibration pass, whose scales persist; a failed allocation or
// memset leaves nothing behind, so the next call tries again instead of handing out an unzeroed buffer.
static const float* zero_vector() {
    static std::mutex mu;
    static float* z[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!z[dev]) {
        float* p = nullptr;
        float probe = 1.f;
        if (hipMalloc((void**)&p, 16384 * 4) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 16384 * 4) != hipSuccess || hipMemcpy(&probe, p + 16383, 4, hipMemcpyDeviceToHost) != hipSuccess || probe != 0.f) {
            (void)hipFree(p);
            return nullptr;
        }
        z[dev] = p;
    }
    return z[dev];
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NP, bool GEGLU>
int launch_lean(const LeanArgs& a0, hipStream_t stream) {
    LeanArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.N / BN;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int smem = 2 * (BM + BN) * 128 + NW * 512 + 2 * BN * 4;
    static_assert(smem <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    auto kern = gemm_lean_kernel<BM, BN, WAVES_M, WAVES_N, NP, GEGLU>;
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    int grid = a.tiles_m * a.tiles_n;
    constexpr int resident = smem > 80 * 1024 ? 256 : 512;      // workgroups that fit the chip at once (1 or 2 per CU)
    if (grid > resident) grid = resident;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Shapes / features the lean kernels take (everything else runs on gemm_conv.hip's gemm_kernel).  epi 0 = std, 1 = GEGLU.
bool sd_gemm_lean_applicable(const GemmArgs& a, int epi) {
    if ((epi != 0 && epi != 1) || a.dt != 0 || a.out_fp8 || a.rows_per_batch || a.subpix || (a.splitk > 1 && a.slab)) return false;
    const int BN = epi ? 256 : 160;
    if (a.N % BN != 0 || a.N > 16384 || a.K % 64 != 0 || a.K < 128 || a.K1 % 64 != 0 || a.M < 1) return false;
    if (a.K1 != a.K && a.X2 == nullptr) return false;
    if ((a.ldc & 7) || (a.ldx & 7) || (a.X2 && (a.ldx2 & 7)) || (a.ldw & 7) || (a.R && (a.ldr & 3))) return false;
    const long big = 1L << 31;
    const long ldw = a.ldw ? a.ldw : a.K;
    if ((long)a.M * a.ldx * 2 >= big || (long)a.M * a.ldc * 2 >= big || (long)a.N * ldw * 2 >= big) return false;
    if (a.X2 && (long)a.M * a.ldx2 * 2 >= big) return false;
    if (a.R && (long)a.M * a.ldr * 2 >= big) return false;
    // producer / consumer side tensors are indexed by row: whole tiles only
    if ((a.stats || a.rowstats || a.ln_rs || a.hm_C) && a.M % (epi ? 256 : 128) != 0) return false;
    if (a.ln_rs) {      // the LN = true instantiations: one K segment, c1 / c2, 4 / 8 / 16 partials, no producer side
        if (a.R || a.bias2 || !a.bias || !a.ln_c1 || a.X2 || a.K1 != a.K || a.stats || a.rowstats) return false;
        if (a.ln_np != 2 && a.ln_np != 4 && a.ln_np != 8 && a.ln_np != 16) return false;
    }
    if (epi == 1 && (a.R || a.bias2 || a.X2 || a.K1 != a.K || a.stats || a.rowstats || a.hm_C)) return false;
    if (a.hm_C) {       // head-major K / V: on the q|k|v projection, which is a LayerNorm-fold consumer in the UNet plan
        if (!a.ln_rs || !a.KV || a.hm_C % 160 != 0 || a.N != 3 * a.hm_C || a.hm_tok < 128 || (a.hm_tok & (a.hm_tok - 1)) ||
            a.M % a.hm_tok != 0 || (long)a.M * a.hm_C * 4 >= big)
            return false;
    }
    return true;
}

int sd_launch_gemm_lean(const GemmArgs& g, int epi, int rows, hipStream_t stream) {
    SD_REQUIRE(sd_gemm_lean_applicable(g, epi), "gemm (lean): problem not supported (M=%d N=%d K=%d epi=%d)", g.M, g.N, g.K, epi);
    SD_REQUIRE(epi == 1 || rows == 128 || (rows == 64 && !g.stats && !g.ln_rs && !g.hm_C),
               "gemm (lean): %d-row tile with block statistics / LayerNorm fold / head-major K|V", rows);
    const long ldw = g.ldw ? g.ldw : g.K;
    LeanArgs a{};
    a.X = g.X; a.X2 = g.X2; a.W = g.W; a.R = g.R; a.C = g.C; a.KV = g.KV;
    a.M = g.M; a.N = g.N; a.K = g.K; a.K1 = g.K1;
    a.ldxb = (int)g.ldx * 2; a.ldx2b = (int)g.ldx2 * 2; a.ldwb = (int)ldw * 2; a.ldrb = (int)g.ldr * 2; a.ldcb = (int)g.ldc * 2;
    // resource ranges: rows beyond M (a partial last M tile) read zeros and their stores are dropped
    a.x_bytes = (unsigned)(((long)g.M - 1) * g.ldx * 2 + (long)g.K1 * 2);
    a.x2_bytes = g.X2 ? (unsigned)(((long)g.M - 1) * g.ldx2 * 2 + (long)(g.K - g.K1) * 2) : 0;
    a.w_bytes = (unsigned)(((long)g.N - 1) * ldw * 2 + (long)g.K * 2);
    a.r_bytes = g.R ? (unsigned)(((long)g.M - 1) * g.ldr * 2 + (long)g.N * 2) : 0;
    const int ncols = epi ? g.N / 2 : g.hm_C ? g.hm_C : g.N;          // columns that go to C
    a.c_bytes = (unsigned)(((long)g.M - 1) * g.ldc * 2 + (long)ncols * 2);
    a.kv_bytes = g.hm_C ? (unsigned)((long)g.M * g.hm_C * 4) : 0;
    const float* zeros = nullptr;
    if (!g.bias || (!g.ln_rs && !g.bias2)) {
        zeros = zero_vector();
        SD_REQUIRE(zeros, "gemm (lean): cannot allocate the zero vector");
    }
    if (g.ln_rs) {
        a.vec0 = g.ln_c1; a.vec1 = g.bias; a.ln_rs = g.ln_rs; a.ln_eps = g.ln_eps; a.inv_k = 1.0f / (float)g.K;
    } else {
        a.vec0 = g.bias ? g.bias : zeros; a.vec1 = g.bias2 ? g.bias2 : zeros;
    }
    a.rowstats = g.rowstats; a.stats = g.stats;
    if (g.hm_C) {
        a.hm_tpc = g.hm_C / 160;
        int sh = 0;
        while ((1 << sh) < g.hm_tok) ++sh;
        a.hm_tok_shift = sh; a.hm_heads = g.hm_C / 40; a.hm_samples = g.M / g.hm_tok;
    }
    if (epi == 1) {
        // (round 5: 128 x 128 tiles on 4 waves, two independent workgroups per CU -- so that one's GELU epilogue overlaps the other's
        // K loop -- measured 8 % SLOWER at every FF shape, bit-identical: 133 -> 144 us at 64x64, 117 -> 128 at 32x32, 117 -> 125 at
        // 16x16, tools/geglu_tile_ab.py; the doubled LDS-fill bytes per flop cost more than the overlap returns)
        if (!g.ln_rs) return launch_lean<256, 256, 4, 2, 0, true>(a, stream);
        return g.ln_np == 2 ? launch_lean<256, 256, 4, 2, 2, true>(a, stream)      // (2 partials: the fused cross-attention's, one slice)
             : g.ln_np == 4 ? launch_lean<256, 256, 4, 2, 4, true>(a, stream)
             : g.ln_np == 8 ? launch_lean<256, 256, 4, 2, 8, true>(a, stream) : launch_lean<256, 256, 4, 2, 16, true>(a, stream);
    }
    if (g.ln_rs)
        return g.ln_np == 2 ? launch_lean<128, 160, 2, 2, 2, false>(a, stream)
             : g.ln_np == 4 ? launch_lean<128, 160, 2, 2, 4, false>(a, stream)
             : g.ln_np == 8 ? launch_lean<128, 160, 2, 2, 8, false>(a, stream) : launch_lean<128, 160, 2, 2, 16, false>(a, stream);
    return rows == 64 ? launch_lean<64, 160, 2, 2, 0, false>(a, stream) : launch_lean<128, 160, 2, 2, 0, false>(a, stream);
}
