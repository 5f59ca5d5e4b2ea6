// Flash-style attention (self and cross) for gfx950, head dims 40 / 80 / 160 (SD-1.5: 8 heads
// at C = 320 / 640 / 1280), arbitrary key count (4096 ... 64 for self-attention, 77 for the
// prompt cross-attention), bf16 in / fp32 accumulate / bf16 out.
//
// Layout: Q, K, V, O are token-major [B, N, heads*D] views (row strides ldq/ldk/ldv/ldo), so
// the fused QKV projection output is consumed in place.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries.
//   S^T = K . Q^T   (v_mfma_f32_32x32x16_bf16, K rows as A operand, Q held in registers)
// puts the query on the lane and 16 keys in the accumulator registers, so the online softmax
// is lane-local (one cross-half exchange for the max) and the bf16-packed probabilities are
// directly the B operand of
//   O^T += V^T . P^T
// with no LDS round trip.  K rows are read in the order pi(r) = r with bits 2,3 swapped so the
// accumulator-register -> k mapping of the second product meets V in natural key order; V^T
// fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose).
//
// The kernel is VALU-bound at d = 40 (exp + max + convert per score), so the design removes
// vector instructions rather than matrix ones:
//   * the softmax denominator is not summed on the VALU: the V tile carries a column of ones in
//     its padding (d = 40 -> 64, 80 -> 96), so row D of O^T accumulates sum(P) on the matrix core;
//   * 64 keys (two S^T tiles) share one max / rescale decision;
//   * MFMA results live in arch VGPRs (built with -amdgpu-mfma-vgpr-form: no accvgpr copies);
//   * per-thread staging offsets are loop invariant; K/V tiles are register-prefetched one tile
//     ahead and double-buffered in LDS (one barrier per 64-key tile).
// LDS rows are padded to an odd number of 16-byte slots (conflict-free ds_read_b128 K fragments).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

// Maximum over the two lanes of a query (lane and lane ^ 32 hold different keys).  The two results of the swap are copied
// to scalars BEFORE the bit cast: `__builtin_bit_cast(float, sw[1])` on the element of an ext-vector reads element 0 with
// this hipcc (ROCm 7.2; `sw[0] + 2 * sw[1]` compiles to v_fmac v1, 2.0, v1), so that -- rounds 1 and 2 -- max(sw[0], sw[1])
// was sw[0] and the running maximum the maximum over the h = 0 half's keys only.  That is still a valid softmax reference
// (both halves used the same one), but one the other half's scores can exceed by more than the fp32 exponent range: rows
// whose largest score sat 128 log2 units above it came out inf / NaN (test_attention_pipelined_kernel_moves_its_stale_reference).
__device__ __forceinline__ float half_pair_max(float x) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned r0 = sw[0], r1 = sw[1];
    return fmaxf(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}

__device__ __forceinline__ int pi_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// XCD-aware work order of the pipelined kernels (round 5).  They are launched on a 1-D grid of nq x heads x B workgroups;
// the hardware deals consecutive workgroups round-robin over the 8 XCDs (MI355X_MICROARCH.md: speed only, never
// correctness), so with (query block, head, sample) taken straight from a 3-D blockIdx the query blocks of ONE (sample, head)
// land on ALL eight XCDs and every XCD's L2 streams every head's K / V from beyond it: rocprofv3 counted 680 MB per launch
// of the 64x64 self-attention = 8 x its 84 MB of K / V (profiles/round5_conv_traffic.json).  Here XCD x takes a CONTIGUOUS
// run of virtual work items, query block fastest: all query blocks of a (sample, head) share one L2 and its K / V is fetched once.
__device__ __forceinline__ void xcd_work_item(int nq, int heads, int xcd_order, int& qblk, int& head, int& b) {
    const int G = gridDim.x, L = blockIdx.x;
    const int q = G >> 3, r = G & 7, xcd = L & 7, idx = L >> 3;
    const int v = xcd_order ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : L;       // bijective for every G
    qblk = v % nq;
    const int rest = v / nq;
    head = rest % heads;
    b = rest / heads;
}

template <int D>
struct AttnCfg {
    static constexpr int DK = (D + 15) / 16 * 16;          // QK^T contraction length (zero padded)
    static constexpr int KQ = DK / 16;
    static constexpr int DVT = (D + 31) / 32;              // 32-row O^T tiles
    static constexpr int CH = D / 8;                       // 16-B chunks per K/V row
    static constexpr int SLOTS = (DK > DVT * 32 ? DK : DVT * 32) / 8;
    static constexpr int RS = (SLOTS | 1) * 16;            // LDS row stride (odd # of 16-B slots)
    static constexpr int NPF = (64 * CH + 255) / 256;      // chunks each thread stages per tile
    static constexpr bool ONES = DVT * 32 > D;             // room for the ones column
    static constexpr bool DBUF = D <= 80;                  // double-buffered LDS tiles
    static constexpr int TILE_BYTES = 2 * 64 * RS;         // K + V
    static constexpr int SMEM = TILE_BYTES * (DBUF ? 2 : 1);
    // waves per SIMD the register allocator must leave room for (one wave of a block per SIMD)
    static constexpr int MIN_WAVES = D <= 40 ? 4 : (D <= 80 ? 3 : 2);
};

template <int D, bool USE_TR>
__global__ __launch_bounds__(256, AttnCfg<D>::MIN_WAVES) void attn_kernel(const AttnArgs a) {
    using Cfg = AttnCfg<D>;
    constexpr int KQ = Cfg::KQ, DVT = Cfg::DVT, CH = Cfg::CH, RS = Cfg::RS, NPF = Cfg::NPF;
    constexpr bool ONES = Cfg::ONES, DBUF = Cfg::DBUF;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + wave * 32 + r;
    const bool qvalid = q < a.Nq;
    const float c = a.q_prescaled ? 1.0f : a.scale * 1.4426950408889634f;

    // ---- zero LDS once (pad columns stay zero), then plant the ones column in every V row ----
    for (int i = tid; i < Cfg::SMEM / 16; i += 256) *(u32x4*)(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (ONES) {
        for (int i = tid; i < 64 * (DBUF ? 2 : 1); i += 256) {
            const int buf = i >> 6, key = i & 63;
            *(bf16_t*)(smem + buf * Cfg::TILE_BYTES + 64 * RS + key * RS + D * 2) = 0x3F80;  // 1.0
        }
    }

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q][16kk + 8h .. +8) ----
    bf16x8 qf[KQ];
    {
        const bf16_t* qp = a.Q + ((long)b * a.Nq + (qvalid ? q : 0)) * a.ldq + head * D;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int col = kk * 16 + h * 8;
            if (qvalid && col < D) qf[kk] = *(const bf16x8*)(qp + col);
            else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        // Retire the Q loads HERE: otherwise hipcc's wait-count pass carries them as pending into
        // the tile loop and puts s_waitcnt vmcnt(0) in front of the first MFMA of every iteration,
        // which also drains the K/V prefetch issued a few instructions earlier.
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) asm volatile("" : "+v"(qf[kk]));
    }

    // ---- loop-invariant staging offsets ----
    const bf16_t* kbase = a.K + (long)b * a.Nk * a.ldk + head * D;
    const bf16_t* vbase = a.V + (long)b * a.Nk * a.ldv + head * D;
    int st_key[NPF], st_lds[NPF];
    long st_k[NPF], st_v[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int idx = tid + i * 256;
        const int key = idx / CH, ch = idx - key * CH;
        st_key[i] = idx < 64 * CH ? key : 1 << 28;      // out-of-range marker
        st_lds[i] = key * RS + ch * 16;
        st_k[i] = (long)key * a.ldk + ch * 8;
        st_v[i] = (long)key * a.ldv + ch * 8;
    }
    u32x4 kreg[NPF], vreg[NPF];
    auto prefetch = [&](int t) {
        const int k0 = t * 64;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            if (k0 + st_key[i] < a.Nk) {
                kreg[i] = *(const u32x4*)(kbase + (long)k0 * a.ldk + st_k[i]);
                vreg[i] = *(const u32x4*)(vbase + (long)k0 * a.ldv + st_v[i]);
            } else {
                kreg[i] = u32x4{0u, 0u, 0u, 0u};
                vreg[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto commit = [&](char* tile) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            if (st_key[i] < 64) {
                *(u32x4*)(tile + st_lds[i]) = kreg[i];
                *(u32x4*)(tile + 64 * RS + st_lds[i]) = vreg[i];
            }
        }
    };

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int kfrag_off = pi_swap23(r) * RS + h * 16;
    // transposed V read: lane 4q4+p4 of each 16-lane group addresses key row q4, 4 columns at 4*p4
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int vtr_off = 64 * RS + (8 * h + q4) * RS + (16 * g16 + 4 * p4) * 2;

    // one 64-key tile: two S^T tiles share the max / rescale decision
    auto compute = [&](const char* tile, int key0) {
        const bool two = key0 + 32 < a.Nk;          // second 32-key half has at least one key
        f32x16 s0, s1;
        {
            const char* kp = tile + kfrag_off;
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < KQ; ++kk)
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + kk * 32), qf[kk], s0, 0, 0, 0);
            if (two) {
                kp += 32 * RS;
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < KQ; ++kk)
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + kk * 32), qf[kk], s1, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s1[i] = -1e30f;
            }
        }
        if (key0 + 64 > a.Nk) {                      // ragged tail: mask keys >= Nk
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kr = pi_swap23((i & 3) + 8 * (i >> 2) + 4 * h);
                if (key0 + kr >= a.Nk) s0[i] = -1e30f;
                if (key0 + 32 + kr >= a.Nk) s1[i] = -1e30f;
            }
        }
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, fmaxf(s0[i], s1[i]));
        {   // max across the two half-waves (the other 16+16 keys of the same query)
            tmax = half_pair_max(tmax);
        }
        const float m_new = fmaxf(m_run, tmax);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            if (!ONES) l_run *= alpha;
#pragma unroll
            for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
        bf16x8 pf[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sv = s2 < 2 ? s0[8 * s2 + j] : s1[8 * (s2 - 2) + j];
                p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -mc));
                if (!ONES) l_run += p[j];
            }
            u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
            pf[s2] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                if (s2 >= 2 && !two) break;
                bf16x8 vf;
                if (USE_TR) {
                    const char* ap = tile + s2 * 16 * RS + vtr_off + tt * 64;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(ap));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) bf16x4*)(ap + 4 * RS));
                    vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        vf[j] = *(const short*)(tile + 64 * RS + (s2 * 16 + 8 * h + j) * RS + (32 * tt + r) * 2);
                }
                o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o[tt], 0, 0, 0);
            }
        }
    };

    const int ntiles = (a.Nk + 63) / 64;
    prefetch(0);
    __syncthreads();  // LDS zero fill + ones column done
    if (DBUF) {
        commit(smem);
        if (ntiles > 1) prefetch(1);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            char* cur = smem + (t & 1) * Cfg::TILE_BYTES;
            if (t + 1 < ntiles) {
                commit(smem + ((t + 1) & 1) * Cfg::TILE_BYTES);   // other buffer: last read in iteration t-1
                if (t + 2 < ntiles) prefetch(t + 2);
            }
            compute(cur, t * 64);
            __syncthreads();
        }
    } else {
        for (int t = 0; t < ntiles; ++t) {
            commit(smem);
            __syncthreads();
            if (t + 1 < ntiles) prefetch(t + 1);
            compute(smem, t * 64);
            __syncthreads();  // everyone done with this tile before it is overwritten
        }
    }

    float l_tot;
    if (ONES) {
        // row D of O^T holds sum(P): tile D/32, row D%32 = (reg&3) + 8*(reg>>2) + 4*h  ->  h = 0 half
        constexpr int TT = D / 32, RR = D % 32;
        static_assert(!ONES || ((RR & 4) == 0), "ones row must sit in the h = 0 half");
        constexpr int REG = (RR & 3) + 4 * (RR >> 3);
        l_tot = __shfl(o[TT][REG], r);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    if (qvalid) {
        bf16_t* op = a.O + ((long)b * a.Nq + q) * a.ldo + head * D;
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = 32 * tt + 8 * g4 + 4 * h;
                if (dv < D) {
                    u32x2 w = {pack2bf(o[tt][4 * g4 + 0] * inv, o[tt][4 * g4 + 1] * inv),
                               pack2bf(o[tt][4 * g4 + 2] * inv, o[tt][4 * g4 + 3] * inv)};
                    *(u32x2*)(op + dv) = w;
                }
            }
        }
    }
}


// =====================================================================================================
// LDS-DMA variant (D = 40, 80): K/V tiles go global -> LDS with 16-byte global_load_lds, no VGPR round
// trip and no ds_write.  Ablation on MI355X showed the register-staged path spending 27 % of the
// N = 4096 kernel in its staging (waiting for the one-tile-ahead loads, then ds_write_b128 issue);
// the DMA ring is 3 tiles deep with a COUNTED vmcnt, so a tile has two iterations to land.
// Row padding is produced by the per-lane SOURCE address (LDS stays lane-linear): pad chunks read a
// zero page, and V's "ones column" (softmax denominator on the matrix core) reads a constant chunk.
//   D = 40: K rows 80 B (5 chunks; the 3rd k-step over-reads 16 B of the next row, multiplied by the
//           zero padding of Q), V rows 96 B (5 data + ones chunk)
//   D = 80: K rows 176 B (10 data + zero chunk), V rows 192 B (10 data + ones chunk + zero chunk)
// both conflict-free for the ds_read_b128 K fragments; 192 B is also conflict-free for the tr reads.
// =====================================================================================================
template <int D>
struct DmaCfg {
    static constexpr int DK = (D + 15) / 16 * 16, KQ = DK / 16, DVT = (D + 31) / 32, CD = D / 8;
    static constexpr int CHK = D == 40 ? 5 : 11;
    static constexpr int CHV = D == 40 ? 6 : 12;
    static constexpr int RSK = CHK * 16, RSV = CHV * 16;
    static constexpr int KBYTES = 64 * RSK, TILE = 64 * (RSK + RSV);
    static constexpr int NI = CHK + CHV;               // 1-KiB DMA wave-instructions per tile
    static constexpr int NW = 8;                       // waves per workgroup (32 queries each)
    static constexpr int NIW = (NI + NW - 1) / NW;     // per wave (waves < NI % NW issue NIW, else NIW - 1)
    static constexpr int NSLOT = 3;
    static constexpr int SMEM = TILE * NSLOT + 256;    // tail pad: the last V row's tr reads run past its 96/192 B
    static_assert(D == 40 || D == 80, "LDS-DMA attention is built for head dims 40 and 80");
};

template <int D>
__global__ __launch_bounds__(DmaCfg<D>::NW * 64, D == 40 ? 4 : 3) void attn_dma_kernel(const AttnArgs a) {
    using Cfg = DmaCfg<D>;
    constexpr int KQ = Cfg::KQ, DVT = Cfg::DVT, CD = Cfg::CD, CHK = Cfg::CHK, CHV = Cfg::CHV;
    constexpr int RSK = Cfg::RSK, RSV = Cfg::RSV, KBYTES = Cfg::KBYTES, TILE = Cfg::TILE, NI = Cfg::NI, NIW = Cfg::NIW;
    constexpr int NW = Cfg::NW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * (NW * 32) + wave * 32 + r;
    const bool qvalid = q < a.Nq;
    const float c = a.q_prescaled ? 1.0f : a.scale * 1.4426950408889634f;
    const char* zero = (const char*)a.consts;
    const char* ones = zero + 256;

    bf16x8 qf[KQ];
    {
        const bf16_t* qp = a.Q + ((long)b * a.Nq + (qvalid ? q : 0)) * a.ldq + head * D;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int col = kk * 16 + h * 8;
            if (qvalid && col < D) qf[kk] = *(const bf16x8*)(qp + col);
            else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) asm volatile("" : "+v"(qf[kk]));   // retire the loads before the DMA ring starts
    }

    // ---- loop-invariant DMA descriptors: this wave issues instructions wave, wave+NW, ... < NI ----
    const bf16_t* kbase = a.K + (long)b * a.Nk * a.ldk + head * D;
    const bf16_t* vbase = a.V + (long)b * a.Nk * a.ldv + head * D;
    int d_key[NIW], d_kind[NIW];        // kind: 0 K data, 1 V data, 2 zero pad, 3 ones chunk
    long d_src[NIW];                    // element offset of the chunk inside a 64-key tile (data kinds)
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int inst = wave + NW * j;
        d_key[j] = 0; d_kind[j] = 2; d_src[j] = 0;
        if (inst < CHK) {
            const int ch = inst * 64 + lane, key = ch / CHK, part = ch - key * CHK;
            d_key[j] = key;
            d_kind[j] = part < CD ? 0 : 2;
            d_src[j] = (long)key * a.ldk + part * 8;
        } else if (inst < NI) {
            const int ch = (inst - CHK) * 64 + lane, key = ch / CHV, part = ch - key * CHV;
            d_key[j] = key;
            d_kind[j] = part < CD ? 1 : (part == CD ? 3 : 2);
            d_src[j] = (long)key * a.ldv + part * 8;
        }
    }
    auto issue = [&](int t, int slot) {
        char* base = smem + slot * TILE;
        const int k0 = t * 64;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int inst = wave + NW * j;
            if (inst < NI) {                                    // wave-uniform
                const bool live = k0 + d_key[j] < a.Nk;
                const void* src = d_kind[j] == 3 ? (const void*)ones
                                : (d_kind[j] == 2 || !live) ? (const void*)zero
                                : d_kind[j] == 0 ? (const void*)(kbase + (long)k0 * a.ldk + d_src[j])
                                                 : (const void*)(vbase + (long)k0 * a.ldv + d_src[j]);
                // K instructions fill [0, KBYTES), V instructions follow; 1 KiB per instruction
                glds16(src, base + inst * 1024);
            }
        }
    };
    static_assert(CHK * 1024 == KBYTES, "K region = CHK DMA instructions");

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -1e30f;

    const int kfrag_off = pi_swap23(r) * RSK + h * 16;
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int vtr_off = KBYTES + (8 * h + q4) * RSV + (16 * g16 + 4 * p4) * 2;

    auto compute = [&](const char* tile, int key0) {
        const bool two = key0 + 32 < a.Nk;
        f32x16 s0, s1;
        {
            const char* kp = tile + kfrag_off;
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < KQ; ++kk)
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + kk * 32), qf[kk], s0, 0, 0, 0);
            if (two) {
                kp += 32 * RSK;
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < KQ; ++kk)
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + kk * 32), qf[kk], s1, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s1[i] = -1e30f;
            }
        }
        if (key0 + 64 > a.Nk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kr = pi_swap23((i & 3) + 8 * (i >> 2) + 4 * h);
                if (key0 + kr >= a.Nk) s0[i] = -1e30f;
                if (key0 + 32 + kr >= a.Nk) s1[i] = -1e30f;
            }
        }
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, fmaxf(s0[i], s1[i]));
        {
            tmax = half_pair_max(tmax);
        }
        const float m_new = fmaxf(m_run, tmax);
        if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
            m_run = m_new;
        }
        const float mc = m_run * c;
        bf16x8 pf[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sv = s2 < 2 ? s0[8 * s2 + j] : s1[8 * (s2 - 2) + j];
                p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -mc));
            }
            u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
            pf[s2] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                if (s2 >= 2 && !two) break;
                const char* ap = tile + s2 * 16 * RSV + vtr_off + tt * 64;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap + 4 * RSV));
                const bf16x8 vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o[tt], 0, 0, 0);
            }
        }
    };

    const int ntiles = (a.Nk + 63) / 64;
    const bool more = (NI % NW == 0) || (wave < NI % NW);   // this wave issues NIW (else NIW - 1) per tile
    issue(0, 0);
    if (ntiles > 1) issue(1, 1);
    int slot = 0;
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) {               // leave tile t+1 in flight
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();       // tile t landed for every wave; tile t-1 fully consumed
        asm volatile("" ::: "memory");
        if (t + 2 < ntiles) issue(t + 2, slot == 0 ? 2 : slot - 1);    // ring slot of tile t-1
        compute(smem + slot * TILE, t * 64);
        slot = slot == 2 ? 0 : slot + 1;
    }

    // row D of O^T holds sum(P) (ones chunk): tile D/32, row D%32 -> register (RR&3) + 4*(RR>>3), h = 0 half
    constexpr int TT = D / 32, RR = D % 32, REG = (RR & 3) + 4 * (RR >> 3);
    static_assert((RR & 4) == 0, "ones row must sit in the h = 0 half");
    const float inv = 1.0f / __shfl(o[TT][REG], r);
    if (qvalid) {
        bf16_t* op = a.O + ((long)b * a.Nq + q) * a.ldo + head * D;
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = 32 * tt + 8 * g4 + 4 * h;
                if (dv < D) {
                    u32x2 w = {pack2bf(o[tt][4 * g4 + 0] * inv, o[tt][4 * g4 + 1] * inv),
                               pack2bf(o[tt][4 * g4 + 2] * inv, o[tt][4 * g4 + 3] * inv)};
                    *(u32x2*)(op + dv) = w;
                }
            }
        }
    }
}

// =====================================================================================================
// Software-pipelined variant for D = 40 and key counts that are a multiple of 64 (the 64x64 self-attention).
// The plain loop above is a serial chain per tile -- QK^T MFMAs, softmax VALU, PV MFMAs -- and on gfx950 the
// matrix and vector pipes of a SIMD barely overlap ACROSS waves (measured: their busy times add up to the
// kernel time), but they do overlap when one wave's instruction stream interleaves them (probe: valu 750 +
// mfma 455 -> 900 cycles).  So each iteration here carries two independent strands:
//     phase A:  exp2/pack of S(t)                 ||  S(t+1) = K(t+1) . Q^T      (6 MFMAs)
//     phase B:  max over S(t+1), new running max  ||  O^T += V(t)^T . P(t)^T     (8 MFMAs)
// and the rare O rescale is applied after PV(t).  Ring of 4 LDS tiles: tiles t (V) and t+1 (K) are in use,
// t+2 and t+3 are in flight.  4 waves per workgroup (one per SIMD), 128 queries.
// =====================================================================================================
struct PipeCfg {
    static constexpr int D = 40, KQ = 3, DVT = 2, CD = 5, CHK = 5, CHV = 6, RSK = 80, RSV = 96;
    static constexpr int KBYTES = 64 * RSK, NI = CHK + CHV, NW = 4;
    static constexpr int NIW = 3;                      // DMA pieces per wave and tile: 11 real + 1 padding piece
    static constexpr int TILE = NIW * NW * 1024;       // 12 KiB slot: K rows, V rows, 1 KiB the padding piece lands in
    static constexpr int NSLOT = 4, SMEM = TILE * NSLOT;
    static_assert(TILE >= 64 * (RSK + RSV) + 256, "slot holds the tile + the tr over-read of the last V row");
};

// V (round 3; 0 = the round-2 kernel, kept for A/B through SD_ATTN_VARIANT):
//   bit 0  V^T fragments by inline-asm ds_read_b64_tr_b16 with hand-counted lgkmcnt waits.  hipcc cannot tell the builtin's
//          LDS read from the LDS-DMA writes in flight and put an `s_waitcnt vmcnt(0)` in front of the first tr read of EVERY
//          tile: the three-deep DMA ring was drained once per iteration.
//   bit 1  the softmax reference is part of the matrix product: Q is scaled by scale * log2(e) once, and the padding column
//          d = 40 of the QK^T contraction carries -M (M = the reference, per query, a bf16 value) against a column of ones
//          (the h = 1 lanes of the third k-step read a {1, 0, ..} chunk instead of the over-read of the next K row), so
//          the probabilities are exp2 of the MFMA result with no multiply-add per score, no register beyond one LDS
//          address; and M is STALE: it moves only when a tile's maximum exceeds it by more than 2^8 (probabilities up to
//          256 are exact in bf16 / fp32; the final division by the matrix-core row sum cancels M), so the O rescale all
//          but disappears after the first tile.
//   bit 2  (with bit 0) two V^T fragment register sets in turn: a group's reads are issued a whole chunk before their wait.
#define SD_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
template <int V>
__global__ __launch_bounds__(PipeCfg::NW * 64, 3) void attn_pipe40_kernel(const AttnArgs a) {
    using Cfg = PipeCfg;
    constexpr int D = Cfg::D, KQ = Cfg::KQ, DVT = Cfg::DVT, CD = Cfg::CD, CHK = Cfg::CHK, CHV = Cfg::CHV;
    constexpr int RSK = Cfg::RSK, RSV = Cfg::RSV, KBYTES = Cfg::KBYTES, TILE = Cfg::TILE, NI = Cfg::NI, NIW = Cfg::NIW;
    constexpr int NW = Cfg::NW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qblk, head, b;
    xcd_work_item((a.Nq + NW * 32 - 1) / (NW * 32), a.heads, a.xcd_order, qblk, head, b);
    const int q = qblk * (NW * 32) + wave * 32 + r;
    const bool qvalid = q < a.Nq;
    const float c = a.q_prescaled ? 1.0f : a.scale * 1.4426950408889634f;
    const char* zero = (const char*)a.consts;
    const char* ones = zero + 256;

    bf16x8 qf[KQ];
    {
        const bf16_t* qp = a.Q + ((long)b * a.Nq + (qvalid ? q : 0)) * a.ldq + head * D;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int col = kk * 16 + h * 8;
            if (qvalid && col < D) qf[kk] = *(const bf16x8*)(qp + col);
            else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if ((V & 2) && !a.q_prescaled) {           // Q <- bf16(Q * scale * log2 e): S^T comes out of the MFMA in exp2 units
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = pack2bf(bf2f((bf16_t)qf[kk][2 * j]) * c, bf2f((bf16_t)qf[kk][2 * j + 1]) * c);
                qf[kk] = __builtin_bit_cast(bf16x8, w);
            }
        }
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) asm volatile("" : "+v"(qf[kk]));   // retire the loads before the DMA ring starts
    }

    // Branch-free LDS-DMA descriptors: per piece a source pointer for tile 0 and a byte step per tile (0 for
    // the zero / ones chunks and the padding piece).  Every wave issues exactly NIW pieces per tile.
    const char* d_ptr[NIW];
    long d_step[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int inst = wave + NW * j;
        d_ptr[j] = zero; d_step[j] = 0;
        if (inst < CHK) {
            const int ch = inst * 64 + lane, key = ch / CHK, part = ch - key * CHK;
            if (part < CD) {
                const long ld = a.kv_head_major ? D : a.ldk;
                const long base = a.kv_head_major ? ((long)b * a.heads + head) * a.Nk * D : (long)b * a.Nk * a.ldk + head * D;
                d_ptr[j] = (const char*)(a.K + base + (long)key * ld + part * 8);
                d_step[j] = 64 * ld * 2;
            }
        } else if (inst < NI) {
            const int ch = (inst - CHK) * 64 + lane, key = ch / CHV, part = ch - key * CHV;
            if (part < CD) {
                const long ld = a.kv_head_major ? D : a.ldv;
                const long base = a.kv_head_major ? ((long)b * a.heads + head) * a.Nk * D : (long)b * a.Nk * a.ldv + head * D;
                d_ptr[j] = (const char*)(a.V + base + (long)key * ld + part * 8);
                d_step[j] = 64 * ld * 2;
            } else if (part == CD) {
                d_ptr[j] = ones;
            }
        } else if (V & 2) {
            d_ptr[j] = ones;                       // the padding piece: 64 {1, 0, ..} chunks (see kfrag2_off)
        }
    }
    const int ntiles = a.Nk / 64;                 // >= 4, all full (checked by the launcher)
    auto issue = [&](int t) {                     // tiles past the end fetch the zero page (their slot is free)
        char* base = smem + (t & 3) * TILE;
        const bool real = t < ntiles;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const char* src = real ? d_ptr[j] + d_step[j] * t : zero;
            glds16(src, base + (wave + NW * j) * 1024);
        }
    };
    auto wait_keep = [&](int tiles_in_flight) {   // the newest 0 / 1 / 2 tiles may still be pending
        if (tiles_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");
        else if (tiles_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;

    const int kfrag_off = pi_swap23(r) * RSK + h * 16;
    // V & 2: third k-step (columns 32..47).  h = 0: the row's columns 32..39 as before.  h = 1 (columns 40..47, Q is zero
    // there except -M in column 40): a {1, 0, ..} chunk -- the ones chunk of V row 37 for keys 0..31, and 32 K rows
    // (2560 B) further on a chunk of the padding piece for keys 32..63 (both immediates are shared with the h = 0 lanes)
    static_assert(KBYTES + 37 * RSV + 2 * CD * 8 + 32 * RSK >= 11 * 1024 && KBYTES + 37 * RSV + 2 * CD * 8 + 32 * RSK + 16 <= 12 * 1024,
                  "second ones chunk inside the padding piece");
    const int kfrag2_off = h ? KBYTES + 37 * RSV + 2 * CD * 8 - 64 : kfrag_off;
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int vtr_off = KBYTES + (8 * h + q4) * RSV + (16 * g16 + 4 * p4) * 2;

    auto qk = [&](const char* tile, f32x16& s0, f32x16& s1) {
        const char* kp = tile + kfrag_off;
        bf16x8 kf[2 * KQ];
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const char* kpp = ((V & 2) && kk == KQ - 1) ? tile + kfrag2_off : kp;
            kf[kk] = *(const bf16x8*)(kpp + kk * 32);
            kf[KQ + kk] = *(const bf16x8*)(kpp + 32 * RSK + kk * 32);
        }
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], f32x16{}, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ], qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
        for (int kk = 1; kk < KQ; ++kk) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + kk], qf[kk], s1, 0, 0, 0);
        }
    };
    auto tile_max = [&](const f32x16& s0, const f32x16& s1) -> float {
        float tmax = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, fmaxf(s0[i], s1[i]));
        return half_pair_max(tmax);
    };
    auto softmax_p = [&](const f32x16& s0, const f32x16& s1, float mc, bf16x8* pf) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sv = s2 < 2 ? s0[8 * s2 + j] : s1[8 * (s2 - 2) + j];
                p[j] = (V & 2) ? __builtin_amdgcn_exp2f(sv) : __builtin_amdgcn_exp2f(__builtin_fmaf(sv, c, -mc));
            }
            u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
            pf[s2] = __builtin_bit_cast(bf16x8, w);
        }
    };
    auto pv = [&](const char* tile, const bf16x8* pf) {
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const char* ap = tile + s2 * 16 * RSV + vtr_off + tt * 64;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap + 4 * RSV));
                const bf16x8 vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o[tt], 0, 0, 0);
            }
        }
    };

    issue(0);
    issue(1);
    issue(2);
    wait_keep(2);                                 // tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16 sa0, sa1, sb0, sb1;                    // S(t) / S(t+1), roles swap every iteration
    qk(smem, sa0, sa1);
    float m_run = tile_max(sa0, sa1);
    // V & 2: M = bf16(max of tile 0); -M into column 40 of Q (h = 1 lanes, first element of the third fragment)
    auto set_ref = [&](float m) -> float {
        const unsigned pb = pack2bf(-m, 0.f) & 0xffffu;
        if (h) qf[KQ - 1][0] = (short)pb;
        return -bf2f((bf16_t)pb);
    };
    if (V & 2) {
        m_run = set_ref(m_run);
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa0[i] -= m_run; sa1[i] -= m_run; }
    }
    constexpr float STALE = 8.f;                  // log2 of the largest probability before M is moved
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)vtr_off;

    // One steady-state iteration (t <= ntiles - 2): consumes S(t) in (c0, c1), produces S(t+1) in (n0, n1).
    // Hand-interleaved in seven fenced chunks so that every chunk carries matrix AND vector work of one wave:
    //   chunks 0-2: two QK^T MFMAs of tile t+1  +  exp2/pack of 8 scores of tile t   (P group 0, 1, 2)
    //   chunk  3  : two PV MFMAs with P group 0 +  exp2/pack of the last 8 scores    (P group 3)
    //   chunks 4-6: two PV MFMAs with P group 1, 2, 3  +  the running max over S(t+1)
    auto exp_group = [&](const f32x16& c0, const f32x16& c1, int s2, float mc) -> bf16x8 {
        float p[8];
        if (V & 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_exp2f(s2 < 2 ? c0[8 * s2 + j] : c1[8 * (s2 - 2) + j]);
            u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
            return __builtin_bit_cast(bf16x8, w);
        }
        const f32x2_t cc = {c, c}, mm = {-mc, -mc};
#pragma unroll
        for (int j = 0; j < 8; j += 2) {          // s*c - m*c two scores at a time: v_pk_fma_f32
            const f32x2_t sv = s2 < 2 ? f32x2_t{c0[8 * s2 + j], c0[8 * s2 + j + 1]}
                                      : f32x2_t{c1[8 * (s2 - 2) + j], c1[8 * (s2 - 2) + j + 1]};
            const f32x2_t e = __builtin_elementwise_fma(sv, cc, mm);
            p[j] = __builtin_amdgcn_exp2f(e[0]);
            p[j + 1] = __builtin_amdgcn_exp2f(e[1]);
        }
        u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
        return __builtin_bit_cast(bf16x8, w);
    };
    auto vfrag = [&](const char* tile, int tt, int s2) -> bf16x8 {
        const char* ap = tile + s2 * 16 * RSV + vtr_off + tt * 64;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(ap + 4 * RSV));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto steady = [&](int t, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        wait_keep(1);                             // tiles t and t+1 landed (t+2 may be in flight)
        __builtin_amdgcn_s_barrier();             // ... and every wave is past PV(t-1): slot (t+3)&3 is free
        asm volatile("" ::: "memory");
        issue(t + 3);
        const char* cur = smem + (t & 3) * TILE;
        const char* kp = smem + ((t + 1) & 3) * TILE + kfrag_off;
        const char* kp2 = (V & 2) ? smem + ((t + 1) & 3) * TILE + kfrag2_off : kp;
        const float mc = m_run * c;
        bf16x8 kf[2 * KQ];
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const char* kpp = kk == KQ - 1 ? kp2 : kp;
            kf[kk] = *(const bf16x8*)(kpp + kk * 32);
            kf[KQ + kk] = *(const bf16x8*)(kpp + 32 * RSK + kk * 32);
        }
        bf16x8 pf[4], va, vb;
        // V & 1: V^T fragments of P group G by inline asm into register set S = x / y (4 reads: tile 0 lo, hi, tile 1 lo, hi);
        // READY_A / READY_B: the tile-0 / tile-1 fragment has landed when at most N LDS operations issued after it are
        // outstanding.  V & 4: two sets in turn, group G + 1 is requested BEFORE the wait for group G (else the four
        // waits per tile expose the LDS latency to this wave); without bit 2 one set, requested after the previous MFMAs.
        const unsigned vaddr = lds_v + (unsigned)((t & 3) * TILE);
        bf16x4 xal, xah, xbl, xbh, yal, yah, ybl, ybh;
#define SD_VREAD(S, G)                                                                                     \
        SD_TR_READ(S##al, vaddr, (G) * 16 * RSV); SD_TR_READ(S##ah, vaddr, (G) * 16 * RSV + 4 * RSV);      \
        SD_TR_READ(S##bl, vaddr, (G) * 16 * RSV + 64); SD_TR_READ(S##bh, vaddr, (G) * 16 * RSV + 64 + 4 * RSV);
#define SD_VREADY_A(S, N)                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(S##al), "+v"(S##ah) : "n"(N));                        \
        va = bf16x8{S##al[0], S##al[1], S##al[2], S##al[3], S##ah[0], S##ah[1], S##ah[2], S##ah[3]};
#define SD_VREADY_B(S, N)                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(S##bl), "+v"(S##bh), "+v"(o[0]) : "n"(N));            \
        vb = bf16x8{S##bl[0], S##bl[1], S##bl[2], S##bl[3], S##bh[0], S##bh[1], S##bh[2], S##bh[3]};
        constexpr bool ASM = V & 1, PING = (V & 5) == 5;
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], f32x16{}, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ], qf[0], f32x16{}, 0, 0, 0);
        pf[0] = exp_group(c0, c1, 0, mc);
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 1], qf[1], n1, 0, 0, 0);
        pf[1] = exp_group(c0, c1, 1, mc);
        __builtin_amdgcn_sched_barrier(0);
        if (PING) { SD_VREAD(x, 0) __builtin_amdgcn_sched_barrier(0); }
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 2], qf[2], n1, 0, 0, 0);
        pf[2] = exp_group(c0, c1, 2, mc);
        if (!ASM) { va = vfrag(cur, 0, 0); vb = vfrag(cur, 1, 0); }
        else if (!PING) { SD_VREAD(x, 0) }
        __builtin_amdgcn_sched_barrier(0);
        if (PING) { SD_VREAD(y, 1) SD_VREADY_A(x, 6) }
        else if (ASM) { SD_VREADY_A(x, 2) }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[0], o[0], 0, 0, 0);
        if (PING) { SD_VREADY_B(x, 4) }
        else if (ASM) { SD_VREADY_B(x, 0) }
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf[0], o[1], 0, 0, 0);
        pf[3] = exp_group(c0, c1, 3, mc);
        if (!ASM) { va = vfrag(cur, 0, 1); vb = vfrag(cur, 1, 1); }
        else if (!PING) { SD_VREAD(x, 1) }
        __builtin_amdgcn_sched_barrier(0);
        if (PING) { SD_VREAD(x, 2) SD_VREADY_A(y, 6) }
        else if (ASM) { SD_VREADY_A(x, 2) }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[1], o[0], 0, 0, 0);
        if (PING) { SD_VREADY_B(y, 4) }
        else if (ASM) { SD_VREADY_B(x, 0) }
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf[1], o[1], 0, 0, 0);
        float tm0 = fmaxf(n0[0], n1[0]);
#pragma unroll
        for (int i = 1; i < 8; ++i) tm0 = (V & 2) ? fmaxf(fmaxf(tm0, n0[i]), n1[i]) : fmaxf(tm0, fmaxf(n0[i], n1[i]));
        if (!ASM) { va = vfrag(cur, 0, 2); vb = vfrag(cur, 1, 2); }
        else if (!PING) { SD_VREAD(x, 2) }
        __builtin_amdgcn_sched_barrier(0);
        if (PING) { SD_VREAD(y, 3) SD_VREADY_A(x, 6) }
        else if (ASM) { SD_VREADY_A(x, 2) }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[2], o[0], 0, 0, 0);
        if (PING) { SD_VREADY_B(x, 4) }
        else if (ASM) { SD_VREADY_B(x, 0) }
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf[2], o[1], 0, 0, 0);
#pragma unroll
        for (int i = 8; i < 16; ++i) tm0 = (V & 2) ? fmaxf(fmaxf(tm0, n0[i]), n1[i]) : fmaxf(tm0, fmaxf(n0[i], n1[i]));
        if (!ASM) { va = vfrag(cur, 0, 3); vb = vfrag(cur, 1, 3); }
        else if (!PING) { SD_VREAD(x, 3) }
        __builtin_amdgcn_sched_barrier(0);
        if (PING) { SD_VREADY_A(y, 2) }
        else if (ASM) { SD_VREADY_A(x, 2) }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[3], o[0], 0, 0, 0);
        if (PING) { SD_VREADY_B(y, 0) }
        else if (ASM) { SD_VREADY_B(x, 0) }
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf[3], o[1], 0, 0, 0);
        const float tmx = half_pair_max(tm0);
        const float m_new = fmaxf(m_run, tmx);
        __builtin_amdgcn_sched_barrier(0);
        if (V & 2) {
            // tmx is relative to M (the MFMAs started at -M).  Rare: some query's tile maximum is more than 2^STALE above
            // its reference -> every lane moves M up to its tile maximum (if that is above M at all)
            if (!__all(tmx <= STALE)) {
                const float m_old = m_run;
                m_run = set_ref(m_run + fmaxf(tmx, 0.f));
                const float d = m_run - m_old;
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { n0[i] -= d; n1[i] -= d; }
            }
        } else if (!__all(m_new == m_run)) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
            m_run = m_new;
        }
    };
    auto final_tile = [&](int t, f32x16& c0, f32x16& c1) {
        wait_keep(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 pf[4];
        softmax_p(c0, c1, m_run * c, pf);
        pv(smem + (t & 3) * TILE, pf);
    };
#undef SD_VREAD
#undef SD_VREADY_A
#undef SD_VREADY_B
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        steady(t, sa0, sa1, sb0, sb1);
        steady(t + 1, sb0, sb1, sa0, sa1);
    }
    if (t + 2 == ntiles) {                        // even tile count: one more steady step, then the last tile
        steady(t, sa0, sa1, sb0, sb1);
        final_tile(t + 1, sb0, sb1);
    } else {
        final_tile(t, sa0, sa1);
    }

    // row D of O^T holds sum(P) (ones chunk): tile D/32, row D%32 -> register (RR&3) + 4*(RR>>3), h = 0 half
    constexpr int TT = D / 32, RR = D % 32, REG = (RR & 3) + 4 * (RR >> 3);
    const float inv = 1.0f / __shfl(o[TT][REG], r);
    if (qvalid) {
        bf16_t* op = a.O + ((long)b * a.Nq + q) * a.ldo + head * D;
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = 32 * tt + 8 * g4 + 4 * h;
                if (dv < D) {
                    u32x2 w = {pack2bf(o[tt][4 * g4 + 0] * inv, o[tt][4 * g4 + 1] * inv),
                               pack2bf(o[tt][4 * g4 + 2] * inv, o[tt][4 * g4 + 3] * inv)};
                    *(u32x2*)(op + dv) = w;
                }
            }
        }
    }
}

int launch_attn_pipe40(const AttnArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_pipe40_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PipeCfg::SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_pipe40_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PipeCfg::SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_pipe40_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, PipeCfg::SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_pipe40_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, PipeCfg::SMEM));
        attr_set = true;
    }
    constexpr int QB = PipeCfg::NW * 32;
    SD_REQUIRE((long)((a.Nq + QB - 1) / QB) * a.heads * a.B < (1l << 31), "attention: grid too large");
    dim3 grid(((a.Nq + QB - 1) / QB) * a.heads * a.B);          // 1-D: the kernel derives an XCD-aware (query block, head, sample)
    // SD_ATTN_VARIANT: 0 = the round-2 kernel, bits see the kernel (A/B: 0, 1, 3, 7)
    static const int variant = getenv("SD_ATTN_VARIANT") ? atoi(getenv("SD_ATTN_VARIANT")) & 7 : 7;
    const dim3 blk(PipeCfg::NW * 64);
    if (variant == 0) hipLaunchKernelGGL(attn_pipe40_kernel<0>, grid, blk, PipeCfg::SMEM, stream, a);
    else if (variant == 1) hipLaunchKernelGGL(attn_pipe40_kernel<1>, grid, blk, PipeCfg::SMEM, stream, a);
    else if (variant == 3) hipLaunchKernelGGL(attn_pipe40_kernel<3>, grid, blk, PipeCfg::SMEM, stream, a);
    else hipLaunchKernelGGL(attn_pipe40_kernel<7>, grid, blk, PipeCfg::SMEM, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

// =====================================================================================================
// Round 5: the same software pipeline for D = 80 (the 32x32 self-attention: N = Nk = 1024).  attn_kernel<80> stages K / V
// through registers one tile ahead and runs its tile as a serial chain: 0.26 MFMA busy, and its 1024 workgroups of the bench
// shape need 1.33 rounds of the 768 slots three 53-KiB workgroups per CU give.  Here: 4 waves x 32 queries, K / V tiles by
// LDS-DMA into a ring of THREE 24-KiB slots (two workgroups per CU = 512 slots: exactly two rounds), tile t+2 in flight
// while  phase A: exp2 / pack of S(t) || S(t+1) = K(t+1) . Q^T (10 MFMAs)  and  phase B: max over S(t+1) || O^T += V(t)^T . P(t)^T
// (12 MFMAs) run.  At d = 80 the tile is MFMA-heavy (22 x 32 cycles against ~550 of VALU), so the softmax reference stays on
// the VALU (one fma per score) but is STALE like the d = 40 kernel's: M moves only when a tile's maximum exceeds it by 2^8
// (decided on S(t+1) after PV(t) has completed, so everything at the old reference is rescaled exactly once).
// V^T fragments by inline-asm ds_read_b64_tr_b16 in two register sets with hand-counted lgkmcnt (asm_lint.py LDS_WAITS checks the
// compiler's code around them): the builtin makes hipcc wait vmcnt(0) -- the tile in flight -- in front of every tile's first read.
// =====================================================================================================
struct Pipe80Cfg {
    static constexpr int D = 80, KQ = 5, DVT = 3, CD = 10, CHK = 11, CHV = 12, RSK = 176, RSV = 192;
    static constexpr int KBYTES = 64 * RSK, NI = CHK + CHV, NW = 4;
    static constexpr int NIW = 6;                      // DMA pieces per wave and iteration: 11 K + 12 V + 1 padding piece
    // K and V tiles have different lifetimes -- iteration t multiplies K(t+1) and V(t) -- so they live in SEPARATE rings of three:
    // K(t+1) / V(t) in use, K(t+2), K(t+3) / V(t+1), V(t+2) in flight: two iterations for a tile to land (one unified 24-KiB
    // slot per tile allows one tile in flight in 80 KiB, and the loop then ran at the DMA latency: 73 us against 82 before)
    static constexpr int VBYTES = 64 * RSV, NSLOT = 3;
    static constexpr int VOFF = NSLOT * KBYTES, PADOFF = VOFF + NSLOT * VBYTES;      // 33 KiB of K, 36 KiB of V, the padding piece
    static constexpr int SMEM = PADOFF + 1024 + 256;   // ... + the tr over-read of the last V row of the last slot
    static_assert(KBYTES == CHK * 1024 && VBYTES == CHV * 1024 && SMEM <= 80 * 1024, "whole DMA pieces; two workgroups per CU");
};

__global__ __launch_bounds__(Pipe80Cfg::NW * 64, 2) void attn_pipe80_kernel(const AttnArgs a) {
    using Cfg = Pipe80Cfg;
    constexpr int D = Cfg::D, KQ = Cfg::KQ, DVT = Cfg::DVT, CD = Cfg::CD, CHK = Cfg::CHK, CHV = Cfg::CHV;
    constexpr int RSK = Cfg::RSK, RSV = Cfg::RSV, KBYTES = Cfg::KBYTES, VBYTES = Cfg::VBYTES, NI = Cfg::NI, NIW = Cfg::NIW;
    constexpr int NW = Cfg::NW, VOFF = Cfg::VOFF, PADOFF = Cfg::PADOFF;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qblk, head, b;
    xcd_work_item((a.Nq + NW * 32 - 1) / (NW * 32), a.heads, a.xcd_order, qblk, head, b);
    const int q = qblk * (NW * 32) + wave * 32 + r;
    const bool qvalid = q < a.Nq;
    const float c = a.q_prescaled ? 1.0f : a.scale * 1.4426950408889634f;
    const char* zero = (const char*)a.consts;
    const char* ones = zero + 256;

    bf16x8 qf[KQ];
    {
        const bf16_t* qp = a.Q + ((long)b * a.Nq + (qvalid ? q : 0)) * a.ldq + head * D;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            if (qvalid) qf[kk] = *(const bf16x8*)(qp + kk * 16 + h * 8);
            else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) asm volatile("" : "+v"(qf[kk]));   // retire the loads before the DMA ring starts
    }

    // branch-free LDS-DMA descriptors (as attn_pipe40_kernel): per piece a source pointer for tile 0 and a byte step per tile
    const char* d_ptr[NIW];
    long d_step[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int inst = wave + NW * j;
        d_ptr[j] = zero; d_step[j] = 0;
        if (inst < CHK) {
            const int ch = inst * 64 + lane, key = ch / CHK, part = ch - key * CHK;
            if (part < CD) {
                d_ptr[j] = (const char*)(a.K + (long)b * a.Nk * a.ldk + head * D + (long)key * a.ldk + part * 8);
                d_step[j] = 64 * a.ldk * 2;
            }
        } else if (inst < NI) {
            const int ch = (inst - CHK) * 64 + lane, key = ch / CHV, part = ch - key * CHV;
            if (part < CD) {
                d_ptr[j] = (const char*)(a.V + (long)b * a.Nk * a.ldv + head * D + (long)key * a.ldv + part * 8);
                d_step[j] = 64 * a.ldv * 2;
            } else if (part == CD) {
                d_ptr[j] = ones;
            }
        }
    }
    const int ntiles = a.Nk / 64;                 // >= 3, all full (checked by the launcher)
    // one batch = the K pieces of tile tk into K slot ks and the V pieces of tile tv into V slot vs (+ the padding piece);
    // tiles past the end fetch the zero page (their slots are dead)
    auto issue = [&](int tk, int ks, int tv, int vs) {
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int inst = wave + NW * j;                   // wave-uniform
            const bool isk = inst < CHK;
            const int t = isk ? tk : tv;
            const char* src = t < ntiles ? d_ptr[j] + d_step[j] * t : zero;
            char* dst = isk ? smem + ks * KBYTES + inst * 1024 : inst < NI ? smem + VOFF + vs * VBYTES + (inst - CHK) * 1024 : smem + PADOFF;
            glds16(src, dst);
        }
    };

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;

    const int kfrag_off = pi_swap23(r) * RSK + h * 16;
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int vtr_off = VOFF + (8 * h + q4) * RSV + (16 * g16 + 4 * p4) * 2;
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)vtr_off;

    auto qk = [&](const char* tile, f32x16& s0, f32x16& s1) {
        const char* kp = tile + kfrag_off;
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp), qf[0], f32x16{}, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + 32 * RSK), qf[0], f32x16{}, 0, 0, 0);
#pragma unroll
        for (int kk = 1; kk < KQ; ++kk) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + kk * 32), qf[kk], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(kp + 32 * RSK + kk * 32), qf[kk], s1, 0, 0, 0);
        }
    };
    auto exp_group = [&](const f32x16& c0, const f32x16& c1, int s2, float mc) -> bf16x8 {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s2 < 2 ? c0[8 * s2 + j] : c1[8 * (s2 - 2) + j], c, -mc));
        u32x4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
        return __builtin_bit_cast(bf16x8, w);
    };

    // invariant at the start of iteration t: K(0 .. t+2) and V(0 .. t+1) have been requested, in batches of NIW pieces per wave
    issue(0, 0, 0, 0);
    issue(1, 1, 0, 0);                            // (no V tile of its own yet: V(0) again, identical bytes into the same slot)
    issue(2, 2, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");   // K(0), V(0) landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16 sa0, sa1, sb0, sb1;                    // S(t) / S(t+1), roles swap every iteration
    qk(smem, sa0, sa1);
    float m_run;                                  // the softmax reference M (a score), exact for tile 0, stale afterwards
    {
        float tm = fmaxf(sa0[0], sa1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tm = fmaxf(fmaxf(tm, sa0[i]), sa1[i]);
        m_run = half_pair_max(tm);
    }
    constexpr float STALE = 8.f;                  // log2 of the largest probability before M is moved

    // V^T fragments of P group G (keys 16 G .. 16 G + 15 of the tile) for the three 32-row O^T tiles: six transposed reads into
    // register set S; READY(S, N): they have landed when at most N LDS operations issued after them are outstanding
#define SD_VREAD80(S, G)                                                                                    \
    SD_TR_READ(S##0l, vaddr, (G) * 16 * RSV); SD_TR_READ(S##0h, vaddr, (G) * 16 * RSV + 4 * RSV);           \
    SD_TR_READ(S##1l, vaddr, (G) * 16 * RSV + 64); SD_TR_READ(S##1h, vaddr, (G) * 16 * RSV + 64 + 4 * RSV); \
    SD_TR_READ(S##2l, vaddr, (G) * 16 * RSV + 128); SD_TR_READ(S##2h, vaddr, (G) * 16 * RSV + 128 + 4 * RSV);
#define SD_VREADY80(S, N)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(S##0l), "+v"(S##0h), "+v"(S##1l), "+v"(S##1h), "+v"(S##2l), "+v"(S##2h) : "n"(N)); \
    v0 = bf16x8{S##0l[0], S##0l[1], S##0l[2], S##0l[3], S##0h[0], S##0h[1], S##0h[2], S##0h[3]};            \
    v1 = bf16x8{S##1l[0], S##1l[1], S##1l[2], S##1l[3], S##1h[0], S##1h[1], S##1h[2], S##1h[3]};            \
    v2 = bf16x8{S##2l[0], S##2l[1], S##2l[2], S##2l[3], S##2h[0], S##2h[1], S##2h[2], S##2h[3]};
#define SD_PV80(G)                                                                                          \
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf[G], o[0], 0, 0, 0);                               \
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf[G], o[1], 0, 0, 0);                               \
    o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, pf[G], o[2], 0, 0, 0);

    // One steady-state iteration (t <= ntiles - 2): consumes S(t) in (c0, c1), produces S(t+1) in (n0, n1).
    auto steady = [&](int t, int slot, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");     // K(t+1), V(t) landed; K(t+2), V(t+1) may be in flight
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave is past QK(t) and PV(t-1)
        asm volatile("" ::: "memory");
        const int nslot = slot == 2 ? 0 : slot + 1, fslot = slot == 0 ? 2 : slot - 1;
        issue(t + 3, slot, t + 2, fslot);                     // K(t+3) into the slot of K(t), V(t+2) into the slot of V(t-1)
        const char* kp = smem + nslot * KBYTES + kfrag_off;
        const float mc = m_run * c;
        bf16x8 kf[2 * KQ];
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            kf[kk] = *(const bf16x8*)(kp + kk * 32);
            kf[KQ + kk] = *(const bf16x8*)(kp + 32 * RSK + kk * 32);
        }
        bf16x8 pf[4], v0, v1, v2;
        const unsigned vaddr = lds_v + (unsigned)(slot * VBYTES);
        bf16x4 x0l, x0h, x1l, x1h, x2l, x2h, y0l, y0h, y1l, y1h, y2l, y2h;
        __builtin_amdgcn_sched_barrier(0);
        // phase A: two QK^T MFMAs of tile t+1 + exp2 / pack of 8 scores of tile t, four times; then the last k-step
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], f32x16{}, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ], qf[0], f32x16{}, 0, 0, 0);
        pf[0] = exp_group(c0, c1, 0, mc);
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 1], qf[1], n1, 0, 0, 0);
        pf[1] = exp_group(c0, c1, 1, mc);
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 2], qf[2], n1, 0, 0, 0);
        pf[2] = exp_group(c0, c1, 2, mc);
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3], qf[3], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 3], qf[3], n1, 0, 0, 0);
        pf[3] = exp_group(c0, c1, 3, mc);
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 0)
        __builtin_amdgcn_sched_barrier(0);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4], qf[4], n0, 0, 0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[KQ + 4], qf[4], n1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // phase B: three PV MFMAs per P group + a quarter of the running max over S(t+1)
        SD_VREAD80(y, 1) SD_VREADY80(x, 6)
        SD_PV80(0)
        float tm0 = fmaxf(n0[0], n1[0]);
#pragma unroll
        for (int i = 1; i < 4; ++i) tm0 = fmaxf(fmaxf(tm0, n0[i]), n1[i]);
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 2) SD_VREADY80(y, 6)
        SD_PV80(1)
#pragma unroll
        for (int i = 4; i < 8; ++i) tm0 = fmaxf(fmaxf(tm0, n0[i]), n1[i]);
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(y, 3) SD_VREADY80(x, 6)
        SD_PV80(2)
#pragma unroll
        for (int i = 8; i < 12; ++i) tm0 = fmaxf(fmaxf(tm0, n0[i]), n1[i]);
        __builtin_amdgcn_sched_barrier(0);
        SD_VREADY80(y, 0)
        SD_PV80(3)
#pragma unroll
        for (int i = 12; i < 16; ++i) tm0 = fmaxf(fmaxf(tm0, n0[i]), n1[i]);
        const float tmx = half_pair_max(tm0);
        __builtin_amdgcn_sched_barrier(0);
        // rare: some query's tile maximum is more than 2^STALE above its reference -> every lane moves M up to its maximum so
        // far; O (and the row sum in it) is rescaled once, S(t+1) is exponentiated against the new reference next iteration
        if (!__all((tmx - m_run) * c <= STALE)) {
            const float m_new = fmaxf(m_run, tmx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
            m_run = m_new;
        }
    };
    auto final_tile = [&](int slot, f32x16& c0, f32x16& c1) {
        // (what is still in flight fetched the zero page into dead slots)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float mc = m_run * c;
        bf16x8 pf[4], v0, v1, v2;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) pf[s2] = exp_group(c0, c1, s2, mc);
        const unsigned vaddr = lds_v + (unsigned)(slot * VBYTES);
        bf16x4 x0l, x0h, x1l, x1h, x2l, x2h;
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 0) SD_VREADY80(x, 0) SD_PV80(0)
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 1) SD_VREADY80(x, 0) SD_PV80(1)
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 2) SD_VREADY80(x, 0) SD_PV80(2)
        __builtin_amdgcn_sched_barrier(0);
        SD_VREAD80(x, 3) SD_VREADY80(x, 0) SD_PV80(3)
        __builtin_amdgcn_sched_barrier(0);
    };
#undef SD_VREAD80
#undef SD_VREADY80
#undef SD_PV80
    int t = 0, slot = 0;
    for (; t + 2 < ntiles; t += 2) {
        steady(t, slot, sa0, sa1, sb0, sb1);
        slot = slot == 2 ? 0 : slot + 1;
        steady(t + 1, slot, sb0, sb1, sa0, sa1);
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (t + 2 == ntiles) {                        // even tile count: one more steady step, then the last tile
        steady(t, slot, sa0, sa1, sb0, sb1);
        slot = slot == 2 ? 0 : slot + 1;
        final_tile(slot, sb0, sb1);
    } else {
        final_tile(slot, sa0, sa1);
    }

    // row D of O^T holds sum(P) (ones chunk): tile D/32, row D%32 -> register (RR&3) + 4*(RR>>3), h = 0 half
    constexpr int TT = D / 32, RR = D % 32, REG = (RR & 3) + 4 * (RR >> 3);
    static_assert((RR & 4) == 0, "ones row must sit in the h = 0 half");
    const float inv = 1.0f / __shfl(o[TT][REG], r);
    if (qvalid) {
        bf16_t* op = a.O + ((long)b * a.Nq + q) * a.ldo + head * D;
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = 32 * tt + 8 * g4 + 4 * h;
                if (dv < D) {
                    u32x2 w = {pack2bf(o[tt][4 * g4 + 0] * inv, o[tt][4 * g4 + 1] * inv),
                               pack2bf(o[tt][4 * g4 + 2] * inv, o[tt][4 * g4 + 3] * inv)};
                    *(u32x2*)(op + dv) = w;
                }
            }
        }
    }
}

int launch_attn_pipe80(const AttnArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_pipe80_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Pipe80Cfg::SMEM));
        attr_set = true;
    }
    constexpr int QB = Pipe80Cfg::NW * 32;
    SD_REQUIRE((long)((a.Nq + QB - 1) / QB) * a.heads * a.B < (1l << 31), "attention: grid too large");
    dim3 grid(((a.Nq + QB - 1) / QB) * a.heads * a.B);          // 1-D: XCD-aware work order inside the kernel
    hipLaunchKernelGGL(attn_pipe80_kernel, grid, dim3(Pipe80Cfg::NW * 64), Pipe80Cfg::SMEM, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int D>
int launch_attn_dma(const AttnArgs& a, hipStream_t stream) {
    // SD_ATTN_LDS_PAD (KiB): occupancy experiment knob -- extra dynamic LDS limits workgroups per CU
    static const int pad = getenv("SD_ATTN_LDS_PAD") ? atoi(getenv("SD_ATTN_LDS_PAD")) * 1024 : 0;
    const int smem = DmaCfg<D>::SMEM + pad;
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    constexpr int QB = DmaCfg<D>::NW * 32;
    dim3 grid((a.Nq + QB - 1) / QB, a.heads, a.B);
    hipLaunchKernelGGL((attn_dma_kernel<D>), grid, dim3(DmaCfg<D>::NW * 64), smem, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
    const int smem = AttnCfg<D>::SMEM;
    static const bool no_tr = getenv("SD_ATTN_NO_TR") != nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid((a.Nq + 127) / 128, a.heads, a.B);
    if (no_tr) hipLaunchKernelGGL((attn_kernel<D, false>), grid, dim3(256), smem, stream, a);
    else hipLaunchKernelGGL((attn_kernel<D, true>), grid, dim3(256), smem, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int sd_launch_attention(const AttnArgs& a0, hipStream_t stream) {
    AttnArgs a = a0;
    {
        const char* e = getenv("SD_ATTN_XCD");          // per call: A/B of the work order in one process
        if (e && atoi(e) == 0) a.xcd_order = 0;
    }
    SD_REQUIRE(a.Q && a.K && a.V && a.O, "attention: null operand");
    SD_REQUIRE(a.B > 0 && a.heads > 0 && a.Nq > 0 && a.Nk > 0, "attention: empty problem");
    if (a.kv_head_major) {
        SD_REQUIRE(a.D == 40 && a.Nk % 64 == 0 && a.Nk >= 256 && a.consts != nullptr && a.ldq % 8 == 0 && a.ldo % 4 == 0 &&
                       a.ldq >= (long)a.heads * a.D && a.ldo >= (long)a.heads * a.D,
                   "attention: head-major K / V is built for d = 40 and key counts that are multiples of 64 (>= 256)");
        return launch_attn_pipe40(a, stream);
    }
    SD_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0,
               "attention: row strides must keep 16-byte alignment");
    SD_REQUIRE(a.ldq >= (long)a.heads * a.D && a.ldk >= (long)a.heads * a.D && a.ldv >= (long)a.heads * a.D &&
                   a.ldo >= (long)a.heads * a.D, "attention: row stride smaller than heads*D");
    SD_REQUIRE(((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V) % 16 == 0 && (uintptr_t)a.O % 8 == 0,
               "attention: operands must be 16-byte aligned");
    SD_REQUIRE(a.B <= 65535 && a.heads <= 65535, "attention: grid too large");
    static const bool no_dma = getenv("SD_ATTN_NO_DMA") != nullptr;
    const bool dma = !no_dma && a.consts != nullptr;
    switch (a.D) {
        case 40: {
            static const bool no_pipe = getenv("SD_ATTN_NO_PIPE") != nullptr;
            if (dma && !no_pipe && a.Nk % 64 == 0 && a.Nk >= 256) return launch_attn_pipe40(a, stream);
            return dma ? launch_attn_dma<40>(a, stream) : launch_attn<40>(a, stream);
        }
        case 80: {
            // measured at the 32x32 level (UNet batch 16): register-staged 79 us vs LDS-DMA 87 us self, 21.5 vs 23.6 us
            // cross -- the 8-wave DMA workgroups only pay off at d = 40; SD_ATTN_DMA80 re-selects the DMA kernel
            static const bool dma80 = getenv("SD_ATTN_DMA80") != nullptr;
            // round 5: the software-pipelined kernel for key counts that are a multiple of 64 (the 32x32 self-attention);
            // SD_ATTN_NO_PIPE keeps everything on attn_kernel<80>
            static const bool no_pipe80 = getenv("SD_ATTN_NO_PIPE") != nullptr;
            if (dma && !no_pipe80 && !a.kv_head_major && a.Nk % 64 == 0 && a.Nk >= 192) return launch_attn_pipe80(a, stream);
            return (dma && dma80) ? launch_attn_dma<80>(a, stream) : launch_attn<80>(a, stream);
        }
        case 160: return launch_attn<160>(a, stream);
        default: sd_set_error("attention: head dim %d not supported (40, 80, 160)", a.D); return -1;
    }
}
