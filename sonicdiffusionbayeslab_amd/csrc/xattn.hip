// Fused prompt cross-attention for gfx950: one launch per transformer block.
//
//   Y = R + sum_h softmax_L( X . A_h ) . B_h + b_o            (src/models.py:227-235 -> diffusers Attention with
//                                                                the L = 77 prompt keys, 8 heads)
// The prompt is step-invariant, so per sample and head  A_h = scale * W_q,h^T K_h^T  [C x 80]  and
// B_h = V_h W_o,h^T  [80 x C]  are computed once per sampling run (sd_unet_set_context); what is left per step is
// "attention with 8 x 80 keys whose head dimension is C".  to_q, Q K^T, softmax, P V and to_out never leave the chip:
// X, R are read once and Y is written once (3 x M x C x 2 bytes), against the five round trips of
// to_q -> attention -> to_out.  Executed work: 2 x (2 M x 640 x C) flop on the matrix cores.
//
// One workgroup = 4 waves = 128 tokens of one sample, one wave per SIMD with the whole 512-entry register file:
//   phase 1   S^T[320 keys, 32 tokens] = A^T . X^T          v_mfma_f32_32x32x16_bf16, A^T rows as the A operand, the
//             wave's 32 tokens on the lanes; two passes of 4 heads = 10 key tiles each, accumulated at once in 160
//             accumulation registers (pinned to the AGPR half of the file: the VALU never touches them inside the K
//             loop), K = C streamed in 32-channel tiles: A^T tile (320 x 64 B) + X tile (128 x 64 B) = 28 KiB by
//             LDS-DMA, ring of 4 (two tiles of lead), counted vmcnt.
//   softmax   per head over its 80 key slots: lane-local (a lane holds 40 of them, its partner lane ^ 32 the rest);
//             slots >= L are masked; probabilities are normalised and packed to bf16 IN PLACE -- a 32x32 accumulator
//             tile is directly the B operand of the next product (k order inside a 16-step: 8(j>>2) + 4h + (j&3)).
//   phase 2   Y^T[32 channels, 32 tokens] = Bw . P^T         per 32-channel tile: 40 MFMAs over the 640 key slots, Bw
//             rows (channel-major, key-contiguous, stored in the permuted k order above) by LDS-DMA, ring of 3;
//             epilogue: + b_o + R in fp32, bf16, v_permlane32_swap pairs the lane halves into 16-byte row stores.
// LDS rows are 64 B; the four 16-byte chunks of a row are XOR-swizzled by (row >> 2) & 3 on the DMA source address, which
// makes the ds_read_b128 fragment reads (32 consecutive rows, one chunk) bank-conflict free.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

constexpr int TOK = 128;                    // tokens per workgroup
constexpr int KEYS = 640;                   // 8 heads x 80 key slots
constexpr int HKEYS = KEYS / 2;             // key slots of one pass (4 heads)
constexpr int NKT = HKEYS / 32;             // 10 key tiles of 32 per pass
constexpr int STAGE1 = (HKEYS + TOK) * 64;  // 28 KiB: A^T half tile + X tile of one 32-channel K tile
constexpr int STAGE2 = 32 * KEYS * 2;       // 40 KiB: Bw rows of one 32-channel tile
constexpr int NSTAGE1 = 4;                  // phase 1: two K tiles of lead
constexpr int NSTAGE2 = 3;
constexpr int SMEM = NSTAGE2 * STAGE2;      // 120 KiB (>= NSTAGE1 * STAGE1 = 112 KiB)
static_assert(NSTAGE1 * STAGE1 <= SMEM, "phase-1 ring must fit the phase-2 ring");
constexpr int P1_PIECES = 7;                // 1-KiB DMA pieces per wave and K tile (28 / 4)
constexpr int P2_PIECES = 10;               // ... per wave and channel tile (40 / 4)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(256, 1) void xattn_fused_kernel(const XattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int C = p.C;
    const long m0 = (long)blockIdx.x * TOK;
    const int sample = (int)(m0 / p.rows_per_sample);
    const char* At = (const char*)p.At + (long)sample * KEYS * C * 2;
    const char* Bw = (const char*)p.Bw + (long)sample * C * KEYS * 2;
    const long rowbytes = (long)C * 2;

    // ---- LDS-DMA source pointers (16 rows x 64 B per piece; lane -> row lane >> 2, chunk lane & 3, swizzled) ----
    const int prow = lane >> 2;
    const int sch = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;              // source chunk byte offset
    const char* aptr = At + (long)(16 * wave + prow) * rowbytes + sch;   // piece i: + i * 64 rows; pass g: + 320 rows
    const char* xptr = (const char*)p.X + (m0 + 16 * wave + prow) * rowbytes + sch;   // pieces 5, 6: + 64 rows
    auto issue1 = [&](int pass, int kt, char* st) {
        const char* ap = aptr + (long)pass * HKEYS * rowbytes + kt * 64;
#pragma unroll
        for (int i = 0; i < 5; ++i) glds16(ap + (long)i * 64 * rowbytes, st + (wave + 4 * i) * 1024);
        glds16(xptr + kt * 64, st + (20 + wave) * 1024);
        glds16(xptr + 64 * rowbytes + kt * 64, st + (24 + wave) * 1024);
    };
    // phase 2: piece q = wave + 4 i covers sub-tile kt = q >> 1 (32 key slots = 64 B), rows 16 (q & 1) .. + 15
    const char* bptr = Bw + (long)(16 * (wave & 1) + prow) * (KEYS * 2) + (wave >> 1) * 64 + sch;   // piece i: + 128 B
    auto issue2 = [&](int j, char* st) {
        const char* src = bptr + (long)j * 32 * (KEYS * 2);
#pragma unroll
        for (int i = 0; i < P2_PIECES; ++i) glds16(src + i * 128, st + (wave + 4 * i) * 1024);
    };

    // ---- fragment read offsets: row r of a 32-row tile, k-step ks: chunk (2 ks + h) ^ ((r >> 2) & 3) ----
    const int fo0 = r * 64 + (((0 + h) ^ ((r >> 2) & 3)) << 4);
    const int fo1 = r * 64 + (((2 + h) ^ ((r >> 2) & 3)) << 4);
    const int xrow = (HKEYS + 32 * wave) * 64;

    const int KT = C / 32, NT = C / 32;
    bf16x8 P[4 * NKT];              // P^T fragments of all 40 half-tiles (16 key slots each)
    const float c = 1.4426950408889634f;

    // Fragment prefetch: the LDS reads of k-step 1 are issued before the MFMAs of k-step 0, and those of the NEXT
    // tile's k-step 0 before the MFMAs of k-step 1, so that with one wave per SIMD no MFMA waits for an LDS read issued
    // just before it.  The one barrier per K tile sits BETWEEN the two k-steps: by then every wave has left tile kt-1
    // (its stage is refilled right after the barrier) and has retired its own DMA of tile kt+1 (read after the barrier).
    auto frags = [&](const char* sb, int fo, bf16x8* af, bf16x8& xf) {
        xf = *(const bf16x8*)(sb + xrow + fo);
#pragma unroll
        for (int t = 0; t < NKT; ++t) af[t] = *(const bf16x8*)(sb + t * 2048 + fo);
    };
    auto prologue1 = [&](int pass) {
        issue1(pass, 0, smem);
        if (KT > 1) issue1(pass, 1, smem + STAGE1);
        if (KT > 2) issue1(pass, 2, smem + 2 * STAGE1);
    };
    prologue1(0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // =============================== phase 1: S^T = A^T . X^T (4 heads) ===============================
        f32x16 S[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
            asm("" : "+a"(S[t]));
        }
        bf16x8 a0[NKT], a1[NKT], x0, x1;
        if (KT > 2) wait_vmcnt<2 * P1_PIECES>();        // tile 0 landed, tiles 1 and 2 may still be in flight
        else if (KT > 1) wait_vmcnt<P1_PIECES>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frags(smem, fo0, a0, x0);
        for (int kt = 0; kt < KT; ++kt) {
            const char* sb = smem + (kt & 3) * STAGE1;
            // hipcc drains lgkmcnt at a loop head; with the k-step-1 reads issued AFTER the first MFMAs that wait only
            // covers the k-step-0 fragments requested 10 MFMAs earlier
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[t], x0, S[t], 0, 0, 0);
                asm("" : "+a"(S[t]));                   // accumulators stay in the AGPR half: no VGPR<->AGPR shuttling
                if (t == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    frags(sb, fo1, a1, x1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT) {
                if (kt + 2 < KT) wait_vmcnt<P1_PIECES>();   // own pieces of tile kt+1 landed, tile kt+2 may be in flight
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 3 < KT) issue1(pass, kt + 3, smem + ((kt + 3) & 3) * STAGE1);   // the stage tile kt-1 was read from
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], x1, S[t], 0, 0, 0);
                asm("" : "+a"(S[t]));
                if (t == 1) {                           // (same reason: the block-entry wait must not cover fresh reads)
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < KT) frags(smem + ((kt + 1) & 3) * STAGE1, fo0, a0, x0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave is done reading this pass's stages -> the next operand stream starts under the softmax
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (pass == 0) {
            prologue1(1);
        } else {
            issue2(0, smem);
            if (NT > 1) issue2(1, smem + STAGE2);
        }

        // =============================== softmax over each head's 80 key slots ===============================
        // lane (token r, half h) holds key 32 t + (i & 3) + 8 (i >> 2) + 4 h in S[t][i]; head hd = half-tiles 5 hd .. 5 hd + 4
        // (a half-tile = registers 8 s .. 8 s + 7 of tile t, u = 2 t + s); key slots >= L of a head sit in its last half-tile.
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            float v[40];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int u = 5 * hd + q;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[q * 8 + j] = S[u >> 1][(u & 1) * 8 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int slot = 64 + 8 * (j >> 2) + 4 * h + (j & 3);
                if (slot >= p.L) v[32 + j] = -1e30f;
            }
            float mx = v[0];
#pragma unroll
            for (int i = 1; i < 40; ++i) mx = fmaxf(mx, v[i]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mc = mx * c;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 40; ++i) {
                v[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], c, -mc));
                sum += v[i];
            }
            sum += __shfl_xor(sum, 32);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const u32x4 pk = {pack2bf(v[q * 8 + 0] * inv, v[q * 8 + 1] * inv), pack2bf(v[q * 8 + 2] * inv, v[q * 8 + 3] * inv),
                                  pack2bf(v[q * 8 + 4] * inv, v[q * 8 + 5] * inv), pack2bf(v[q * 8 + 6] * inv, v[q * 8 + 7] * inv)};
                P[20 * pass + 5 * hd + q] = __builtin_bit_cast(bf16x8, pk);
            }
        }
    }

    // =============================== phase 2: Y^T = Bw . P^T, 32 channels at a time ===============================
    const long trow = m0 + 32 * wave + r;                       // this lane's token
    const char* Rrow = (const char*)p.R + trow * rowbytes;
    char* Yrow = (char*)p.Y + trow * rowbytes;
    // Same structure: 8-deep ring of Bw fragments, the barrier of tile j+1 between MFMA 20 and 21 of tile j, the first
    // fragments of tile j+1 requested before the epilogue of tile j.
    auto bfrag = [&](const char* sb, int g) { return *(const bf16x8*)(sb + (g >> 1) * 2048 + ((g & 1) ? fo1 : fo0)); };
    constexpr int RING = 8, NG = 4 * NKT;
    bf16x8 bq[RING];
    wait_vmcnt<0>();                                    // Bw tiles 0 and 1 (issued before the softmax)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < RING; ++g) bq[g] = bfrag(smem, g);
    int st = 0;
    for (int j = 0; j < NT; ++j) {
        const char* sb = smem + st * STAGE2;
        const int sn = st == 2 ? 0 : st + 1;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g == NG / 2 && j + 1 < NT) {
                // Own DMA of tile j+1 is older than the residual / bias loads of the previous tile, whose data has been
                // consumed; only that tile's two stores may still be in flight.
                if (j == 0) wait_vmcnt<0>();
                else wait_vmcnt<2>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (j + 2 < NT) issue2(j + 2, smem + (st == 0 ? 2 : st - 1) * STAGE2);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g % RING], P[g], acc, 0, 0, 0);
            if (g + RING < NG) bq[g % RING] = bfrag(sb, g + RING);
            else if (j + 1 < NT) bq[g % RING] = bfrag(smem + sn * STAGE2, g + RING - NG);
        }
        st = sn;
        // ---- epilogue: lane holds channels 32 j + 8 g + 4 h + (0..3), g = 0..3, of its token ----
        const int cb = 32 * j + 4 * h;
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 rr = *(const u32x2*)(Rrow + (cb + 8 * g) * 2);
            const f32x4 bv = *(const f32x4*)(p.bias + cb + 8 * g);
            const float y0 = acc[4 * g + 0] + bv[0] + bflo(rr[0]), y1 = acc[4 * g + 1] + bv[1] + bfhi(rr[0]);
            const float y2 = acc[4 * g + 2] + bv[2] + bflo(rr[1]), y3 = acc[4 * g + 3] + bv[3] + bfhi(rr[1]);
            pk[g] = u32x2{pack2bf(y0, y1), pack2bf(y2, y3)};
        }
        // lanes r / r + 32 hold channels 8 g + (0..3) / 8 g + (4..7): swap pairs of groups -> 16 contiguous bytes per lane
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * q][0], pk[2 * q + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * q][1], pk[2 * q + 1][1], false, false);
            // h = 0: [own 2q | partner's 2q] = channels 16 q .. 16 q + 7;  h = 1: [partner's 2q+1 | own 2q+1] = 16 q + 8 .. + 15
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            *(u32x4*)(Yrow + (32 * j + 16 * q + 8 * h) * 2) = o;
        }
    }
}

}  // namespace

bool sd_xattn_fused_applicable(int rows_per_sample, int C, int heads, int L) {
    static const char* env = getenv("SD_XATTN_FUSED");
    if (env && atoi(env) == 0) return false;
    return heads == 8 && L > 64 && L <= 80 && C % 32 == 0 && C >= 64 && rows_per_sample % TOK == 0;
}

int sd_launch_xattn_fused(const XattnArgs& a, hipStream_t stream) {
    SD_REQUIRE(a.X && a.R && a.Y && a.At && a.Bw && a.bias, "xattn: null operand");
    SD_REQUIRE(a.C % 32 == 0 && a.C >= 64, "xattn: C=%d must be a multiple of 32", a.C);
    SD_REQUIRE(a.rows_per_sample % TOK == 0 && a.M % a.rows_per_sample == 0 && a.M > 0,
               "xattn: %d tokens per sample must be a multiple of %d (M=%d)", a.rows_per_sample, TOK, a.M);
    SD_REQUIRE(a.L > 64 && a.L <= 80, "xattn: %d prompt keys (65..80 are built)", a.L);
    SD_REQUIRE((long)KEYS * a.C * 2 * (a.M / a.rows_per_sample) < (1l << 40), "xattn: operand too large");
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    hipLaunchKernelGGL(xattn_fused_kernel, dim3(a.M / TOK), dim3(256), SMEM, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
