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
//             LDS-DMA, ring of 5 (three tiles of lead), counted vmcnt.
//   softmax   per head over its 80 key slots: lane-local (a lane holds 40 of them, its partner lane ^ 32 the rest);
//             slots >= L are masked; probabilities are normalised and packed to bf16 IN PLACE -- a 32x32 accumulator
//             tile is directly the B operand of the next product (k order inside a 16-step: 8(j>>2) + 4h + (j&3)).
//   phase 2   Y^T[32 channels, 32 tokens] = Bw . P^T         per 32-channel tile: 40 MFMAs over the 640 key slots, Bw
//             rows (channel-major, key-contiguous, stored in the permuted k order above) and the residual tile by
//             LDS-DMA, ring of 3; epilogue from LDS only: + b_o + R in fp32, bf16, v_permlane32_swap pairs the lane
//             halves into 16-byte row stores.  grid.y splits the channel tiles when the token grid alone cannot fill
//             the chip (each slice repeats phase 1).
// With one wave per SIMD nothing hides a wave's issue stalls, so (a) every LDS-DMA piece (~60-100 issue cycles) is
// placed singly between two MFMAs instead of in a burst after the barrier, (b) the fragments of the next k-step are
// requested a few MFMAs into the current one, and (c) the loops contain no ordinary global load (hipcc would drain
// the whole DMA pipeline with vmcnt(0) at its first use): residual and bias come through LDS.
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
constexpr int BWBYTES = 32 * KEYS * 2;      // 40 KiB: Bw rows of one 32-channel tile
constexpr int STAGE2 = BWBYTES + TOK * 64;  // + the residual tile (128 tokens x 32 channels) = 48 KiB
constexpr int NSTAGE1 = 5;                  // phase 1: three K tiles of lead
constexpr int NSTAGE2 = 3;
constexpr int RING = NSTAGE2 * STAGE2;      // 144 KiB (>= NSTAGE1 * STAGE1 = 112 KiB)
constexpr int MAXC = 1280;
constexpr int SMEM = RING + MAXC * 4;       // + the to_out bias
static_assert(NSTAGE1 * STAGE1 <= RING && SMEM <= 160 * 1024, "LDS budget");
constexpr int P1_PIECES = 7;                // 1-KiB DMA pieces per wave and K tile (28 / 4)
constexpr int P2_PIECES = 12;               // ... per wave and channel tile (40 + 8) / 4

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
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of token tiles,
    // i.e. whole samples, so that the 32 workgroups that stream one sample's A^T / Bw (800 KiB at C = 320) share one L2
    // (with the plain order every L2 sees the operands of all 8 samples in flight: 6.4 MiB against 4 MiB).  Speed only.
    int wg = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = wg & 7, idx = wg >> 3;
        wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const long m0 = (long)wg * TOK;
    const int sample = (int)(m0 / p.rows_per_sample);
    const char* At = (const char*)p.At + (long)sample * KEYS * C * 2;
    const char* Bw = (const char*)p.Bw + (long)sample * C * KEYS * 2;
    const long rowbytes = (long)C * 2;
    const int KT = C / 32;
    const int NTall = C / 32, nsl = gridDim.y;
    const int jbeg = (int)((long)NTall * blockIdx.y / nsl), NT = (int)((long)NTall * (blockIdx.y + 1) / nsl) - jbeg;

    // diagnostic phase stamps (only with a stamp buffer: tools/xattn_stamps.py); never read by the kernel
    auto stamp = [&](int i) {
        if (p.stamps && tid == 0) p.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    // to_out bias -> LDS (before any DMA is in flight: these are the kernel's only ordinary global loads)
    float* sbias = (float*)(smem + RING);
    for (int i = tid; i < C; i += 256) sbias[i] = p.bias[i];

    // ---- LDS-DMA source pointers (16 rows x 64 B per piece; lane -> row lane >> 2, chunk lane & 3, swizzled) ----
    const int prow = lane >> 2;
    const int sch = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;              // source chunk byte offset
    const char* aptr = At + (long)(16 * wave + prow) * rowbytes + sch;   // A piece i: + i * 64 rows; pass g: + 320 rows
    const char* xptr = (const char*)p.X + (m0 + 16 * wave + prow) * rowbytes + sch;   // X piece i: + i * 64 rows
    // `real` = false: nothing left to fetch -- the piece still issues (the K loop stays one branch-free block with
    // constant DMA counts; a join would make hipcc drain lgkmcnt in front of the next MFMA) but reads 16 hot bytes
    const char* dummy = (const char*)p.bias;
    auto piece1 = [&](int pass, int kt, int i, char* st, bool real) {  // i = 0..4: A^T rows, 5..6: X rows
        const char* src = i < 5 ? aptr + ((long)pass * HKEYS + i * 64) * rowbytes + kt * 64
                                : xptr + (long)(i - 5) * 64 * rowbytes + kt * 64;
        glds16(real ? src : dummy, st + (i < 5 ? wave + 4 * i : 20 + 4 * (i - 5) + wave) * 1024);
    };
    auto issue1 = [&](int pass, int kt, char* st) {
#pragma unroll
        for (int i = 0; i < P1_PIECES; ++i) piece1(pass, kt, i, st, true);
    };
    // phase 2: Bw piece q = wave + 4 i (i < 10) covers sub-tile kt = q >> 1 (32 key slots = 64 B), rows 16 (q & 1) .. + 15;
    // pieces 10, 11: residual rows 16 (wave + 4 (i - 10)) .. + 15 of the token tile
    const char* bptr = Bw + (long)(16 * (wave & 1) + prow) * (KEYS * 2) + (wave >> 1) * 64 + sch;   // piece i: + 128 B
    const char* rptr = (const char*)p.R + (m0 + 16 * wave + prow) * rowbytes + sch;                  // piece i: + 64 rows
    auto piece2 = [&](int j, int i, char* st, bool real) {
        const char* src = i < 10 ? bptr + (long)j * 32 * (KEYS * 2) + i * 128 : rptr + (long)(i - 10) * 64 * rowbytes + j * 64;
        glds16(real ? src : dummy, st + (i < 10 ? (wave + 4 * i) * 1024 : BWBYTES + (wave + 4 * (i - 10)) * 1024));
    };
    auto issue2 = [&](int j, char* st) {
#pragma unroll
        for (int i = 0; i < P2_PIECES; ++i) piece2(j, i, st, true);
    };

    // ---- fragment read offsets: row r of a 32-row tile, k-step ks: chunk (2 ks + h) ^ ((r >> 2) & 3) ----
    const int fsw = (r >> 2) & 3;
    const int fo0 = r * 64 + (((0 + h) ^ fsw) << 4);
    const int fo1 = r * 64 + (((2 + h) ^ fsw) << 4);
    const int xrow = (HKEYS + 32 * wave) * 64;

    bf16x8 P[4 * NKT];              // P^T fragments of all 40 half-tiles (16 key slots each)
    const float c = 1.4426950408889634f;

    auto frags = [&](const char* sb, int fo, bf16x8* af, bf16x8& xf) {
        xf = *(const bf16x8*)(sb + xrow + fo);
#pragma unroll
        for (int t = 0; t < NKT; ++t) af[t] = *(const bf16x8*)(sb + t * 2048 + fo);
    };
    auto prologue1 = [&](int pass) {
        issue1(pass, 0, smem);
        if (KT > 1) issue1(pass, 1, smem + STAGE1);
        if (KT > 2) issue1(pass, 2, smem + 2 * STAGE1);
        if (KT > 3) issue1(pass, 3, smem + 3 * STAGE1);
    };
    __builtin_amdgcn_s_waitcnt(0);          // bias loads + LDS writes retired
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
        if (KT > 3) wait_vmcnt<3 * P1_PIECES>();        // tile 0 landed, tiles 1..3 may still be in flight
        else if (KT > 2) wait_vmcnt<2 * P1_PIECES>();
        else if (KT > 1) wait_vmcnt<P1_PIECES>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frags(smem, fo0, a0, x0);
        int st = 0;                                     // stage of tile kt (kt mod 5)
        for (int kt = 0; kt < KT; ++kt) {
            const char* sb = smem + st * STAGE1;
            const int sn = st == NSTAGE1 - 1 ? 0 : st + 1;
            // hipcc drains lgkmcnt at a loop head / block entry; with the next fragments requested AFTER the first
            // MFMAs that wait only covers reads issued 10 MFMAs earlier
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
            // The one barrier per K tile sits BETWEEN the two k-steps: by then every wave has left tile kt-1 (its
            // stage is refilled after the barrier) and has retired its own DMA of tile kt+1 (read after the barrier).
            // in flight at this point: tiles kt+1, kt+2, kt+3 (or their dummies) -> tile kt+1 has landed once at most
            // 2 x P1_PIECES remain; during the first iterations fewer were issued, the count is then conservative
            if (KT > 3) wait_vmcnt<2 * P1_PIECES>();    // (tiles 0 .. kt+3 have been issued: prologue + one per iteration)
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            char* refill = smem + (st == 0 ? NSTAGE1 - 1 : st - 1) * STAGE1;   // the stage tile kt-1 was read from = stage of kt+4
            const bool more = kt + 4 < KT;
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[t], x1, S[t], 0, 0, 0);
                asm("" : "+a"(S[t]));
                __builtin_amdgcn_sched_barrier(0);
                if (t == 1 && kt + 1 < KT) frags(smem + sn * STAGE1, fo0, a0, x0);
                if (t >= 2 && t < 2 + P1_PIECES) piece1(pass, kt + 4, t - 2, refill, more);   // one DMA piece per MFMA gap
                __builtin_amdgcn_sched_barrier(0);
            }
            st = sn;
        }
        stamp(1 + 2 * pass);
        // every wave is done reading this pass's stages -> the next operand stream starts under the softmax
        wait_vmcnt<0>();                                // (trailing dummy pieces: no two DMAs to one LDS address in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (pass == 0) {
            prologue1(1);
        } else {
            issue2(jbeg, smem);
            if (NT > 1) issue2(jbeg + 1, smem + STAGE2);
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
        stamp(2 + 2 * pass);
    }

    // =============================== phase 2: Y^T = Bw . P^T, 32 channels at a time ===============================
    const long trow = m0 + 32 * wave + r;                       // this lane's token
    char* Yrow = (char*)p.Y + trow * rowbytes;
    // residual tile in LDS: row 32 wave + r, 16-byte chunk g (channels 8 g .. 8 g + 7) at position g ^ swizzle, + 8 h
    const int roff = BWBYTES + (32 * wave + r) * 64 + 8 * h;
    // 8-deep ring of Bw fragments; the barrier of tile j+1 between MFMA 20 and 21 of tile j, one DMA piece of tile j+2
    // per MFMA gap after it, the first fragments of tile j+1 requested before the epilogue of tile j.
    auto bfrag = [&](const char* sb, int g) { return *(const bf16x8*)(sb + (g >> 1) * 2048 + ((g & 1) ? fo1 : fo0)); };
    constexpr int NRING = 8, NG = 4 * NKT;
    bf16x8 bq[NRING];
    wait_vmcnt<0>();                                    // Bw / R tiles 0 and 1 (issued before the softmax)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < NRING; ++g) bq[g] = bfrag(smem, g);
    int st = 0;                                         // (phase-2 stage index)
    for (int jj = 0; jj < NT; ++jj) {
        const int j = jbeg + jj;
        const char* sb = smem + st * STAGE2;
        const int sn = st == 2 ? 0 : st + 1;
        char* refill = smem + (st == 0 ? 2 : st - 1) * STAGE2;
        const bool more = jj + 2 < NT;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g == NG / 2 && jj + 1 < NT) {
                // Own DMA of tile j+1 is older than the previous tile's two stores, the only younger vector-memory ops.
                if (jj == 0) wait_vmcnt<0>();
                else wait_vmcnt<2>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[g % NRING], P[g], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + NRING < NG) bq[g % NRING] = bfrag(sb, g + NRING);
            else if (jj + 1 < NT) bq[g % NRING] = bfrag(smem + sn * STAGE2, g + NRING - NG);
            if (g > NG / 2 && g <= NG / 2 + P2_PIECES) piece2(j + 2, g - NG / 2 - 1, refill, more);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue (LDS operands only): lane holds channels 32 j + 8 g + 4 h + (0..3), g = 0..3, of its token ----
        const int cb = 32 * j + 4 * h;
        u32x2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2 rr = *(const u32x2*)(sb + roff + ((g ^ fsw) << 4));
            const f32x4 bv = *(const f32x4*)(sbias + cb + 8 * g);
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
        st = sn;
    }
    stamp(5);
}

}  // namespace

bool sd_xattn_fused_applicable(int rows_per_sample, int C, int heads, int L) {
    static const char* env = getenv("SD_XATTN_FUSED");
    if (env && atoi(env) == 0) return false;
    return heads == 8 && L > 64 && L <= 80 && C % 32 == 0 && C >= 64 && C <= MAXC && rows_per_sample % TOK == 0;
}

int sd_launch_xattn_fused(const XattnArgs& a, hipStream_t stream) {
    SD_REQUIRE(a.X && a.R && a.Y && a.At && a.Bw && a.bias, "xattn: null operand");
    SD_REQUIRE(a.C % 32 == 0 && a.C >= 64 && a.C <= MAXC, "xattn: C=%d must be a multiple of 32 in [64, %d]", a.C, MAXC);
    SD_REQUIRE(a.rows_per_sample % TOK == 0 && a.M % a.rows_per_sample == 0 && a.M > 0,
               "xattn: %d tokens per sample must be a multiple of %d (M=%d)", a.rows_per_sample, TOK, a.M);
    SD_REQUIRE(a.L > 64 && a.L <= 80, "xattn: %d prompt keys (65..80 are built)", a.L);
    SD_REQUIRE((long)KEYS * a.C * 2 * (a.M / a.rows_per_sample) < (1l << 40), "xattn: operand too large");
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    // channel slices (each repeats phase 1): only when the token grid leaves most of the 256 CUs idle
    const int wgs = a.M / TOK, nt = a.C / 32;
    int nsl = 1;
    static const int force = getenv("SD_XATTN_SLICES") ? atoi(getenv("SD_XATTN_SLICES")) : 0;
    if (force > 0) nsl = force;
    else if (wgs <= 128) nsl = 2;
    if (nsl > nt) nsl = nt;
    hipLaunchKernelGGL(xattn_fused_kernel, dim3(wgs, nsl), dim3(256), SMEM, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
