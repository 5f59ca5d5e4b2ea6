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
// One workgroup = 8 waves = 128 tokens of one sample, TWO waves per SIMD (256 registers each): wave w owns the token
// block tb = w & 3 (32 tokens, on the MFMA lanes) and, in phase 1, the head group hg = w >> 2 (4 heads = 320 key slots).
// A first version with one 512-register wave per SIMD measured 18 % MFMA busy: with nothing else resident on the SIMD
// every LDS-DMA piece (~9 cycles per cache line it touches, ~145 per 16-row piece) and every row-scattered store stalls
// the matrix pipe for its whole issue time (s_memtime stamps: phase 2 at 4x its MFMA time).  Two waves per SIMD overlap
// one wave's issue stalls with the partner's MFMAs; splitting the HEADS between the partners keeps each wave at 160
// accumulators.
//   phase 1   S^T[320 keys, 32 tokens] = A^T . X^T   v_mfma_f32_32x32x16_bf16, A^T rows as the A operand; 10 key tiles
//             accumulated at once in 160 registers (built with -amdgpu-mfma-vgpr-form: MFMA results in arch VGPRs, so
//             the softmax reads them in place and all 256 registers of the wave are one pool).  K = C is
//             streamed in 32-channel tiles: A^T tile (640 x 64 B) + X tile (128 x 64 B) = 48 KiB by LDS-DMA, ring of 3,
//             counted vmcnt, 6 one-KiB pieces per wave and tile, one piece per MFMA gap.
//   softmax   per head over its 80 key slots: lane-local (a lane holds 40 of them, its partner lane ^ 32 the rest);
//             slots >= L are masked; probabilities are normalised and packed to bf16 IN PLACE -- a 32x32 accumulator
//             tile is directly the B operand of the next product (k order inside a 16-step: 8(j>>2) + 4h + (j&3)).
//   exchange  the two waves of a token block swap their halves of P through LDS (2 rounds of 10 KiB per wave): each
//             then holds the probabilities of all 8 heads (160 registers) ...
//   phase 2   ... and they split the CHANNEL tiles instead: Y^T[32 channels, 32 tokens] = Bw . P^T, wave hg takes tile
//             hg of every pair of 32-channel tiles.  A stage = (tile pair, half of the 640 key slots) = 40 KiB of Bw rows
//             (channel-major, key-contiguous, stored in the permuted k order above) + 1/2 of the pair's residual tile,
//             ring of 3, again 6 pieces per wave and stage; 20 MFMAs per wave and stage.  Epilogue from LDS only
//             (residual, bias): + b_o + R in fp32, bf16, v_permlane32_swap pairs the lane halves into 16-byte row stores.
//             grid.y splits the tile pairs when the token grid alone cannot fill the chip (each slice repeats phase 1).
// The loops contain no ordinary global load (hipcc would drain the whole DMA pipeline with vmcnt(0) at its first use)
// and no branch around a DMA piece (a dummy source keeps the counts constant).  XCD-aware workgroup order.
// LDS rows are 64 B; the four 16-byte chunks of a row are XOR-swizzled by (row >> 2) & 3 on the DMA source address, which
// makes the ds_read_b128 fragment reads (32 consecutive rows, one chunk) bank-conflict free.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int TOK = 128;                    // tokens per workgroup
constexpr int KEYS = 640;                   // 8 heads x 80 key slots
constexpr int HKEYS = KEYS / 2;             // key slots of one head group (4 heads)
constexpr int NKT = HKEYS / 32;             // 10 key tiles of 32 per head group
constexpr int STAGE1 = (KEYS + TOK) * 64;   // 48 KiB: A^T tile + X tile of one 32-channel K tile
constexpr int STAGE2 = 2 * 32 * HKEYS * 2;  // 40 KiB: Bw rows of a pair of 32-channel tiles x half the key slots
constexpr int NSTAGE = 3;
constexpr int RING = NSTAGE * STAGE1;       // 144 KiB; phase 2: 3 x 40 KiB + residual ring 2 x 16 KiB = 152 KiB -> see ROFF
constexpr int RTILE = 2 * TOK * 64;         // 16 KiB: residual of a tile pair, [tile][128 tokens][64 B]
constexpr int ROFF = NSTAGE * STAGE2;       // 120 KiB
constexpr int EXOFF = STAGE2;               // P exchange buffer: 8 waves x 10 KiB behind phase-2 stage 0
constexpr int MAXC = 1280;
constexpr int BIASOFF = ROFF + 2 * RTILE;   // 152 KiB
constexpr int C2OFF = RING;                 // V & 32: ln_c2 of the sample (2.5 KiB + DMA slack) and the token blocks' rstd (512 B) in
constexpr int RSOFF = C2OFF + 3072;         // the 8 KiB between the phase-1 ring and the bias: phase 2 reaches them (second residual
                                            // tile) only after the exchange barriers, when every wave is past the softmax
static_assert(RSOFF + TOK * 4 <= BIASOFF, "LDS budget (LayerNorm fold)");
constexpr int SMEM = BIASOFF + MAXC * 4;    // 157 KiB
static_assert(RING <= BIASOFF && EXOFF + 8 * 10240 <= ROFF && SMEM <= 160 * 1024, "LDS budget");
constexpr int PIECES = 6;                   // 1-KiB DMA pieces per wave and stage, both phases

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// V (round 3; 0 = the round-2 kernel, kept for A/B through SD_XATTN_VARIANT):
//   bit 0  phase 1: the last group of 5 MFMAs of a K tile (k-step 1 of key tiles 5-9) is issued after the NEXT barrier, from
//          fragments carried in registers, so that a wave has matrix work the moment it leaves the barrier -- until now all
//          eight waves first read 12 KiB of fragments each (~400 LDS cycles + latency) while the matrix pipe idled.  Same
//          MFMAs in the same per-accumulator order: bit-identical.
//   bit 1  phase 2: the same across the barrier between the two key halves of a tile pair (the kh = 1 stage ends in the
//          epilogue and carries nothing).
//   bit 3  phase 2: the stage of parity kh holds key half kh for tile 0 of the pair and key half kh ^ 1 for tile 1, so a wave
//          (tile hg) multiplies its OWN probabilities at kh = 0 and the exchanged ones at kh = 1: P is indexed statically --
//          80 v_cndmask per stage and wave (select P[i] / P[20 + i] by kh == hg) disappear.  The waves of tile 1 sum the key
//          halves in the other order (fp32 reassociation: not bit-identical with bit 3 clear).
//   bit 2  the to_out bias reaches LDS by LDS-DMA with the first stage instead of ordinary loads + a full wait in front
//          of the first DMA piece (the first operand tile used to land ~5 k cycles into the workgroup).
//   bit 5  norm2 folded in (XattnArgs::ln_rs): X is the un-normalised residual stream, the row's rstd comes from its producer's
//          partials (16 loads per lane in the prologue, in flight under the first two operand tiles) and the beta term of every
//          key slot from LDS; a score is rstd * raw + c2 -- one FMA per score in the softmax, and the LayerNorm launch with
//          its 2 x M x C x 2 bytes is gone.
template <int V>
__global__ __launch_bounds__(512, 2) void xattn_fused_kernel(const XattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb = wave & 3, hg = wave >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int C = p.C;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of token tiles,
    // i.e. whole samples, so that the workgroups that stream one sample's A^T / Bw share one L2.  Speed only.
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
    const int NPall = C / 64, nsl = gridDim.y;         // channel-tile PAIRS of this slice
    const int pbeg = (int)((long)NPall * blockIdx.y / nsl), NP = (int)((long)NPall * (blockIdx.y + 1) / nsl) - pbeg;

    // diagnostic phase stamps (only with a stamp buffer: tools/xattn_stamps.py); never read by the kernel
    auto stamp = [&](int i) {
        if (p.stamps && tid == 0) p.stamps[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    // to_out bias -> LDS.  V & 4: by LDS-DMA (C floats = C / 256 one-KiB pieces, waves 0 .. C / 256 - 1 one piece each; C is a
    // multiple of 64, a partial last piece re-reads the vector's tail -- clamped source, identical bytes land in the slack
    // behind sbias[C]); else ordinary loads, the kernel's only ones, retired before any DMA is in flight.
    float* sbias = (float*)(smem + BIASOFF);
    float* sc2 = (float*)(smem + C2OFF);
    float* srstd = (float*)(smem + RSOFF);
    f32x2_t lnp[16];
    if (V & 32) {       // this lane's token: row partials of the producer (consumed below, after the first DMA pieces are issued)
        const long lrow = (m0 + 32 * tb + r) % p.ln_rows;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            lnp[i] = *(const f32x2_t*)(p.ln_rs + ((long)(i < p.ln_np ? i : 0) * p.ln_rows + lrow) * 2);
    }
    if (V & 4) {
        if (wave * 256 < C) {
            const int i = min(wave * 256 + lane * 4, C - 4);
            glds16(p.bias + i, (char*)sbias + wave * 1024);
        }
    } else {
        for (int i = tid; i < C; i += 512) sbias[i] = p.bias[i];
    }
    if (V & 32) {       // ln_c2 of this sample: 640 floats = 2.5 one-KiB pieces (waves 0..2; the third clamps its source)
        if (wave < 3) {
            const int i = min(wave * 256 + lane * 4, KEYS - 4);
            glds16(p.ln_c2 + (long)sample * KEYS + i, (char*)sc2 + wave * 1024);
        }
    }

    // ---- LDS-DMA pieces (16 rows x 64 B; lane -> row lane >> 2, chunk lane & 3, swizzled on the source) ----
    // `real` = false: nothing left to fetch -- the piece still issues (constant DMA counts, no branch) but reads 16 hot bytes
    const int prow = lane >> 2;
    const int sch = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const char* dummy = (const char*)p.bias;
    // A^T and Bw are stored TILED (set_context / sd_launch_retile32): [K tile][640 rows][64 B] and [channel tile][32-slot
    // slice][32 rows][64 B], so an operand piece is one contiguous KiB = 8 full cache lines (16 half lines at the row
    // stride cost twice the address-unit time and twice the L2 requests)
    const char* aptr = At + (long)(16 * wave + prow) * 64 + sch;                         // A piece i (0..4): + i * 128 rows
    const char* xptr = (const char*)p.X + (m0 + 16 * wave + prow) * rowbytes + sch;      // piece 5: X rows 16 wave ..
    auto piece1 = [&](int kt, int i, char* st, bool real) {
        const char* src = i < 5 ? aptr + ((long)kt * KEYS + i * 128) * 64 : xptr + kt * 64;
        glds16(real ? src : dummy, st + (i < 5 ? wave + 8 * i : 40 + wave) * 1024);
    };
    // phase 2, stage s = 2 * pair + kh: Bw piece q = wave + 8 i (i < 5): tile q / 20 of the pair, sub-tile (q % 20) >> 1 of
    // the 10 sub-tiles (32 key slots = 64 B) of key half kh, rows 16 (q & 1) .. + 15; piece 5: residual rows
    // 16 (wave & 7).. of tile kh of the pair (so one pair's residual arrives with its two stages)
    const char* rptr = (const char*)p.R + (m0 + 16 * wave + prow) * rowbytes + sch;
    auto piece2 = [&](int pi, int kh, int i, char* st, char* rst, bool real) {
        const char* src;
        char* dst;
        if (i < 5) {
            const int q = wave + 8 * i, tile = q / 20, rem = q - tile * 20;
            const int khs = (V & 8) ? (kh ^ tile) : kh;       // key half staged for this tile
            src = Bw + ((long)((2 * pi + tile) * 20 + 10 * khs + (rem >> 1)) * 32 + 16 * (rem & 1) + prow) * 64 + sch;
            dst = st + q * 1024;
        } else {
            src = rptr + (2 * pi + kh) * 64;
            dst = rst + kh * (TOK * 64) + wave * 1024;
        }
        glds16(real ? src : dummy, dst);
    };

    // ---- fragment read offsets: row r of a 32-row tile, k-step ks: chunk (2 ks + h) ^ ((r >> 2) & 3) ----
    const int fsw = (r >> 2) & 3;
    const int fo0 = r * 64 + (((0 + h) ^ fsw) << 4);
    const int fo1 = r * 64 + (((2 + h) ^ fsw) << 4);
    const int xrow = (KEYS + 32 * tb) * 64;
    const int arow = hg * HKEYS * 64;
    const float c = 1.4426950408889634f;

    if (!(V & 4)) __builtin_amdgcn_s_waitcnt(0);          // bias loads + LDS writes retired
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece1(0, i, smem, true);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece1(1, i, smem + STAGE1, KT > 1);
    if (V & 32) {       // rstd of this lane's token -> LDS (no register held through phase 1); read back in the softmax
        float ls = 0.f, lq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ls += i < p.ln_np ? lnp[i][0] : 0.f;
            lq += i < p.ln_np ? lnp[i][1] : 0.f;
        }
        const float mean = ls / (float)C;
        srstd[32 * tb + r] = rsqrtf(fmaxf(lq / (float)C - mean * mean, 0.f) + p.ln_eps);      // (both lane halves: same value)
    }

    // =============================== phase 1: S^T = A^T . X^T (this wave's 4 heads) ===============================
    // V & 16: no zeroing of the 160 accumulators -- the first K tile's first k-step multiplies onto a constant-zero C operand
    f32x16 S[NKT];
    if (!(V & 16)) {
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
        }
    }
    {
        int st = 0;
        bf16x8 fa[5], fb[5], xf0, xf1;
        // V & 1: fragments of the deferred group (k-step 1 of key tiles 5-9) of the previous K tile; zeros before the first
        bf16x8 xfp = {0, 0, 0, 0, 0, 0, 0, 0};
        if (V & 1) {
#pragma unroll
            for (int t = 0; t < 5; ++t) fb[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        auto ktile = [&](int kt, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;        // K tile 0 of a V & 16 kernel
            wait_vmcnt<PIECES>();                       // own pieces of tile kt landed, tile kt+1 may be in flight
            __builtin_amdgcn_s_barrier();               // ... for every wave; all waves are past tile kt-1
            asm volatile("" ::: "memory");
            const char* sb = smem + st * STAGE1;
            char* refill = smem + (st == 0 ? 2 : st - 1) * STAGE1;      // stage of tile kt-1 = stage of tile kt+2
            const bool more = kt + 2 < KT;
            // Software pipeline in groups of 5 key tiles: the reads of group g+1 are in flight under the MFMAs of group g
            // (two 20-register fragment sets; 160 accumulators leave no room for whole k-steps).  Without it every wave of
            // the workgroup reads, then every wave multiplies: LDS and matrix pipe alternate instead of overlapping
            // (measured: the K loop took 0.75 of its time with the MFMAs removed).
            auto rd = [&](bf16x8* f, int grp) {         // grp = 2 ks + (tile half)
                const int fo = (grp & 2) ? fo1 : fo0;
#pragma unroll
                for (int t = 0; t < 5; ++t) f[t] = *(const bf16x8*)(sb + arow + (5 * (grp & 1) + t) * 2048 + fo);
            };
            auto mm = [&](const bf16x8* f, const bf16x8& xf, int grp, bool dma) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int tt = 5 * (grp & 1) + t;
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    S[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[t], xf, FIRST && grp < 2 ? zero : S[tt], 0, 0, 0);
                    // one DMA piece per MFMA gap (groups 0 and 1: pieces 0..2 and 3..5)
                    if (dma && grp < 2 && t >= 1 && t < 4) piece1(kt + 2, 3 * grp + t - 1, refill, more);
                }
            };
            xf0 = *(const bf16x8*)(sb + xrow + fo0);
            xf1 = *(const bf16x8*)(sb + xrow + fo1);
            rd(fa, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((V & 1) && !FIRST) {
                // the previous tile's last group first: its operands are in registers, the matrix pipe starts at once
                mm(fb, xfp, 3, false);
                __builtin_amdgcn_sched_barrier(0);
            }
            rd(fb, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa, xf0, 0, true);
            __builtin_amdgcn_sched_barrier(0);
            rd(fa, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(fb, xf0, 1, true);
            __builtin_amdgcn_sched_barrier(0);
            rd(fb, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa, xf1, 2, true);
            __builtin_amdgcn_sched_barrier(0);
            if (V & 1) {
                // group 3 is multiplied after the next barrier: its fragments must have left the LDS before this wave
                // arrives there (the stage is refilled behind it).  The builtin, so that hipcc's wait bookkeeping knows.
                xfp = xf1;
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
            } else {
                mm(fb, xf1, 3, true);
            }
            __builtin_amdgcn_sched_barrier(0);
            st = st == 2 ? 0 : st + 1;
        };
        if (V & 16) ktile(0, std::true_type{});
        for (int kt = (V & 16) ? 1 : 0; kt < KT; ++kt) ktile(kt, std::false_type{});
        if (V & 1) {
#pragma unroll
            for (int t = 0; t < 5; ++t) S[5 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[t], xfp, S[5 + t], 0, 0, 0);
        }
    }
    stamp(1);
    wait_vmcnt<0>();                                    // (trailing dummy pieces)
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the phase-1 stages
    asm volatile("" ::: "memory");
    char* const rring = smem + ROFF;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece2(pbeg, 0, i, smem, rring, true);          // stage 0 streams in under the softmax

    // =============================== softmax over each head's 80 key slots ===============================
    // lane (token r, half h) holds key 32 t + (i & 3) + 8 (i >> 2) + 4 h in S[t][i]; head hd = half-tiles 5 hd .. 5 hd + 4
    // (a half-tile = registers 8 s .. 8 s + 7 of tile t, u = 2 t + s); key slots >= L of a head sit in its last half-tile.
    bf16x8 P[4 * NKT];              // P^T fragments of all 40 half-tiles (16 key slots each); own half first
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        float v[40];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int u = 5 * hd + q;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[q * 8 + j] = S[u >> 1][(u & 1) * 8 + j];
        }
        if (V & 32) {   // score = rstd(token) * raw + c2(key slot); slot of v[8 q + j] = 16 q + 8 (j >> 2) + 4 h + (j & 3)
            const float rstd = srstd[32 * tb + r];
            const float* c2h = sc2 + (4 * hg + hd) * 80 + 4 * h;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const f32x4 cv = *(const f32x4*)(c2h + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * q + j] = __builtin_fmaf(v[4 * q + j], rstd, cv[j]);
            }
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
        if (V & 16) {
            // packed f32 pairs (v_pk_fma / v_pk_add / v_pk_mul): half the VALU issue slots of the scalar chain, and four
            // independent partial sums instead of one 40-deep dependent add chain
            f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
            const f32x2 cc = {c, c}, mm2 = {-mc, -mc};
#pragma unroll
            for (int i = 0; i < 40; i += 4) {
                const f32x2 a = __builtin_elementwise_fma(f32x2{v[i], v[i + 1]}, cc, mm2);
                const f32x2 b = __builtin_elementwise_fma(f32x2{v[i + 2], v[i + 3]}, cc, mm2);
                const f32x2 ea = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                const f32x2 eb = {__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
                s0 += ea;
                s1 += eb;
                v[i] = ea[0], v[i + 1] = ea[1], v[i + 2] = eb[0], v[i + 3] = eb[1];
            }
            s0 += s1;
            // the horizontal add as a SCALAR add of two opaque halves: left to hipcc it becomes `v_pk_add_f32 v, v, v op_sel:[0,1]
            // op_sel_hi:[1,0]` -- the packed form with a lo-lane op_sel that MI355X mis-executes beside the SIMD partner's MFMAs
            // (asm_lint.py PK_OPSEL; profiles/round5_notes.md)
            float s_lo = s0[0], s_hi = s0[1];
            asm volatile("" : "+v"(s_lo), "+v"(s_hi));
            sum = s_lo + s_hi;
        } else {
#pragma unroll
            for (int i = 0; i < 40; ++i) {
                v[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], c, -mc));
                sum += v[i];
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        if (V & 16) {
            const f32x2 ii = {inv, inv};
#pragma unroll
            for (int i = 0; i < 40; i += 2) {
                const f32x2 t = f32x2{v[i], v[i + 1]} * ii;
                v[i] = t[0], v[i + 1] = t[1];
            }
        }
        const float inv1 = (V & 16) ? 1.f : inv;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const u32x4 pk = {pack2bf(v[q * 8 + 0] * inv1, v[q * 8 + 1] * inv1), pack2bf(v[q * 8 + 2] * inv1, v[q * 8 + 3] * inv1),
                              pack2bf(v[q * 8 + 4] * inv1, v[q * 8 + 5] * inv1), pack2bf(v[q * 8 + 6] * inv1, v[q * 8 + 7] * inv1)};
            P[5 * hd + q] = __builtin_bit_cast(bf16x8, pk);     // local half-tile index; global = 20 hg + local
        }
    }
    stamp(2);
    // ---- exchange: the partner wave (same tokens, other head group) needs this half, and this wave needs the partner's ----
    {
        char* mine = smem + EXOFF + wave * 10240 + lane * 16;
        const char* theirs = smem + EXOFF + (wave ^ 4) * 10240 + lane * 16;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
            for (int f = 0; f < 10; ++f) *(bf16x8*)(mine + f * 1024) = P[10 * rnd + f];
            // raw barrier + lgkmcnt only: __syncthreads() would also wait vmcnt(0), i.e. for the stage-0 DMA in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int f = 0; f < 10; ++f) P[20 + 10 * rnd + f] = *(const bf16x8*)(theirs + f * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // P[0..19] = own head group (global half-tiles 20 hg ..), P[20..39] = the other one (20 (1 - hg) ..)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) piece2(pbeg, 1, i, smem + STAGE2, rring, true);
    stamp(3);

    // =============================== phase 2: Y^T = Bw . P^T, one 32-channel tile of every pair ===============================
    const long trow = m0 + 32 * tb + r;                         // this lane's token
    char* Yrow = (char*)p.Y + trow * rowbytes;
    // residual tile in LDS: [tile hg][row 32 tb + r][16-byte chunk g (channels 8 g .. 8 g + 7) at position g ^ swizzle] + 8 h
    const int roff = hg * (TOK * 64) + (32 * tb + r) * 64 + 8 * h;
    const int brow = hg * (32 * HKEYS * 2);                     // this wave's tile inside a stage
    const int NS = 2 * NP;
    f32x16 acc;
    bf16x8 fa2[5], fb2[5];            // (declared outside the stage loop: V & 2 carries fb2 from a kh = 0 stage into the next)
    float ysum = 0.f, ysq = 0.f;      // LayerNorm partials of this lane's token over its channels (p.rowstats)
    int st = 0;
    // a stage = (tile pair, key half kh); kh is a compile-time constant of the body (the loop below runs the two stages of a
    // pair back to back), so the carried group of V & 2 and the epilogue are straight-line code
    auto stage = [&](const int s, auto kh_tag) {
        constexpr int kh = decltype(kh_tag)::value;
        const int pi = pbeg + (s >> 1);
        // Own DMA of stage s is older than stage s+1's six pieces and, after an epilogue, its two stores.
        if (kh == 0 && s > 0) wait_vmcnt<PIECES + 2>();
        else wait_vmcnt<PIECES>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sb = smem + st * STAGE2 + brow;
        char* refill = smem + (st == 0 ? 2 : st - 1) * STAGE2;
        const bool more = s + 2 < NS;
        const int pi2 = pbeg + ((s + 2) >> 1);
        char* rst = rring + (((s + 2) >> 1) & 1) * RTILE;
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
        {
            auto rd = [&](bf16x8* f, int grp) {         // 4 groups of 5 k-steps (16 key slots each)
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int i = 5 * grp + t;
                    f[t] = *(const bf16x8*)(sb + (i >> 1) * 2048 + ((i & 1) ? fo1 : fo0));
                }
            };
            auto mm = [&](const bf16x8* f, int grp, int khm, bool dma) {
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int i = 5 * grp + t;
                    // global half-tile 20 khm + i: this wave's own half when khm == hg; V & 8: own half at parity 0 (static)
                    const bf16x8 pf = (V & 8) ? P[20 * khm + i] : (khm == hg ? P[i] : P[20 + i]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[t], pf, acc, 0, 0, 0);
                    if (dma && grp < 2 && t >= 1 && t < 4) piece2(pi2, kh, 3 * grp + t - 1, refill, rst, more);
                }
            };
            rd(fa2, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((V & 2) && kh == 1) {                   // the kh = 0 stage's last group, carried across the barrier in fb2
                mm(fb2, 3, 0, false);
                __builtin_amdgcn_sched_barrier(0);
            }
            rd(fb2, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa2, 0, kh, true);
            __builtin_amdgcn_sched_barrier(0);
            rd(fa2, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(fb2, 1, kh, true);
            __builtin_amdgcn_sched_barrier(0);
            rd(fb2, 3);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa2, 2, kh, true);
            __builtin_amdgcn_sched_barrier(0);
            if ((V & 2) && kh == 0) {
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): fb2's reads have left the stage before the next barrier
            } else {
                mm(fb2, 3, kh, true);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kh == 1) {
            // ---- epilogue (LDS operands only): lane holds channels 32 j + 8 g + 4 h + (0..3), g = 0..3, of its token ----
            const int j = 2 * pi + hg;
            const char* rt = rring + ((s >> 1) & 1) * RTILE + roff;
            const int cb = 32 * j + 4 * h;
            u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x2 rr = *(const u32x2*)(rt + ((g ^ fsw) << 4));
                const f32x4 bv = *(const f32x4*)(sbias + cb + 8 * g);
                const float y0 = acc[4 * g + 0] + bv[0] + bflo(rr[0]), y1 = acc[4 * g + 1] + bv[1] + bfhi(rr[0]);
                const float y2 = acc[4 * g + 2] + bv[2] + bflo(rr[1]), y3 = acc[4 * g + 3] + bv[3] + bfhi(rr[1]);
                pk[g] = u32x2{pack2bf(y0, y1), pack2bf(y2, y3)};
                if (p.rowstats) {
                    const float v0 = bflo(pk[g][0]), v1 = bfhi(pk[g][0]), v2 = bflo(pk[g][1]), v3 = bfhi(pk[g][1]);
                    ysum += (v0 + v1) + (v2 + v3);
                    ysq = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, ysq))));
                }
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
        st = st == 2 ? 0 : st + 1;
    };
    for (int pr = 0; pr < NP; ++pr) {
        stage(2 * pr, std::integral_constant<int, 0>{});
        stage(2 * pr + 1, std::integral_constant<int, 1>{});
    }
    if (p.rowstats) {       // partial (channel slice, tile parity hg): the two lane halves hold channels 4 h + (0..3) of every 8
        ysum += __shfl_xor(ysum, 32);
        ysq += __shfl_xor(ysq, 32);
        if (h == 0) *(f32x2_t*)(p.rowstats + (((long)blockIdx.y * 2 + hg) * p.M + trow) * 2) = f32x2_t{ysum, ysq};
    }
    stamp(5);
}

}  // namespace

bool sd_xattn_fused_applicable(int rows_per_sample, int C, int heads, int L) {
    static const char* env = getenv("SD_XATTN_FUSED");
    if (env && atoi(env) == 0) return false;
    return heads == 8 && L > 64 && L <= 80 && C % 64 == 0 && C >= 128 && C <= MAXC && rows_per_sample % TOK == 0;
}

// channel slices (each repeats phase 1): only when the token grid leaves most of the 256 CUs idle
int sd_xattn_slices(int M, int C) {
    static const int force = getenv("SD_XATTN_SLICES") ? atoi(getenv("SD_XATTN_SLICES")) : 0;
    const int wgs = M / TOK, np = C / 64;
    int nsl = 1;
    if (force > 0) nsl = force;
    else if (wgs <= 128) nsl = 2;
    return nsl > np ? np : nsl;
}

int sd_launch_xattn_fused(const XattnArgs& a, hipStream_t stream) {
    SD_REQUIRE(a.X && a.R && a.Y && a.At && a.Bw && a.bias, "xattn: null operand");
    SD_REQUIRE(a.C % 64 == 0 && a.C >= 128 && a.C <= MAXC, "xattn: C=%d must be a multiple of 64 in [128, %d]", a.C, MAXC);
    SD_REQUIRE(a.rows_per_sample % TOK == 0 && a.M % a.rows_per_sample == 0 && a.M > 0,
               "xattn: %d tokens per sample must be a multiple of %d (M=%d)", a.rows_per_sample, TOK, a.M);
    SD_REQUIRE(a.L > 64 && a.L <= 80, "xattn: %d prompt keys (65..80 are built)", a.L);
    SD_REQUIRE((long)KEYS * a.C * 2 * (a.M / a.rows_per_sample) < (1l << 40), "xattn: operand too large");
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel<15>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel<31>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)xattn_fused_kernel<63>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    // SD_XATTN_VARIANT=0: the round-2 kernel (A/B); 15 = the first four round-3 changes; default 31 = all round-3 changes (measured one by one on one box, 64x64
    // launch of the bench: 72.2 us -> bias by DMA 70.9 -> + phase-1 carried group 69.9 -> + phase 2 68.5 -> + static P 68.0)
    static const int variant = getenv("SD_XATTN_VARIANT") ? atoi(getenv("SD_XATTN_VARIANT")) : 31;
    const int wgs = a.M / TOK, nsl = sd_xattn_slices(a.M, a.C);
    if (a.ln_rs) {      // norm2 folded in: the product variant only
        SD_REQUIRE(a.ln_c2 && a.ln_np >= 1 && a.ln_np <= 16 && a.ln_rows > 0 && a.M % a.ln_rows == 0 && a.ln_eps > 0.f && !a.stamps,
                   "xattn with the LayerNorm fold: 1..16 row partials, M a multiple of the partials' rows, c2");
        hipLaunchKernelGGL(xattn_fused_kernel<63>, dim3(wgs, nsl), dim3(512), SMEM, stream, a);
        SD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 0) hipLaunchKernelGGL(xattn_fused_kernel<0>, dim3(wgs, nsl), dim3(512), SMEM, stream, a);
    else if (variant == 15) hipLaunchKernelGGL(xattn_fused_kernel<15>, dim3(wgs, nsl), dim3(512), SMEM, stream, a);
    else hipLaunchKernelGGL(xattn_fused_kernel<31>, dim3(wgs, nsl), dim3(512), SMEM, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
