// GroupNorm(+SiLU) over NHWC activations and LayerNorm over token rows, gfx950.
// Both are HBM-bound: every access is a 16-byte (8 x bf16) vector, rows are contiguous, and a
// thread keeps a FIXED 8-channel chunk so its per-channel scale/shift live in registers.
// GroupNorm is two launches: (1) deterministic partial sums per (batch, pixel-split, group),
// (2) finalise + normalise (+SiLU).  A two-tensor channel concat is read in place and written
// as one contiguous tensor (the only place the up-path concat is ever materialised).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace {

struct GnGeom {
    int nchunks, rows_par, threads, cpg, C;
};

__host__ __device__ inline GnGeom gn_geom(int C, int groups) {
    GnGeom g;
    g.C = C;
    g.nchunks = C / 8;
    g.rows_par = g.nchunks >= 256 ? 1 : 256 / g.nchunks;
    const int act = g.rows_par * g.nchunks;
    g.threads = (act + 63) / 64 * 64;
    g.cpg = C / groups;
    return g;
}

__device__ __forceinline__ const bf16_t* gn_src(const GroupNormArgs& a, long pix, int c0) {
    return c0 < a.C1 ? a.x1 + pix * a.C1 + c0 : a.x2 + pix * a.C2 + (c0 - a.C1);
}

__global__ void gn_stats_kernel(const GroupNormArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sh = (float4*)smem;
    const GnGeom g = gn_geom(a.C1 + a.C2, a.groups);
    const int tid = threadIdx.x;
    const int split = blockIdx.x, b = blockIdx.y;
    const bool active = tid < g.rows_par * g.nchunks;
    const int prow = tid / g.nchunks, ch = tid - prow * g.nchunks;
    const int c0 = ch * 8;
    const int g0 = c0 / g.cpg;
    const int nb = min(8, (g0 + 1) * g.cpg - c0);  // elements of this chunk that belong to g0
    const int per = (a.HW + a.nsplit - 1) / a.nsplit;
    const int pbeg = split * per, pend = min(a.HW, pbeg + per);
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    if (active) {
        auto accum = [&](const u32x4 v) {
            float f[8] = {bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1]),
                          bflo(v[2]), bfhi(v[2]), bflo(v[3]), bfhi(v[3])};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < nb) { s0 += f[j]; q0 += f[j] * f[j]; }
                else        { s1 += f[j]; q1 += f[j] * f[j]; }
            }
        };
        const long base = (long)b * a.HW;
        const int rp = g.rows_par;
        int p = pbeg + prow;
        for (; p + 3 * rp < pend; p += 4 * rp) {   // 4 independent 16-B loads in flight per lane
            const u32x4 v0 = *(const u32x4*)gn_src(a, base + p, c0);
            const u32x4 v1 = *(const u32x4*)gn_src(a, base + p + rp, c0);
            const u32x4 v2 = *(const u32x4*)gn_src(a, base + p + 2 * rp, c0);
            const u32x4 v3 = *(const u32x4*)gn_src(a, base + p + 3 * rp, c0);
            accum(v0); accum(v1); accum(v2); accum(v3);
        }
        for (; p < pend; p += rp) accum(*(const u32x4*)gn_src(a, base + p, c0));
    }
    sh[tid] = make_float4(s0, q0, s1, q1);
    __syncthreads();
    // deterministic two-level reduction: over the pixel rows of each chunk, then chunks -> groups
    float4* sh2 = sh + blockDim.x;
    if (tid < g.nchunks) {
        float4 acc = sh[tid];
        for (int r = 1; r < g.rows_par; ++r) {
            const float4 v = sh[r * g.nchunks + tid];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        sh2[tid] = acc;
    }
    __syncthreads();
    if (tid < a.groups) {
        const int c_lo = (tid * g.cpg) / 8, c_hi = ((tid + 1) * g.cpg - 1) / 8;
        float s = 0.f, q = 0.f;
        for (int c = c_lo; c <= c_hi; ++c) {
            const float4 v = sh2[c];
            if ((c * 8) / g.cpg == tid) { s += v.x; q += v.y; }   // chunk's leading part is ours
            else                        { s += v.z; q += v.w; }   // chunk straddles into our group
        }
        float* out = a.partial + (((long)b * a.nsplit + split) * a.groups + tid) * 2;
        out[0] = s;
        out[1] = q;
    }
}

// one wave per (batch, group): sum the pixel-split partials -> mean, rstd
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GroupNormArgs a) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= a.B * a.groups) return;
    const int b = idx / a.groups, gi = idx - b * a.groups;
    float s = 0.f, q = 0.f;
    for (int i = lane; i < a.nsplit; i += 64) {
        const float* pp = a.partial + (((long)b * a.nsplit + i) * a.groups + gi) * 2;
        s += pp[0];
        q += pp[1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
        const float cnt = (float)a.HW * (float)((a.C1 + a.C2) / a.groups);
        const float mean = s / cnt;
        const float var = fmaxf(q / cnt - mean * mean, 0.f);
        float* st = a.partial + (long)a.B * a.nsplit * a.groups * 2 + (long)idx * 2;
        st[0] = mean;
        st[1] = rsqrtf(var + a.eps);
    }
}

// Statistics delivered by the producers (GemmArgs::stats): per 64-row block and channel (sum, sum of squares).  One
// wave per (batch, group) sums its channels over the blocks of its sample in a fixed order -> mean, rstd, written where
// gn_apply_kernel reads them.  Replaces gn_stats_kernel + gn_finalize_kernel (no pass over the tensor).
__global__ __launch_bounds__(256) void gn_finalize_stats_kernel(const GroupNormArgs a) {
    // one 256-thread block per (batch, group): the (64-row block, channel) pairs of the group are strided over the
    // threads, then a fixed-shape reduction (wave, then the four waves in order): deterministic
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x;
    const int b = idx / a.groups, gi = idx - b * a.groups;
    const int C = a.C1 + a.C2, cpg = C / a.groups, nblk = a.HW >> 6;
    const int total = nblk * cpg;                       // (block, channel) pairs of this group
    float s = 0.f, q = 0.f;
    for (int i = tid; i < total; i += 256) {
        const int blk = i / cpg, c = gi * cpg + (i - blk * cpg);
        const long gb = (long)b * nblk + blk;
        const f32x2_t v = *(const f32x2_t*)(c < a.C1 ? a.stats1 + (gb * a.C1 + c) * 2 : a.stats2 + (gb * a.C2 + (c - a.C1)) * 2);
        s += v[0];
        q += v[1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        s = (red[0] + red[2]) + (red[4] + red[6]);
        q = (red[1] + red[3]) + (red[5] + red[7]);
        const float cnt = (float)a.HW * (float)cpg;
        const float mean = s / cnt;
        const float var = fmaxf(q / cnt - mean * mean, 0.f);
        float* st = a.partial + (long)a.B * a.nsplit * a.groups * 2 + (long)idx * 2;
        st[0] = mean;
        st[1] = rsqrtf(var + a.eps);
    }
}

// SILU / FP8 are template parameters: as run-time flags hipcc kept a branch per ELEMENT in the loop (eight blocks of 18
// instructions per 16-byte vector)
template <bool SILU, bool FP8>
__global__ void gn_apply_kernel(const GroupNormArgs a) {
    const GnGeom g = gn_geom(a.C1 + a.C2, a.groups);
    const int tid = threadIdx.x;
    const int split = blockIdx.x, b = blockIdx.y;
    const bool active = tid < g.rows_par * g.nchunks;
    if (!active) return;
    const int prow = tid / g.nchunks, ch = tid - prow * g.nchunks;
    const int c0 = ch * 8;
    const float* st = a.partial + (long)a.B * a.nsplit * a.groups * 2 + (long)b * a.groups * 2;
    const int g0 = c0 / g.cpg, g1 = (c0 + 7) / g.cpg;
    const int nb = min(8, (g0 + 1) * g.cpg - c0);
    const float m0 = st[g0 * 2], r0 = st[g0 * 2 + 1], m1 = st[g1 * 2], r1 = st[g1 * 2 + 1];
    float sc[8], sf[8];
    {
        const f32x4 ga = *(const f32x4*)(a.gamma + c0), gb = *(const f32x4*)(a.gamma + c0 + 4);
        const f32x4 ba = *(const f32x4*)(a.beta + c0), bb = *(const f32x4*)(a.beta + c0 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gm = j < 4 ? ga[j & 3] : gb[j & 3], bt = j < 4 ? ba[j & 3] : bb[j & 3];
            const float mean = j < nb ? m0 : m1, rstd = j < nb ? r0 : r1;
            sc[j] = rstd * gm;
            sf[j] = bt - mean * sc[j];
        }
    }
    const int per = (a.HW + a.nsplit - 1) / a.nsplit;
    const int pbeg = split * per, pend = min(a.HW, pbeg + per);
    auto apply = [&](const u32x4 v, long pix) {
        float f[8] = {bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1]),
                      bflo(v[2]), bfhi(v[2]), bflo(v[3]), bfhi(v[3])};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f[j] = f[j] * sc[j] + sf[j];
            if (SILU) f[j] = silu_f(f[j]);
        }
        if (FP8) {       // e4m3 bytes, rows of Cpad bytes; the owner of the last chunk zeroes the K-tail padding
            char* yr = (char*)a.y + pix * a.Cpad;
            const u32x2 o = {pack4fp8(f[0] * a.oscale, f[1] * a.oscale, f[2] * a.oscale, f[3] * a.oscale),
                             pack4fp8(f[4] * a.oscale, f[5] * a.oscale, f[6] * a.oscale, f[7] * a.oscale)};
            *(u32x2*)(yr + c0) = o;
            if (ch == g.nchunks - 1)
                for (int k = g.C; k < a.Cpad; k += 16) *(u32x4*)(yr + k) = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        u32x4 o = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
        *(u32x4*)(a.y + pix * g.C + c0) = o;
    };
    const long base = (long)b * a.HW;
    const int rp = g.rows_par;
    int p = pbeg + prow;
    for (; p + 3 * rp < pend; p += 4 * rp) {
        const u32x4 v0 = *(const u32x4*)gn_src(a, base + p, c0);
        const u32x4 v1 = *(const u32x4*)gn_src(a, base + p + rp, c0);
        const u32x4 v2 = *(const u32x4*)gn_src(a, base + p + 2 * rp, c0);
        const u32x4 v3 = *(const u32x4*)gn_src(a, base + p + 3 * rp, c0);
        apply(v0, base + p); apply(v1, base + p + rp); apply(v2, base + p + 2 * rp); apply(v3, base + p + 3 * rp);
    }
    for (; p < pend; p += rp) apply(*(const u32x4*)gn_src(a, base + p, c0), base + p);
}

// Single-launch GroupNorm(+SiLU) for small images (a group of <= 10240 values: the 8x8 level, and the 16x16
// level up to 1280 channels), one workgroup per
// (batch item, group).  The three-launch path above costs ~3 x 5 us of launch floor there for a few hundred KB of
// data; here the group's HW x cpg values (<= 10 KB) are read twice by the same workgroup (the second time from
// L1/L2), with a block reduction in between (measured 18 -> 8 us at 8x8, 20 -> 16 us at 16x16 x 1280; wider 16x16 tensors stay on the split path).  Accesses are 8 bytes (4 channels): cpg is a multiple of 4 at these
// levels (20 / 40 / 60 / 80) but not of 8.
// MAXU: units of 4 channels per thread (3 / 5 / 10: ceil(HW * cpg / 4 / 256) rounded up; 10 = 10 240 values per group).
// The per-unit arithmetic lives in three helpers shared with gn_slab_kernel below, which reproduces this kernel's statistics
// and outputs bit for bit.
__device__ __forceinline__ float2 gn_unit_sums(const u32x2 v) {         // (sum, sum of squares) of a unit's four bf16 values
    const float f0 = bflo(v[0]), f1 = bfhi(v[0]), f2 = bflo(v[1]), f3 = bfhi(v[1]);
    return make_float2((f0 + f1) + (f2 + f3), (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3));
}
__device__ __forceinline__ void gn_mean_rstd(float ts, float tq, float cnt, float eps, float& mean, float& rstd) {
    mean = ts / cnt;
    rstd = rsqrtf(fmaxf(tq / cnt - mean * mean, 0.f) + eps);
}
template <bool SILU, bool FP8>
__device__ __forceinline__ void gn_unit_store(const GroupNormArgs& a, const u32x2 v, float mean, float rstd, long pix, int c, int C) {
    const f32x4 gm = *(const f32x4*)(a.gamma + c), bt = *(const f32x4*)(a.beta + c);
    float f[4] = {bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1])};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        f[jj] = (f[jj] - mean) * rstd * gm[jj] + bt[jj];
        if (SILU) f[jj] = silu_f(f[jj]);
    }
    if (FP8) {
        *(unsigned*)((char*)a.y + pix * a.Cpad + c) = pack4fp8(f[0] * a.oscale, f[1] * a.oscale, f[2] * a.oscale, f[3] * a.oscale);
    } else {
        u32x2 o = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3])};
        *(u32x2*)(a.y + pix * C + c) = o;
    }
}

template <bool SILU, bool FP8, int MAXU>
__global__ __launch_bounds__(256) void gn_small_kernel(const GroupNormArgs a) {
    __shared__ float2 red[4];
    const int C = a.C1 + a.C2, cpg = C / a.groups, upp = cpg >> 2;      // 4-channel units per pixel
    const int grp = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = a.HW * upp;
    const long base = (long)b * a.HW;
    auto src = [&](int p, int c) -> const bf16_t* {
        return c < a.C1 ? a.x1 + (base + p) * a.C1 + c : a.x2 + (base + p) * a.C2 + (c - a.C1);
    };
    // unit i = tid + 256 j  <->  (pixel p, unit k): stepped incrementally, no division in the loops.  The group's values
    // (<= MAXU units of 4 channels per thread) are loaded ONCE, all loads in flight together, and stay in registers.
    const int p0 = tid / upp, k0 = tid - p0 * upp, dp = 256 / upp, dk = 256 - dp * upp;
    u32x2 vals[MAXU];
    int pj[MAXU], kj[MAXU];
    {
        int p = p0, k = k0;
#pragma unroll
        for (int j = 0; j < MAXU; ++j) {
            const bool live = tid + 256 * j < total;
            pj[j] = live ? p : 0; kj[j] = live ? k : 0;                  // dead units: unconditional loads from a valid address
            vals[j] = *(const u32x2*)src(pj[j], grp * cpg + kj[j] * 4);
            p += dp; k += dk;
            if (k >= upp) { k -= upp; ++p; }
        }
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
        const bool live = tid + 256 * j < total;
        const float2 u = gn_unit_sums(vals[j]);
        s += live ? u.x : 0.f;
        q += live ? u.y : 0.f;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) red[wave] = make_float2(s, q);
    __syncthreads();
    const float ts = (red[0].x + red[1].x) + (red[2].x + red[3].x);
    const float tq = (red[0].y + red[1].y) + (red[2].y + red[3].y);
    float mean, rstd;
    gn_mean_rstd(ts, tq, (float)a.HW * (float)cpg, a.eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < MAXU; ++j) {
        if (tid + 256 * j >= total) break;
        gn_unit_store<SILU, FP8>(a, vals[j], mean, rstd, base + pj[j], grp * cpg + kj[j] * 4, C);
    }
    if (FP8 && grp == a.groups - 1)       // K-tail padding of every pixel row
        for (int i = tid; i < a.HW * ((a.Cpad - C) >> 4); i += 256) {
            const int pc = (a.Cpad - C) >> 4, p = i / pc, k = i - p * pc;
            *(u32x4*)((char*)a.y + (base + p) * a.Cpad + C + k * 16) = u32x4{0u, 0u, 0u, 0u};
        }
}

// The same GroupNorm when its first source has NOT been written yet: the producer ran split-K and deferred the reduce
// (GroupNormArgs::slab).  A unit's four channels are  bf16(slab[0] + slab[1] + ... + bias + bias2 + R)  in splitk_reduce_kernel's
// order of additions, stored to x1w on the way; the second source of a channel concat stays bf16.  The slabs are 16x the bytes
// of the bf16 tensor, so the layout follows THEM: a workgroup owns G = max(1, 80 / cpg) whole groups of one sample -- 80 (60)
// consecutive channels = 320 (240) contiguous bytes of every slab row -- as 32 pixel rows x ncol unit columns of threads; a
// thread keeps ONE column (its gamma / beta / group are fixed) and MAXI pixels, with MAXI x UNR sixteen-byte loads in flight
// per trip.  (One 256-thread workgroup per group, as gn_small_kernel, read 160-byte pieces of the slab rows: slower than the
// two launches it replaced.)  Statistics: every unit's (sum, sum of squares) goes through the LDS and is added up by "virtual
// threads" in gn_small_kernel's exact order (unit i of a group -> thread i % 256, ascending i; wave butterfly; four waves),
// so the two paths agree bit for bit.
template <bool SILU, int MAXI, int UNR>
__global__ __launch_bounds__(640) void gn_slab_kernel(const GroupNormArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C1 + a.C2, cpg = C / a.groups, upp = cpg >> 2;
    const int G = cpg >= 80 ? 1 : 80 / cpg, ncol = G * upp;              // groups / unit columns of this workgroup
    const int b = blockIdx.y, grp0 = blockIdx.x * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = a.HW * upp;                                        // units of one group
    float2* su = (float2*)smem;                                          // [G][total]
    float2* red = su + G * total;                                        // [G][4]
    const long base = (long)b * a.HW;
    const int row = tid / ncol, col = tid - row * ncol;
    const bool active = row < 32;
    const int gl = col / upp, kk = col - gl * upp;                       // this thread's group (local) and unit inside it
    const int c = grp0 * cpg + col * 4;
    const bool from_slab = c < a.C1;
    u32x2 vals[MAXI];
    if (active) {
        if (from_slab) {
            const long MN = (long)a.B * a.HW * a.C1;
            long off[MAXI];
            f32x4 acc[MAXI];
#pragma unroll
            for (int it = 0; it < MAXI; ++it) {
                const int p = row + 32 * it;
                off[it] = (base + (p < a.HW ? p : 0)) * a.C1 + c;        // dead pixels: unconditional loads from a valid address
                acc[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int s0 = 0; s0 < a.splitk; s0 += UNR) {                 // (splitk % UNR == 0: checked by the launcher)
                f32x4 t[MAXI][UNR];
#pragma unroll
                for (int it = 0; it < MAXI; ++it)
#pragma unroll
                    for (int uu = 0; uu < UNR; ++uu) t[it][uu] = *(const f32x4*)(a.slab + (s0 + uu) * MN + off[it]);
#pragma unroll
                for (int it = 0; it < MAXI; ++it)
#pragma unroll
                    for (int uu = 0; uu < UNR; ++uu) acc[it] = (s0 + uu == 0) ? t[it][uu] : acc[it] + t[it][uu];
            }
            f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
            const bool hb = a.sbias != nullptr, hb2 = a.sbias2 != nullptr;
            const f32x4 b1 = hb ? *(const f32x4*)(a.sbias + c) : bsum, b2 = hb2 ? *(const f32x4*)(a.sbias2 + c) : bsum;
#pragma unroll
            for (int it = 0; it < MAXI; ++it) {
                const int p = row + 32 * it;
                f32x4 v = acc[it];
                if (hb) v += b1;
                if (hb2) v += b2;
                if (a.sR) {
                    const u32x2 r = *(const u32x2*)(a.sR + (base + (p < a.HW ? p : 0)) * a.sldr + c);
                    v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
                }
                vals[it] = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                if (p < a.HW) *(u32x2*)(a.x1w + off[it]) = vals[it];
            }
        } else {
#pragma unroll
            for (int it = 0; it < MAXI; ++it) {
                const int p = row + 32 * it;
                vals[it] = *(const u32x2*)(a.x2 + (base + (p < a.HW ? p : 0)) * a.C2 + (c - a.C1));
            }
        }
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int p = row + 32 * it;
            if (p < a.HW) su[gl * total + p * upp + kk] = gn_unit_sums(vals[it]);
        }
    }
    __syncthreads();
    // gn_small_kernel's reduction, replayed: four waves per group, virtual thread v = 64 (wave & 3) + lane adds units v, v + 256, ...
    for (int g0 = 0; g0 < G; g0 += 2) {
        const int g = g0 + (wave >> 2);
        if (wave < 8 && g < G) {
            const int v = (wave & 3) * 64 + lane;
            float s = 0.f, q = 0.f;
            for (int i = v; i < total; i += 256) {
                const float2 u = su[g * total + i];
                s += u.x;
                q += u.y;
            }
            s = wave_sum(s);
            q = wave_sum(q);
            if (lane == 0) red[g * 4 + (wave & 3)] = make_float2(s, q);
        }
    }
    __syncthreads();
    if (!active) return;
    const float2 r0 = red[gl * 4], r1 = red[gl * 4 + 1], r2 = red[gl * 4 + 2], r3 = red[gl * 4 + 3];
    float mean, rstd;
    gn_mean_rstd((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (float)a.HW * (float)cpg, a.eps, mean, rstd);
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int p = row + 32 * it;
        if (p < a.HW) gn_unit_store<SILU, false>(a, vals[it], mean, rstd, base + p, c, C);
    }
}

// one wave per token row; up to 3 chunks of 8 channels per lane (C <= 1536)
// FP8: y holds e4m3 bytes of sat(out * oscale) in rows of Cpad bytes (the pad is zeroed)
template <bool FP8>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                        int rows, int C, float eps, int Cpad, float oscale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = C >> 3;
    const bf16_t* xr = x + (long)row * C;
    float f[3][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const u32x4 v = *(const u32x4*)(xr + ch * 8);
            f[i][0] = bflo(v[0]); f[i][1] = bfhi(v[0]); f[i][2] = bflo(v[1]); f[i][3] = bfhi(v[1]);
            f[i][4] = bflo(v[2]); f[i][5] = bfhi(v[2]); f[i][6] = bflo(v[3]); f[i][7] = bfhi(v[3]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[i][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    bf16_t* yr = y + (long)row * C;
    char* yq = (char*)y + (long)row * Cpad;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const f32x4 g0 = *(const f32x4*)(gamma + ch * 8), g1 = *(const f32x4*)(gamma + ch * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + ch * 8), b1 = *(const f32x4*)(beta + ch * 8 + 4);
            float o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
                o[j + 4] = (f[i][j + 4] - mean) * rstd * g1[j] + b1[j];
            }
            if (FP8) {
                const u32x2 ov = {pack4fp8(o[0] * oscale, o[1] * oscale, o[2] * oscale, o[3] * oscale),
                                  pack4fp8(o[4] * oscale, o[5] * oscale, o[6] * oscale, o[7] * oscale)};
                *(u32x2*)(yq + ch * 8) = ov;
            } else {
                u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
                *(u32x4*)(yr + ch * 8) = ov;
            }
        }
    }
    if (FP8 && lane < ((Cpad - C) >> 4)) *(u32x4*)(yq + C + lane * 16) = u32x4{0u, 0u, 0u, 0u};
}

// LayerNorm for C = 40 * LPR (320 / 640 / 1280: every SD-1.5 transformer width): LPR = 8 / 16 / 32 lanes share a
// row, 5 chunks of 8 channels per lane, so all 64 lanes work (one wave per row leaves 24 of 64 lanes idle at
// C = 320) and a wave covers 64 / LPR rows.  Lane j of a row group reads chunks j, j + LPR, ...: 16 * LPR contiguous
// bytes per load instruction and row.  Two-pass statistics in registers, reductions by xor-shuffles inside the group.
template <int LPR, bool FP8>
__global__ __launch_bounds__(256) void layernorm_grouped_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                int rows, float eps, float oscale) {
    constexpr int C = 40 * LPR, RPW = 64 / LPR;                  // channels, rows per wave
    constexpr int CPAD = (C + 127) / 128 * 128;                  // fp8 row bytes (K tail of the consuming GEMM)
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const bf16_t* xr = x + (live ? row : 0) * C;
    float f[5][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const u32x4 v = *(const u32x4*)(xr + (sub + i * LPR) * 8);
        f[i][0] = bflo(v[0]); f[i][1] = bfhi(v[0]); f[i][2] = bflo(v[1]); f[i][3] = bfhi(v[1]);
        f[i][4] = bflo(v[2]); f[i][5] = bfhi(v[2]); f[i][6] = bflo(v[3]); f[i][7] = bfhi(v[3]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += f[i][j];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    bf16_t* yr = y + row * C;
    char* yq = (char*)y + row * CPAD;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int ch = sub + i * LPR;
        const f32x4 g0 = *(const f32x4*)(gamma + ch * 8), g1 = *(const f32x4*)(gamma + ch * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(beta + ch * 8), b1 = *(const f32x4*)(beta + ch * 8 + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
            o[j + 4] = (f[i][j + 4] - mean) * rstd * g1[j] + b1[j];
        }
        if (FP8) {
            const u32x2 ov = {pack4fp8(o[0] * oscale, o[1] * oscale, o[2] * oscale, o[3] * oscale),
                              pack4fp8(o[4] * oscale, o[5] * oscale, o[6] * oscale, o[7] * oscale)};
            *(u32x2*)(yq + ch * 8) = ov;
        } else {
            u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
            *(u32x4*)(yr + ch * 8) = ov;
        }
    }
    if (FP8 && sub < ((CPAD - C) >> 4)) *(u32x4*)(yq + C + sub * 16) = u32x4{0u, 0u, 0u, 0u};
}

// bf16 [rows, C] -> e4m3 [rows, Cpad] of sat(x * scale); one 8-element chunk per thread, pad chunks zero
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_t* __restrict__ x, char* __restrict__ y, long rows,
                                                           int C, int Cpad, float scale) {
    const int cpr = Cpad >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cpr) return;
    const long row = i / cpr;
    const int c = (int)(i - row * cpr) * 8;
    u32x2 o = {0u, 0u};
    if (c < C) {
        const u32x4 v = *(const u32x4*)(x + row * C + c);
        o[0] = pack4fp8(bflo(v[0]) * scale, bfhi(v[0]) * scale, bflo(v[1]) * scale, bfhi(v[1]) * scale);
        o[1] = pack4fp8(bflo(v[2]) * scale, bfhi(v[2]) * scale, bflo(v[3]) * scale, bfhi(v[3]) * scale);
    }
    *(u32x2*)(y + row * Cpad + c) = o;
}

// one wave per row; the row lives in registers (cols <= 4096)
// largest e4m3 magnitude CODE (byte & 0x7f: monotonic in |value| on the OCP e4m3fn grid, 0x7f = NaN) of a tensor, by
// atomicMax into *out (zeroed by the caller) -- the measurement behind sd_unet_calibrate_fp8
__global__ __launch_bounds__(256) void amax_e4m3_kernel(const u32x4* __restrict__ y, long n16, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 v = y[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned w = v[k] & 0x7f7f7f7fu;
            m = max(m, max(max(w & 0xff, (w >> 8) & 0xff), max((w >> 16) & 0xff, w >> 24)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ s, long rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* r = s + row * cols;
    const int nchunks = cols >> 3;
    const float c = scale * 1.4426950408889634f;
    float f[8][8];
    float mx = -1e30f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            const u32x4 v = *(const u32x4*)(r + ch * 8);
            f[i][0] = bflo(v[0]); f[i][1] = bfhi(v[0]); f[i][2] = bflo(v[1]); f[i][3] = bfhi(v[1]);
            f[i][4] = bflo(v[2]); f[i][5] = bfhi(v[2]); f[i][6] = bflo(v[3]); f[i][7] = bfhi(v[3]);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, f[i][j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { f[i][j] = __builtin_amdgcn_exp2f((f[i][j] - mx) * c); sum += f[i][j]; }
        }
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunks) {
            u32x4 o = {pack2bf(f[i][0] * inv, f[i][1] * inv), pack2bf(f[i][2] * inv, f[i][3] * inv),
                       pack2bf(f[i][4] * inv, f[i][5] * inv), pack2bf(f[i][6] * inv, f[i][7] * inv)};
            *(u32x4*)(r + ch * 8) = o;
        }
    }
}

}  // namespace

int sd_launch_softmax_rows(bf16_t* s, long rows, int cols, float scale, hipStream_t stream) {
    SD_REQUIRE(s && rows > 0, "softmax_rows: empty");
    SD_REQUIRE(cols % 8 == 0 && cols > 0 && cols <= 4096, "softmax_rows: cols=%d must be a multiple of 8 and <= 4096", cols);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, s, rows, cols, scale);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t sd_groupnorm_scratch_bytes(int B, int HW, int groups) {
    return ((size_t)B * sd_groupnorm_nsplit(B, HW) * groups * 2 + (size_t)B * groups * 2) * sizeof(float);
}

int sd_groupnorm_nsplit(int B, int HW) {
    // enough blocks to cover 256 CUs several times over, but >= 8 pixels per block
    int n = (2048 + B - 1) / B;      // (512 .. 4096 target blocks measure the same within 2 %, round 4)
    if (n > HW / 8) n = HW / 8;
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    return n;
}

// the single-launch small-image kernel takes this instance (it needs no producer statistics)
bool sd_groupnorm_uses_small(int B, int HW, int C1, int C2, int groups) {
    static const bool no_small = getenv("SD_GN_NO_SMALL") != nullptr;
    const int C = C1 + C2;
    return !no_small && HW <= 256 && (long)HW * (C / groups) <= 10240 && (C / groups) % 4 == 0 && C1 % 4 == 0 && B <= 65535;
}

// the shapes gn_slab_kernel takes (a GroupNorm that finishes its producer's deferred split-K reduce): the single-launch
// kernel's shapes with whole groups per 80- (60-)channel workgroup and the per-unit sums of a workgroup in <= 64 KiB of LDS
bool sd_groupnorm_slab_ok(int B, int HW, int C1, int C2, int groups) {
    if (!sd_groupnorm_uses_small(B, HW, C1, C2, groups)) return false;
    const int cpg = (C1 + C2) / groups, G = cpg >= 80 ? 1 : 80 / cpg;
    return cpg >= 20 && cpg <= 80 && groups % G == 0 && C2 % 4 == 0 && (long)B * HW * C1 < (1L << 29);
}

int sd_launch_groupnorm(const GroupNormArgs& a, hipStream_t stream) {
    const int C = a.C1 + a.C2;
    SD_REQUIRE(a.x1 && a.y && a.gamma && a.beta && a.partial, "groupnorm: null operand");
    SD_REQUIRE(!a.out_fp8 || (a.Cpad >= C && a.Cpad % 128 == 0 && C % 16 == 0 && a.oscale > 0.f),
               "groupnorm fp8 out: Cpad=%d must be a multiple of 128 >= C=%d (C a multiple of 16)", a.Cpad, C);
    SD_REQUIRE(a.C1 % 8 == 0 && a.C2 % 8 == 0 && (a.C2 == 0 || a.x2), "groupnorm: C1=%d C2=%d must be multiples of 8", a.C1, a.C2);
    // an 8-channel chunk may straddle at most two groups: cpg >= 8, or cpg = 4 (aligned halves)
    SD_REQUIRE(a.groups > 0 && a.groups <= 64 && C % a.groups == 0 && (C / a.groups >= 8 || C / a.groups == 4),
               "groupnorm: C=%d groups=%d needs C/groups >= 8 (or == 4)", C, a.groups);
    SD_REQUIRE(a.nsplit >= 1 && a.nsplit <= a.HW && a.B > 0 && a.HW > 0, "groupnorm: bad split %d for HW=%d", a.nsplit, a.HW);
    const GnGeom g = gn_geom(C, a.groups);
    SD_REQUIRE(g.threads <= 1024, "groupnorm: C=%d too wide", C);
    if (a.slab) {
        const int cpg = C / a.groups, G = cpg >= 80 ? 1 : 80 / cpg, ncol = G * (cpg / 4);
        SD_REQUIRE(a.splitk > 1 && a.splitk <= 64 && a.x1w && !a.out_fp8 && (!a.sR || a.sldr >= a.C1) && (a.C2 == 0 || a.x2),
                   "groupnorm with a deferred split-K reduce: splitk=%d, bf16 output, x1w and a residual stride >= C1", a.splitk);
        SD_REQUIRE(sd_groupnorm_slab_ok(a.B, a.HW, a.C1, a.C2, a.groups), "groupnorm with a deferred split-K reduce: HW=%d C=%d+%d "
                   "is not a shape gn_slab_kernel takes", a.HW, a.C1, a.C2);
        const dim3 grid(a.groups / G, a.B), blk((32 * ncol + 63) / 64 * 64);
        const size_t smem = ((size_t)G * a.HW * (cpg / 4) + 4 * G) * sizeof(float2);
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, blk, smem, stream, a); };
        auto pick = [&](auto mi, auto un) {
            constexpr int MI = decltype(mi)::value, UN = decltype(un)::value;
            if (a.silu) go(gn_slab_kernel<true, MI, UN>); else go(gn_slab_kernel<false, MI, UN>);
        };
        using std::integral_constant;
        const int iters = (a.HW + 31) / 32, k = a.splitk;
        if (iters <= 2) {
            if (k % 8 == 0) pick(integral_constant<int, 2>{}, integral_constant<int, 8>{});
            else if (k % 4 == 0) pick(integral_constant<int, 2>{}, integral_constant<int, 4>{});
            else if (k % 2 == 0) pick(integral_constant<int, 2>{}, integral_constant<int, 2>{});
            else pick(integral_constant<int, 2>{}, integral_constant<int, 1>{});
        } else if (iters <= 4) {
            if (k % 4 == 0) pick(integral_constant<int, 4>{}, integral_constant<int, 4>{});
            else if (k % 2 == 0) pick(integral_constant<int, 4>{}, integral_constant<int, 2>{});
            else pick(integral_constant<int, 4>{}, integral_constant<int, 1>{});
        } else {
            if (k % 2 == 0) pick(integral_constant<int, 8>{}, integral_constant<int, 2>{});
            else pick(integral_constant<int, 8>{}, integral_constant<int, 1>{});
        }
        SD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (sd_groupnorm_uses_small(a.B, a.HW, a.C1, a.C2, a.groups)) {
        const int units = (a.HW * (C / a.groups / 4) + 255) / 256;       // 4-channel units per thread
        const dim3 grid(a.groups, a.B);
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, a); };
        auto pick = [&](auto mu) {
            constexpr int MU = decltype(mu)::value;
            if (a.out_fp8) { if (a.silu) go(gn_small_kernel<true, true, MU>); else go(gn_small_kernel<false, true, MU>); }
            else { if (a.silu) go(gn_small_kernel<true, false, MU>); else go(gn_small_kernel<false, false, MU>); }
        };
        if (units <= 3) pick(std::integral_constant<int, 3>{});
        else if (units <= 5) pick(std::integral_constant<int, 5>{});
        else pick(std::integral_constant<int, 10>{});
        SD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    dim3 grid(a.nsplit, a.B);
    if (a.stats1 != nullptr) {
        SD_REQUIRE(a.HW % 64 == 0 && (a.C2 == 0 || a.stats2 != nullptr), "groupnorm: producer statistics need HW %% 64 == 0 and both sources");
        hipLaunchKernelGGL(gn_finalize_stats_kernel, dim3(a.B * a.groups), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(g.threads), (g.threads + g.nchunks) * sizeof(float4), stream, a);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((a.B * a.groups + 3) / 4), dim3(256), 0, stream, a);
    }
    if (a.out_fp8) {
        if (a.silu) hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, dim3(g.threads), 0, stream, a);
        else hipLaunchKernelGGL((gn_apply_kernel<false, true>), grid, dim3(g.threads), 0, stream, a);
    } else {
        if (a.silu) hipLaunchKernelGGL((gn_apply_kernel<true, false>), grid, dim3(g.threads), 0, stream, a);
        else hipLaunchKernelGGL((gn_apply_kernel<false, false>), grid, dim3(g.threads), 0, stream, a);
    }
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

static int launch_layernorm(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int rows, int C, float eps,
                            bool fp8, int Cpad, float oscale, hipStream_t stream) {
    SD_REQUIRE(x && y && gamma && beta, "layernorm: null operand");
    SD_REQUIRE(C % 8 == 0 && C > 0 && C <= 1536, "layernorm: C=%d must be a multiple of 8 and <= 1536", C);
    SD_REQUIRE(rows > 0, "layernorm: no rows");
    SD_REQUIRE(!fp8 || (Cpad == (C + 127) / 128 * 128 && C % 16 == 0 && oscale > 0.f && Cpad - C <= 64 * 16),
               "layernorm fp8 out: Cpad=%d must be C=%d rounded up to 128 (C a multiple of 16)", Cpad, C);
    auto grouped = [&](auto kern, int lpr) {
        const int rows_per_block = 4 * (64 / lpr);
        hipLaunchKernelGGL(kern, dim3((rows + rows_per_block - 1) / rows_per_block), dim3(256), 0, stream, x, gamma, beta, y,
                           rows, eps, oscale);
    };
    if (fp8) {
        if (C == 320) grouped(layernorm_grouped_kernel<8, true>, 8);
        else if (C == 640) grouped(layernorm_grouped_kernel<16, true>, 16);
        else if (C == 1280) grouped(layernorm_grouped_kernel<32, true>, 32);
        else
            hipLaunchKernelGGL(layernorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, rows, C,
                               eps, Cpad, oscale);
    } else {
        if (C == 320) grouped(layernorm_grouped_kernel<8, false>, 8);
        else if (C == 640) grouped(layernorm_grouped_kernel<16, false>, 16);
        else if (C == 1280) grouped(layernorm_grouped_kernel<32, false>, 32);
        else
            hipLaunchKernelGGL(layernorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, rows, C,
                               eps, C, 1.0f);
    }
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_layernorm(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int rows, int C,
                        float eps, hipStream_t stream) {
    return launch_layernorm(x, gamma, beta, y, rows, C, eps, false, C, 1.0f, stream);
}

int sd_launch_layernorm_fp8(const bf16_t* x, const float* gamma, const float* beta, void* y, int rows, int C, int Cpad,
                            float eps, float oscale, hipStream_t stream) {
    return launch_layernorm(x, gamma, beta, (bf16_t*)y, rows, C, eps, true, Cpad, oscale, stream);
}

int sd_launch_quantize_fp8(const bf16_t* x, void* y, long rows, int C, int Cpad, float scale, hipStream_t stream) {
    SD_REQUIRE(x && y && rows > 0, "quantize_fp8: null operand");
    SD_REQUIRE(C % 8 == 0 && Cpad % 8 == 0 && Cpad >= C && scale > 0.f, "quantize_fp8: C=%d Cpad=%d", C, Cpad);
    const long n = rows * (Cpad >> 3);
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, (char*)y, rows, C,
                       Cpad, scale);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_amax_e4m3(const void* y, long nbytes, unsigned* out, hipStream_t stream) {
    SD_REQUIRE(y && out && nbytes > 0 && nbytes % 16 == 0 && ((uintptr_t)y & 15) == 0, "amax_e4m3: %ld bytes, 16-byte granules", nbytes);
    const long n16 = nbytes / 16;
    const unsigned grid = (unsigned)std::min<long>((n16 + 255) / 256, 2048);
    hipLaunchKernelGGL(amax_e4m3_kernel, dim3(grid), dim3(256), 0, stream, (const u32x4*)y, n16, out);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
