// bf16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950.
//
//   C[M,N] = X[M,K] . W[N,K]^T  (+bias +bias2 +residual | GEGLU)
//
// * X rows are either plain matrix rows (GEMM; optionally two K-segments from two tensors =
//   a channel concat that is never materialised) or NHWC pixels gathered on the fly for a 3x3
//   convolution (stride 1/2, optional fused nearest-2x upsample); K = 9*Cin ordered
//   (64-channel slice, tap, channel) so the 9 shifted reads of a slice are adjacent in time.
// * Tiles are staged global -> LDS with 16-byte `global_load_lds` (no VGPR round trip), double
//   buffered, one barrier per 64-deep K tile.  LDS rows are 128 B (64 bf16) with the 16-byte
//   chunk index XOR-swizzled by (row & 7) -- applied on the per-lane SOURCE address, LDS stays
//   lane-linear -- so the ds_read_b128 fragment reads are bank-conflict free.
// * Out-of-range rows (M/N tails, conv zero padding) read a zero page instead of branching.
// * MFMA v_mfma_f32_16x16x32_bf16 with W as the A operand: a lane then owns 4 consecutive
//   output columns of one row, i.e. one 8-byte packed bf16 store per 16x16 tile.
// * DT = 1: fp8-e4m3 (OCP) operands -- X quantised by its producer (GroupNorm / LayerNorm / GEGLU epilogue,
//   static per-tensor scale), W quantised per output channel on the host.  The same 128-byte LDS rows then hold
//   128 K elements and one v_mfma_f32_16x16x128_f8f6f4 (32 cycles) replaces two bf16 MFMAs (2 x 16 cycles) per
//   16x16 tile and K tile: twice the flops per LDS-fill byte, fragment-read byte and matrix-core cycle.  A lane
//   reads its 32 K bytes of a row as two ds_read_b128; the chunk swizzle is f(row) = bit1(row) | (row & 4)
//   (conflict-free for this read pattern, tools/lds_swizzle_search.py).  A and B fragments are read with the
//   same (lane group, byte) -> k map, so the contraction does not depend on the instruction's internal k order.
//   The epilogue multiplies by wscale[n] / xscale before bias / residual.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

// Timing ablations (SD_GEMM_TUNE bits: 8 no stores, 16 one K tile per item, 64 narrow stores -- bits 8 and 16 give WRONG
// results by design) exist only in the SD_ABLATE build (libsdhip_ablate.so, sonicdiffusionbayeslab_amd/build.py): the
// product library carries neither the branches nor the environment variable.
#ifdef SD_ABLATE
#define SD_TUNE(p) ((p).tune)
#else
#define SD_TUNE(p) 0
#endif

namespace {

constexpr int AMODE_GEMM = 0;
constexpr int AMODE_CONV = 1;
constexpr int EPI_STD = 0;
constexpr int EPI_GEGLU = 1;
constexpr int EPI_SOFTMAX = 2;   // row softmax over every 80-column group (one group per N-wave), std stores

constexpr int kPersistentGrid = 512;   // 2 workgroups per CU x 256 CUs

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// STAGES = 2: classic double buffer (wait for everything, one barrier per K tile).
// STAGES = 3: two K tiles in flight; the wait before the barrier is a COUNTED vmcnt that leaves the
//             newest tile's LDS-DMA outstanding, so HBM/L2 latency spans a whole tile of MFMA work.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int AMODE, int EPI, int DT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_kernel(const GemmArgs p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int ESZ = DT ? 1 : 2;               // bytes per operand element
    constexpr int KTE = 128 / ESZ;                // K elements per 128-byte LDS row = per K tile
    static_assert(!DT || STAGES == 2, "fp8 path: 2-stage loop only");
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int XI = BM / 8, WI = BN / 8;  // 1-KiB glds wave-instructions per tile
    static_assert(XI % NW == 0, "X tile rows must split evenly over waves");
    constexpr int XPW = XI / NW;
    constexpr int WPW = (WI + NW - 1) / NW;          // waves < WI % NW issue WPW, the rest WPW - 1
    constexpr int WREM = WI % NW;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    static_assert(EPI != EPI_GEGLU || (TN % 2 == 0), "GEGLU pairs 16-col tiles");
    static_assert(STAGES == 2 || STAGES == 3, "2 or 3 LDS stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Persistent workgroups: the grid is at most 2 workgroups per CU and each walks the work items
    // (output tile x K split) with stride gridDim.x.  XCD-aware order: blocks that share an XCD (same
    // blockIdx % 8) get consecutive items, n fastest, so the X panel of an m-tile is re-read from
    // that XCD's L2.  The first K tile of the NEXT item is prefetched before the epilogue of the
    // current one, so neither the prologue latency nor the store tail is exposed between tiles.
    int work = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = work & 7, idx = work >> 3;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nwork = ntiles * p.splitk;
    const int KTall = p.K / KTE;
    int split = 0, m0 = 0, n0 = 0, kt_begin = 0, KT = 0;
    auto decode = [&](int w) {
        // (scalar work per item: keep it to two 32-bit divisions -- the 64-bit K-range arithmetic this replaced was ~250
        // instructions per item, against ~330 in the whole K loop of a K = 320 GEMM; KTall * splitk < 2^31 by far)
        split = p.splitk > 1 ? w / ntiles : 0;
        const int t = w - split * ntiles;
        const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
        if (p.splitk > 1) {
            kt_begin = KTall * split / p.splitk;
            KT = KTall * (split + 1) / p.splitk - kt_begin;
        } else {
            kt_begin = 0;
            KT = KTall;
        }
        if (SD_TUNE(p) & 16) KT = 1;      // (SD_ABLATE build only) epilogue + fixed costs only: wrong results
    };

    // ---- per-lane source descriptors -------------------------------------------------
    const int lrow8 = lane >> 3;                 // row inside an 8-row glds piece
    // global 16-B chunk this lane fetches (swizzle): bf16 rows by (row & 7), fp8 rows by bit1(row) | (row & 4)
    const int gch = (lane & 7) ^ (DT ? (((lrow8 >> 1) & 1) | (lrow8 & 4)) : lrow8);
    const char* zero = (const char*)p.zero_page;

    // Everything that does not change along K is folded into one pointer per row here, so the
    // per-tile work is a scalar offset add + a select (the K loop is otherwise address-VALU bound).
    //   GEMM: row pointers into X (and X2 for the second K segment); null = zero page.
    //   CONV: pointer of the tap (0,0) input pixel + a 9-bit validity mask of the taps.
    const char* xptr[XPW];        // byte pointers: element size is 2 (bf16) or 1 (fp8)
    const char* xptr2[XPW];
    int xmask[XPW];
    int xoy[XPW], xox[XPW], xb[XPW];   // only the fused-upsample conv path recomputes pixels per tap
    const char* wsrc[WPW];
    const int Hv = p.Hin << p.up, Wv = p.Win << p.up;
    auto setup = [&]() {
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int inst = wave + i * NW;
        const int m = m0 + inst * 8 + lrow8;
        xptr2[i] = nullptr; xmask[i] = 0; xoy[i] = xox[i] = xb[i] = 0;
        if (AMODE == AMODE_GEMM) {
            xptr[i] = m < p.M ? (const char*)p.X + (long)m * p.ldx * ESZ + gch * 16 : nullptr;
            if (p.X2) xptr2[i] = m < p.M ? (const char*)p.X2 + (long)m * p.ldx2 * ESZ + gch * 16 : nullptr;
        } else if (p.subpix) {
            // sub-pixel upsample conv: row m = (sample, phase, low-res pixel); 2x2 taps from (y - 1 + py, x - 1 + px)
            const int hwl = p.Hin * p.Win;
            const int b = m / (4 * hwl), rem = m - b * 4 * hwl;
            const int ph = rem / hwl, ml = rem - ph * hwl;
            const int y = ml / p.Win, x = ml - y * p.Win;
            const int y0 = y - 1 + (ph >> 1), x0 = x - 1 + (ph & 1);
            xoy[i] = y0; xox[i] = x0; xb[i] = m < p.M ? b * hwl : -1;
            xptr[i] = (const char*)p.X + ((long)b * hwl + (long)y0 * p.Win + x0) * p.Cin * ESZ + gch * 16;
            int mask = 0;
            if (m < p.M) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ty = y0 + (t >> 1), tx = x0 + (t & 1);
                    if (ty >= 0 && ty < p.Hin && tx >= 0 && tx < p.Win) mask |= 1 << t;
                }
            }
            xmask[i] = mask;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const int y0 = oy * p.stride - 1, x0 = ox * p.stride - 1;
            xoy[i] = y0; xox[i] = x0; xb[i] = m < p.M ? b * p.Hin * p.Win : -1;
            xptr[i] = (const char*)p.X + ((long)b * p.Hin * p.Win + (long)y0 * p.Win + x0) * p.Cin * ESZ + gch * 16;
            int mask = 0;
            if (m < p.M) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ty = y0 + t / 3, tx = x0 + t % 3;
                    if (ty >= 0 && ty < Hv && tx >= 0 && tx < Wv) mask |= 1 << t;
                }
            }
            xmask[i] = mask;
        }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int inst = wave + i * NW;
        const int n = n0 + inst * 8 + lrow8;
        const long wb = p.subpix ? (long)((m0 / (p.Hin * p.Win)) & 3) * p.w_batch_stride
                                 : p.rows_per_batch > 0 ? (long)(m0 / p.rows_per_batch) * p.w_batch_stride : 0;
        wsrc[i] = (inst < WI && n < p.N) ? (const char*)p.W + (wb + (long)n * p.ldw) * ESZ + gch * 16 : nullptr;
    }
    };

    // conv K order: k = (channel slice of 64, tap, channel) -- the 9 taps of one slice are
    // consecutive K tiles, so the shifted re-reads of the same pixels hit L2 while still hot.
    auto stage = [&](int kt, int buf) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + BM * 128;
        const int k0 = kt * KTE;
        if (AMODE == AMODE_GEMM) {
            const bool seg2 = k0 >= p.K1;
            const int kk = (seg2 ? k0 - p.K1 : k0) * ESZ;
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
                const int inst = wave + i * NW;
                const char* rp = seg2 ? xptr2[i] : xptr[i];
                const void* src = rp ? (const void*)(rp + kk) : (const void*)zero;
                glds16(src, xs + inst * 1024);
            }
        } else {
            const int nt = p.subpix ? 4 : 9, tw = p.subpix ? 2 : 3;        // taps per channel slice, taps per row
            const int cs = kt / nt, tap = kt - cs * nt;
            const int dy = tap / tw, dx = tap - dy * tw;
            if (!p.up) {
                const long toff = (((long)dy * p.Win + dx) * p.Cin + cs * KTE) * ESZ;   // wave-uniform
#pragma unroll
                for (int i = 0; i < XPW; ++i) {
                    const int inst = wave + i * NW;
                    const void* src = ((xmask[i] >> tap) & 1) ? (const void*)(xptr[i] + toff) : (const void*)zero;
                    glds16(src, xs + inst * 1024);
                }
            } else {
                const int coff = cs * 128 + gch * 16;
#pragma unroll
                for (int i = 0; i < XPW; ++i) {
                    const int inst = wave + i * NW;
                    const int ty = xoy[i] + dy, tx = xox[i] + dx;
                    const bool ok = (xmask[i] >> tap) & 1;
                    const int iy = ty >> 1, ix = tx >> 1;
                    const void* src = ok ? (const void*)((const char*)p.X + ((long)(xb[i] + iy * p.Win + ix)) * p.Cin * ESZ + coff)
                                         : (const void*)zero;
                    glds16(src, xs + inst * 1024);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int inst = wave + i * NW;
            if (inst < WI) {     // wave-uniform
                const void* src = wsrc[i] ? (const void*)(wsrc[i] + k0 * ESZ) : (const void*)zero;
                glds16(src, ws + inst * 1024);
            }
        }
    };

    // ---- main loop ---------------------------------------------------------------------
    f32x4 acc[TN][TM];

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lrow = lane & 15, lq = lane >> 4;
    // bf16: k-step ks reads chunk (4 ks + lq) ^ (row & 7); fp8: the lane's 32 K bytes are chunks 2 lq, 2 lq + 1, each ^ f(row)
    const int f8sw = ((lane >> 1) & 1) | (lane & 4);
    const int swz[2] = {DT ? ((2 * lq) ^ f8sw) << 4 : ((lq) ^ (lane & 7)) << 4,
                        DT ? ((2 * lq + 1) ^ f8sw) << 4 : ((4 + lq) ^ (lane & 7)) << 4};
    const int xoff = (wm * WTM + lrow) * 128;
    const int woff = BM * 128 + (wn * WTN + lrow) * 128;

    // One K tile = two 32-deep k-steps.  The fragments of k-step 0 are read BEFORE the next tile's
    // LDS-DMA is issued (a glds costs the issuing wave ~60 cycles each, which hides the ds_read
    // latency), k-step 1's fragments are read under k-step 0's MFMAs.
    constexpr bool FRAG_DB = !DT && TM * TN * 4 + (TM + TN) * 8 <= 200;   // accumulators + two fragment sets
    bf16x8 xf0[DT ? 1 : TM], wf0[DT ? 1 : TN], xf1[FRAG_DB ? TM : 1], wf1[FRAG_DB ? TN : 1];
    // fp8: all W fragments of the tile (TN x 32 B) + a two-deep ring of X fragments, MFMAs walk the M tiles
    i32x8 wq[DT ? TN : 1], xq[2];
    auto ld32 = [&](const char* ptr) -> i32x8 {
        return cat_u32x4(*(const u32x4*)(ptr + swz[0]), *(const u32x4*)(ptr + swz[1]));
    };
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

    // LayerNorm fold (consumer side): lane l owns row l of this wave's 64 rows and fetches that row's partials.
    //  * std epilogue (registers to spare): all partials are requested under the first K tile (ordinary loads older than
    //    every later LDS-DMA: retired in order long before the epilogue) -> no latency left in the epilogue;
    //  * GEGLU (the 256 x 256 tile has no register to spare): four partials at a time in the epilogue, the first four
    //    together with the c1 / bias vectors.
    // (EPI_SOFTMAX, round 5: norm2 folded into the first GEMM of the two-GEMM prompt cross-attention -- per-sample weights, so c1 / c2
    // are per-sample vectors too: GemmArgs::ln_per_sample)
    constexpr bool LN_OK = (EPI == EPI_STD || EPI == EPI_GEGLU || EPI == EPI_SOFTMAX) && AMODE == AMODE_GEMM && WTM == 64 && !DT && STAGES == 2;
    constexpr bool LN_PRE = LN_OK && EPI != EPI_GEGLU;
    constexpr int LN_MAXP = 16;
    char* lnbuf = smem + STAGES * STAGE_BYTES + wave * 512;
    // c1 | c2 of this N-wave's WTN columns, fetched by LDS-DMA under the first K tile (waves wm == 0; the K loop's own
    // wait + barrier publish them): the epilogue of a folded GEMM then has no global load at all
    char* lnvec = smem + STAGES * STAGE_BYTES + NW * 512 + (wave / WAVES_M) * (2 * WTN * 4);
    f32x2_t lnp[LN_PRE ? LN_MAXP : 4];
    auto ln_vec_dma = [&]() {
        if (wave % WAVES_M == 0 && lane < WTN / 4) {
            const int n = min(n0 + (wave / WAVES_M) * WTN + lane * 4, p.N - 4);
            const long vo = (p.ln_per_sample && p.rows_per_batch > 0) ? (long)(m0 / p.rows_per_batch) * p.N : 0;
            glds16(p.ln_c1 + vo + n, lnvec);
            glds16(p.bias + vo + n, lnvec + WTN * 4);
        }
    };
    auto ln_issue = [&](int i0, int cnt) {           // partials i0 .. i0 + cnt - 1 of this lane's row -> lnp[0 ..]
        const int mrow = min(m0 + wm * WTM + lane, p.M - 1);
        const float* src = p.ln_rs + (long)mrow * 2;
        // unconditional loads from a clamped index (a select on a loaded value makes hipcc branch around the load and
        // wait for it at the join: sixteen serial round trips); absent partials are masked when they are summed
#pragma unroll
        for (int i = 0; i < (LN_OK ? LN_MAXP : 0); ++i)
            if (i < cnt) lnp[i] = *(const f32x2_t*)(src + (long)min(i0 + i, p.ln_np - 1) * p.M * 2);
    };

    if (work >= nwork) return;
    decode(work);
    setup();
    int g = 0;             // running K-tile count of this workgroup: LDS buffer parity (2-stage path)
    bool primed = false;   // first K tile of the current item already in flight
    bool stores_pending = false;   // ... and exactly EPI_STORES epilogue stores were issued after it
    // stores per wave for a full tile: adjacent tiles are paired into 16-byte stores
    constexpr int OUT_TILES = EPI == EPI_GEGLU ? TN / 2 : TN;
    constexpr int EPI_STORES = (OUT_TILES / 2 + OUT_TILES % 2) * TM;
    while (true) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (STAGES == 2) {
        if (!primed) stage(kt_begin, g & 1);
        for (int kt = 0; kt < KT; ++kt) {
            // vmcnt counts stores too and retires in order.  The first K tile of a prefetched item was
            // issued BEFORE the previous item's epilogue stores, so a COUNTED wait (= that epilogue's
            // store count) retires the DMA and leaves the stores in flight under this tile's MFMAs.
            if (kt == 0 && stores_pending) wait_vmcnt<EPI_STORES>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sb = smem + ((g + kt) & 1) * STAGE_BYTES;
            if (DT) {
#pragma unroll
                for (int t = 0; t < TN; ++t) wq[t] = ld32(sb + woff + t * 2048);
                xq[0] = ld32(sb + xoff);
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < KT) stage(kt_begin + kt + 1, (g + kt + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    if (b + 1 < TM) xq[(b + 1) & 1] = ld32(sb + xoff + (b + 1) * 2048);
#pragma unroll
                    for (int a = 0; a < TN; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq[a], xq[b & 1], acc[a][b], 0, 0, 0, 0, 0, 0);
                }
                continue;
            }
            load_frags(sb, 0, xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT) stage(kt_begin + kt + 1, (g + kt + 1) & 1);
            if (LN_OK && kt == 0 && p.ln_rs) {
                if (LN_PRE) ln_issue(0, LN_MAXP);
                ln_vec_dma();
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
        g += KT;
    } else {
        // loads this wave issues per K tile (wave-uniform): XPW + (wave < WREM ? WPW : WPW - 1)
        const bool more = (WREM == 0) || (wave < WREM);
        stage(kt_begin, 0);
        if (KT > 1) stage(kt_begin + 1, 1);
        int buf = 0;
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) {           // leave tile kt+1 in flight
                if (more) wait_vmcnt<XPW + WPW>();
                else wait_vmcnt<XPW + WPW - 1>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();   // tile kt landed for every wave; tile kt-1 fully consumed
            asm volatile("" ::: "memory");
            const char* sb = smem + buf * STAGE_BYTES;
            load_frags(sb, 0, xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < KT) stage(kt_begin + kt + 2, buf == 0 ? 2 : buf - 1);   // reuses tile kt-1's buffer
            __builtin_amdgcn_sched_barrier(0);
            load_frags(sb, 1, xf1, wf1);
            mfmas(xf0, wf0);
            mfmas(xf1, wf1);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    // ---- epilogue, phase A (before the next item's DMA is issued): every LOAD the epilogue needs --
    // bias / time-embedding / residual are folded into the fp32 accumulators now, so that no ordinary
    // load is outstanding once the LDS-DMA prefetch is in flight (hipcc would drain everything with
    // vmcnt(0) at their first use) and phase B is stores only.
    const int em0 = m0, en0 = n0, esplit = split;
    if (DT && p.splitk == 1) {     // dequantise: per-output-channel weight scale / per-tensor activation scale
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
            const f32x4 sv = *(const f32x4*)(p.wscale + n) * p.xscale_inv;
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] *= sv;
        }
    }
    if (LN_OK && p.ln_rs) {
        // LayerNorm fold: rstd * (acc - mean * c1[n]) + c2[n]  (c2 = W beta + b arrives as p.bias)
        if (!LN_PRE) ln_issue(0, 4);
        constexpr int TNH = LN_PRE ? TN : (TN + 1) / 2;      // c1 / c2 vectors in flight at a time
        f32x4 cv[TNH], bv[TNH];
        auto vec_load = [&](int a0) {
#pragma unroll
            for (int a = 0; a < TNH; ++a) {
                cv[a] = *(const f32x4*)(lnvec + ((a0 + a) * 16 + lq * 4) * 4);
                bv[a] = *(const f32x4*)(lnvec + WTN * 4 + ((a0 + a) * 16 + lq * 4) * 4);
            }
        };
        vec_load(0);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < (LN_PRE ? LN_MAXP : 4); ++i) {
            s += i < p.ln_np ? lnp[i][0] : 0.f;
            q += i < p.ln_np ? lnp[i][1] : 0.f;
        }
        if (!LN_PRE) {
            for (int i0 = 4; i0 < p.ln_np; i0 += 4) {
                ln_issue(i0, 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s += i0 + i < p.ln_np ? lnp[i][0] : 0.f;
                    q += i0 + i < p.ln_np ? lnp[i][1] : 0.f;
                }
            }
        }
        const float invk = 1.0f / (float)p.K;
        *(f32x2_t*)(lnbuf + lane * 8) = ln_mean_rstd(s, q, invk, p.ln_eps);      // row = lane; every lane needs rows b * 16 + lrow
        f32x2_t mr[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) mr[b] = *(const f32x2_t*)(lnbuf + (b * 16 + lrow) * 8);
#pragma unroll
        for (int a0 = 0; a0 < TN; a0 += TNH) {
            if (a0 > 0) vec_load(a0);
#pragma unroll
            for (int a = 0; a < TNH; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    if (a0 + a < TN) acc[a0 + a][b] = ln_fold(acc[a0 + a][b], cv[a], mr[b][0], mr[b][1], bv[a]);
        }
    }
    if (p.splitk == 1) {
        if (EPI == EPI_STD && p.R) {
            // all residual loads of the tile issued back to back (rows clamped instead of branched, so
            // the compiler keeps them in flight together), then folded into the accumulators
            u32x2 rr[TN][TM];
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = min(em0 + wm * WTM + b * 16 + lrow, p.M - 1);
                    rr[a][b] = *(const u32x2*)(p.R + (long)m * p.ldr + n);
                }
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    acc[a][b][0] += bflo(rr[a][b][0]); acc[a][b][1] += bfhi(rr[a][b][0]);
                    acc[a][b][2] += bflo(rr[a][b][1]); acc[a][b][3] += bfhi(rr[a][b][1]);
                }
        }
        if (p.bias && !(LN_OK && p.ln_rs)) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                // (GEGLU keeps the packed [16 value | 16 gate] column order for its bias)
                const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
                f32x4 bv = *(const f32x4*)(p.bias + n);
                if (EPI == EPI_STD && p.bias2) bv += *(const f32x4*)(p.bias2 + n);
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] += bv;
            }
        }
    }

    if (EPI == EPI_SOFTMAX) {
        // Each N-wave owns one 80-column group (WTN = 80): softmax over its first sm_valid columns, the rest -> 0.
        // A lane holds 4 consecutive columns per 16-column tile; the 4 lane groups lq of a row sit 16 lanes apart.
        static_assert(EPI != EPI_SOFTMAX || WTN == 80, "softmax epilogue: one 80-column group per N-wave");
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            float mx = -1e30f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (a * 16 + lq * 4 + j < p.sm_valid) mx = fmaxf(mx, acc[a][b][j]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = a * 16 + lq * 4 + j < p.sm_valid
                                        ? __builtin_amdgcn_exp2f((acc[a][b][j] - mx) * 1.4426950408889634f) : 0.f;
                    acc[a][b][j] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int a = 0; a < TN; ++a) acc[a][b] *= inv;
        }
    }

    // ---- next work item: descriptors + first K tile in flight before the stores of this one -------
    work += gridDim.x;
    const bool more_work = work < nwork;
    if (more_work) {
        decode(work);
        setup();
        if (STAGES == 2) {       // buffer g&1 was last read two K tiles ago, before the last barrier
            stage(kt_begin, g & 1);
            primed = true;
            // phase B issues exactly EPI_STORES stores per wave iff the finished tile is full
            stores_pending = (em0 + BM <= p.M) && (en0 + BN <= p.N) && (p.ldc & 7) == 0 && p.splitk == 1 &&
                             !(SD_TUNE(p) & (32 | 64 | 8)) && !p.out_fp8;   // (statistics stores come before them: only more to wait for)
        }
    }

    // ---- epilogue, phase B: convert + store; lane owns columns n..n+3 of row m per 16x16 tile -----
    if (EPI == EPI_STD && p.splitk > 1) {
        // split-K: raw fp32 partial sums into this split's slab; splitk_reduce_kernel finishes
        float* slab = p.slab + (long)esplit * p.M * p.N;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int n = en0 + wn * WTN + a * 16 + lq * 4;
            if (n >= p.N) continue;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = em0 + wm * WTM + b * 16 + lrow;
                if (m >= p.M) continue;
                *(f32x4*)(slab + (long)m * p.N + n) = acc[a][b];
            }
        }
    } else if (EPI == EPI_STD || EPI == EPI_SOFTMAX) {
        // GroupNorm statistics of this tile for the consuming GroupNorm (64-row blocks = this wave's rows; WTM == 64)
        if (EPI == EPI_STD && WTM == 64 && p.stats)
            tile_channel_stats<TN, TM>(acc, p.stats, (em0 + wm * WTM) >> 6, p.N, en0 + wn * WTN, p.N, em0 + wm * WTM, p.M, lane);
        if (EPI == EPI_STD && p.rowstats)
            tile_row_stats<TN, TM>(acc, p.rowstats + ((long)(en0 / BN) * WAVES_N + wn) * p.M * 2, en0 + wn * WTN, p.N,
                                   em0 + wm * WTM, p.M, lane);
        // Stores are issue-bound (one 8-B store per lane per 16x16 tile): v_permlane16_swap pairs two
        // adjacent tiles so that every lane owns 16 contiguous bytes -> half the store instructions,
        // 64 contiguous bytes per row per instruction.  After the swap lane group lq holds
        // lq=0: tile a cols 0-7, lq=1: tile a+1 cols 0-7, lq=2: tile a cols 8-15, lq=3: tile a+1 cols 8-15.
        const bool wide_ok = (p.ldc & 7) == 0 && !(SD_TUNE(p) & 64);
        // destination of columns col.. of row m: C, or the head-major K / V block of this tile (GemmArgs::KV)
        const bool hm = EPI == EPI_STD && p.hm_C > 0 && en0 >= p.hm_C;            // tile-uniform
        bf16_t* hm_base = nullptr;
        if (hm) {
            const int which = en0 / p.hm_C - 1, head0 = (en0 - (which + 1) * p.hm_C) / 40, heads = p.hm_C / 40;
            const int smp = em0 / p.hm_tok;
            hm_base = p.KV + ((((long)which * (p.M / p.hm_tok) + smp) * heads + head0) * p.hm_tok - (long)smp * p.hm_tok) * 40;
        }
        auto out_ptr = [&](int m, int col) -> bf16_t* {
            if (AMODE == AMODE_CONV && p.subpix) {          // (sample, phase, low-res pixel) -> output pixel (2y+py, 2x+px)
                const int hwl = p.Hin * p.Win;
                const int b = m / (4 * hwl), rem = m - b * 4 * hwl;
                const int ph = rem / hwl, ml = rem - ph * hwl;
                const int y = ml / p.Win, x = ml - y * p.Win;
                const long orow = ((long)b * p.Hout + 2 * y + (ph >> 1)) * p.Wout + 2 * x + (ph & 1);
                return p.C + orow * p.ldc + col;
            }
            if (!hm) return p.C + (long)m * p.ldc + col;
            const int cl = col - en0;                       // 0 .. 159: (head, channel) = (cl / 40, cl % 40)
            const int hl = (cl * 205) >> 13;
            return hm_base + ((long)hl * p.hm_tok + m) * 40 + (cl - hl * 40);
        };
        auto store_narrow = [&](int a) {
            const int n = en0 + wn * WTN + a * 16 + lq * 4;
            if (n >= p.N) return;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = em0 + wm * WTM + b * 16 + lrow;
                const f32x4 v = acc[a][b];
                u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                if (m < p.M && !(SD_TUNE(p) & 8)) *(u32x2*)out_ptr(m, n) = o;
            }
        };
#pragma unroll
        for (int a = 0; a < TN; a += 2) {
            const int nb = en0 + wn * WTN + a * 16;
            if (a + 1 < TN && wide_ok && nb + 32 <= p.N) {            // wave-uniform: both tiles in range
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = em0 + wm * WTM + b * 16 + lrow;
                    const f32x4 vx = acc[a][b], vy = acc[a + 1 < TN ? a + 1 : a][b];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[0], vx[1]), pack2bf(vy[0], vy[1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[2], vx[3]), pack2bf(vy[2], vy[3]), false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    const int col = nb + (lq & 1) * 16 + (lq >> 1) * 8;
                    if (m < p.M && !(SD_TUNE(p) & 8)) *(u32x4*)out_ptr(m, col) = o;
                }
            } else {
                store_narrow(a);
                if (a + 1 < TN) store_narrow(a + 1);
            }
        }
    } else {
        // GEGLU: weight rows were packed so that every 32-column group is [16 value | 16 gate];
        // two output tiles (four accumulator tiles) are paired for 16-byte stores as above
        const bool wide_ok = (p.ldc & 7) == 0 && !(SD_TUNE(p) & 64) && !p.out_fp8;
        if (p.out_fp8) {
            // fp8 output (input of ff.net.2): a lane owns 4 consecutive bytes of a row per output tile, the four lane
            // groups of a row 16 contiguous bytes
#pragma unroll
            for (int a = 0; a < TN; a += 2) {
                const int n = en0 + wn * WTN + a * 16;
                if (n >= p.N) continue;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = em0 + wm * WTM + b * 16 + lrow;
                    const f32x4 va = acc[a][b], vg = acc[a + 1][b];
                    const f32x2_t lo = geglu_pair(f32x2_t{va[0], va[1]}, f32x2_t{vg[0], vg[1]}) * p.oscale;
                    const f32x2_t hi = geglu_pair(f32x2_t{va[2], va[3]}, f32x2_t{vg[2], vg[3]}) * p.oscale;
                    if (m < p.M) *(unsigned*)((char*)p.C + (long)m * p.ldc + (n >> 1) + lq * 4) = pack4fp8(lo[0], lo[1], hi[0], hi[1]);
                }
            }
        } else {
        auto geglu_tile = [&](int a, int b) -> u32x2 {      // output tile of accumulator pair (a, a+1)
            const f32x4 va = acc[a][b], vg = acc[a + 1][b];
#ifdef GEGLU_NOGELU
            return u32x2{pack2bf(va[0] * vg[0], va[1] * vg[1]), pack2bf(va[2] * vg[2], va[3] * vg[3])};
#endif
            const f32x2_t lo = geglu_pair(f32x2_t{va[0], va[1]}, f32x2_t{vg[0], vg[1]});
            const f32x2_t hi = geglu_pair(f32x2_t{va[2], va[3]}, f32x2_t{vg[2], vg[3]});
            return u32x2{pack2bf(lo[0], lo[1]), pack2bf(hi[0], hi[1])};
        };
        auto geglu_narrow = [&](int a) {
            const int n = en0 + wn * WTN + a * 16;
            if (n >= p.N) return;
            const int no = (n >> 1) + lq * 4;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = em0 + wm * WTM + b * 16 + lrow;
                const u32x2 o = geglu_tile(a, b);
                if (m < p.M && !(SD_TUNE(p) & 8)) *(u32x2*)(p.C + (long)m * p.ldc + no) = o;
            }
        };
#pragma unroll
        for (int a = 0; a < TN; a += 4) {
            const int n = en0 + wn * WTN + a * 16;  // first packed column of the (pair of) pairs
            if (a + 3 < TN && wide_ok && n + 64 <= p.N) {
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = em0 + wm * WTM + b * 16 + lrow;
                    const u32x2 ox = geglu_tile(a, b), oy = geglu_tile(a + 2 < TN ? a + 2 : a, b);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(ox[0], oy[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(ox[1], oy[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    const int col = (n >> 1) + (lq & 1) * 16 + (lq >> 1) * 8;
                    if (m < p.M && !(SD_TUNE(p) & 8)) *(u32x4*)(p.C + (long)m * p.ldc + col) = o;
                }
            } else {
                geglu_narrow(a);
                if (a + 2 < TN) geglu_narrow(a + 2);
            }
        }
        }
    }
    if (!more_work) break;
    if (STAGES != 2) __syncthreads();   // 3-stage path re-runs its prologue: all LDS reads must be done
    }   // while (work)
}

// out[m, n..n+3] = sum_s slab[s][m][n..] + bias + bias2 + R   (deterministic split-K finish)
__global__ void splitk_reduce_kernel(const GemmArgs p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = p.N / 4;
    if (i >= (long)p.M * nq) return;
    const int m = (int)(i / nq), n = (int)(i - (long)m * nq) * 4;
    const long off = (long)m * p.N + n;
    f32x4 v = *(const f32x4*)(p.slab + off);
    for (int s = 1; s < p.splitk; ++s) v += *(const f32x4*)(p.slab + (long)s * p.M * p.N + off);
    if (p.wscale) v *= *(const f32x4*)(p.wscale + n) * p.xscale_inv;      // fp8 operands: dequantise the raw sums
    if (p.bias) v += *(const f32x4*)(p.bias + n);
    if (p.bias2) v += *(const f32x4*)(p.bias2 + n);
    if (p.R) {
        const u32x2 r = *(const u32x2*)(p.R + (long)m * p.ldr + n);
        v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
    }
    u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(u32x2*)(p.C + (long)m * p.ldc + n) = o;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int AMODE, int EPI, int DT = 0>
int launch(const GemmArgs& a0, hipStream_t stream) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    if (a.ldw == 0) a.ldw = a.K;
    if (EPI != EPI_STD || a.slab == nullptr || a.splitk < 1) a.splitk = 1;
    SD_REQUIRE(!a.defer_reduce || a.splitk > 1, "gemm: defer_reduce needs split-K (the consumer would read an unwritten slab)");
#ifdef SD_ABLATE
    static const int tune = getenv("SD_GEMM_TUNE") ? atoi(getenv("SD_GEMM_TUNE")) : 0;
    a.tune = tune;
#else
    a.tune = 0;
#endif
    // + 512 B per wave: (mean, rstd) slots of the LayerNorm fold (2-stage bf16 kernels)
    // + (c1 | c2) of the tile's BN columns
    constexpr int smem = STAGES * (BM + BN) * 128 + (STAGES == 2 && DT == 0 ? WAVES_M * WAVES_N * 512 + 2 * BN * 4 : 0);
    static_assert(smem <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    auto kern = gemm_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, AMODE, EPI, DT>;
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    int grid = a.tiles_m * a.tiles_n * a.splitk;
    constexpr int persistent = smem > 80 * 1024 ? kPersistentGrid / 2 : kPersistentGrid;   // workgroups that fit a CU
    if (STAGES == 2 && grid > persistent) grid = persistent;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES_M * WAVES_N * 64), smem, stream, a);
    if (a.splitk > 1 && !a.defer_reduce) sd_launch_splitk_reduce(a, stream);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

void sd_launch_splitk_reduce(const GemmArgs& a, hipStream_t stream) {
    const long n4 = (long)a.M * (a.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a);
}

// The 256-row, 8-wave, 3-stage (counted-vmcnt) tile, one workgroup per CU.  Measured on MI355X at
// the SD-1.5 shapes it ties or loses to two independent 128-row workgroups per CU (its 8 waves
// issue their LDS-DMA in lockstep after the barrier), so it is opt-in: SD_GEMM_BIG=1.
static int big_tile_mode() {
    static const char* env = getenv("SD_GEMM_BIG");
    return env ? atoi(env) : 0;
}
static bool big_tile_ok(int M, int N, int BN) {
    (void)M; (void)N; (void)BN;
    return big_tile_mode() == 1;
}

// Split-K factor for a (M, N, K) problem on the 128x160 tile: only when the tile grid cannot fill
// the 256 CUs and K is long enough to amortise the fp32 slab round trip.
// Rows of the plain GEMM's output tile: 64 x 160 when 128-row tiles would leave the 512 workgroup slots (2 per CU) less
// than 3/4 full (the 16x16 and 8x8 levels: M = 4096 / 1024), else 128 x 160.  SD_GEMM_SMALL=0: always 128.
// Not for a long K on >= 4096 rows (K = 0: unknown): 100 K tiles in ONE 64-row item lose to 128-row tiles with split-K 2
// (the merged ff.net.2 + proj_out of the 16x16 level, 4096 x 1280 x 6400: 99.8 -> 85.4 us; every K <= 2560 shape and
// the 1024-row level measured faster on 64 rows, tools/op_times.py with SD_GEMM_SMALL=0).
int sd_gemm_tile_rows(int M, int N, int K) {
    static const bool off = getenv("SD_GEMM_SMALL") && atoi(getenv("SD_GEMM_SMALL")) == 0;
    if (K >= 4096 && M >= 4096) return 128;
    return (!off && M > 64 && ((M + 127) / 128) * ((N + 159) / 160) < 384) ? 64 : 128;
}

int sd_gemm_splitk(int M, int N, int K, int rows) {
    static const char* env = getenv("SD_SPLITK");
    if (env) return atoi(env) > 1 ? atoi(env) : 1;
    if (rows == 0) rows = sd_gemm_tile_rows(M, N, K);
    const int tiles = ((M + rows - 1) / rows) * ((N + 159) / 160);
    const int KT = K / 64;
    if (tiles >= 448) return 1;           // already ~2 workgroups per CU
    int want = (512 + tiles - 1) / tiles; // aim at two workgroups per CU (measured: 740 -> 980 TF/s at 16x16)
    int maxs = KT / 12;                   // every split keeps >= 12 K tiles to amortise the slab round trip
    int s = want < maxs ? want : maxs;
    if (s > 8) s = 8;
    return s < 1 ? 1 : s;
}

static int check_fp8(const GemmArgs& a, const char* what) {
    SD_REQUIRE(a.wscale != nullptr && a.xscale_inv > 0.f, "%s fp8: weight scales / activation scale missing", what);
    SD_REQUIRE(a.K % 128 == 0 && a.K >= 128, "%s fp8: K=%d must be a positive multiple of 128", what, a.K);
    SD_REQUIRE(a.X2 == nullptr && a.K1 == a.K, "%s fp8: no second K segment", what);
    SD_REQUIRE(a.rows_per_batch == 0, "%s fp8: per-sample weights are bf16 only", what);
    SD_REQUIRE(((uintptr_t)a.X & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "%s fp8: operands must be 16-byte aligned", what);
    return 0;
}

static int check_stats(const GemmArgs& a, int epi) {
    if (!a.stats) return 0;
    SD_REQUIRE(epi == EPI_STD && a.M % 64 == 0 && !(a.splitk > 1 && a.slab) && !a.out_fp8,
               "producer statistics: plain epilogue, M %% 64 == 0 and no split-K (M=%d splitk=%d)", a.M, a.splitk);
    return 0;
}
static int check_headmajor(const GemmArgs& a, int epi) {
    if (!a.hm_C) return 0;
    SD_REQUIRE(epi == EPI_STD && a.KV && a.hm_C % 160 == 0 && a.N == 3 * a.hm_C && a.hm_tok % 128 == 0 && a.M % a.hm_tok == 0 &&
                   !(a.splitk > 1 && a.slab) && big_tile_mode() == 0 && sd_gemm_tile_rows(a.M, a.N) == 128,
               "head-major K / V: N = 3 C with C %% 160 == 0, tokens per sample %% 128 == 0, 128 x 160 tiles, no split-K");
    return 0;
}
static int check_ln(const GemmArgs& a, int epi) {
    if (a.rowstats) SD_REQUIRE(epi == EPI_STD && !(a.splitk > 1 && a.slab), "LayerNorm partials: plain epilogue without split-K");
    if (!a.ln_rs) return 0;
    SD_REQUIRE((epi == EPI_STD || epi == EPI_GEGLU || epi == EPI_SOFTMAX) && a.dt == 0 && !(a.splitk > 1 && a.slab) && a.ln_c1 && a.ln_np > 0 &&
               a.ln_np <= 16 && a.X2 == nullptr && a.K1 == a.K && a.K >= 128 && big_tile_mode() == 0 && a.bias && !a.R && !a.bias2,
               "LayerNorm fold: bf16 operands, std / GEGLU epilogue, one K segment, no split-K (epi=%d np=%d)", epi, a.ln_np);
    return 0;
}

int sd_launch_gemm(const GemmArgs& a, int epi, hipStream_t stream) {
    if (check_stats(a, epi) || check_ln(a, epi) || check_headmajor(a, epi)) return -1;
    if (a.dt == 1) {
        if (check_fp8(a, "gemm")) return -1;
        SD_REQUIRE(a.N % 4 == 0 && a.M > 0 && a.N > 0 && a.zero_page, "gemm fp8: bad problem");
        SD_REQUIRE(a.ldx % 16 == 0 && (a.ldw == 0 || a.ldw % 16 == 0), "gemm fp8: row strides must be multiples of 16 bytes");
        if (epi == EPI_GEGLU) {
            SD_REQUIRE(a.N % 32 == 0, "geglu gemm: N=%d must be a multiple of 32", a.N);
            SD_REQUIRE(!a.out_fp8 || (a.ldc % 4 == 0), "geglu gemm fp8 out: ldc=%ld must be a multiple of 4 bytes", a.ldc);
            if (a.N % 256 == 0) return launch<256, 256, 4, 2, 2, AMODE_GEMM, EPI_GEGLU, 1>(a, stream);
            return launch<128, 128, 2, 2, 2, AMODE_GEMM, EPI_GEGLU, 1>(a, stream);
        }
        SD_REQUIRE(epi == EPI_STD && !a.out_fp8, "gemm fp8: epilogue %d not built", epi);
        return launch<128, 160, 2, 2, 2, AMODE_GEMM, EPI_STD, 1>(a, stream);
    }
    SD_REQUIRE(!a.out_fp8, "gemm: fp8 output needs fp8 operands");
    SD_REQUIRE(a.K % 64 == 0 && a.K >= 64, "gemm: K=%d must be a positive multiple of 64", a.K);
    SD_REQUIRE(a.N % 4 == 0, "gemm: N=%d must be a multiple of 4", a.N);
    SD_REQUIRE(a.K1 % 64 == 0 && a.K1 <= a.K, "gemm: K1=%d must be a multiple of 64 and <= K", a.K1);
    SD_REQUIRE(a.K1 == a.K || a.X2 != nullptr, "gemm: second K segment needs X2");
    SD_REQUIRE(a.M > 0 && a.N > 0, "gemm: empty problem");
    SD_REQUIRE(a.zero_page != nullptr, "gemm: zero page missing");
    // (read per call, not cached: tests/test_ops_gpu.py compares the two kernels bit for bit inside one process)
    const char* lean_env = getenv("SD_GEMM_LEAN");
    const bool lean_off = lean_env && atoi(lean_env) == 0;
    if (epi == EPI_GEGLU) {
        SD_REQUIRE(a.N % 32 == 0, "geglu gemm: N=%d must be a multiple of 32", a.N);
        if (!lean_off && big_tile_mode() == 0 && sd_gemm_lean_applicable(a, 1)) return sd_launch_gemm_lean(a, 1, 256, stream);
        if (big_tile_ok(a.M, a.N, 128)) return launch<256, 128, 4, 2, 3, AMODE_GEMM, EPI_GEGLU>(a, stream);
        // 256 x 256 (one 8-wave workgroup per CU): half the LDS-fill bytes per flop of the 128 x 128 tile,
        // measured 5-20 % faster at every SD-1.5 GEGLU shape (N = 2560 / 5120 / 10240)
        if (big_tile_mode() != 3 && a.N % 256 == 0) return launch<256, 256, 4, 2, 2, AMODE_GEMM, EPI_GEGLU>(a, stream);
        return launch<128, 128, 2, 2, 2, AMODE_GEMM, EPI_GEGLU>(a, stream);
    }
    SD_REQUIRE(a.rows_per_batch == 0 || a.rows_per_batch % 128 == 0,
               "gemm: rows_per_batch=%d must be a multiple of the 128-row tile", a.rows_per_batch);
    if (epi == EPI_SOFTMAX) {
        SD_REQUIRE(a.N % 80 == 0 && a.sm_valid > 0 && a.sm_valid <= 80 && a.R == nullptr && (a.bias == nullptr || a.ln_rs),
                   "softmax gemm: N=%d must be a multiple of 80, 0 < sm_valid=%d <= 80, no bias/residual", a.N, a.sm_valid);
        SD_REQUIRE(!a.ln_per_sample || (a.ln_rs && a.rows_per_batch > 0), "gemm: per-sample LayerNorm vectors need per-sample weights");
        return launch<128, 160, 2, 2, 2, AMODE_GEMM, EPI_SOFTMAX>(a, stream);
    }
    if (a.rows_per_batch) return launch<128, 160, 2, 2, 2, AMODE_GEMM, EPI_STD>(a, stream);   // tiles must not straddle samples
    // the UNet's full-tile projections: the lean kernel (gemm_lean.hip; same tile and accumulation order, bit-identical
    // results, a fraction of the per-item instructions).  SD_GEMM_LEAN=0 keeps everything on gemm_kernel (A/B).
    if (!lean_off && big_tile_mode() == 0 && sd_gemm_lean_applicable(a, epi)) {
        const int rows = (!a.stats && !a.ln_rs && !a.hm_C) ? sd_gemm_tile_rows(a.M, a.N, a.K) : 128;
        return sd_launch_gemm_lean(a, 0, rows, stream);
    }
    if (big_tile_ok(a.M, a.N, 160)) return launch<256, 160, 4, 2, 3, AMODE_GEMM, EPI_STD>(a, stream);
    if (big_tile_mode() == 2) return launch<256, 160, 4, 2, 2, AMODE_GEMM, EPI_STD>(a, stream);
    // (the 64-row tile's waves own 32 rows: no GroupNorm block statistics, no LayerNorm-fold consumer on it)
    if (!a.stats && !a.ln_rs && sd_gemm_tile_rows(a.M, a.N, a.K) == 64) return launch<64, 160, 2, 2, 2, AMODE_GEMM, EPI_STD>(a, stream);
    return launch<128, 160, 2, 2, 2, AMODE_GEMM, EPI_STD>(a, stream);
}

int sd_launch_conv3x3(const GemmArgs& a, hipStream_t stream) {
    if (check_stats(a, EPI_STD)) return -1;
    if (a.dt == 1) {
        if (check_fp8(a, "conv3x3")) return -1;
        SD_REQUIRE(a.Cin % 128 == 0, "conv3x3 fp8: Cin=%d must be a multiple of 128 (pad the channels)", a.Cin);
    }
    SD_REQUIRE(a.Cin % 64 == 0, "conv3x3: Cin=%d must be a multiple of 64", a.Cin);
    if (a.subpix) {     // nearest-2x upsample + 3x3 conv as four 2x2 convs on the low-res input (GemmArgs::subpix)
        SD_REQUIRE(a.dt == 0 && a.K == 4 * a.Cin && a.N % 4 == 0 && a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && a.stride == 1 &&
                       a.up == 0 && (a.Hin * a.Win) % 64 == 0 && a.M % (4 * a.Hin * a.Win) == 0 && a.R == nullptr &&
                       a.zero_page != nullptr && a.w_batch_stride > 0 && !(a.splitk > 1 && a.slab),
                   "conv3x3 sub-pixel upsample: bf16, K = 4 Cin, low-res pixels per sample %% 64 == 0, no residual, no split-K");
        if (sd_conv_halo_subpix_applicable(a)) return sd_launch_conv3x3_halo(a, stream);     // the halo kernel's 4-tap mode
        // a tile must not straddle two phases: 64-row tiles for 8x8 inputs (no GroupNorm block statistics on that tile)
        if ((a.Hin * a.Win) % 128 != 0) {
            SD_REQUIRE(!a.stats, "conv3x3 sub-pixel upsample: block statistics need low-res pixels per sample %% 128 == 0");
            return launch<64, 160, 2, 2, 2, AMODE_CONV, EPI_STD>(a, stream);
        }
        return launch<128, 160, 2, 2, 2, AMODE_CONV, EPI_STD>(a, stream);
    }
    SD_REQUIRE(a.K == 9 * a.Cin, "conv3x3: K must be 9*Cin");
    SD_REQUIRE(a.N % 4 == 0, "conv3x3: Cout=%d must be a multiple of 4", a.N);
    SD_REQUIRE(a.stride == 1 || a.stride == 2, "conv3x3: stride %d", a.stride);
    SD_REQUIRE(!(a.up && a.stride != 1), "conv3x3: upsample needs stride 1");
    SD_REQUIRE(a.zero_page != nullptr, "conv3x3: zero page missing");
    const int hv = a.Hin << a.up, wv = a.Win << a.up;
    SD_REQUIRE(a.Hout == (hv + 2 - 3) / a.stride + 1 && a.Wout == (wv + 2 - 3) / a.stride + 1,
               "conv3x3: output size %dx%d inconsistent with input %dx%d stride %d up %d", a.Hout, a.Wout,
               a.Hin, a.Win, a.stride, a.up);
    SD_REQUIRE(a.M % (a.Hout * a.Wout) == 0, "conv3x3: M not a multiple of Hout*Wout");
    if (sd_conv_halo_applicable(a)) return sd_launch_conv3x3_halo(a, stream);
    if (a.dt == 1) return launch<128, 160, 2, 2, 2, AMODE_CONV, EPI_STD, 1>(a, stream);
    if (big_tile_ok(a.M, a.N, 160)) return launch<256, 160, 4, 2, 3, AMODE_CONV, EPI_STD>(a, stream);
    return launch<128, 160, 2, 2, 2, AMODE_CONV, EPI_STD>(a, stream);
}
