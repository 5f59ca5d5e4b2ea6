// 3x3 stride-1 convolution with an LDS-resident input halo (gfx950).
//
// The implicit-GEMM kernel of gemm_conv.hip stages the shifted input pixels of every tap again:
// per 64-channel slice 9 x (128 x 128 B) of X next to 9 x (160 x 128 B) of W, and its LDS fill
// (16-B `global_load_lds`, ~64 B/clk/CU) is as long as the MFMA work of the tile.  Here one
// workgroup owns 256 consecutive output pixels (whole image rows) x 160 output channels and
// keeps the input HALO of those pixels -- (rows + 2) x (width + 2) pixel slots of 128 B -- in
// LDS for a whole channel slice; the nine taps read it at shifted addresses.  Per slice that is
// <= 56 KiB of X once plus 9 x 20 KiB of W for 256 rows: 2.8x fewer fill bytes per flop.
//
//   LDS: 2 halo buffers (slice s / s+1) x 400 slots x 128 B + 3 W stages x 160 rows x 128 B = 160 KiB exactly,
//        one persistent workgroup of 8 waves (4 along pixels x 2 along channels) per CU.  W runs two K tiles
//        ahead (counted vmcnt): inside a UNet forward the weights come cold from HBM, and with one tile of
//        lead every K tile waited for them (3x3 conv time per forward 5.89 -> 5.61 ms).
//   Slots and W rows hold their eight 16-B chunks XOR-swizzled by (slot & 7) / (row & 7), applied to
//   the per-lane SOURCE address of the LDS-DMA; a fragment read of 16 consecutive pixels is then
//   bank-conflict free for every tap shift.
//   K order = (slice, tap, channel), weights packed [Cout][Cin/64][9][64] as for the implicit GEMM.
//   Zero padding and M/N tails read the zero page.
//   Split-K runs over channel slices (fp32 slabs + splitk_reduce_kernel of gemm_conv.hip).
//   DT = 1: fp8-e4m3 operands.  A 128-byte pixel slot / W row then holds 128 channels (a "slice" is 128 channels,
//   Cin padded to a multiple of 128 by the producer), the byte geometry of halo, W stages and DMA pieces is unchanged,
//   and one v_mfma_f32_16x16x128_f8f6f4 replaces the two bf16 k-steps of a tap: twice the flops per staged byte and
//   per matrix-core cycle.  Chunk swizzle f(slot) = bit1(slot) | (slot & 4) (conflict-free for the two 16-byte reads
//   of a lane's 32 K bytes at every tap shift); the epilogue dequantises with wscale[n] / xscale.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 160;
constexpr int TN = 5, WTN = 80;               // 16x16 tiles per wave along the channels: 80 channels
// NWV waves per workgroup: 8 = 4 along pixels x 2 along channels, a wave owns 64 pixels x 80 channels (TM = 4), two waves
// per SIMD; 4 = 2 x 2, 128 pixels x 80 channels (TM = 8), one wave per SIMD with up to 512 registers: 13 instead of 18
// fragment reads per 40 MFMAs (see the kernel).
constexpr int HSLOTS = 400;                   // 50 one-KiB DMA pieces; a wave issues 7 (8 waves) or 13 (4 waves) -- 6 / 2 re-issues
constexpr int HBYTES = HSLOTS * 128;
constexpr int WPIECES = BN / 8;               // 20 one-KiB pieces per W stage; a wave issues 3 (8 waves, 4 re-issues) or 5
constexpr int WBYTES = WPIECES * 1024;
constexpr int WSTAGES = 3;                    // W(kt+1) may still be in flight while tile kt is multiplied
constexpr int SMEM = 2 * HBYTES + WSTAGES * WBYTES;   // = 160 KiB exactly
static_assert(SMEM <= 160 * 1024, "halo tile does not fit the 160 KiB LDS");
constexpr unsigned NOSRC = 0xffffffffu;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int swz_of(int row, int dt) { return dt ? (((row >> 1) & 1) | (row & 4)) : (row & 7); }

// DIAG (timing ablations, WRONG results by design; instantiated in the SD_ABLATE build only and reached through
// sd_op_conv3x3_ablate, never from the environment; tools/conv_ab.py): 1 = the K loop never waits
// for its LDS-DMA (vmcnt), 2 = it issues no LDS-DMA at all, 4 = no tap barrier either -- what is left of the tap time says
// whether DMA latency, DMA issue or barrier skew parks the waves; 8 = no fragment reads either (the bare MFMA stream).
//
// NWV = 4 (bf16 only).  With 64 x 80 per wave a tap is 2 x 9 KiB of fragment reads per wave for 40 MFMAs: 8 waves read
// 144 KiB per tap and the LDS-DMA adds 26 KiB -- 1360 LDS cycles at 128 B/clk against 1280 cycles of MFMAs per SIMD, so the
// LDS co-limits the matrix pipe.  128 x 80 per wave needs 2 x 13 KiB for 80 MFMAs: 104 + 26 KiB = 1016 LDS cycles per tap.
// One wave per SIMD has no partner to hide the barrier and the first fragment reads of a tap, so the second k-step's
// MFMAs of a tap are issued after the NEXT tap's barrier from fragments carried in registers (same per-accumulator order
// as the plain loop: bit-identical).
//
// SUB = 1 (round 4): the nearest-2x upsample + 3x3 conv as FOUR 2x2 convs on the low-res input (GemmArgs::subpix; 4/9 of the
// multiply-adds).  The tile grid runs over the low-res pixels of ONE output phase (py, px); a work item is (pixel tile,
// phase, channel tile), a channel slice has FOUR taps -- halo offsets (py + ty - 1, px + tx - 1), ty, tx in {0, 1} -- with the
// phase's own weights ([4 phases][Cout][Cin/64][4 taps][64]), and the epilogue scatters row (b, y, x) to output pixel
// (2 y + py, 2 x + px).  The halo of a slice is still 7 pieces per wave, now spread 2-2-2-1 over the four taps (56 KiB of halo
// + 4 x 20 KiB of W per slice against the implicit-GEMM kernel's 4 x (32 + 20) KiB).  Same K order (slice, tap, channel) and
// the same 16x16x32 accumulation chains as the implicit-GEMM form: bit-identical results.
template <int DT, int DIAG = 0, int NWV = 8, int SUB = 0>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? 2 : 1) void conv_halo_kernel(const GemmArgs p) {
    static_assert(NWV == 8 || (NWV == 4 && DT == 0), "4-wave layout: bf16 only");
    static_assert(!SUB || (DT == 0 && DIAG == 0 && NWV == 8), "sub-pixel upsampler: bf16, 8 waves");
    constexpr int NTAP = SUB ? 4 : 9;             // taps = K tiles per channel slice
    constexpr int NT = NWV * 64;
    constexpr int WMW = NWV == 8 ? 4 : 2;         // waves along the pixels
    constexpr int TM = BM / WMW / 16, WTM = TM * 16;
    constexpr int HPIECES = NWV == 8 ? 7 : 13;    // halo DMA pieces per wave and slice
    constexpr int WPW = NWV == 8 ? 3 : 5;         // W DMA pieces per wave and stage
    constexpr bool CARRY = NWV == 4;
    constexpr int ESZ = DT ? 1 : 2;               // bytes per operand element; a slice = 128 / ESZ channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Persistent workgroups, one per CU, walking the work items (output tile x K split) with stride
    // gridDim.x.  XCD-aware order: workgroups of one XCD (blockIdx % 8) take consecutive items, n fastest,
    // so the n-tiles that share an input halo read it from the same L2.
    int work = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = work & 7, idx = work >> 3;
        work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nwork = ntiles * (SUB ? 4 : p.splitk);
    if (work >= nwork) return;
    const int S = (p.Cin * ESZ) >> 7;
    const int Mrows = SUB ? p.M >> 2 : p.M;      // rows of the tile grid (SUB: the low-res pixels, = the rows of one phase)

    // tile geometry: BM / PIX pieces of RW whole OUTPUT rows (more than one piece only when an image has
    // fewer than 256 pixels).  With the fused nearest-2x upsample (p.up = 1) the halo lives in the
    // low-resolution input: (RW / 2 + 2) x (Win + 2) slots, and a tap reads slot ((y + dy - 1) >> 1, ...).
    const int up = SUB ? 0 : p.up;
    const int Ho = SUB ? p.Hin : p.Hout, Wo = SUB ? p.Win : p.Wout, HWo = Ho * Wo;
    const int RW = min(Ho, BM / Wo);
    const int PW = p.Win + 2, PP = ((RW >> up) + 2) * PW;
    const int PIX = RW * Wo;                     // output pixels per piece
    const int HP = (BM / PIX) * PP;              // halo slots in use (<= HSLOTS, checked on the host)

    const char* Xb = (const char*)p.X;
    const char* Wg = (const char*)p.W;

    // ---- LDS-DMA source descriptors ---------------------------------------------------------------------
    // Item-independent part, computed once (runtime divisions by the halo pitch): for DMA piece j this lane
    // fills chunk position (L & 7) of halo slot L >> 3 = (piece i, halo row hy, halo column hx).
    //   hrel[j] = byte offset of that chunk relative to the piece's first input row (may be negative; a
    //   multiple of 16) | piece i in bits 0-1 | bit 2: first halo row | bit 3: last halo row.
    //   Slots beyond the halo and the left/right padding columns: HREL_ZERO.
    // Output width and image size are powers of two (checked on the host), so the per-item part is shifts.
    constexpr int HREL_ZERO = (int)0x80000000;
    const int lgW = __builtin_ctz(Wo), lgHW = __builtin_ctz(HWo);
    const int RWin = RW >> up;                   // input rows under one piece
    int hrel[HPIECES];
    // The 50 halo pieces do not divide over 8 waves x 7: waves >= 2 re-issue their piece 5 as piece 6 (same source,
    // same LDS destination -- harmless), so that every wave issues the same number of DMAs and the counted
    // vmcnt waits of the K loop are compile-time constants.
    const int jdup = wave < 2 ? HPIECES - 1 : HPIECES - 2;      // (50 pieces = 6 x 8 + 2 = 12 x 4 + 2)
#pragma unroll
    for (int j = 0; j < HPIECES; ++j) {
        const int L = (j == HPIECES - 1 ? jdup : j) * NT + tid;
        const int slot = L >> 3, cpos = L & 7;
        const int i = slot / PP, rem = slot - i * PP;
        const int hy = rem / PW, hx = rem - hy * PW;
        const bool ok = slot < HP && hx >= 1 && hx <= p.Win;
        hrel[j] = ok ? ((((hy - 1) * p.Win + (hx - 1)) * p.Cin * ESZ + ((cpos ^ swz_of(slot, DT)) << 4)) | i |
                        (hy == 0 ? 4 : 0) | (hy == RWin + 1 ? 8 : 0))
                     : HREL_ZERO;
    }

    // W stage pieces: wave w issues pieces w, w + 8 and w + 16; waves 4-7 have no third piece and re-issue w + 8
    auto wpiece = [&](int i) { return NWV == 4 ? wave + 4 * i : (i < 2 ? wave + 8 * i : (wave < 4 ? wave + 16 : wave + 8)); };

    // ---- per-item state: tile origin, K range, source offsets -------------------------------------------
    int split = 0, m0 = 0, n0 = 0, s_begin = 0, s_end = 0, phase = 0;
    unsigned hoff[HPIECES];                      // byte offsets into X; NOSRC = zero page
    unsigned woffs[WPW];                         // byte offsets into W
    auto setup = [&](int w) {
        int tile_n, tile_m;
        if (SUB) {           // no split-K
            // Item order (round 5): with 256 workgroups the 32 of one XCD take 32 consecutive items per round.  Pixel tile
            // slowest (the round-3 order: n fastest, then the four phases of a pixel tile) made those 32 items one or two pixel
            // tiles x EVERY (phase, channel tile): each XCD streamed the whole 4-phase weight tensor through its L2 every
            // round (counters: 630 MB fetched per launch for 48 MB of operands).  Now 32 consecutive items are a rectangle of
            // 8 pixel tiles x 4 (phase, channel tile) panels -- 8 halos + 4 weight panels per XCD-round, the minimum of
            // a * halo + (32 / a) * panel at both UNet shapes -- and an XCD keeps its panels from round to round where the
            // pixel tiles allow.  Pixel tiles beyond the last full block of 8 keep the old order.
            const int C = 4 * p.tiles_n, rect = (p.tiles_m >> 3) * (C >> 2) * 32;
            const bool rects = gridDim.x == 256 && p.tune == 0;        // tune = 1: SD_SUBPIX_ORDER=0, the round-3 order (A/B)
            int combo;
            if (rects && w < rect) {
                const int r = w >> 5, i = w & 31, bc = r % (C >> 2);
                tile_m = (r / (C >> 2)) * 8 + (i >> 2);
                combo = bc * 4 + (i & 3);
            } else {
                const int w2 = rects ? w - rect : w;
                tile_m = (rects ? (p.tiles_m & ~7) : 0) + w2 / C;
                combo = w2 % C;
            }
            tile_n = combo % p.tiles_n;
            phase = combo / p.tiles_n;
            split = 0; s_begin = 0; s_end = S;
        } else {
            split = w / ntiles;
            const int t = w - split * ntiles;
            tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
            s_begin = (int)((long)S * split / p.splitk);
            s_end = (int)((long)S * (split + 1) / p.splitk);
        }
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int j = 0; j < HPIECES; ++j) {
            const int mp = m0 + (hrel[j] & 3) * PIX;                   // first output pixel of the piece
            const int b = mp >> lgHW;
            const int ybase = (mp & (HWo - 1)) >> (lgW + up);          // its first input row
            const bool ok = hrel[j] != HREL_ZERO && mp < Mrows && !((hrel[j] & 4) && ybase == 0) &&
                            !((hrel[j] & 8) && ybase + RWin == p.Hin);
            const unsigned base = (unsigned)(((long)b * p.Hin + ybase) * p.Win * p.Cin * ESZ);
            hoff[j] = ok ? base + (unsigned)(hrel[j] & ~15) : NOSRC;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int pc = wpiece(i);
            const int n = n0 + pc * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_of(lane >> 3, DT);
            woffs[i] = n < p.N ? (unsigned)(((SUB ? (long)phase * p.w_batch_stride : 0l) + (long)n * p.ldw) * ESZ + (c << 4)) : NOSRC;
        }
    };
    // LDS-DMA pieces as BUFFER loads (round 4, as gemm_lean.hip): wave-uniform resource + this lane's 32-bit byte offset
    // (computed once per item) + a scalar offset per slice / K tile.  A piece costs SALU + one VMEM instruction and NO vector
    // ALU work -- the 64-bit pointer add and the two selects per piece of the flat form were ~24 of the ~55 VALU instructions
    // a wave issues per tap beside its 40 MFMAs (each of which holds the vector issue port for 8 of its 16 cycles).  Lanes
    // without a source (padding, tails: offset NOSRC) are out of range and read zeros; "nothing to fetch" (s / kk < 0) is a
    // resource of zero bytes.  No zero page.
    const unsigned xbytes = (unsigned)((long)(Mrows / HWo) * p.Hin * p.Win * p.Cin * ESZ);
    const unsigned wbytes = (unsigned)((long)p.N * p.ldw * ESZ * (SUB ? 4 : 1));
    auto issue_h = [&](int j, int s, char* hb) {  // s < 0: nothing to fetch, keeps the loop branch-free
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)Xb, 0, (int)(s >= 0 ? xbytes : 0u), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(hb + ((j == HPIECES - 1 ? jdup : j) * NWV + wave) * 1024), 16, (int)hoff[j],
                                                 s >= 0 ? s * 128 : 0, 0, 0);
    };
    auto issue_w1 = [&](int i, int kk, char* wb) {   // kk = slice * NTAP + tap; < 0: nothing to fetch
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)(kk >= 0 ? wbytes : 0u), 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(wb + wpiece(i) * 1024), 16, (int)woffs[i], kk >= 0 ? kk * 128 : 0, 0, 0);
    };
    auto issue_h_tap = [&](int tap, int s, char* hb) {      // the halo pieces that go out with tap `tap` of a slice
        if (SUB) {
            if (tap < 3) {
                issue_h(2 * tap, s, hb);
                issue_h(2 * tap + 1, s, hb);
            } else {
                issue_h(6, s, hb);
            }
        } else if (NWV == 8) {
            if (tap < 7) issue_h(tap, s, hb);
        } else if (tap < 5) {
            issue_h(2 * tap, s, hb);
            issue_h(2 * tap + 1, s, hb);
        } else if (tap < 8) {
            issue_h(tap + 5, s, hb);
        }
    };
    auto issue_w = [&](int kk, char* wb) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) issue_w1(i, kk, wb);
    };

    // ---- fragment addressing (the same for every item) --------------------------------------------------
    const int wm = wave % WMW, wn = wave / WMW;
    const int lrow = lane & 15, lq = lane >> 4;
    int hbase[TM], hrc[TM];                       // this lane's pixels: piece base slot (+1,+1), row << 16 | column
#pragma unroll
    for (int f = 0; f < TM; ++f) {
        const int pt = wm * WTM + f * 16 + lrow;
        const int i = pt / PIX, rem = pt - i * PIX;
        hrc[f] = (rem >> lgW) << 16 | (rem & (Wo - 1));
        hbase[f] = i * PP + PW + 1;
    }
    const int wfrag = (wn * WTN + lrow) * 128;
    // bf16: k-step 0 reads chunk lq ^ (row & 7), k-step 1 = ^ 64; fp8: chunks 2 lq and 2 lq + 1 (= ^ 16), each ^ f(row)
    const int wswz0 = DT ? ((2 * lq) ^ swz_of(lane, 1)) << 4 : (lq ^ (lane & 7)) << 4;

    auto frag_addr = [&](int f, int tp) -> int {     // this lane's pixel of tile f at tap tp: halo slot -> byte offset
        const int hp = hbase[f] + (((hrc[f] >> 16) + (tp / 3 - 1)) >> up) * PW + (((hrc[f] & 0xffff) + (tp % 3 - 1)) >> up);
        return hp * 128 + (DT ? (((2 * lq) ^ swz_of(hp, 1)) << 4) : ((lq ^ (hp & 7)) << 4));
    };

    auto frag_addr_sub = [&](int f, int t) -> int {  // SUB: tap t = (ty, tx) of this item's phase
        const int hp = hbase[f] + ((hrc[f] >> 16) + (phase >> 1) + (t >> 1) - 1) * PW + ((hrc[f] & 0xffff) + (phase & 1) + (t & 1) - 1);
        return hp * 128 + ((lq ^ (hp & 7)) << 4);
    };

    f32x4 acc[TN][TM];
    bf16x8 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    int xan[TM];                                  // CARRY: fragment addresses of the NEXT tap (computed under this tap's MFMAs)
#pragma unroll
    for (int f = 0; f < TM; ++f) xan[f] = frag_addr(f, 0);
    auto mfmas = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    };
    // fp8: all W fragments of the tap (TN x 32 B) + a two-deep ring of X fragments; the MFMAs walk the pixel tiles
    i32x8 wq[DT ? TN : 1], xq[2];
    auto ld32 = [&](const char* base, int off) -> i32x8 {
        return cat_u32x4(*(const u32x4*)(base + off), *(const u32x4*)(base + (off ^ 16)));
    };

    char* const Hb0 = smem;
    char* const Hb1 = smem + HBYTES;
    char* const Wb = smem + 2 * HBYTES;          // WSTAGES stages of WBYTES

    // Buffer parities run on across items: the halo buffer / W stage an item starts in is the one its
    // predecessor was not reading in its last K tile, so the next item's first halo and W stage can be in
    // flight under the predecessor's epilogue stores.
    int hsel = 0, wst = 0;                        // halo buffer / W stage of the tile about to be multiplied
    setup(work);
#pragma unroll
    for (int j = 0; j < HPIECES; ++j) issue_h(j, s_begin, Hb0);
    issue_w(s_begin * NTAP, Wb);
    issue_w(s_begin * NTAP + 1, Wb + WBYTES);     // every item has >= NTAP K tiles
    bool stores_pending = false;                  // exactly FULL_STORES stores were issued after that DMA
    constexpr int FULL_STORES = (TN / 2 + TN % 2) * TM;

    // Rejected after measurement (round 3, profiles/round3_notes.md; all bit-identical to this loop): carrying the second
    // k-step's fragments across the tap barrier so that a wave has MFMAs to issue the moment it leaves it -- for every wave
    // (+5 % time), or only for waves 4-7, the SIMD partners of waves 0-3, as two straight-line loops (+9 %) or with
    // wave-uniform branches inside the taps (+21 %: spills reloaded in the loop behind vmcnt(0)).  The fragment-read wait
    // and the DMA issue after the barrier are therefore NOT what idles the matrix pipe for half of a tap.
    while (true) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (CARRY) {                              // nothing carried into an item's first tap: zero fragments
#pragma unroll
            for (int f = 0; f < TM; ++f) xf1[f] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int a = 0; a < TN; ++a) wf1[a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }

        for (int s = s_begin; s < s_end; ++s) {
            const char* hcur = hsel ? Hb1 : Hb0;
            char* hnext = hsel ? Hb0 : Hb1;
            const bool more = s + 1 < s_end;
            // The 36 fragment addresses (9 taps x 4 pixel tiles) do not depend on the slice and hipcc hoists them out of
            // this loop (36 VGPRs).  bf16: keep the hoist -- measured 3-5 % faster than recomputing them per tap although
            // it costs 14 spilled descriptors (tools/ab_halo.sh, same box, interleaved).  fp8: the 32-byte fragments leave
            // no room (73 spills, reloads inside the tap loop drain the counted vmcnt pipeline), so there the values are
            // made opaque per slice and the addresses are recomputed under the MFMAs.
            // 4-wave layout: 72 hoisted addresses do not fit next to 104 fragment registers in the 256 arch VGPRs either, and
            // one wave per SIMD has the issue slots to recompute them under its 80 MFMAs per tap.
            if (DT || NWV == 4) {
#pragma unroll
                for (int f = 0; f < TM; ++f) asm volatile("" : "+v"(hbase[f]), "+v"(hrc[f]));
            }
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                // In-order vmcnt: tile kt's W stage (issued two K tiles ago) and, at tap 0, the slice's halo have
                // landed once only the DMAs of the PREVIOUS iteration -- its halo piece, if any, and the 3 pieces of
                // W(kt+1) -- may still be pending; on an item's first tile also the predecessor's epilogue stores,
                // which were issued after this item's prologue DMA.  After the barrier every wave is done reading
                // the W stage of tile kt-1 and the halo of slice s-1, which are refilled below.
                // (4 waves: 13 halo pieces per wave go out two per tap at taps 0-4 and one at taps 5-7, 5 W pieces every tap:
                // the previous tap issued 0 + 5 before tap 0, 2 + 5 before taps 1-5, 1 + 5 before taps 6-8.)
                if (SUB) {
                    // 7 halo pieces go out 2-2-2-1 with the four taps, 3 W pieces with every tap: the previous iteration issued
                    // 1 + 3 before tap 0 (the halo piece first: only the W pieces may be pending), 2 + 3 before taps 1-3
                    if (tap == 0) {
                        if (s == s_begin && stores_pending) wait_vmcnt<3 + FULL_STORES>();
                        else wait_vmcnt<3>();
                    } else {
                        wait_vmcnt<5>();
                    }
                } else if (DIAG == 0) {
                if (NWV == 8) {
                if (tap == 0) {
                    if (s == s_begin && stores_pending) wait_vmcnt<3 + FULL_STORES>();
                    else wait_vmcnt<3>();
                } else if (tap == 8) {
                    wait_vmcnt<3>();
                } else {
                    wait_vmcnt<4>();
                }
                } else {
                if (tap == 0) {
                    if (s == s_begin && stores_pending) wait_vmcnt<5 + FULL_STORES>();
                    else wait_vmcnt<5>();
                } else if (tap <= 5) {
                    wait_vmcnt<7>();
                } else {
                    wait_vmcnt<6>();
                }
                }
                }
                if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char* wcur = Wb + wst * WBYTES;
                char* wnext = Wb + (wst == 0 ? 2 : wst - 1) * WBYTES;      // stage of tile kt+2 = stage of tile kt-1
                int xa[TM];
#pragma unroll
                for (int f = 0; f < TM; ++f) xa[f] = CARRY ? xan[f] : SUB ? frag_addr_sub(f, tap) : frag_addr(f, tap);
                if (CARRY) {
                    // One wave per SIMD: nothing else fills the matrix pipe while this wave issues DMAs or computes
                    // addresses, so both are placed INSIDE the MFMA streams -- a DMA piece every 5 MFMAs of the carried
                    // group, the next tap's fragment addresses (they do not depend on the slice) under the second group.
                    if (!(DIAG & 8) || (tap == 0 && s == s_begin)) {        // (DIAG 8: some operand, read once per item)
#pragma unroll
                    for (int f = 0; f < TM; ++f) xf0[f] = *(const bf16x8*)(hcur + xa[f]);
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf0[a] = *(const bf16x8*)(wcur + wfrag + a * 2048 + wswz0);
                    }
                    if ((DIAG & 8) && tap == 0 && s == s_begin) {
#pragma unroll
                        for (int f = 0; f < TM; ++f) xf1[f] = xf0[f];
#pragma unroll
                        for (int a = 0; a < TN; ++a) wf1[a] = wf0[a];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int hs = more ? s + 1 : -1, kw = s * 9 + tap + 2 < s_end * 9 ? s * 9 + tap + 2 : -1;
#pragma unroll
                    for (int a = 0; a < TN; ++a) {
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[a], xf1[b], acc[a][b], 0, 0, 0);
                            const int idx = a * TM + b;
                            if (idx % 5 == 4 && !(DIAG & 2)) {          // pieces 0..7 of this tap: 5 of W, then the halo's
                                const int pc = idx / 5;
                                if (pc < WPW) issue_w1(pc, kw, wnext);
                                else if (tap < 5 && pc < WPW + 2) issue_h(2 * tap + (pc - WPW), hs, hnext);
                                else if (tap < 8 && pc == WPW) issue_h(tap + 5, hs, hnext);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DIAG & 8)) {
#pragma unroll
                    for (int f = 0; f < TM; ++f) xf1[f] = *(const bf16x8*)(hcur + (xa[f] ^ 64));
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf1[a] = *(const bf16x8*)(wcur + wfrag + a * 2048 + (wswz0 ^ 64));
                    }
#pragma unroll
                    for (int f = 0; f < TM; ++f) xan[f] = frag_addr(f, tap == 8 ? 0 : tap + 1);
                    mfmas(xf0, wf0);
                    // the second k-step is multiplied after the next barrier: its fragments must have left the LDS before this
                    // wave arrives there (stage and halo are refilled behind it).  The builtin, so that hipcc's wait
                    // bookkeeping knows.
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0)
                } else if (DT) {
#pragma unroll
                    for (int a = 0; a < TN; ++a) wq[a] = ld32(wcur, wfrag + a * 2048 + wswz0);
                    xq[0] = ld32(hcur, xa[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_h_tap(tap, more ? s + 1 : -1, hnext);
                    issue_w(s * 9 + tap + 2 < s_end * 9 ? s * 9 + tap + 2 : -1, wnext);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        if (b + 1 < TM) xq[(b + 1) & 1] = ld32(hcur, xa[b + 1]);
#pragma unroll
                        for (int a = 0; a < TN; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq[a], xq[b & 1], acc[a][b], 0, 0, 0, 0, 0, 0);
                    }
                } else {
                    if (!(DIAG & 8)) {
#pragma unroll
                    for (int f = 0; f < TM; ++f) xf0[f] = *(const bf16x8*)(hcur + xa[f]);
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf0[a] = *(const bf16x8*)(wcur + wfrag + a * 2048 + wswz0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DIAG & 2)) {
                    issue_h_tap(tap, more ? s + 1 : -1, hnext);
                    issue_w(s * NTAP + tap + 2 < s_end * NTAP ? s * NTAP + tap + 2 : -1, wnext);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DIAG & 8)) {
#pragma unroll
                    for (int f = 0; f < TM; ++f) xf1[f] = *(const bf16x8*)(hcur + (xa[f] ^ 64));
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf1[a] = *(const bf16x8*)(wcur + wfrag + a * 2048 + (wswz0 ^ 64));
                    } else if (tap == 0 && s == s_begin) {       // (MFMA-only ablation: some operand, read once per item)
#pragma unroll
                        for (int f = 0; f < TM; ++f) xf0[f] = xf1[f] = *(const bf16x8*)(hcur + xa[f]);
#pragma unroll
                        for (int a = 0; a < TN; ++a) wf0[a] = wf1[a] = *(const bf16x8*)(wcur + wfrag + a * 2048 + wswz0);
                    }
                    mfmas(xf0, wf0);
                    mfmas(xf1, wf1);
                }
                wst = wst == 2 ? 0 : wst + 1;
            }
            hsel ^= 1;
        }
        if (CARRY) mfmas(xf1, wf1);               // the last tap's second k-step

        // ---- epilogue, phase A: every LOAD the epilogue needs, folded into the accumulators now so that
        // no ordinary load is outstanding once the next item's LDS-DMA is in flight ----------------------
        // (accumulator layout as gemm_kernel: a lane owns 4 consecutive channels of one pixel per 16x16 tile)
        const int em0 = m0, en0 = n0, esplit = split, ephase = phase;
        // row of C that holds tile row m (SUB: low-res pixel (b, y, x) of phase (py, px) -> output pixel (2 y + py, 2 x + px))
        auto crow = [&](int m) -> long {
            if (!SUB) return m;
            const int b = m >> lgHW, ml = m & (HWo - 1);
            return ((long)b * p.Hout + 2 * (ml >> lgW) + (ephase >> 1)) * p.Wout + 2 * (ml & (Wo - 1)) + (ephase & 1);
        };
        if (DT && p.splitk == 1) {     // dequantise: per-output-channel weight scale / per-tensor activation scale
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
                const f32x4 sv = *(const f32x4*)(p.wscale + n) * p.xscale_inv;
#pragma unroll
                for (int b = 0; b < TM; ++b) acc[a][b] *= sv;
            }
        }
        if (p.splitk == 1) {
            if (p.R) {
                u32x2 rr[TN][TM];
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const int m = min(em0 + wm * WTM + b * 16 + lrow, Mrows - 1);
                        rr[a][b] = *(const u32x2*)(p.R + crow(m) * p.ldr + n);
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
            if (p.bias) {
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int n = min(en0 + wn * WTN + a * 16 + lq * 4, p.N - 4);
                    f32x4 bv = *(const f32x4*)(p.bias + n);
                    if (p.bias2) bv += *(const f32x4*)(p.bias2 + n);
#pragma unroll
                    for (int b = 0; b < TM; ++b) acc[a][b] += bv;
                }
            }
        }

        // ---- next item: descriptors + its first halo and W stage in flight before this item's stores ----
        work += gridDim.x;
        const bool more_work = work < nwork;
        const bool wide_ok = (p.ldc & 7) == 0;
        if (more_work) {
            setup(work);
            // The buffers refilled here were last read one slice / one K tile before the final one, i.e.
            // before a barrier every wave has passed: no extra barrier.
            // the two W stages refilled here received zero-page fills in this item's last two iterations: retire
            // them first (two DMAs to one LDS address must not be in flight together)
            wait_vmcnt<0>();
            char* hb = hsel ? Hb1 : Hb0;
#pragma unroll
            for (int j = 0; j < HPIECES; ++j) issue_h(j, s_begin, hb);
            issue_w(s_begin * NTAP, Wb + wst * WBYTES);
            issue_w(s_begin * NTAP + 1, Wb + (wst == 2 ? 0 : wst + 1) * WBYTES);
            stores_pending = p.splitk == 1 && wide_ok && (em0 + BM <= Mrows) && (en0 + BN <= p.N);
        }

        // ---- epilogue, phase B: stores only ----------------------------------------------------------------
        if (p.splitk > 1) {
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
        } else {
            // GroupNorm statistics of this tile for the consuming GroupNorm (64-row blocks = this wave's rows)
            if (p.stats) {
                // (SUB: the blocks of a sample are contiguous in the row order (sample, phase, low-res pixel) the consumer
                // expects of a sub-pixel conv: block = (b * 4 + phase) * HW / 64 + block within the phase)
                const int r0 = em0 + wm * WTM;
                const int blk0 = SUB ? ((((r0 >> lgHW) << 2) + ephase) << (lgHW - 6)) + ((r0 & (HWo - 1)) >> 6) : r0 >> 6;
#pragma unroll
                for (int hb64 = 0; hb64 < TM / 4; ++hb64)       // one 64-row block per 4 pixel tiles
                    tile_channel_stats<TN, TM>(acc, p.stats, blk0 + hb64, p.N, en0 + wn * WTN, p.N,
                                               em0 + wm * WTM, Mrows, lane, 4 * hb64, 4);
            }
            // v_permlane16_swap pairs two adjacent 16-channel tiles: 16 contiguous bytes per lane and store
            auto store_narrow = [&](int a) {
                const int n = en0 + wn * WTN + a * 16 + lq * 4;
                if (n >= p.N) return;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = em0 + wm * WTM + b * 16 + lrow;
                    const f32x4 v = acc[a][b];
                    u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                    if (m < Mrows) *(u32x2*)(p.C + crow(m) * p.ldc + n) = o;
                }
            };
#pragma unroll
            for (int a = 0; a < TN; a += 2) {
                const int nb = en0 + wn * WTN + a * 16;
                if (a + 1 < TN && wide_ok && nb + 32 <= p.N) {
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const int m = em0 + wm * WTM + b * 16 + lrow;
                        const f32x4 vx = acc[a][b], vy = acc[a + 1 < TN ? a + 1 : a][b];
                        const auto s0 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[0], vx[1]), pack2bf(vy[0], vy[1]), false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(pack2bf(vx[2], vx[3]), pack2bf(vy[2], vy[3]), false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        const int col = nb + (lq & 1) * 16 + (lq >> 1) * 8;
                        if (m < Mrows) *(u32x4*)(p.C + crow(m) * p.ldc + col) = o;
                    }
                } else {
                    store_narrow(a);
                    if (a + 1 < TN) store_narrow(a + 1);
                }
            }
        }
        if (!more_work) break;
    }
}

bool halo_enabled() {
    static const char* env = getenv("SD_CONV_HALO");
    return env ? atoi(env) != 0 : true;
}

}  // namespace

// The halo kernel takes stride-1 convs (optionally with the fused nearest-2x upsample) whose 256-pixel tiles
// are whole output rows and whose halo fits the 448 slots: every stride-1 conv of the SD-1.5 UNet.  Strided
// convs and wide images (the VAE decoder) stay on the implicit-GEMM kernel.  At width 8 a 16-pixel fragment
// spans two rows and its LDS read is 2-way bank conflicted on a few lanes; it still beats re-staging X.
bool sd_conv_halo_applicable(const GemmArgs& a) {
    if (!halo_enabled() || a.stride != 1) return false;
    const int H = a.Hin << a.up, W = a.Win << a.up;          // output = upsampled input size
    if (W < 8 || W > BM || BM % W) return false;
    const int RW = H < BM / W ? H : BM / W;
    const int PIX = RW * W;
    if (BM % PIX || (H * W) % PIX || (a.up && (RW & 1))) return false;
    if ((W & (W - 1)) || ((H * W) & (H * W - 1))) return false;   // the kernel decodes pixels with shifts
    if ((BM / PIX) * ((RW >> a.up) + 2) * (a.Win + 2) > HSLOTS) return false;
    const long ldw = a.ldw ? a.ldw : a.K;                     // (the launchers default ldw to K after this check)
    if ((long)a.M * a.Cin * 2 >= (1l << 32) || (long)a.N * ldw * 2 >= (1l << 32)) return false;
    if (a.dt == 1 && a.Cin % 128) return false;
    return true;
}

// The sub-pixel upsampler (GemmArgs::subpix: four 2x2 convs on the low-res input) on the halo kernel's 4-tap mode: the tile
// grid is the LOW-RES image; only where (pixel tiles x 4 phases x channel tiles) fills the 256 CUs -- the 8x8 -> 16x16
// upsampler (128 items at UNet batch 16) stays on the implicit-GEMM kernel's 64-row tiles.  SD_SUBPIX_HALO=0: never.
bool sd_conv_halo_subpix_applicable(const GemmArgs& a) {
    const char* env = getenv("SD_SUBPIX_HALO");         // per call: the tests compare the two kernels in one process
    const bool off = env && atoi(env) == 0;
    if (off || !a.subpix || a.dt != 0 || a.R != nullptr || (a.splitk > 1 && a.slab) || (a.Hin * a.Win) % 64 != 0) return false;
    GemmArgs g = a;
    g.up = 0; g.stride = 1; g.M = a.M / 4;
    if (!sd_conv_halo_applicable(g)) return false;
    const long ldw = a.ldw ? a.ldw : a.K;
    if ((long)a.N * 4 * ldw * 2 >= (1l << 32)) return false;              // 32-bit W offsets over the four phases
    if (a.w_batch_stride != (long)a.N * ldw) return false;                // the kernel's W range is 4 x N x ldw: phases packed back to back
    return (long)((g.M + BM - 1) / BM) * 4 * ((a.N + BN - 1) / BN) >= 256;
}

// Split-K over channel slices: one 8-wave workgroup per CU, so aim at >= 256 work items.
int sd_conv_halo_splitk(int M, int N, int Cin, int dt) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (tiles >= 192) return 1;
    const int S = Cin / (dt ? 128 : 64);
    int want = (256 + tiles - 1) / tiles;
    int maxs = tiles < 64 ? S / 2 : S / 4;       // >= 4 slices (36 K tiles) per split; >= 2 on tiny grids
    int s = want < maxs ? want : maxs;
    if (s > 8) s = 8;
    return s < 1 ? 1 : s;
}

// Split factor for a 3x3 conv, whichever kernel will run it (the plan sizes the slab with this).
int sd_conv3x3_splitk(int M, int N, int Cin, int Hin, int Win, int stride, int up, int dt) {
    static const char* env = getenv("SD_SPLITK");
    GemmArgs a;
    a.M = M; a.N = N; a.Cin = Cin; a.K = 9 * Cin; a.ldw = a.K; a.Hin = Hin; a.Win = Win; a.stride = stride; a.up = up; a.dt = dt;
    if (!env && sd_conv_halo_applicable(a)) return sd_conv_halo_splitk(M, N, Cin, dt);
    return sd_gemm_splitk(M, N, dt ? 9 * Cin / 2 : 9 * Cin, 128);      // the heuristic counts 128-byte K tiles
}

int sd_launch_conv3x3_halo(const GemmArgs& a0, hipStream_t stream) {
    GemmArgs a = a0;
    a.tiles_m = ((a.subpix ? a.M / 4 : a.M) + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    if (a.ldw == 0) a.ldw = a.K;
    if (a.slab == nullptr || a.splitk < 1) a.splitk = 1;
    SD_REQUIRE(!a.defer_reduce || a.splitk > 1, "conv3x3 halo: defer_reduce needs split-K (the consumer would read an unwritten slab)");
    if (a.subpix) {
        SD_REQUIRE(a.dt == 0 && a.splitk == 1 && a.R == nullptr && a.up == 0 && a.K == 4 * a.Cin && a.w_batch_stride > 0 &&
                       a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && a.M % (4 * a.Hin * a.Win) == 0 && a0.tune == 0,
                   "conv3x3 halo, sub-pixel upsampler: bf16, no split-K, no residual, K = 4 Cin");
        // the buffer range of W covers N * ldw * 4 elements: a phase stride other than N * ldw would put later phases
        // out of range, and the range check would then hand the kernel zeros without an error
        SD_REQUIRE(a.w_batch_stride == (long)a.N * a.ldw, "conv3x3 halo, sub-pixel upsampler: phase stride %ld != N * ldw = %ld",
                   a.w_batch_stride, (long)a.N * a.ldw);
        static bool sub_attr_set = false;
        if (!sub_attr_set) {
            SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 0, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
            sub_attr_set = true;
        }
        int g = a.tiles_m * a.tiles_n * 4;
        if (g > 256) g = 256;
        static const bool old_order = getenv("SD_SUBPIX_ORDER") && atoi(getenv("SD_SUBPIX_ORDER")) == 0;
        a.tune = old_order ? 1 : 0;
        hipLaunchKernelGGL((conv_halo_kernel<0, 0, 8, 1>), dim3(g), dim3(512), SMEM, stream, a);
        SD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int slices = a.Cin / (a.dt ? 128 : 64);
    SD_REQUIRE(a.splitk <= slices, "conv3x3 halo: splitk %d exceeds the %d channel slices", a.splitk, slices);
    static bool attr_set = false;
    if (!attr_set) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
#ifdef SD_ABLATE
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 15>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<0, 15, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
#endif
        attr_set = true;
    }
    int grid = a.tiles_m * a.tiles_n * a.splitk;
    if (grid > 256) grid = 256;                  // persistent: one 8-wave workgroup per CU
#ifdef SD_ABLATE
    // The SD_ABLATE build only (libsdhip_ablate.so): the DIAG instantiations (timing ablations, WRONG results by design) are
    // reached through sd_op_conv3x3_ablate (GemmArgs::tune) alone -- never from the environment; tune bit 8 (+ 256) or
    // SD_CONV_WAVES=4 selects the 4-wave layout (128 x 80 per wave, one wave per SIMD; bf16): bit-identical to the 8-wave
    // layout and 12 % slower at every UNet shape (round 3).
    const int tune = a0.tune;
    a.tune = tune;
    static const int env_waves = getenv("SD_CONV_WAVES") ? atoi(getenv("SD_CONV_WAVES")) : 8;
    const int waves = (tune & 256) ? 4 : env_waves;
    if (a.dt) hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(grid), dim3(512), SMEM, stream, a);
    else if (waves == 4 && (tune & 15) == 8) hipLaunchKernelGGL((conv_halo_kernel<0, 15, 4>), dim3(grid), dim3(256), SMEM, stream, a);
    else if (waves == 4 && (tune & 15) == 0) hipLaunchKernelGGL((conv_halo_kernel<0, 0, 4>), dim3(grid), dim3(256), SMEM, stream, a);
    else if ((tune & 15) == 1) hipLaunchKernelGGL((conv_halo_kernel<0, 1>), dim3(grid), dim3(512), SMEM, stream, a);
    else if ((tune & 15) == 2) hipLaunchKernelGGL((conv_halo_kernel<0, 3>), dim3(grid), dim3(512), SMEM, stream, a);
    else if ((tune & 15) == 4) hipLaunchKernelGGL((conv_halo_kernel<0, 7>), dim3(grid), dim3(512), SMEM, stream, a);
    else if ((tune & 15) == 8) hipLaunchKernelGGL((conv_halo_kernel<0, 15>), dim3(grid), dim3(512), SMEM, stream, a);
    else hipLaunchKernelGGL(conv_halo_kernel<0>, dim3(grid), dim3(512), SMEM, stream, a);
#else
    SD_REQUIRE(a0.tune == 0, "conv3x3 (halo): this build carries no ablation kernels (load libsdhip_ablate.so)");
    a.tune = 0;
    if (a.dt) hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(grid), dim3(512), SMEM, stream, a);
    else hipLaunchKernelGGL(conv_halo_kernel<0>, dim3(grid), dim3(512), SMEM, stream, a);
#endif
    if (a.splitk > 1 && !a.defer_reduce) sd_launch_splitk_reduce(a, stream);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
