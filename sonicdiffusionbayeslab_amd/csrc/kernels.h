// Internal launcher interface shared by the kernel translation units and unet.hip.
// Every launcher validates its operand shapes on the host BEFORE launching (a faulting kernel
// can take the whole GPU host down), enqueues on the caller's stream only, never synchronises,
// and returns 0 / negative with the message available through sd_last_error().
#pragma once
#include "common.h"

struct GemmArgs {
    const bf16_t* X = nullptr;   // GEMM: [M, K1] rows (ldx);  CONV: NHWC input [B, Hin, Win, Cin]
    long ldx = 0;
    const bf16_t* X2 = nullptr;  // optional second K segment [M, K-K1] (channel concat), GEMM only
    long ldx2 = 0;
    int K1 = 0;                  // split point (== K when there is no second segment)
    const bf16_t* W = nullptr;   // [N, K] rows of stride ldw (conv: K = (slice, tap, channel))
    long ldw = 0;                // 0 = K (dense); a larger stride lets W be a column slice of an activation
    const float* bias = nullptr;   // [N] fp32, optional
    const float* bias2 = nullptr;  // [N] fp32, optional (time-embedding projection)
    const bf16_t* R = nullptr;     // residual [M, ldr], optional
    long ldr = 0;
    bf16_t* C = nullptr;           // output [M, ldc]
    long ldc = 0;
    int M = 0, N = 0, K = 0;
    // conv geometry
    int Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, stride = 1, up = 0;
    int splitk = 1;                   // > 1: K is split over blocks, fp32 partials go through `slab`
    float* slab = nullptr;            // [splitk, M, N] fp32 scratch (required when splitk > 1)
    // split-K only: leave the partial sums in `slab` and launch no splitk_reduce_kernel -- the consuming single-launch
    // GroupNorm finishes them (GroupNormArgs::slab: same sums in the same order, + bias + bias2 + R, and writes C as well)
    int defer_reduce = 0;
    const void* zero_page = nullptr;  // >= 16 zero bytes in device memory
    // per-sample W (cross-attention folded into two GEMMs): rows [b*rows_per_batch, (b+1)*rows_per_batch) of X use
    // W + b * w_batch_stride; rows_per_batch must be a multiple of the 128-row tile (0 = one W for all rows)
    long w_batch_stride = 0;
    int rows_per_batch = 0;
    int sm_valid = 0;                 // EPI 2: softmax over the first sm_valid of every 80 output columns
    int tiles_m = 0, tiles_n = 0;     // filled by the launcher
    int tune = 0;                     // experiment knobs, filled by the launcher from SD_GEMM_TUNE
    // GroupNorm statistics of the output from this kernel's epilogue (split-K = 1 only): [M / 64][N][2] fp32 partial
    // sums / sums of squares per 64-row block and channel (common.h::tile_channel_stats); null = off
    float* stats = nullptr;
    // LayerNorm folded into the consuming GEMM.  PRODUCER side (std epilogue, split-K = 1): rowstats = [2 tiles_n][M][2]
    // fp32, per row the (sum, sum of squares) of every N-wave's 80 output columns (common.h::tile_row_stats).
    // CONSUMER side: X holds the un-normalised rows, W = bf16(W * gamma), ln_c1[n] = sum_k W'[n][k], bias = W beta + b;
    // the epilogue turns the raw sums into  rstd_m * (acc - mean_m * c1[n]) + bias[n]  with mean / rstd of row m
    // from its ln_np partials ln_rs[part][M][2] (K = the normalised width).
    float* rowstats = nullptr;
    // q|k|v projection with HEAD-MAJOR K / V (std epilogue): columns [0, hm_C) (Q) go to C as usual, the 160-column tiles of
    // K (columns [hm_C, 2 hm_C)) and V are stored as KV[which][sample][head][token][40]: a 64-key tile of one head is one
    // contiguous 5 KiB block for the self-attention's LDS-DMA.  Needs head dim 40, hm_C % 160 == 0, hm_tok % 128 == 0.
    bf16_t* KV = nullptr;
    int hm_C = 0, hm_tok = 0;
    // CONV, nearest-2x upsample + 3x3 conv as FOUR 2x2 convs on the low-res input (sub-pixel phases (py, px) of the output):
    //   out[2y+py, 2x+px] = sum_{dy,dx in {0,1}} W4[ph][dy][dx] . in[y-1+py+dy, x-1+px+dx],  W4 = sums of the 3x3 taps that
    //   read the same low-res pixel -- 4/9 of the multiply-adds, exact.  Rows are ordered (sample, phase, low-res pixel);
    //   W holds [4 phases][Cout][Cin/64][4 taps][64] (w_batch_stride = one phase), K = 4 Cin, Hout = 2 Hin, M = 4 B Hin Win.
    int subpix = 0;
    const float* ln_rs = nullptr;
    const float* ln_c1 = nullptr;
    int ln_np = 0;
    float ln_eps = 1e-5f;
    int ln_per_sample = 0;            // per-sample weights (rows_per_batch > 0): ln_c1 and bias (c2) are [samples][N] as well
    // fp8-e4m3 operands (dt = 1): X and W hold OCP e4m3 bytes, K (and ldx / ldw / Cin) count fp8 elements and are
    // multiples of 128; the epilogue multiplies the fp32 sums by wscale[n] * xscale_inv before bias / residual.
    int dt = 0;
    const float* wscale = nullptr;    // [N] fp32: per-output-channel weight scale (W = W_fp8 * wscale[n])
    float xscale_inv = 1.0f;          // 1 / activation scale (X_fp8 = sat(X * xscale))
    int out_fp8 = 0;                  // GEGLU epilogue only: store e4m3 bytes (ldc in bytes) ...
    float oscale = 1.0f;              // ... of out * oscale
};

int sd_gemm_tile_rows(int M, int N, int K = 0);   // 64 or 128: M tile of the plain (std epilogue) GEMM (K = 0: not known)
// heuristic split factor (1 = none) for the std epilogue; rows = M tile of the kernel that will run (0 = the plain bf16 GEMM's choice)
int sd_gemm_splitk(int M, int N, int K, int rows = 0);   // heuristic split factor (1 = none) for the std epilogue
int sd_launch_gemm(const GemmArgs& a, int epi /*0 std, 1 geglu, 2 softmax over 80-column groups*/, hipStream_t stream);
int sd_launch_conv3x3(const GemmArgs& a, hipStream_t stream);
// gemm_lean.hip: the std-epilogue (epi 0; rows = 64 or 128: the M tile) and GEGLU (epi 1, 256 x 256 tile) GEMMs on full N
// tiles with buffer addressing
bool sd_gemm_lean_applicable(const GemmArgs& a, int epi);
int sd_launch_gemm_lean(const GemmArgs& a, int epi, int rows, hipStream_t stream);
void sd_launch_splitk_reduce(const GemmArgs& a, hipStream_t stream);   // slab -> C (+bias +bias2 +R)
// conv_halo.hip: LDS-resident-halo kernel for stride-1 convs on whole-row tiles
bool sd_conv_halo_applicable(const GemmArgs& a);
int sd_conv3x3_splitk(int M, int N, int Cin, int Hin, int Win, int stride, int up, int dt = 0);
bool sd_conv_halo_subpix_applicable(const GemmArgs& a);      // the sub-pixel upsampler on the halo kernel's 4-tap mode
int sd_launch_conv3x3_halo(const GemmArgs& a, hipStream_t stream);

// GroupNorm over NHWC (optionally a two-tensor channel concat) -> bf16 [B, HW, C1+C2]
struct GroupNormArgs {
    const bf16_t* x1 = nullptr; int C1 = 0;
    const bf16_t* x2 = nullptr; int C2 = 0;   // optional
    const float* gamma = nullptr; const float* beta = nullptr;  // [C1+C2]
    bf16_t* y = nullptr;
    // out_fp8: y holds e4m3 bytes of sat(out * oscale), rows of Cpad >= C1+C2 bytes (a multiple of 128; the
    // pad channels are written as zeros: they are the K tail of the fp8 GEMM / conv that consumes y)
    int out_fp8 = 0, Cpad = 0;
    float oscale = 1.0f;
    // statistics delivered by the producers of x1 / x2 ([B * HW / 64][C1 or C2][2], see GemmArgs::stats): the statistics
    // pass over the tensor is skipped (both must be given when C2 > 0; HW a multiple of 64)
    const float* stats1 = nullptr; const float* stats2 = nullptr;
    // Single-launch kernel only: x1 has NOT been written yet -- its producer ran split-K with GemmArgs::defer_reduce and left
    // `splitk` fp32 partial slabs [splitk][B * HW][C1].  The kernel forms x1 = bf16(sum_s slab[s] + sbias + sbias2 + sR) exactly
    // as splitk_reduce_kernel would (same order of additions), normalises THAT, and stores it to x1w for x1's other readers:
    // one launch and one pass over the slabs instead of two launches (round 5: 27 such pairs per SD-1.5 forward at UNet batch 16).
    const float* slab = nullptr; int splitk = 0;
    const float* sbias = nullptr; const float* sbias2 = nullptr;   // [C1] fp32, optional
    const bf16_t* sR = nullptr; long sldr = 0;                     // residual rows, optional
    bf16_t* x1w = nullptr;                                         // [B * HW][C1]
    float* partial = nullptr;   // workspace of sd_groupnorm_scratch_bytes(): [B,nsplit,groups,2] partials + [B,groups,2] stats
    int B = 0, HW = 0, groups = 32, nsplit = 0;
    float eps = 1e-5f;
    int silu = 0;
};
int sd_groupnorm_nsplit(int B, int HW);
bool sd_groupnorm_uses_small(int B, int HW, int C1, int C2, int groups);
bool sd_groupnorm_slab_ok(int B, int HW, int C1, int C2, int groups);     // ... and may finish a deferred split-K reduce
size_t sd_groupnorm_scratch_bytes(int B, int HW, int groups);
int sd_launch_groupnorm(const GroupNormArgs& a, hipStream_t stream);

// in-place row softmax of bf16 scores: p = softmax(scale * s) over `cols` (multiple of 8)
int sd_launch_softmax_rows(bf16_t* s, long rows, int cols, float scale, hipStream_t stream);
// VAE post_quant_conv: y[b,co,p] = sum_ci W[co,ci] * (x[b,ci,p] * in_scale) + bias[co]  (fp32 NCHW, 4 -> 4)
int sd_launch_pqconv(const float* x, const float* W, const float* bias, float* y, int B, int HW, float in_scale,
                     hipStream_t stream);

int sd_launch_layernorm(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int rows, int C,
                        float eps, hipStream_t stream);
// same, writing e4m3 bytes of sat(out * oscale) into rows of Cpad bytes (pad channels zero)
int sd_launch_layernorm_fp8(const bf16_t* x, const float* gamma, const float* beta, void* y, int rows, int C, int Cpad,
                            float eps, float oscale, hipStream_t stream);
// bf16 [rows, C] -> e4m3 [rows, Cpad] of sat(x * scale), pad zero (tests; operands whose producer has no fp8 epilogue)
// largest e4m3 magnitude code (byte & 0x7f) of `nbytes` e4m3 bytes -> atomicMax into *out (the caller zeroes it)
int sd_launch_amax_e4m3(const void* y, long nbytes, unsigned* out, hipStream_t stream);
int sd_launch_quantize_fp8(const bf16_t* x, void* y, long rows, int C, int Cpad, float scale, hipStream_t stream);

struct AttnArgs {
    const bf16_t* Q = nullptr; long ldq = 0;  // [B, Nq, heads*D] rows with stride ldq
    const bf16_t* K = nullptr; long ldk = 0;  // [B, Nk, ...]
    const bf16_t* V = nullptr; long ldv = 0;
    bf16_t* O = nullptr; long ldo = 0;
    int B = 0, heads = 0, Nq = 0, Nk = 0, D = 0;
    float scale = 0.f;
    const void* consts = nullptr;   // device page: 16 zero bytes at +0, the bf16 chunk {1,0,0,0,0,0,0,0} at +256
    // HEAD-MAJOR K / V ([B][heads][Nk][D] contiguous: a 64-key tile of one head is 64 * D * 2 contiguous bytes, so the
    // LDS-DMA pieces of the 64x64 self-attention touch 8 cache lines instead of ~26); 0 = token-major as Q (ldk / ldv)
    int kv_head_major = 0;
    // Q already carries scale * log2(e) (the UNet folds it into W_q before the weight's one rounding): the kernels take
    // the QK^T product as the exp2 argument as it is; `scale` is then ignored
    int q_prescaled = 0;
    // pipelined kernels: XCD-aware work order (all query blocks of a (sample, head) on one XCD); 0 = dispatch order (A/B: SD_ATTN_XCD=0)
    int xcd_order = 1;
};
int sd_launch_attention(const AttnArgs& a, hipStream_t stream);

// y[n] = sum_k W[n,k] * act_in(x[k]) + b[n]   (M = 1 path: timestep embedding MLP + projections)
int sd_launch_gemv(const float* x, const bf16_t* W, const float* b, float* y, int N, int K, int silu_in,
                   hipStream_t stream);
int sd_launch_timestep_sinusoid(float t, float* out, int dim, hipStream_t stream);
int sd_launch_f32_to_bf16(const float* src, bf16_t* dst, long n, hipStream_t stream);
// dst[b][c][r'] = src[b][r][c]; perm16: r' = r with bits 2 and 3 swapped (the k order in which a 32x32 MFMA
// accumulator tile is consumed as the next product's operand, see xattn.hip)
int sd_launch_transpose_bf16(const bf16_t* src, bf16_t* dst, int B, int R, int Cc, hipStream_t stream, int perm16 = 0);

int sd_launch_retile32(const bf16_t* src, bf16_t* dst, int B, int R, int K, int RT, hipStream_t stream);
int sd_launch_replicate(const void* src, void* dst, long bytes, int rep, hipStream_t stream);   // dst[r][i] = src[i], r < rep

// xattn.hip: fused prompt cross-attention  Y = R + sum_h softmax_L(X A_h) B_h + b_o  (one launch per block)
struct XattnArgs {
    const bf16_t* X = nullptr;    // [M, C] LayerNorm output
    const bf16_t* R = nullptr;    // [M, C] residual
    bf16_t* Y = nullptr;          // [M, C]
    // TILED operands (sd_launch_retile32): a 16-row x 64-byte DMA piece is one contiguous KiB
    const bf16_t* At = nullptr;   // [samples][C/32][640][32]: row (head, key slot) = scale * K_h[key] . W_q,h, zero rows for slots >= L
    const bf16_t* Bw = nullptr;   // [samples][C/32][20][32][32]: (channel tile, 32-slot slice, channel, slot in PERMUTED k order)
    const float* bias = nullptr;  // [C] to_out bias
    int M = 0, C = 0, rows_per_sample = 0, L = 0;
    float* rowstats = nullptr;    // [2 * channel slices][M][2]: LayerNorm partials of Y (GemmArgs::rowstats layout), null = off
    // LayerNorm (norm2) folded into the kernel: X holds the UN-normalised rows (usually X == R), ln_rs = their row partials
    // [ln_np][ln_rows][2] (sum, sum of squares; GemmArgs::rowstats of the producer; row m reads row m % ln_rows: a CFG pair
    // replicated after the producer ran shares them), At = rows of  scale K_h W_q,h diag(gamma)  CENTRED over the channel
    // (sum_c (x_c - mean) w_c = sum_c x_c (w_c - mean_c w): the row mean drops out of the product), ln_c2 [samples][640] fp32 =
    // the beta term of every key slot.  A score is then  rstd_m * (X . At)[m][n] + ln_c2[n].  null = X is already normalised.
    const float* ln_rs = nullptr; int ln_np = 0; long ln_rows = 0;
    const float* ln_c2 = nullptr;
    float ln_eps = 1e-5f;
    unsigned long long* stamps = nullptr;   // diagnostic: 8 s_memtime stamps per workgroup (SD_XATTN_STAMPS), else null
};
bool sd_xattn_fused_applicable(int rows_per_sample, int C, int heads, int L);
int sd_xattn_slices(int M, int C);     // channel slices of the launch (LayerNorm partials per row = 2 * slices)
int sd_launch_xattn_fused(const XattnArgs& a, hipStream_t stream);
int sd_launch_xattn_expand(const bf16_t* kv, bf16_t* out, int B, int L, int C, int NH, int col_off, float scale,
                           hipStream_t stream);
// clip.hip: CLIP text encoder pieces
int sd_launch_clip_embed(const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int rows, int L, int H,
                         int vocab, hipStream_t stream);
int sd_launch_clip_attention(const bf16_t* qkv, bf16_t* out, int B, int L, int H, int heads, hipStream_t stream);
int sd_launch_quick_gelu(bf16_t* x, long n, hipStream_t stream);
int sd_launch_bf16_to_f32(const bf16_t* src, float* dst, long n, hipStream_t stream);

// conv_in: NCHW fp32 latents [Bsrc,4,H,W] (batch index taken modulo Bsrc: CFG duplication is
// fused) -> NHWC bf16 [B,H,W,Cout]
int sd_launch_conv_in(const float* x, int Bsrc, const float* Wt /*[Cin*9][Cout] fp32*/, const float* bias,
                      bf16_t* y, int B, int H, int W, int Cin, int Cout, hipStream_t stream);
// conv_out: NHWC bf16 [B,H,W,Cin] -> NCHW fp32 [B,Cout,H,W]  (Cout <= 4)
int sd_launch_conv_out(const bf16_t* x, const bf16_t* Wp /*[Cout][9][Cin]*/, const float* bias, float* y, int B,
                       int H, int W, int Cin, int Cout, hipStream_t stream);

// fused CFG combine + linear multistep scheduler update (DDIM / DPM-Solver(++) / LCM)
struct StepCoef {
    float px, pe, p1, p2, pn;  // prev = px*x + pe*eps + p1*m1 + p2*m2 + p3*m3 + pn*noise
    float yx, ye;              // y2   = yx*x + ye*eps           (x0_pred / denoised)
    float mx, me;              // m0   = mx*x + me*eps           (history entry)
    float p3;                  // third history term (PNDM/PLMS order 4)
};
int sd_launch_sched_step(const float* eps, int cfg, float guidance, const float* x, const float* m1,
                         const float* m2, const float* m3, const float* noise, float* prev, float* y2, float* m_out,
                         StepCoef c, long n, hipStream_t stream);
