/* libsdhip -- C ABI of the MI355X-native (gfx950) Stable Diffusion sampling hot path.
 *
 * The reference (Kotstantinovskiy/SonicDiffusionBayesLab) is pure Python over diffusers and has
 * no FFI of its own; each entry point below names the reference call site it replaces.  The
 * reference-side binding a maintainer would add (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / C++ types.
 *   - every function returns 0 on success, a negative code on failure; the message is available
 *     through sd_last_error() (thread-local).  No exception crosses the ABI.
 *   - `stream` is a hipStream_t passed as void*; work is only ever enqueued on that stream and
 *     the library never synchronises the device inside a forward/step call.
 *   - device buffers handed in are BORROWED for the duration of the call; the caller (PyTorch in
 *     this repo) owns all I/O tensors and the workspace.  UNet weights are owned by the handle.
 *   - activations are bf16 NHWC inside the library; latents / noise predictions cross the ABI as
 *     fp32 NCHW, exactly the tensors the reference loop holds (src/models.py:173-182,227-261).
 */
#ifndef SD_HIP_H
#define SD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sd_unet sd_unet;

/* diffusers UNet2DConditionModel config subset used by SD-1.5 */
typedef struct sd_unet_config {
    int sample_size;             /* latent H = W (64 for 512x512) */
    int in_channels;             /* 4 */
    int out_channels;            /* 4 */
    int num_levels;              /* 4 */
    int block_out_channels[8];   /* 320,640,1280,1280 */
    int layers_per_block;        /* 2 */
    int attn_levels[8];          /* 1,1,1,0 : level carries Transformer2DModel blocks */
    int cross_attention_dim;     /* 768 */
    int num_heads;               /* 8 (diffusers attention_head_dim=8 is used as head COUNT) */
    int norm_num_groups;         /* 32 */
    float norm_eps;              /* 1e-5 */
    int context_len;             /* 77 */
    /* Weight / operand type of the MFMA contractions (BASELINE configs[4]: "fp8 MFMA weights"):
     *   SD_DTYPE_BF16     - bf16 operands everywhere (v_mfma_f32_16x16x32_bf16);
     *   SD_DTYPE_FP8_E4M3 - OCP e4m3 weights with one fp32 scale per output channel, and e4m3 activations with a
     *                       static per-tensor scale written by the producing GroupNorm / LayerNorm / GEGLU epilogue,
     *                       for the 3x3 resnet convs, proj_in, the self-attention QKV projection and both
     *                       feed-forward GEMMs (v_mfma_f32_16x16x128_f8f6f4, fp32 accumulate, dequantised in the
     *                       epilogue).  Attention, the prompt cross-attention, to_out / proj_out / shortcut GEMMs,
     *                       the down/upsampler convs and conv_in / conv_out stay bf16.
     * fp8_act_scale_*: DEFAULT activation scales x_fp8 = sat(x * scale) (0 = 8 and 2) of tensors that
     * sd_unet_calibrate_fp8 / sd_unet_set_fp8_scale have not given a scale of their own. */
    int weight_dtype;
    float fp8_act_scale_norm;    /* GroupNorm(+SiLU) / LayerNorm outputs */
    float fp8_act_scale_ff;      /* GEGLU outputs (input of ff.net.2) */
} sd_unet_config;
enum { SD_DTYPE_BF16 = 0, SD_DTYPE_FP8_E4M3 = 1 };

const char* sd_last_error(void);
int sd_abi_version(void);

/* ---- UNet handle: replaces `self.unet` of StableDiffusionPipeline (src/models.py:227-235) ---- */
int sd_unet_create(const sd_unet_config* cfg, sd_unet** out);
void sd_unet_destroy(sd_unet* u);

/* Parameters use the diffusers state_dict names and layouts (conv OIHW, linear [out,in]), fp32 on
 * the HOST; the library repacks (OHWI bf16, fused QKV, GEGLU interleave ...) and uploads them in
 * sd_unet_finalize().  Replaces `from_pretrained(...).to(device)` for the UNet
 * (src/experiments/base_experiment.py:55-64). */
int sd_unet_num_params(const sd_unet* u);
int sd_unet_param_info(const sd_unet* u, int index, char* name, int name_cap, long long shape[4], int* ndim);
int sd_unet_load_param(sd_unet* u, const char* name, const float* host_data, long long numel);
int sd_unet_finalize(sd_unet* u);

/* Parity hook for the HOST side of sd_unet_finalize (weight repacking / fp8 quantisation): finalize packs into a host
 * staging blob first and uploads second, so on a box without a GPU it fails at the upload with the blob intact; this
 * copies `nbytes` of the packed item `key` (a diffusers parameter name; fp8 items: name + ".fp8" / ".scale") out of
 * it.  Returns the item's byte offset, < 0 on error (unknown key, blob already released after a successful upload). */
long long sd_unet_debug_packed(const sd_unet* u, const char* key, void* host_out, long long nbytes);

/* ---- fp8 activation scales (SD_DTYPE_FP8_E4M3 handles; no reference counterpart: the reference runs fp16,
 * configs/consistency_model_config.yaml:1-34).  Every e4m3 ACTIVATION tensor of the plan is named after the module that
 * writes it -- "<resnet>.norm1|norm2", "<Transformer2DModel>.norm", "<block>.norm1|norm3", "<block>.ff.net.0" (the GEGLU
 * product) -- and carries one per-tensor scale, x_fp8 = sat(x * scale); e4m3 is a floating-point format, so the scale only
 * positions the +-448 .. 2^-9 range over the tensor's values.
 *   sd_unet_calibrate_fp8: ONE forward on the caller's inputs (sd_unet_set_context first, as for sd_unet_forward); every
 *     producer of an e4m3 tensor is first run with a probe scale under which nothing saturates, the tensor's largest
 *     |value| is read back, and its scale becomes the largest power of two <= 448 / (margin * amax) -- amax is the
 *     maximum over all calibration calls so far (call it for a few timesteps).  Synchronises the stream ~125 times; all
 *     plans are rebuilt afterwards (same workspace layout).  margin in [1, 64].
 *   sd_unet_fp8_scale_count / _info: the named tensors known so far (names appear when a plan is built), their scale and
 *     observed amax (0 = never calibrated);  sd_unet_set_fp8_scale: restore a saved calibration. */
int sd_unet_calibrate_fp8(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch, float timestep,
                          float margin, void* workspace, long long workspace_bytes);
int sd_unet_fp8_scale_count(const sd_unet* u);
int sd_unet_fp8_scale_info(const sd_unet* u, int index, char* name, int name_cap, float* scale, float* amax);
int sd_unet_set_fp8_scale(sd_unet* u, const char* name, float scale);

/* Workspace the caller must provide for a given UNet batch (2*B with CFG).  `cache_branch_id`
 * < 0 disables the DeepCache plan; >= 0 reserves the cached tensors of that branch
 * (DeepCacheSDHelper.set_params, src/experiments/deep_cache.py:25-28). The SAME workspace must be
 * passed to set_context and to every forward of one sampling run. */
long long sd_unet_workspace_bytes(sd_unet* u, int unet_batch, int cache_branch_id);

/* Prompt conditioning: device fp32 [unet_batch, context_len, cross_attention_dim]
 * (`prompt_embeds` after the CFG concat, src/models.py:154-155).  Projects K/V of all
 * cross-attention layers once -- they are step-invariant. */
int sd_unet_set_context(sd_unet* u, void* stream, const float* encoder_hidden_states, int unet_batch,
                        int cache_branch_id, void* workspace, long long workspace_bytes);

enum { SD_CACHE_OFF = 0, SD_CACHE_FULL_AND_STORE = 1, SD_CACHE_SKIP = 2 };

/* eps = UNet(latent_model_input, t, encoder_hidden_states)  (src/models.py:217-235).
 * `latents` is fp32 NCHW [latent_batch,4,H,W]; with unet_batch = 2*latent_batch the CFG
 * duplication `torch.cat([latents]*2)` (src/models.py:217) is fused: the two halves of the batch then see the same latents
 * and timestep, so everything before the first prompt cross-attention (conv_in, down_blocks.0.resnets.0,
 * attentions.0 up to attn1.to_out) is computed once per latent and copied to both halves -- the results are those of
 * the duplicated batch (SD_CFG_DEDUP=0 in the environment computes the prefix twice instead).
 * `eps_out` is fp32 NCHW [unet_batch,4,H,W].  cache_mode selects the DeepCache plan
 * (full step that refreshes the cache / skip step that reuses it). */
int sd_unet_forward(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                    float timestep, float* eps_out, void* workspace, long long workspace_bytes, int cache_mode,
                    int cache_branch_id);

/* ---- AutoencoderKL decoder: replaces `self.vae.decode(latents / scaling_factor)`
 * (src/models.py:287-302).  `sd_vae` IS the `sd_unet` handle type: parameters are enumerated / loaded /
 * finalised and the workspace is sized through the sd_unet_* functions above (diffusers AutoencoderKL
 * names: post_quant_conv.*, decoder.*); cfg: sample_size = latent size, block_out_channels =
 * 128,256,512,512, layers_per_block = 2, in_channels = 4, out_channels = 3, norm_num_groups = 32.
 * decode: fp32 NCHW latents [batch,4,h,w] * latent_scale -> fp32 NCHW images [batch,3,8h,8w]. */
typedef struct sd_unet sd_vae;
int sd_vae_create(const sd_unet_config* cfg, sd_vae** out);
int sd_vae_decode(sd_vae* v, void* stream, const float* latents, int batch, float latent_scale, float* images_out,
                  void* workspace, long long workspace_bytes);

/* ---- CLIP text encoder: replaces `self.text_encoder(text_input_ids)[0]` inside
 * `encode_prompt` (src/models.py:139-155; transformers CLIPTextModel, quick_gelu, causal mask).  `sd_clip` IS the
 * `sd_unet` handle type: parameters (transformers names, `text_model.` prefix: embeddings.token_embedding.weight,
 * encoder.layers.N.self_attn.{q,k,v,out}_proj.*, layer_norm1/2.*, mlp.fc1/fc2.*, final_layer_norm.*) are
 * enumerated / loaded / finalised and the workspace is sized (unet_batch = prompts, cache_branch_id = -1) through the
 * sd_unet_* functions.  encode: device int32 token ids [batch, max_positions] -> device fp32
 * last_hidden_state [batch, max_positions, hidden_size] (what `prompt_embeds` is, src/models.py:139-150).
 * Tokenisation (CLIP BPE) is host-side text processing and stays in the host language. */
typedef struct sd_clip_config {
    int vocab_size;          /* 49408 */
    int hidden_size;         /* 768 */
    int num_layers;          /* 12 */
    int num_heads;           /* 12 */
    int intermediate_size;   /* 3072 */
    int max_positions;       /* 77 */
    float layer_norm_eps;    /* 1e-5 */
} sd_clip_config;
typedef struct sd_unet sd_clip;
int sd_clip_create(const sd_clip_config* cfg, sd_clip** out);
int sd_clip_encode(sd_clip* c, void* stream, const int* input_ids, int batch, float* hidden_out, void* workspace,
                   long long workspace_bytes);

/* Measurement hook for bench.py: the same forward with a hipEvent pair around every launch.  Per
 * op kind (0 sinusoid, 1 gemv, 2 conv_in, 3 groupnorm, 4 conv3x3, 5 gemm, 6 layernorm,
 * 7 attention, 8 conv_out; 16 conv3x3 with fp8 operands, 17 gemm with fp8 operands, 18 fused prompt cross-attention)
 * it returns summed
 * milliseconds, launch count, algorithmic FLOPs and algorithmic HBM bytes in arrays of SD_PROFILE_KINDS = 32
 * entries.  Synchronises the stream; never used inside a timed region. */
#define SD_PROFILE_KINDS 32
int sd_unet_forward_profiled(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                             float timestep, float* eps_out, void* workspace, long long workspace_bytes, int cache_mode,
                             int cache_branch_id, double kind_ms[SD_PROFILE_KINDS], long long kind_launches[SD_PROFILE_KINDS],
                             double kind_flops[SD_PROFILE_KINDS], double kind_bytes[SD_PROFILE_KINDS]);

/* The same measurement, one text line per launch: "op-index kind M N K milliseconds GFLOP MB" (kinds as above before the
 * 16+ regrouping; development: which shapes carry a group's time).  Returns the bytes written to `text`, < 0 on error. */
long long sd_unet_forward_op_times(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                                   float timestep, float* eps_out, void* workspace, long long workspace_bytes, int cache_mode,
                                   int cache_branch_id, char* text, long long cap);

/* Debug/parity hook: copy a named intermediate (bf16 NHWC) of the LAST full forward into `out`
 * as fp32; names: "conv_in", "down0".."down3", "mid", "up0".."up3".  Synchronises the stream. */
int sd_unet_debug_tensor(sd_unet* u, void* stream, const char* name, float* host_out, long long numel,
                         void* workspace, int unet_batch, int cache_branch_id);

/* ---- fused CFG combine + scheduler.step (src/models.py:238-261; src/schedulers.py:98-187) ----
 *   eps   = cfg ? eps[0:n] + guidance*(eps[n:2n]-eps[0:n]) : eps[0:n]
 *   prev  = coef[0]*x + coef[1]*eps + coef[2]*m1 + coef[3]*m2 + coef[9]*m3 + coef[4]*noise
 *   y2    = coef[5]*x + coef[6]*eps      (x0_pred / LCM "denoised"), optional
 *   m_out = coef[7]*x + coef[8]*eps      (multistep history entry), optional
 * One launch serves DDIM, DPM-Solver / DPM-Solver++ (orders 1-3), LCM and PNDM/PLMS (4 history
 * terms); the host computes the scalar coefficients from its sigma tables, so the kernel needs no
 * host sync.  m1..m3, noise, y2, m_out may be NULL. */
int sd_sched_step(void* stream, const float* eps, int cfg, float guidance, const float* x, const float* m1,
                  const float* m2, const float* m3, const float* noise, float* prev, float* y2, float* m_out,
                  const float coef[10], long long n);

/* ---- operator-level entry points (each is one hot kernel; used by the parity tests) ----------- */
/* C[M,N] = [X|X2][M,K] . W[N,K]^T + bias + bias2 + R ; epi=1: GEGLU on interleaved W (N -> N/2) */
int sd_op_gemm(void* stream, const void* X, long long ldx, const void* X2, long long ldx2, int K1, const void* W,
               const float* bias, const float* bias2, const void* R, long long ldr, void* C, long long ldc, int M,
               int N, int K, int epi);
/* Same GEMM with a PER-SAMPLE weight matrix: rows [b*rows_per_batch, (b+1)*rows_per_batch) use W + b*w_batch_stride
 * (elements; rows_per_batch a multiple of 128).  epi=2: row softmax over the first sm_valid of every 80 output
 * columns, the rest written as 0.  The two halves of the folded prompt cross-attention
 * (Y = R + sum_h softmax_77(X A_h) B_h, src/models.py:227 -> diffusers Attention with 77 keys) are these two calls. */
int sd_op_gemm_batched(void* stream, const void* X, long long ldx, const void* W, long long w_batch_stride,
                       int rows_per_batch, const float* bias, const void* R, long long ldr, void* C, long long ldc,
                       int M, int N, int K, int epi, int sm_valid);
/* The softmax form (epi 2) with the block's norm2 folded in -- the first GEMM of the two-GEMM prompt cross-attention at the
 * 16x16 level: X = un-normalised rows, W = per-sample operands scaled by gamma, c1 / c2 = per-sample [N] fp32 vectors (row
 * sums of the rounded W; beta term), rowstats [parts][M][2] = (sum, sum of squares) partials of X's rows:
 * P = softmax over every 80-column group of  rstd_m * (X W^T - mean_m * c1) + c2. */
int sd_op_gemm_batched_softmax_ln(void* stream, const void* X, long long ldx, const void* W, long long w_batch_stride,
                                  int rows_per_batch, void* C, long long ldc, int M, int N, int K, int sm_valid,
                                  const float* rowstats, int parts, const float* c1, const float* c2, float eps);
/* NHWC 3x3 conv, pad 1, stride 1|2, optional fused nearest-2x upsample; W is bf16
 * [Cout][Cin/64][3*3][64] (K runs over 64-channel slice, tap, channel) */
int sd_op_conv3x3(void* stream, const void* X, const void* W, const float* bias, const float* bias2, const void* R,
                  void* Y, int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample);
/* Upsample2D (nearest 2x, then 3x3 conv; diffusers resnet.py, the UNet's three upsamplers) as four 2x2 convs on the
 * LOW-RES input -- one per output sub-pixel phase -- with the 3x3 taps that read the same low-res pixel summed: 4/9 of
 * the multiply-adds, the same linear map.  W4 = bf16 [4 phases (py, px)][Cout][Cin/64][4 taps (dy, dx)][64];
 * needs Hin * Win % 128 == 0.  Y = [B, 2 Hin, 2 Win, Cout]. */
int sd_op_conv3x3_upsample_subpixel(void* stream, const void* X, const void* W4, const float* bias, void* Y, int B, int Hin,
                                    int Win, int Cin, int Cout);
/* The same followed by the GroupNorm(+SiLU) of the next resnet, as the plan runs the pair: the conv epilogue delivers the
 * per-64-row-block channel statistics (row order (sample, phase, low-res pixel)) and the GroupNorm skips its statistics pass.
 * Where (pixel tiles x 4 phases x channel tiles) >= 256 the conv runs on the halo kernel's 4-tap mode (csrc/conv_halo.hip,
 * SD_SUBPIX_HALO=0: the implicit-GEMM kernel; bit-identical).  Y: conv output, Yn: normalised output. */
int sd_op_conv3x3_upsample_subpixel_groupnorm(void* stream, const void* X, const void* W4, const float* bias, void* Y, int B,
                                              int Hin, int Win, int Cin, int Cout, const float* gamma, const float* beta,
                                              void* Yn, int groups, float eps, int silu);
int sd_op_groupnorm(void* stream, const void* x1, int C1, const void* x2, int C2, const float* gamma,
                    const float* beta, void* y, int B, int HW, int groups, float eps, int silu);
/* conv3x3 (stride 1) -> GroupNorm(+SiLU) the way the forward plan runs every resnet's conv -> norm pair
 * (diffusers ResnetBlock2D, reached from src/models.py:227): the conv's epilogue leaves per-64-pixel-block channel sums
 * and the GroupNorm takes its statistics from them instead of re-reading Y.  Y = conv output, Yn = normalised output. */
int sd_op_conv3x3_groupnorm(void* stream, const void* X, const void* W, const float* bias, const float* bias2,
                            const void* R, void* Y, int B, int Hin, int Win, int Cin, int Cout, const float* gamma,
                            const float* beta, void* Yn, int groups, float eps, int silu);
/* Small images (the single-launch GroupNorm: <= 256 pixels and <= 10240 values per group) take the same entry point: a
 * split-K conv then leaves its fp32 partial slabs to the GroupNorm kernel, which finishes the reduce (+ bias + bias2 + R, in
 * splitk_reduce's order of additions), stores Y and normalises it in ONE launch (SD_GN_SLAB=0: conv + reduce, then the
 * GroupNorm; bit-identical).  sd_op_conv3x3_splitk: the split factor the library picks for a 3x3 conv (1 = none). */
int sd_op_conv3x3_splitk(int M, int Cout, int Cin, int Hin, int Win, int stride, int upsample);
int sd_op_layernorm(void* stream, const void* x, const float* gamma, const float* beta, void* y, int rows, int C,
                    float eps);
int sd_op_attention(void* stream, const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                    long long ldv, void* O, long long ldo, int B, int heads, int Nq, int Nk, int D, float scale);
/* the same with HEAD-MAJOR K / V, [B][heads][Nk][D] contiguous (how the plan's fused q|k|v projection stores K and V at
 * the 64x64 level: a 64-key tile of one head is one contiguous 5 KiB block for the LDS-DMA); d = 40, Nk % 64 == 0 */
int sd_op_gemm_qkv_headmajor(void* stream, const void* X, long long ldx, const void* W, void* Q, void* KV, int M, int C,
                             int tokens, int K);   /* the producer: Q [M][C], KV [2][M/tokens][C/40][tokens][40] */
int sd_op_attention_headmajor(void* stream, const void* Q, long long ldq, const void* K, const void* V, void* O,
                              long long ldo, int B, int heads, int Nq, int Nk, int D, float scale);
/* causal self-attention of the CLIP text tower (transformers CLIPAttention, reached from src/models.py:139-155) on the fused
 * projection output qkv [B * L][3 H] (q | k | v, bf16) -> out [B * L][H]; L <= 128, head dim 64 (16 in the reduced tests) */
int sd_op_clip_attention(void* stream, const void* qkv, void* out, int B, int L, int H, int heads);
int sd_op_conv_in(void* stream, const float* x, int Bsrc, const float* Wt, const float* bias, void* y, int B, int H,
                  int W, int Cin, int Cout);
int sd_op_conv_out(void* stream, const void* x, const void* Wp, const float* bias, float* y, int B, int H, int W,
                   int Cin, int Cout);
int sd_op_time_embedding(void* stream, float t, const void* W1, const float* b1, const void* W2, const float* b2,
                         float* scratch, float* temb, int dim_in, int dim);

/* Fused prompt cross-attention of one transformer block (src/models.py:227-235 -> diffusers Attention over the 77 prompt
 * keys): Y = R + sum_h softmax_L(X A_h) B_h + b_o in ONE launch, probabilities kept in registers.  8 heads x 80 key
 * slots.  A^T row (head, slot) = scale * K_h[slot] . W_q,h over the C channels (zero rows for slots >= L); Bw row =
 * output channel, columns (head, slot) with bits 2 and 3 of the slot index swapped inside every group of 16 (the order in
 * which an MFMA accumulator tile is consumed as the next product's operand).  Both are passed TILED so that every
 * LDS-DMA piece of the kernel is one contiguous KiB: At [samples][C/32][640][32], Bw [samples][C/32][20][32][32] (bf16).
 * sd_unet_set_context builds both from the prompt.  M tokens, rows_per_sample tokens per sample (multiple of 128). */
/* LayerNorm folded into the GEMM that consumes it (BasicTransformerBlock norm1 -> attn1 q|k|v, norm3 -> ff GEGLU; diffusers
 * attention.py, reached from src/models.py:227).  The producer of the residual stream leaves per-row (sum, sum of squares)
 * partials of its stored bf16 output, [parts][M][2] fp32 (sd_op_ln_partials: kind 0 = a GEMM with N columns, kind 1 = the
 * fused cross-attention of M rows x N channels); the consumer runs on the UN-normalised rows with Wg = bf16(W * gamma),
 * c1[n] = sum_k Wg[n][k], c2[n] = sum_k W[n][k] beta[k] + b[n] and finishes  rstd_m (acc - mean_m c1[n]) + c2[n]. */
int sd_op_ln_partials(int kind, int M, int N);
int sd_op_gemm_rowstats(void* stream, const void* X, long long ldx, const void* W, const float* bias, const void* R,
                        long long ldr, void* C, long long ldc, int M, int N, int K, float* rowstats);
int sd_op_gemm_ln(void* stream, const void* X, long long ldx, const void* Wg, const float* c1, const float* c2,
                  const float* rowstats, int parts, float eps, void* C, long long ldc, int M, int N, int K, int epi);
/* the std-epilogue GEMM with every side input / output the UNet plan combines on it (second K segment, bias + bias2, residual,
 * LayerNorm row partials [2 N/160][M][2] and GroupNorm block statistics [M/64][N][2] of the stored output, the LayerNorm
 * fold, head-major K / V) -- the entry through which tests compare the lean projection kernel (csrc/gemm_lean.hip) with
 * the general one bit for bit (environment SD_GEMM_LEAN=0 selects the general kernel; results are identical) */
int sd_op_gemm_plan(void* stream, const void* X, long long ldx, const void* X2, long long ldx2, int K1, const void* W,
                    const float* bias, const float* bias2, const void* R, long long ldr, void* C, long long ldc, int M, int N,
                    int K, float* rowstats, float* stats, const float* ln_rs, int ln_parts, const float* ln_c1, float ln_eps,
                    void* KV, int hm_tokens);
int sd_op_xattn_fused_rowstats(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                               const float* bias, int M, int C, int rows_per_sample, int L, float* rowstats);
int sd_op_xattn_fused(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                      const float* bias, int M, int C, int rows_per_sample, int L);
/* The same with the block's norm2 folded in (how the plan runs it at the 64x64 / 32x32 levels): X holds the UN-normalised rows
 * (normally X == R), ln_rowstats [ln_parts][ln_rows][2] their (sum, sum of squares) partials (row m reads row m % ln_rows),
 * At = tiled rows of scale K_h W_q,h diag(gamma) centred over the channel, c2 [samples][640] fp32 the beta term of every key
 * slot: score = rstd_m (X . At)[m][n] + c2[n].  rowstats: optional partials of Y as in sd_op_xattn_fused_rowstats. */
int sd_op_xattn_fused_ln(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                         const float* bias, int M, int C, int rows_per_sample, int L, const float* ln_rowstats, int ln_parts,
                         long long ln_rows, const float* c2, float eps, float* rowstats);
/* diagnostic twin (tools/xattn_stamps.py; no reference counterpart): same result, and the kernel stores 8 s_memtime
 * stamps per workgroup to `stamps` (caller-owned device memory, 8 * (M / 128) * slices 64-bit words) */
/* timing ablations of the dominant kernel (the stride-1 3x3 conv with the LDS-resident halo); results are WRONG by design
 * and Y is scratch: ablate 0 = the product kernel, 1 = no LDS-DMA waits, 2 = no LDS-DMA, 4 = no tap barrier either, 8 = no
 * fragment reads either (the bare MFMA stream: the rate the matrix pipe sustains at the clock the chip holds under load).
 * + 256 (with 0 or 8): the 4-wave layout of the kernel -- 128 x 80 outputs per wave, one wave per SIMD; with 0 it is a
 * CORRECT kernel, bit-identical to the product layout, measured 12 % slower (profiles/round3_notes.md) and kept for A/B.
 * Every mode but 0 exists ONLY in the SD_ABLATE build of these sources (libsdhip_ablate.so, a second library next to
 * libsdhip.so): the product library carries none of the ablation kernels and returns an error for them. */
int sd_op_conv3x3_ablate(void* stream, const void* X, const void* W, void* Y, int B, int Hin, int Win, int Cin, int Cout,
                         int ablate);
int sd_op_xattn_fused_stamps(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                             const float* bias, int M, int C, int rows_per_sample, int L, unsigned long long* stamps);

/* ---- fp8-e4m3 operand path (SD_DTYPE_FP8_E4M3), operator level ------------------------------------------------
 * X, W hold OCP e4m3 bytes; K / Cin count fp8 elements and are multiples of 128 (zero padded); wscale [N] fp32 is the
 * per-output-channel weight scale, xscale the static activation scale: C = (X_fp8 . W_fp8^T) * wscale[n] / xscale
 * + bias + bias2 + R.  epi = 1: GEGLU on interleaved W; with out_fp8 the result is stored as e4m3 bytes of
 * sat(out * oscale) (ldc in bytes) -- the input of the next fp8 GEMM. */
int sd_op_gemm_fp8(void* stream, const void* X, long long ldx, const void* W, const float* wscale, float xscale,
                   const float* bias, const void* R, long long ldr, void* C, long long ldc, int M, int N, int K, int epi,
                   int out_fp8, float oscale);
/* W is e4m3 [Cout][Cin/128][3*3][128] */
int sd_op_conv3x3_fp8(void* stream, const void* X, const void* W, const float* wscale, float xscale, const float* bias,
                      const float* bias2, const void* R, void* Y, int B, int Hin, int Win, int Cin, int Cout, int stride,
                      int upsample);
/* producers: GroupNorm(+SiLU) / LayerNorm writing e4m3 rows of Cpad = roundup(C, 128) bytes (pad zero), and a plain
 * bf16 -> e4m3 conversion */
int sd_op_groupnorm_fp8(void* stream, const void* x1, int C1, const void* x2, int C2, const float* gamma,
                        const float* beta, void* y, int B, int HW, int groups, float eps, int silu, int Cpad, float oscale);
int sd_op_layernorm_fp8(void* stream, const void* x, const float* gamma, const float* beta, void* y, int rows, int C,
                        int Cpad, float eps, float oscale);
int sd_op_quantize_fp8(void* stream, const void* x_bf16, void* y_fp8, long long rows, int C, int Cpad, float scale);

#ifdef __cplusplus
}
#endif
#endif /* SD_HIP_H */
