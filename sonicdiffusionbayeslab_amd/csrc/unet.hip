// libsdhip host side: UNet handle (parameter packing, execution plan with lifetime-based
// workspace assignment, DeepCache plan filtering) and the C ABI declared in include/sd_hip.h.
//
// Replaces diffusers' UNet2DConditionModel.forward as called at src/models.py:227-235 of the
// reference, plus DeepCacheSDHelper's skip path (src/experiments/deep_cache.py:24-29) and the
// CFG + scheduler.step glue (src/models.py:238-261).
#include "../../include/sd_hip.h"
#include "common.h"
#include "kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void sd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sd_last_error(void) { return g_err; }
extern "C" int sd_abi_version(void) { return 3; }   // 2: sd_unet_config.weight_dtype, fp8 entry points; 3: round-2 fusion entry points

static void* g_zero_page = nullptr;
// grow-only device scratch for the operator-level entry points (tests / micro-benchmarks only;
// the UNet plan carries its own slabs inside the caller's workspace)
static void* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
static void* op_scratch(size_t bytes) {
    if (bytes > g_scratch_bytes) {
        (void)hipDeviceSynchronize();
        if (g_scratch) (void)hipFree(g_scratch);
        g_scratch = nullptr;
        g_scratch_bytes = 0;
        if (hipMalloc(&g_scratch, bytes) != hipSuccess) return nullptr;
        g_scratch_bytes = bytes;
    }
    return g_scratch;
}
static int ensure_zero_page() {
    if (g_zero_page) return 0;
    SD_CHECK_HIP(hipMalloc(&g_zero_page, 4096));
    SD_CHECK_HIP(hipMemset(g_zero_page, 0, 4096));
    const unsigned short one_chunk[8] = {0x3F80, 0, 0, 0, 0, 0, 0, 0};   // bf16 1.0 then zeros (attention ones column)
    SD_CHECK_HIP(hipMemcpy((char*)g_zero_page + 256, one_chunk, sizeof(one_chunk), hipMemcpyHostToDevice));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
namespace {

constexpr size_t NOFF = (size_t)-1;
constexpr int T_LATENTS = -2, T_EPS = -3;

struct ParamSpec {
    std::string name;
    std::vector<long long> shape;
    std::vector<float> data;
    bool loaded = false;
    long long numel() const {
        long long n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct Wrap {  // one DeepCache-wrapped module enclosing an op
    int type;  // 0 down, 1 mid, 2 up
    int block_i, layer_i;
};

enum OpKind { OP_SINUSOID, OP_GEMV, OP_CONV_IN, OP_GN, OP_CONV3, OP_GEMM, OP_LN, OP_ATTN, OP_CONV_OUT, OP_SOFTMAX, OP_PQCONV,
              OP_CLIP_EMBED, OP_CLIP_ATTN, OP_QGELU, OP_TO_F32, OP_XATTN, OP_REPLICATE };

struct Op {
    int kind = 0;
    int x1 = -1, x2 = -1, r = -1, out = -1, aux = -1, b2t = -1;
    size_t w = NOFF, b = NOFF, g = NOFF, be = NOFF;
    long b2idx = 0;
    int M = 0, N = 0, K = 0, K1 = 0, epi = 0;
    int B = 0, Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, stride = 1, up = 0;
    int C1 = 0, C2 = 0, HW = 0, silu = 0, nsplit = 0;
    float eps = 0.f;
    int heads = 0, D = 0, Nq = 0, Nk = 0;
    long ldq = 0, ldk = 0, ldv = 0, ldo = 0, qoff = 0, koff = 0, voff = 0;
    int silu_in = 0, splitk = 1;
    // split-K producer + single-launch GroupNorm as ONE reduce (fuse_deferred_reduce): the producer (CONV3 / GEMM) sets `defer`
    // and launches no splitk_reduce_kernel; the GroupNorm reads the producer's slabs (slab_t, slab_k of them) with its bias /
    // time-embedding row / residual and writes the producer's output tensor on the way (GroupNormArgs::slab)
    int defer = 0, slab_t = -1, slab_k = 0, slab_r = -1, slab_b2t = -1;
    size_t slab_b = NOFF;
    long slab_b2idx = 0;
    // generalised GEMM operands (VAE attention): W operand taken from an activation tensor, X taken
    // from the weight blob, element offsets into tensors and explicit row strides
    int wt = -1;
    size_t wx = NOFF;
    long xoff = 0, woff_el = 0, coff = 0, ldx_o = 0, ldw_o = 0, ldc_o = 0;
    float scale = 0.f;
    long wbs = 0;                 // per-sample W: batch stride (elements), rows per sample, softmax width
    int rpb = 0, sm_valid = 0;
    // fp8-e4m3 operands (SD_DTYPE_FP8_E4M3): this op's X and W are e4m3 (K / Cin padded to 128), wsc = offset of
    // the per-output-channel weight scales, xs = activation scale of its input; out_fp8: the op WRITES e4m3
    // (rows of Cpad bytes) with scale os.  Kalg: unpadded contraction length (algorithmic FLOPs).
    int dt = 0, out_fp8 = 0, Cpad = 0, Kalg = 0;
    int sname = -1;                   // fp8 producer: index into sd_unet::act_names of the tensor it writes (its scale = os)
    // GroupNorm statistics from the producer's epilogue: `stats` = tensor this op writes ([M/64][N][2] fp32),
    // s1 / s2 = the statistics tensors of a GroupNorm's sources (it then skips its statistics pass)
    int stats = -1, s1 = -1, s2 = -1;
    // LayerNorm fold: rs = row partials this op writes ([np][M][2] fp32); lnrs / lnnp / c1 = partials and column sums
    // this GEMM normalises with (its x1 is the un-normalised tensor, its weights carry gamma, its bias W beta + b)
    int subpix = 0;                   // CONV3 with up: four 2x2 convs on the low-res input (GemmArgs::subpix)
    int hm = 0;                       // GEMM: q|k|v with head-major K / V (HW = tokens per sample); ATTN: K / V are head-major
    int qps = 0;                      // ATTN: Q arrives multiplied by scale * log2 e (folded into W_q at pack time)
    int rs = -1, lnrs = -1, lnnp = 0;
    size_t c1 = NOFF;
    size_t wsc = NOFF;
    float xs = 1.f, os = 1.f;
    int nwrap = 0;
    Wrap wraps[3];
};

// Host staging buffer of the packed weights: ONE anonymous mapping reserved up front and populated by the kernel in bulk
// (MAP_POPULATE, transparent huge pages where available).  A std::vector paid ~20 us per 4 KiB first-touch fault here:
// 11-19 s of a 22 s finalize for the 2.3 GB UNet blob.
struct HostBlob {
    unsigned char* p = nullptr;
    size_t n = 0, cap = 0;
    unsigned char* data() { return p; }
    const unsigned char* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    bool reserve(size_t bytes) {
        if (bytes <= cap) return true;
        void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        if (q == MAP_FAILED) return false;
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        if (p) { memcpy(q, p, n); munmap(p, cap); }
        p = (unsigned char*)q; cap = bytes;
        return true;
    }
    void resize(size_t bytes) {             // (new bytes are zero: fresh anonymous pages)
        // a failed mapping surfaces as an error of sd_unet_finalize (which catches this), never as an abort of the host process
        if (bytes > cap && !reserve(std::max(bytes, cap * 2))) throw std::bad_alloc();
        n = bytes;
    }
    void release() { if (p) munmap(p, cap); p = nullptr; n = cap = 0; }
    ~HostBlob() { release(); }
    HostBlob() = default;
    HostBlob(const HostBlob&) = delete;
    HostBlob& operator=(const HostBlob&) = delete;
};

struct Tn {
    size_t bytes = 0;
    int def = -1, last = -1;
    bool persistent = false;
    bool ctx = false;            // written by sd_unet_set_context: same offset in every plan variant of a (batch, branch)
    size_t off = NOFF;
};

struct Plan {
    int UB = 0, branch = -1;
    int rep = 1;                          // 2: the prompt-independent prefix runs once per latent (CFG pair), see Builder::build
    std::vector<Tn> tensors;
    std::vector<Op> ops;
    std::vector<char> skipped;            // per op: skipped on a DeepCache skip step
    std::vector<int> ctx_kv;              // tensor id of the [UB*L, 2C] K|V cache per cross-attn layer
    std::vector<size_t> ctx_w;            // packed [2C, 768] weight offset per layer
    std::vector<int> ctx_c;               // C per layer
    int ctx_bf16 = -1;                    // bf16 copy of encoder_hidden_states
    // folded prompt cross-attention (per layer): A^T [UB][heads*80][C] and B [UB][C][heads*80], see transformer()
    struct Fold { int kv, at, bw, C; size_t wqT, wo; bool perm; int c2 = -1; size_t lnu = NOFF; int c1 = -1; size_t ones = NOFF; };   // perm: Bw in the fused kernel's k order; c2 >= 0: norm2 folded (wqT = the .ln weights, c2 = tensor of the beta terms)
    std::vector<Fold> ctx_fold;
    int ctx_fold_scratch = -1;            // masked K / V expansions [2][UB][heads*80][Cmax]
    std::map<std::string, int> taps;
    size_t total_bytes = 0;
};

}  // namespace

struct sd_unet {
    int kind = 0;   // 0 = UNet2DConditionModel, 1 = AutoencoderKL decoder, 2 = CLIP text encoder (same handle type)
    sd_unet_config cfg;
    sd_clip_config clip;
    std::vector<ParamSpec> params;
    std::unordered_map<std::string, int> pindex;
    std::unordered_map<std::string, size_t> woff;  // packed item -> byte offset into dweights
    HostBlob hblob;                                // host staging of the packed blob
    char* dweights = nullptr;
    bool finalized = false;
    bool debug_taps = false;
    bool fp8 = false;                  // cfg.weight_dtype == SD_DTYPE_FP8_E4M3
    float s_norm = 8.f, s_ff = 2.f;    // fp8 activation scales (GroupNorm / LayerNorm outputs, GEGLU outputs): the DEFAULTS
    // per-tensor activation scales (sd_unet_calibrate_fp8 / sd_unet_set_fp8_scale): every e4m3 activation tensor is named
    // after the module that writes it ("<resnet>.norm1", "<attn>.norm", "<block>.norm1|3", "<block>.ff.net.0"); a tensor
    // without an entry uses the default of its kind.  act_amax: largest |value| calibration has seen (0 = never calibrated).
    std::vector<std::string> act_names;
    std::unordered_map<std::string, int> act_index;
    std::vector<float> act_scale, act_amax;
    int act_id(const std::string& name, float dflt) {
        auto it = act_index.find(name);
        if (it != act_index.end()) return it->second;
        act_index[name] = (int)act_names.size();
        act_names.push_back(name); act_scale.push_back(dflt); act_amax.push_back(0.f);
        return (int)act_names.size() - 1;
    }
    std::map<std::tuple<int, int, int>, Plan> plans;     // (UNet batch, DeepCache branch, prefix replication)
    int last_rep = 1;                                    // variant of the last forward (sd_unet_debug_tensor)
    std::unordered_map<std::string, long> tproj_off;  // resnet prefix -> float index into tproj vector
    long tproj_total = 0;
};

namespace {

// ---------------------------------------------------------------------------------------------
// parameter enumeration (diffusers state_dict names)
// ---------------------------------------------------------------------------------------------
struct Enum {
    sd_unet* u;
    void add(const std::string& n, std::vector<long long> shape) {
        ParamSpec p;
        p.name = n;
        p.shape = std::move(shape);
        u->pindex[n] = (int)u->params.size();
        u->params.push_back(std::move(p));
    }
    void resnet(const std::string& p, int cin, int cout, int temb) {
        add(p + "norm1.weight", {cin});
        add(p + "norm1.bias", {cin});
        add(p + "conv1.weight", {cout, cin, 3, 3});
        add(p + "conv1.bias", {cout});
        add(p + "time_emb_proj.weight", {cout, temb});
        add(p + "time_emb_proj.bias", {cout});
        add(p + "norm2.weight", {cout});
        add(p + "norm2.bias", {cout});
        add(p + "conv2.weight", {cout, cout, 3, 3});
        add(p + "conv2.bias", {cout});
        if (cin != cout) {
            add(p + "conv_shortcut.weight", {cout, cin, 1, 1});
            add(p + "conv_shortcut.bias", {cout});
        }
    }
    void transformer(const std::string& p, int c, int ctx) {
        add(p + "norm.weight", {c});
        add(p + "norm.bias", {c});
        add(p + "proj_in.weight", {c, c, 1, 1});
        add(p + "proj_in.bias", {c});
        const std::string t = p + "transformer_blocks.0.";
        for (int i = 1; i <= 3; ++i) {
            add(t + "norm" + std::to_string(i) + ".weight", {c});
            add(t + "norm" + std::to_string(i) + ".bias", {c});
        }
        add(t + "attn1.to_q.weight", {c, c});
        add(t + "attn1.to_k.weight", {c, c});
        add(t + "attn1.to_v.weight", {c, c});
        add(t + "attn1.to_out.0.weight", {c, c});
        add(t + "attn1.to_out.0.bias", {c});
        add(t + "attn2.to_q.weight", {c, c});
        add(t + "attn2.to_k.weight", {c, ctx});
        add(t + "attn2.to_v.weight", {c, ctx});
        add(t + "attn2.to_out.0.weight", {c, c});
        add(t + "attn2.to_out.0.bias", {c});
        add(t + "ff.net.0.proj.weight", {8 * c, c});
        add(t + "ff.net.0.proj.bias", {8 * c});
        add(t + "ff.net.2.weight", {c, 4 * c});
        add(t + "ff.net.2.bias", {c});
        add(p + "proj_out.weight", {c, c, 1, 1});
        add(p + "proj_out.bias", {c});
    }
};

void enumerate_params(sd_unet* u) {
    const sd_unet_config& c = u->cfg;
    Enum e{u};
    const int c0 = c.block_out_channels[0], temb = 4 * c0, nl = c.num_levels;
    e.add("time_embedding.linear_1.weight", {temb, c0});
    e.add("time_embedding.linear_1.bias", {temb});
    e.add("time_embedding.linear_2.weight", {temb, temb});
    e.add("time_embedding.linear_2.bias", {temb});
    e.add("conv_in.weight", {c0, c.in_channels, 3, 3});
    e.add("conv_in.bias", {c0});
    int ch = c0;
    std::vector<int> skip_ch{c0};
    for (int i = 0; i < nl; ++i) {
        const int co = c.block_out_channels[i];
        const std::string bp = "down_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block; ++j) {
            e.resnet(bp + "resnets." + std::to_string(j) + ".", ch, co, temb);
            ch = co;
            if (c.attn_levels[i]) e.transformer(bp + "attentions." + std::to_string(j) + ".", co, c.cross_attention_dim);
            skip_ch.push_back(co);
        }
        if (i < nl - 1) {
            e.add(bp + "downsamplers.0.conv.weight", {co, co, 3, 3});
            e.add(bp + "downsamplers.0.conv.bias", {co});
            skip_ch.push_back(co);
        }
    }
    e.resnet("mid_block.resnets.0.", ch, ch, temb);
    e.transformer("mid_block.attentions.0.", ch, c.cross_attention_dim);
    e.resnet("mid_block.resnets.1.", ch, ch, temb);
    for (int i = 0; i < nl; ++i) {
        const int lev = nl - 1 - i, co = c.block_out_channels[lev];
        const std::string bp = "up_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            const int sc = skip_ch.back();
            skip_ch.pop_back();
            e.resnet(bp + "resnets." + std::to_string(j) + ".", ch + sc, co, temb);
            ch = co;
            if (c.attn_levels[lev]) e.transformer(bp + "attentions." + std::to_string(j) + ".", co, c.cross_attention_dim);
        }
        if (i < nl - 1) {
            e.add(bp + "upsamplers.0.conv.weight", {co, co, 3, 3});
            e.add(bp + "upsamplers.0.conv.bias", {co});
        }
    }
    e.add("conv_norm_out.weight", {c0});
    e.add("conv_norm_out.bias", {c0});
    e.add("conv_out.weight", {c.out_channels, c0, 3, 3});
    e.add("conv_out.bias", {c.out_channels});
}

// AutoencoderKL decoder (diffusers names): post_quant_conv + decoder.*  
// transformers CLIPTextModel state_dict names (4.48.0 layout, `text_model.` prefix)
std::string clip_layer(int i) { return "text_model.encoder.layers." + std::to_string(i) + "."; }

void enumerate_params_clip(sd_unet* u) {
    const sd_clip_config& c = u->clip;
    Enum e{u};
    const int H = c.hidden_size, I = c.intermediate_size;
    e.add("text_model.embeddings.token_embedding.weight", {c.vocab_size, H});
    e.add("text_model.embeddings.position_embedding.weight", {c.max_positions, H});
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = clip_layer(i);
        for (const char* n : {"k_proj", "v_proj", "q_proj", "out_proj"}) {
            e.add(p + "self_attn." + n + ".weight", {H, H});
            e.add(p + "self_attn." + n + ".bias", {H});
        }
        e.add(p + "layer_norm1.weight", {H}); e.add(p + "layer_norm1.bias", {H});
        e.add(p + "mlp.fc1.weight", {I, H}); e.add(p + "mlp.fc1.bias", {I});
        e.add(p + "mlp.fc2.weight", {H, I}); e.add(p + "mlp.fc2.bias", {H});
        e.add(p + "layer_norm2.weight", {H}); e.add(p + "layer_norm2.bias", {H});
    }
    e.add("text_model.final_layer_norm.weight", {H});
    e.add("text_model.final_layer_norm.bias", {H});
}

void enumerate_params_vae(sd_unet* u) {
    const sd_unet_config& c = u->cfg;
    Enum e{u};
    const int nl = c.num_levels, top = c.block_out_channels[nl - 1];
    auto resnet = [&](const std::string& p, int cin, int cout) {
        e.add(p + "norm1.weight", {cin}); e.add(p + "norm1.bias", {cin});
        e.add(p + "conv1.weight", {cout, cin, 3, 3}); e.add(p + "conv1.bias", {cout});
        e.add(p + "norm2.weight", {cout}); e.add(p + "norm2.bias", {cout});
        e.add(p + "conv2.weight", {cout, cout, 3, 3}); e.add(p + "conv2.bias", {cout});
        if (cin != cout) { e.add(p + "conv_shortcut.weight", {cout, cin, 1, 1}); e.add(p + "conv_shortcut.bias", {cout}); }
    };
    e.add("post_quant_conv.weight", {c.in_channels, c.in_channels, 1, 1});
    e.add("post_quant_conv.bias", {c.in_channels});
    e.add("decoder.conv_in.weight", {top, c.in_channels, 3, 3});
    e.add("decoder.conv_in.bias", {top});
    resnet("decoder.mid_block.resnets.0.", top, top);
    const std::string a = "decoder.mid_block.attentions.0.";
    e.add(a + "group_norm.weight", {top}); e.add(a + "group_norm.bias", {top});
    for (const char* n : {"to_q", "to_k", "to_v", "to_out.0"}) {
        e.add(a + n + ".weight", {top, top});
        e.add(a + n + ".bias", {top});
    }
    resnet("decoder.mid_block.resnets.1.", top, top);
    int ch = top;
    for (int i = 0; i < nl; ++i) {
        const int co = c.block_out_channels[nl - 1 - i];
        const std::string bp = "decoder.up_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            resnet(bp + "resnets." + std::to_string(j) + ".", ch, co);
            ch = co;
        }
        if (i < nl - 1) {
            e.add(bp + "upsamplers.0.conv.weight", {co, co, 3, 3});
            e.add(bp + "upsamplers.0.conv.bias", {co});
        }
    }
    e.add("decoder.conv_norm_out.weight", {ch}); e.add("decoder.conv_norm_out.bias", {ch});
    e.add("decoder.conv_out.weight", {c.out_channels, ch, 3, 3});
    e.add("decoder.conv_out.bias", {c.out_channels});
}

// ---------------------------------------------------------------------------------------------
// weight packing (host)
// ---------------------------------------------------------------------------------------------
inline unsigned short f32_to_bf16_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u = u + 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// f32 -> OCP e4m3fn byte, round to nearest even, saturating at +-448 (no infinities; 0x7f = NaN)
inline unsigned char f32_to_e4m3_host(float f) {
    if (f != f) return 0x7f;
    const unsigned char sign = std::signbit(f) ? 0x80 : 0;
    const float a = fabsf(f);
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0.015625f) {                                  // subnormal range: multiples of 2^-9
        const int q = (int)nearbyintf(a * 512.f);         // 0..8 (8 = the smallest normal)
        return sign | (unsigned char)q;
    }
    int e;
    const float m = frexpf(a, &e);                        // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;
    int q = (int)nearbyintf((m * 2.f - 1.f) * 8.f);       // 0..8
    if (q == 8) { q = 0; ++ex; }
    const int biased = ex + 7;
    if (biased > 15 || (biased == 15 && q == 7)) return sign | 0x7e;
    return sign | (unsigned char)((biased << 3) | q);
}

// host-side packing runs over up to 8 threads (0.86 G parameters: the single-threaded pack took ~20 s) -- of this RANK's
// share of the host cores: under a launcher every rank of the node packs its own replica at the same time
// (LOCAL_WORLD_SIZE ranks; 8 ranks x 8 threads on shared cores was round 3's start-up), SD_AMD_PACK_THREADS overrides
static int pack_threads() {
    static const int nt = [] {
        if (const char* e = getenv("SD_AMD_PACK_THREADS")) return std::max(1, std::min(64, atoi(e)));
        int ranks = 1;
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
        else if (const char* e2 = getenv("WORLD_SIZE")) ranks = std::max(1, atoi(e2));
        const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
        return std::max(1, std::min(8, cores / ranks));
    }();
    return nt;
}
template <class F>
static void parallel_for(long n, F&& body) {
    const int nt = (int)std::max(1l, std::min<long>(pack_threads(), n));
    if (nt == 1) { body(0l, n); return; }
    std::vector<std::thread> th;
    const long per = (n + nt - 1) / nt;
    for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { body(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    body(0l, std::min(n, per));
    for (auto& x : th) x.join();
}
// dst[i][k] += a[i] * row[k] for NB rows (the inner kernel of Packer::ff_out_merge); the AVX2 + FMA clone is picked at run
// time (the library is built for the generic x86-64 baseline)
template <int NB>
__attribute__((target("avx2,fma"))) static void axpy_rows_avx2(float* const* dst, const float* a, const float* __restrict__ row, int K) {
    for (int i = 0; i < NB; ++i) {
        float* __restrict__ d = dst[i];
        const float ai = a[i];
#pragma clang loop vectorize(enable) interleave(enable)
        for (int k = 0; k < K; ++k) d[k] += ai * row[k];
    }
}
template <int NB>
static void axpy_rows_base(float* const* dst, const float* a, const float* __restrict__ row, int K) {
    for (int i = 0; i < NB; ++i) {
        float* __restrict__ d = dst[i];
        const float ai = a[i];
#pragma clang loop vectorize(enable) interleave(enable)
        for (int k = 0; k < K; ++k) d[k] += ai * row[k];
    }
}

static double g_alloc_s = 0;
struct Packer {
    sd_unet* u;
    // rows [N][K] fp32 -> e4m3 [N][Kp] (K zero padded to Kp) + one fp32 scale per row (amax / 448)
    void quant_rows(const std::string& key, const float* w, int N, int K, int Kp) {
        size_t off = alloc(key + ".fp8", (size_t)N * Kp);
        size_t soff = alloc(key + ".scale", (size_t)N * 4);
        unsigned char* o = u->hblob.data() + off;
        float* sc = (float*)(u->hblob.data() + soff);
        for (int n = 0; n < N; ++n) {
            float amax = 0.f;
            for (int k = 0; k < K; ++k) amax = std::max(amax, fabsf(w[(size_t)n * K + k]));
            const float scale = amax > 0.f ? amax / 448.f : 1.f;
            sc[n] = scale;
            const float inv = 1.f / scale;
            for (int k = 0; k < K; ++k) o[(size_t)n * Kp + k] = f32_to_e4m3_host(w[(size_t)n * K + k] * inv);
            for (int k = K; k < Kp; ++k) o[(size_t)n * Kp + k] = 0;
        }
    }
    void fp8_same(const std::string& n, int N, int K) { quant_rows(n, P(n).data(), N, K, (K + 127) / 128 * 128); }
    // scale0: factor on the rows of the FIRST matrix (the self-attention's to_q carries softmax scale * log2 e, see qscale())
    void fp8_concat_rows(const std::string& key, const std::vector<std::string>& names, int K, float scale0 = 1.f) {
        std::vector<float> all;
        for (auto& n : names) all.insert(all.end(), P(n).begin(), P(n).end());
        for (size_t i = 0; i < P(names[0]).size(); ++i) all[i] *= scale0;
        quant_rows(key, all.data(), (int)(all.size() / K), K, (K + 127) / 128 * 128);
    }
    // OIHW -> e4m3 [O][Ip/128][tap][128] (Ip = I padded to 128) + per-output-channel scale
    void conv3_fp8(const std::string& n, int O, int I) {
        const auto& d = P(n);
        const int Ip = (I + 127) / 128 * 128;
        size_t off = alloc(n + ".fp8", (size_t)O * 9 * Ip);
        size_t soff = alloc(n + ".scale", (size_t)O * 4);
        unsigned char* o = u->hblob.data() + off;
        float* sc = (float*)(u->hblob.data() + soff);
        memset(o, 0, (size_t)O * 9 * Ip);
        for (int oc = 0; oc < O; ++oc) {
            float amax = 0.f;
            for (int k = 0; k < I * 9; ++k) amax = std::max(amax, fabsf(d[(size_t)oc * I * 9 + k]));
            const float scale = amax > 0.f ? amax / 448.f : 1.f, inv = 1.f / scale;
            sc[oc] = scale;
            for (int ic = 0; ic < I; ++ic)
                for (int t = 0; t < 9; ++t)
                    o[(((size_t)oc * (Ip / 128) + ic / 128) * 9 + t) * 128 + (ic % 128)] =
                        f32_to_e4m3_host(d[((size_t)oc * I + ic) * 9 + t] * inv);
        }
    }
    const std::vector<float>& P(const std::string& n) { return u->params[u->pindex.at(n)].data; }
    size_t alloc(const std::string& key, size_t bytes) {
        size_t off = (u->hblob.size() + 255) / 256 * 256;
        const auto t0 = std::chrono::steady_clock::now();
        u->hblob.resize(off + bytes);
        g_alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        u->woff[key] = off;
        return off;
    }
    void f32(const std::string& n) {
        const auto& d = P(n);
        size_t off = alloc(n, d.size() * 4);
        memcpy(u->hblob.data() + off, d.data(), d.size() * 4);
    }
    void bf16_same(const std::string& n) {
        const auto& d = P(n);
        size_t off = alloc(n, d.size() * 2);
        unsigned short* o = (unsigned short*)(u->hblob.data() + off);
        const float* src = d.data();
        parallel_for((long)d.size(), [=](long b, long e) { for (long i = b; i < e; ++i) o[i] = f32_to_bf16_host(src[i]); });
    }
    // OIHW -> [O][I/64][tap][64]: K index = (64-channel slice, tap, channel) as the conv kernel walks it
    void conv3(const std::string& n, int O, int I) {
        const auto& d = P(n);
        size_t off = alloc(n, d.size() * 2);
        unsigned short* o = (unsigned short*)(u->hblob.data() + off);
        const float* src = d.data();
        parallel_for(O, [=](long b, long e) {
            for (long oc = b; oc < e; ++oc)
                for (int ic = 0; ic < I; ++ic)
                    for (int t = 0; t < 9; ++t)
                        o[(((size_t)oc * (I / 64) + ic / 64) * 9 + t) * 64 + (ic % 64)] =
                            f32_to_bf16_host(src[((size_t)oc * I + ic) * 9 + t]);
        });
    }
    // nearest-2x upsample + 3x3 conv == four 2x2 convs on the low-res input (GemmArgs::subpix): phase (py, px) of the output
    // reads low-res rows {y - 1 + py, y + py}; the 3x3 taps that land on the same low-res pixel are summed (in fp32):
    //   py = 0: row 0 <- tap row 0, row 1 <- tap rows 1 + 2;   py = 1: row 0 <- tap rows 0 + 1, row 1 <- tap row 2
    // OIHW -> [4 phases][O][I/64][4 taps (dy, dx)][64]
    void conv3_subpixel(const std::string& n, int O, int I) {
        const auto& d = P(n);
        const size_t per = (size_t)O * I * 4;
        size_t off = alloc(n + ".sub", 4 * per * 2);
        unsigned short* o = (unsigned short*)(u->hblob.data() + off);
        auto lo = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2); };      // first 3x3 tap of 2x2 tap t
        auto hi = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2); };      // last one
        for (int ph = 0; ph < 4; ++ph)
            for (int oc = 0; oc < O; ++oc)
                for (int ic = 0; ic < I; ++ic)
                    for (int t = 0; t < 4; ++t) {
                        const int py = ph >> 1, px = ph & 1, dy = t >> 1, dx = t & 1;
                        float acc = 0.f;
                        for (int ty = lo(py, dy); ty <= hi(py, dy); ++ty)
                            for (int tx = lo(px, dx); tx <= hi(px, dx); ++tx) acc += d[((size_t)oc * I + ic) * 9 + ty * 3 + tx];
                        o[ph * per + (((size_t)oc * (I / 64) + ic / 64) * 4 + t) * 64 + (ic % 64)] = f32_to_bf16_host(acc);
                    }
    }
    void conv3_ohwi(const std::string& n, int O, int I) {  // OIHW -> [O][tap][I] (conv_out kernel)
        const auto& d = P(n);
        size_t off = alloc(n, d.size() * 2);
        unsigned short* o = (unsigned short*)(u->hblob.data() + off);
        for (int oc = 0; oc < O; ++oc)
            for (int ic = 0; ic < I; ++ic)
                for (int t = 0; t < 9; ++t)
                    o[((size_t)oc * 9 + t) * I + ic] = f32_to_bf16_host(d[((size_t)oc * I + ic) * 9 + t]);
    }
    void concat_rows(const std::string& key, const std::vector<std::string>& names, float scale0 = 1.f) {
        size_t total = 0;
        for (auto& n : names) total += P(n).size();
        size_t off = alloc(key, total * 2);
        unsigned short* o = (unsigned short*)(u->hblob.data() + off);
        for (auto& n : names)
            for (float v : P(n)) *o++ = f32_to_bf16_host(&n == &names[0] ? v * scale0 : v);
    }
    // LayerNorm folded into the GEMM that consumes it (GemmArgs::ln_rs): rows [N][K] fp32 -> key.ln = bf16(W * gamma),
    // key.c1[n] = sum_k of the ROUNDED row (what the kernel's raw sums contain per unit of the row mean),
    // key.c2[n] = sum_k W[n][k] beta[k] + bias[n]
    void ln_fold(const std::string& key, const float* w, int N, int K, const std::vector<float>& gamma,
                 const std::vector<float>& beta, const float* bias) {
        const size_t woff = alloc(key + ".ln", (size_t)N * K * 2);
        const size_t c1off = alloc(key + ".c1", (size_t)N * 4);
        const size_t c2off = alloc(key + ".c2", (size_t)N * 4);
        unsigned short* o = (unsigned short*)(u->hblob.data() + woff);
        float* c1 = (float*)(u->hblob.data() + c1off);
        float* c2 = (float*)(u->hblob.data() + c2off);
        for (int n = 0; n < N; ++n) {
            double s1 = 0.0, s2 = bias ? (double)bias[n] : 0.0;
            for (int k = 0; k < K; ++k) {
                const float wv = w[(size_t)n * K + k];
                const unsigned short r = f32_to_bf16_host(wv * gamma[k]);
                o[(size_t)n * K + k] = r;
                const unsigned bits = (unsigned)r << 16;
                float rf;
                memcpy(&rf, &bits, 4);
                s1 += rf;
                s2 += (double)wv * beta[k];
            }
            c1[n] = (float)s1;
            c2[n] = (float)s2;
        }
    }
    // ff.net.2 and proj_out are two linear maps with only a residual add between them:
    //   out = (ff W2^T + b2 + h2) Wpo^T + bpo + x  =  [ff | h2] . [Wpo W2 | Wpo]^T + (Wpo b2 + bpo) + x
    // -> ONE GEMM with a two-segment K (4C + C) instead of two launches and the h3 round trip.  The product Wpo W2 is
    // formed here once, in fp32 from the fp32 parameters, and rounded to bf16 like every other weight.
    void ff_out_merge(const std::string& p, const std::string& t, int C) {
        const auto& w2 = P(t + "ff.net.2.weight");      // [C][4C]
        const auto& b2 = P(t + "ff.net.2.bias");
        const auto& wpo = P(p + "proj_out.weight");     // [C][C]
        const auto& bpo = P(p + "proj_out.bias");
        const int K4 = 4 * C, KT = 5 * C;
        std::vector<float> prod((size_t)C * K4, 0.f);
        const int nthreads = pack_threads();
        const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
        auto work = [&](int tid) {
            constexpr int NB = 8;                        // output rows per pass over W2 (C is a multiple of 8)
            for (int n0 = tid * NB; n0 + NB <= C; n0 += nthreads * NB) {
                float* dst[NB];
                float a[NB];
                for (int i = 0; i < NB; ++i) dst[i] = &prod[(size_t)(n0 + i) * K4];
                for (int c = 0; c < C; ++c) {
                    for (int i = 0; i < NB; ++i) a[i] = wpo[(size_t)(n0 + i) * C + c];
                    if (avx2) axpy_rows_avx2<NB>(dst, a, &w2[(size_t)c * K4], K4);
                    else axpy_rows_base<NB>(dst, a, &w2[(size_t)c * K4], K4);
                }
            }
        };
        std::vector<std::thread> th;
        for (int i = 1; i < nthreads; ++i) th.emplace_back(work, i);
        work(0);
        for (auto& x : th) x.join();
        const size_t woff = alloc(p + "ff_out.weight", (size_t)C * KT * 2);
        const size_t boff = alloc(p + "ff_out.bias", (size_t)C * 4);
        unsigned short* o = (unsigned short*)(u->hblob.data() + woff);
        float* bo = (float*)(u->hblob.data() + boff);
        for (int n = 0; n < C; ++n) {
            for (int k = 0; k < K4; ++k) o[(size_t)n * KT + k] = f32_to_bf16_host(prod[(size_t)n * K4 + k]);
            for (int c = 0; c < C; ++c) o[(size_t)n * KT + K4 + c] = f32_to_bf16_host(wpo[(size_t)n * C + c]);
            double acc = bpo[n];
            for (int c = 0; c < C; ++c) acc += (double)wpo[(size_t)n * C + c] * b2[c];
            bo[n] = (float)acc;
        }
    }
    void geglu(const std::string& t, int C) {  // rows: every 32 = [16 value | 16 gate]
        const auto& w = P(t + "ff.net.0.proj.weight");
        const auto& b = P(t + "ff.net.0.proj.bias");
        const int H = 4 * C;
        size_t boff = alloc(t + "ff.geglu.bias", b.size() * 4);
        float* bo = (float*)(u->hblob.data() + boff);
        std::vector<float> perm(u->fp8 ? w.size() : 0);
        unsigned short* o = nullptr;
        if (!u->fp8) {
            const size_t woff = alloc(t + "ff.geglu.weight", w.size() * 2);     // (alloc may move the blob: take data() after it)
            o = (unsigned short*)(u->hblob.data() + woff);
        }
        for (int r = 0; r < 2 * H; ++r) {
            const int grp = r / 32, within = r % 32;
            const int src = within < 16 ? grp * 16 + within : H + grp * 16 + (within - 16);
            if (u->fp8) memcpy(&perm[(size_t)r * C], &w[(size_t)src * C], (size_t)C * 4);
            else for (int k = 0; k < C; ++k) o[(size_t)r * C + k] = f32_to_bf16_host(w[(size_t)src * C + k]);
            ((float*)(u->hblob.data() + boff))[r] = b[src];
        }
        (void)bo;
        if (u->fp8) quant_rows(t + "ff.geglu.weight", perm.data(), 2 * H, C, (C + 127) / 128 * 128);
        else {      // norm3 folded in: same packed row order
            std::vector<float> pw((size_t)2 * H * C), pb(2 * H);
            for (int r = 0; r < 2 * H; ++r) {
                const int grp = r / 32, within = r % 32;
                const int src = within < 16 ? grp * 16 + within : H + grp * 16 + (within - 16);
                memcpy(&pw[(size_t)r * C], &w[(size_t)src * C], (size_t)C * 4);
                pb[r] = b[src];
            }
            ln_fold(t + "ff.geglu.weight", pw.data(), 2 * H, C, P(t + "norm3.weight"), P(t + "norm3.bias"), pb.data());
        }
    }
    void resnet(const std::string& p, int cin, int cout) {
        f32(p + "norm1.weight"); f32(p + "norm1.bias");
        f32(p + "conv1.bias");
        f32(p + "norm2.weight"); f32(p + "norm2.bias");
        f32(p + "conv2.bias");
        if (u->fp8) { conv3_fp8(p + "conv1.weight", cout, cin); conv3_fp8(p + "conv2.weight", cout, cout); }
        else { conv3(p + "conv1.weight", cout, cin); conv3(p + "conv2.weight", cout, cout); }
        if (cin != cout) { bf16_same(p + "conv_shortcut.weight"); f32(p + "conv_shortcut.bias"); }
    }
    void transformer(const std::string& p, int c) {
        f32(p + "norm.weight"); f32(p + "norm.bias");
        f32(p + "proj_in.bias");
        const std::string t = p + "transformer_blocks.0.";
        for (int i = 1; i <= 3; ++i) { f32(t + "norm" + std::to_string(i) + ".weight"); f32(t + "norm" + std::to_string(i) + ".bias"); }
        // The self-attention's softmax scale and the exp -> exp2 factor live in W_q (fp32, before the one rounding every weight
        // gets): S = (c W_q x) . k comes out of the attention kernels' QK^T product in exp2 units (AttnArgs::q_prescaled)
        const float qs = 1.4426950408889634f / sqrtf((float)(c / u->cfg.num_heads));
        if (u->fp8) {
            fp8_same(p + "proj_in.weight", c, c);
            fp8_concat_rows(t + "attn1.qkv.weight", {t + "attn1.to_q.weight", t + "attn1.to_k.weight", t + "attn1.to_v.weight"}, c, qs);
            fp8_same(t + "ff.net.2.weight", c, 4 * c);
        } else {
            bf16_same(p + "proj_in.weight");
            concat_rows(t + "attn1.qkv.weight", {t + "attn1.to_q.weight", t + "attn1.to_k.weight", t + "attn1.to_v.weight"}, qs);
            {   // norm1 folded into the q|k|v projection
                std::vector<float> all;
                for (const char* n : {"attn1.to_q.weight", "attn1.to_k.weight", "attn1.to_v.weight"})
                    all.insert(all.end(), P(t + n).begin(), P(t + n).end());
                for (size_t i = 0; i < (size_t)c * c; ++i) all[i] *= qs;
                ln_fold(t + "attn1.qkv.weight", all.data(), 3 * c, c, P(t + "norm1.weight"), P(t + "norm1.bias"), nullptr);
            }
            bf16_same(t + "ff.net.2.weight");
        }
        bf16_same(t + "attn1.to_out.0.weight"); f32(t + "attn1.to_out.0.bias");
        bf16_same(t + "attn2.to_q.weight");
        {   // to_q transposed ([in][out]): the W operand of A^T = (scale K_h) . W_q,h of the folded cross-attention
            const auto& d = P(t + "attn2.to_q.weight");
            size_t off = alloc(t + "attn2.to_q.weight.T", d.size() * 2);
            unsigned short* o = (unsigned short*)(u->hblob.data() + off);
            for (int r = 0; r < c; ++r)
                for (int k = 0; k < c; ++k) o[(size_t)r * c + k] = f32_to_bf16_host(d[(size_t)k * c + r]);
        }
        if (!u->fp8) {
            // norm2 folded into the fused cross-attention (XattnArgs::ln_rs): A^T = (scale K_h) . W with
            //   W[c][j] = W_q[j][c] gamma[c] - (1 / C) sum_c' W_q[j][c'] gamma[c']      (rows c of the ".T" layout, CENTRED over c:
            //   sum_c (x_c - mean) w_c = sum_c x_c (w_c - mean_c w), so the kernel never needs the row mean), and
            //   u[j] = sum_c W_q[j][c] beta[c]: the beta term of key slot n is  (scale K_h[n]) . u  (set_context, fp32)
            const auto& d = P(t + "attn2.to_q.weight");
            const auto& gm = P(t + "norm2.weight");
            const auto& bt = P(t + "norm2.bias");
            size_t off = alloc(t + "attn2.to_q.weight.T.ln", d.size() * 2);
            size_t uoff = alloc(t + "attn2.to_q.lnu", (size_t)c * 4);
            size_t ooff = alloc(t + "attn2.to_q.ones", (size_t)c * 4);      // (x of the GEMV that sums the rounded rows of A^T: c1)
            for (int j = 0; j < c; ++j) ((float*)(u->hblob.data() + ooff))[j] = 1.0f;
            unsigned short* o = (unsigned short*)(u->hblob.data() + off);
            float* uu = (float*)(u->hblob.data() + uoff);
            for (int j = 0; j < c; ++j) {
                double m = 0.0, ub = 0.0;
                for (int cc = 0; cc < c; ++cc) {
                    m += (double)d[(size_t)j * c + cc] * gm[cc];
                    ub += (double)d[(size_t)j * c + cc] * bt[cc];
                }
                m /= c;
                uu[j] = (float)ub;
                for (int cc = 0; cc < c; ++cc) o[(size_t)cc * c + j] = f32_to_bf16_host((float)((double)d[(size_t)j * c + cc] * gm[cc] - m));
            }
        }
        if (!u->fp8)      // ... and into the plain to_q GEMM of the levels that run the 77-key flash kernel (the 8x8 level at UNet batch 16)
            ln_fold(t + "attn2.to_q.weight", P(t + "attn2.to_q.weight").data(), c, c, P(t + "norm2.weight"), P(t + "norm2.bias"), nullptr);
        concat_rows(t + "attn2.kv.weight", {t + "attn2.to_k.weight", t + "attn2.to_v.weight"});
        bf16_same(t + "attn2.to_out.0.weight"); f32(t + "attn2.to_out.0.bias");
        geglu(t, c);
        f32(t + "ff.net.2.bias");
        bf16_same(p + "proj_out.weight"); f32(p + "proj_out.bias");
        if (!u->fp8) ff_out_merge(p, t, c);
    }
};

// walks the architecture once; F gets (kind, prefix, cin, cout/c) callbacks in forward order
template <class FR, class FT>
void walk_blocks(const sd_unet_config& c, FR&& on_resnet, FT&& on_transformer) {
    const int nl = c.num_levels;
    int ch = c.block_out_channels[0];
    std::vector<int> skip_ch{ch};
    for (int i = 0; i < nl; ++i) {
        const int co = c.block_out_channels[i];
        const std::string bp = "down_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block; ++j) {
            on_resnet(bp + "resnets." + std::to_string(j) + ".", ch, co);
            ch = co;
            if (c.attn_levels[i]) on_transformer(bp + "attentions." + std::to_string(j) + ".", co);
            skip_ch.push_back(co);
        }
        if (i < nl - 1) skip_ch.push_back(co);
    }
    on_resnet("mid_block.resnets.0.", ch, ch);
    on_transformer("mid_block.attentions.0.", ch);
    on_resnet("mid_block.resnets.1.", ch, ch);
    for (int i = 0; i < nl; ++i) {
        const int lev = nl - 1 - i, co = c.block_out_channels[lev];
        const std::string bp = "up_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            const int sc = skip_ch.back();
            skip_ch.pop_back();
            on_resnet(bp + "resnets." + std::to_string(j) + ".", ch + sc, co);
            ch = co;
            if (c.attn_levels[lev]) on_transformer(bp + "attentions." + std::to_string(j) + ".", co);
        }
    }
}

int pack_vae(sd_unet* u) {
    const sd_unet_config& c = u->cfg;
    Packer pk{u};
    const int nl = c.num_levels, top = c.block_out_channels[nl - 1];
    pk.f32("post_quant_conv.weight"); pk.f32("post_quant_conv.bias");
    {   // conv_in: [O][I][3][3] -> Wt[k = ic*9+tap][O] fp32
        const auto& d = pk.P("decoder.conv_in.weight");
        const int O = top, I = c.in_channels;
        size_t off = pk.alloc("decoder.conv_in.weight", d.size() * 4);
        float* o = (float*)(u->hblob.data() + off);
        for (int oc = 0; oc < O; ++oc)
            for (int k = 0; k < I * 9; ++k) o[(size_t)k * O + oc] = d[(size_t)oc * I * 9 + k];
        pk.f32("decoder.conv_in.bias");
    }
    pk.resnet("decoder.mid_block.resnets.0.", top, top);
    const std::string a = "decoder.mid_block.attentions.0.";
    pk.f32(a + "group_norm.weight"); pk.f32(a + "group_norm.bias");
    pk.concat_rows(a + "qk.weight", {a + "to_q.weight", a + "to_k.weight"});
    {
        size_t off = pk.alloc(a + "qk.bias", (size_t)2 * top * 4);
        float* o = (float*)(u->hblob.data() + off);
        for (const char* n : {"to_q.bias", "to_k.bias"})
            for (float v : pk.P(a + n)) *o++ = v;
    }
    pk.bf16_same(a + "to_v.weight"); pk.f32(a + "to_v.bias");
    pk.bf16_same(a + "to_out.0.weight"); pk.f32(a + "to_out.0.bias");
    pk.resnet("decoder.mid_block.resnets.1.", top, top);
    int ch = top;
    for (int i = 0; i < nl; ++i) {
        const int co = c.block_out_channels[nl - 1 - i];
        const std::string bp = "decoder.up_blocks." + std::to_string(i) + ".";
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            pk.resnet(bp + "resnets." + std::to_string(j) + ".", ch, co);
            ch = co;
        }
        if (i < nl - 1) { pk.conv3(bp + "upsamplers.0.conv.weight", co, co); pk.f32(bp + "upsamplers.0.conv.bias"); }
    }
    pk.f32("decoder.conv_norm_out.weight"); pk.f32("decoder.conv_norm_out.bias");
    pk.conv3_ohwi("decoder.conv_out.weight", c.out_channels, ch); pk.f32("decoder.conv_out.bias");
    return 0;
}

int pack_clip(sd_unet* u) {
    const sd_clip_config& c = u->clip;
    Packer pk{u};
    pk.bf16_same("text_model.embeddings.token_embedding.weight");
    pk.bf16_same("text_model.embeddings.position_embedding.weight");
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string p = clip_layer(i), a = p + "self_attn.";
        pk.concat_rows(a + "qkv.weight", {a + "q_proj.weight", a + "k_proj.weight", a + "v_proj.weight"});
        {
            size_t off = pk.alloc(a + "qkv.bias", (size_t)3 * c.hidden_size * 4);
            float* o = (float*)(u->hblob.data() + off);
            for (const char* n : {"q_proj.bias", "k_proj.bias", "v_proj.bias"})
                for (float v : pk.P(a + n)) *o++ = v;
        }
        pk.bf16_same(a + "out_proj.weight"); pk.f32(a + "out_proj.bias");
        pk.f32(p + "layer_norm1.weight"); pk.f32(p + "layer_norm1.bias");
        pk.bf16_same(p + "mlp.fc1.weight"); pk.f32(p + "mlp.fc1.bias");
        pk.bf16_same(p + "mlp.fc2.weight"); pk.f32(p + "mlp.fc2.bias");
        pk.f32(p + "layer_norm2.weight"); pk.f32(p + "layer_norm2.bias");
    }
    pk.f32("text_model.final_layer_norm.weight"); pk.f32("text_model.final_layer_norm.bias");
    return 0;
}

int pack_all(sd_unet* u) {
    if (u->kind == 1) return pack_vae(u);
    if (u->kind == 2) return pack_clip(u);
    const sd_unet_config& c = u->cfg;
    Packer pk{u};
    const int c0 = c.block_out_channels[0], temb = 4 * c0, nl = c.num_levels;
    pk.bf16_same("time_embedding.linear_1.weight"); pk.f32("time_embedding.linear_1.bias");
    pk.bf16_same("time_embedding.linear_2.weight"); pk.f32("time_embedding.linear_2.bias");
    {   // conv_in: [O][I][3][3] -> Wt[k = ic*9+tap][O] fp32
        const auto& d = pk.P("conv_in.weight");
        const int O = c0, I = c.in_channels;
        size_t off = pk.alloc("conv_in.weight", d.size() * 4);
        float* o = (float*)(u->hblob.data() + off);
        for (int oc = 0; oc < O; ++oc)
            for (int k = 0; k < I * 9; ++k) o[(size_t)k * O + oc] = d[(size_t)oc * I * 9 + k];
        pk.f32("conv_in.bias");
    }
    std::vector<std::string> tw, tb;
    long toff = 0;
    walk_blocks(c,
        [&](const std::string& p, int cin, int cout) {
            pk.resnet(p, cin, cout);
            tw.push_back(p + "time_emb_proj.weight");
            tb.push_back(p + "time_emb_proj.bias");
            u->tproj_off[p] = toff;
            toff += cout;
        },
        [&](const std::string& p, int cc) { pk.transformer(p, cc); });
    u->tproj_total = toff;
    pk.concat_rows("tproj.weight", tw);
    {
        size_t off = pk.alloc("tproj.bias", (size_t)toff * 4);
        float* o = (float*)(u->hblob.data() + off);
        for (auto& n : tb)
            for (float v : pk.P(n)) *o++ = v;
    }
    (void)temb;
    for (int i = 0; i < nl - 1; ++i) {
        const int co = c.block_out_channels[i];
        const std::string d = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv.";
        pk.conv3(d + "weight", co, co); pk.f32(d + "bias");
        const int lev = nl - 1 - i, cu = c.block_out_channels[lev];
        const std::string up = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv.";
        pk.conv3(up + "weight", cu, cu); pk.f32(up + "bias");
        pk.conv3_subpixel(up + "weight", cu, cu);
    }
    pk.f32("conv_norm_out.weight"); pk.f32("conv_norm_out.bias");
    pk.conv3_ohwi("conv_out.weight", c.out_channels, c0); pk.f32("conv_out.bias");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan builder
// ---------------------------------------------------------------------------------------------
struct Builder {
    sd_unet* u;
    Plan& pl;
    int UB;
    std::vector<Wrap> wrapstack;
    std::map<int, int> stats_of;      // activation tensor -> statistics tensor written by its producer
    std::map<int, float> tscale;      // e4m3 activation tensor -> the scale its producer wrote it with
    static std::string stem(const std::string& key) {      // "....norm1.weight" -> "....norm1"
        const size_t n = key.rfind(".weight");
        return n == std::string::npos ? key : key.substr(0, n);
    }
    // the first inconsistency found while building (a C-ABI library reports it as an error code, it never aborts the host
    // process): the builder carries on with harmless values and get_plan refuses the plan
    std::string error;
    float xscale(int t) {
        auto it = tscale.find(t);
        if (it == tscale.end()) {
            if (error.empty()) error = "fp8 consumer of a tensor without a scale";
            return 1.0f;
        }
        return it->second;
    }
    // SD_GN_PRODUCER_STATS=0: every GroupNorm runs its own statistics pass (round-1 behaviour)
    bool producer_stats = !(getenv("SD_GN_PRODUCER_STATS") && atoi(getenv("SD_GN_PRODUCER_STATS")) == 0);
    void want_stats(Op& o, int M, int N) {      // called for producers whose output feeds a GroupNorm
        if (!producer_stats || u->kind != 0 || o.splitk > 1 || M % 64 != 0 || o.epi != 0 || o.rpb != 0) return;
        o.stats = tensor((size_t)(M / 64) * N * 2 * 4);
        stats_of[o.out] = o.stats;
    }

    // CFG de-duplication (Plan::rep == 2: the UNet batch is [uncond | cond] over the SAME latents and timestep): every op
    // before the first prompt cross-attention -- conv_in, down_blocks.0.resnets.0 and attentions.0 up to attn1.to_out --
    // sees identical inputs in both halves, so it runs once per latent (UB / 2) and its three outputs that live on
    // (conv_in's skip, the block input = proj_out's residual, h1) are copied to both halves.  prefix_rep > 1 while the
    // builder is inside that prefix.
    int prefix_rep = 1;
    int replicate(int t, size_t bytes, int r) {
        Op o; o.kind = OP_REPLICATE; o.x1 = t; o.M = (int)(bytes / 16); o.N = r; o.out = tensor(bytes * r);
        push(o);
        return o.out;
    }
    // SD_LN_FOLD=0: every LayerNorm is its own launch (round-1 behaviour)
    bool ln_fold = !(getenv("SD_LN_FOLD") && atoi(getenv("SD_LN_FOLD")) == 0);
    // Ask the op that produced a residual-stream tensor for per-row LayerNorm partials; returns the partial count (0 =
    // this producer cannot deliver them: split-K, fp8 output, ...) and the tensor in `rs`.
    int want_rowstats(Op& o, int M, int C, int& rs) {
        int np = 0;
        if (o.kind == OP_GEMM && o.epi == 0 && o.splitk == 1 && !o.out_fp8 && o.N == C && o.M == M && o.ldc_o == 0) np = (C + 159) / 160 * 2;
        else if (o.kind == OP_XATTN && o.N == C && o.M == M) np = 2 * sd_xattn_slices(M, C);
        // the consumer's preconditions (gemm_conv.hip::check_ln): K = C >= 128 (two K tiles: with one, the c1 | c2 LDS-DMA is
        // never waited for), at most 16 partials per row, 128-row tiles -- otherwise the plan keeps the separate LayerNorm
        static const bool big_tiles = getenv("SD_GEMM_BIG") != nullptr;
        if (!ln_fold || np == 0 || np > 16 || C < 128 || big_tiles) return 0;
        o.rs = rs = tensor((size_t)np * M * 2 * 4);
        return np;
    }
    // GEMM over the un-normalised rows x with the LayerNorm folded in (weights key.ln / key.c1 / key.c2 of the packer)
    int gemm_ln(int x, int rs, int np, int M, int N, int C, const std::string& w, int epi) {
        Op o; o.kind = OP_GEMM; o.x1 = x; o.K1 = o.K = o.Kalg = C; o.M = M; o.N = N; o.epi = epi; o.splitk = 1;
        o.w = W(w + ".ln"); o.c1 = W(w + ".c1"); o.b = W(w + ".c2"); o.lnrs = rs; o.lnnp = np;
        o.out = tensor((size_t)M * (epi ? N / 2 : N) * 2);
        push(o);
        return o.out;
    }

    int ctx_tensor(size_t bytes) {       // persistent and written by sd_unet_set_context
        const int id = tensor(bytes, true);
        pl.tensors[id].ctx = true;
        return id;
    }
    int tensor(size_t bytes, bool persistent = false) {
        Tn t;
        t.bytes = (bytes + 255) / 256 * 256;
        t.persistent = persistent;
        pl.tensors.push_back(t);
        return (int)pl.tensors.size() - 1;
    }
    size_t W(const std::string& k) {
        auto it = u->woff.find(k);
        if (it == u->woff.end()) {
            if (error.empty()) error = "missing packed weight " + k;
            return 0;
        }
        return it->second;
    }
    Op& push(Op op) {
        op.nwrap = (int)wrapstack.size();
        for (int i = 0; i < op.nwrap; ++i) op.wraps[i] = wrapstack[i];
        pl.ops.push_back(op);
        return pl.ops.back();
    }
    static int pad128(int c) { return (c + 127) / 128 * 128; }
    // fq: the output feeds an fp8 contraction -> e4m3 rows of pad128(C) bytes, scaled by the handle's norm scale
    int gn(int x1, int c1, int x2, int c2, int hw, const std::string& g, const std::string& b, float eps, int silu,
           bool fq = false) {
        Op o; o.kind = OP_GN; o.x1 = x1; o.C1 = c1; o.x2 = x2; o.C2 = c2; o.HW = hw; o.B = UB;
        o.g = W(g); o.be = W(b); o.eps = eps; o.silu = silu;
        o.nsplit = sd_groupnorm_nsplit(UB, hw);
        o.aux = tensor(sd_groupnorm_scratch_bytes(UB, hw, u->cfg.norm_num_groups));
        if (hw % 64 == 0 && !sd_groupnorm_uses_small(UB, hw, c1, c2, u->cfg.norm_num_groups) && stats_of.count(x1) &&
            (x2 < 0 || stats_of.count(x2))) {
            o.s1 = stats_of[x1];
            o.s2 = x2 >= 0 ? stats_of[x2] : -1;
        }
        if (fq) {
            o.out_fp8 = 1; o.Cpad = pad128(c1 + c2); o.sname = u->act_id(stem(g), u->s_norm); o.os = u->act_scale[o.sname];
            o.out = tensor((size_t)UB * hw * o.Cpad); tscale[o.out] = o.os;
        } else o.out = tensor((size_t)UB * hw * (c1 + c2) * 2);
        push(o);
        return o.out;
    }
    // fq: x is an e4m3 tensor of pad128(cin) channels written with activation scale xs
    int conv3(int x, int hin, int cin, int cout, int stride, int up, const std::string& w, const std::string& b,
              long b2idx, int b2t, int r, bool fq = false) {
        Op o; o.kind = OP_CONV3; o.x1 = x; o.B = UB; o.Hin = hin; o.Win = hin; o.Cin = cin; o.N = cout;
        o.stride = stride; o.up = up;
        const int hv = hin << up;
        o.Hout = o.Wout = (hv + 2 - 3) / stride + 1;
        o.M = UB * o.Hout * o.Wout; o.K = 9 * cin; o.Kalg = o.K;
        o.b = W(b); o.b2t = b2t; o.b2idx = b2idx; o.r = r;
        if (fq) { o.dt = 1; o.Cin = pad128(cin); o.K = 9 * o.Cin; o.w = W(w + ".fp8"); o.wsc = W(w + ".scale"); o.xs = xscale(x); }
        else o.w = W(w);
        // upsampler: nearest-2x + 3x3 as four 2x2 convs on the low-res input, 4/9 of the multiply-adds (SD_CONV_SUBPIXEL=0: off)
        static const bool subpix_off = getenv("SD_CONV_SUBPIXEL") && atoi(getenv("SD_CONV_SUBPIXEL")) == 0;
        if (up && !fq && !subpix_off && u->kind == 0 && stride == 1 && r < 0 && b2t < 0 && (hin * hin) % 64 == 0 &&
            u->woff.count(w + ".sub")) {
            o.subpix = 1; o.K = 4 * cin; o.Kalg = 4 * cin;        // (Kalg: the EXECUTED multiply-adds, 4/9 of the 3x3 form)
            o.w = W(w + ".sub"); o.splitk = 1;
            o.out = tensor((size_t)o.M * cout * 2);
            if ((hin * hin) % 128 == 0) want_stats(o, o.M, cout);     // (8x8 inputs run on 64-row tiles: no block statistics)
            push(o);
            return o.out;
        }
        o.splitk = sd_conv3x3_splitk(o.M, o.N, o.Cin, hin, hin, stride, up, o.dt);
        if (o.splitk > 1) o.aux = tensor((size_t)o.splitk * o.M * o.N * 4);
        o.out = tensor((size_t)o.M * cout * 2);
        want_stats(o, o.M, cout);         // every 3x3 conv of the UNet feeds a GroupNorm (directly or as a skip)
        push(o);
        return o.out;
    }
    // fq: x1 is an e4m3 tensor (its scale comes from its producer); oname: name of the e4m3 tensor this GEMM WRITES (GEGLU
    // epilogue only; empty = bf16 output)
    int gemm(int x1, int k1, int x2, int k2, int M, int N, const std::string& w, const std::string& b, int r, int epi,
             bool fq = false, const std::string& oname = "") {
        Op o; o.kind = OP_GEMM; o.x1 = x1; o.x2 = x2; o.K1 = k1; o.K = k1 + k2; o.Kalg = o.K; o.M = M; o.N = N; o.epi = epi;
        o.b = b.empty() ? NOFF : W(b); o.r = r;
        if (fq) { o.dt = 1; o.K = o.K1 = pad128(k1); o.w = W(w + ".fp8"); o.wsc = W(w + ".scale"); o.xs = xscale(x1); }
        else o.w = W(w);
        o.splitk = epi ? 1 : sd_gemm_splitk(M, N, o.dt ? o.K / 2 : o.K, o.dt ? 128 : 0);     // the heuristic counts 128-byte K tiles
        if (o.splitk > 1) o.aux = tensor((size_t)o.splitk * M * N * 4);
        if (!oname.empty()) {
            o.out_fp8 = 1; o.sname = u->act_id(oname, u->s_ff); o.os = u->act_scale[o.sname]; o.Cpad = pad128(N / 2);
            o.out = tensor((size_t)M * o.Cpad); tscale[o.out] = o.os;
        } else o.out = tensor((size_t)M * (epi ? N / 2 : N) * 2);
        push(o);
        return o.out;
    }
    int ln(int x, int M, int C, const std::string& g, const std::string& b, bool fq = false) {
        Op o; o.kind = OP_LN; o.x1 = x; o.M = M; o.N = C; o.g = W(g); o.be = W(b); o.eps = 1e-5f;
        if (fq) {
            o.out_fp8 = 1; o.Cpad = pad128(C); o.sname = u->act_id(stem(g), u->s_norm); o.os = u->act_scale[o.sname];
            o.out = tensor((size_t)M * o.Cpad); tscale[o.out] = o.os;
        } else o.out = tensor((size_t)M * C * 2);
        push(o);
        return o.out;
    }
    int attn(int q, long qoff, long ldq, int kv, long koff, long voff, long ldkv, int nq, int nk, int C) {
        Op o; o.kind = OP_ATTN; o.x1 = q; o.x2 = kv; o.qoff = qoff; o.koff = koff; o.voff = voff;
        o.ldq = ldq; o.ldk = o.ldv = ldkv; o.ldo = C; o.B = UB; o.heads = u->cfg.num_heads; o.D = C / u->cfg.num_heads;
        o.Nq = nq; o.Nk = nk;
        o.out = tensor((size_t)UB * nq * C * 2);
        push(o);
        return o.out;
    }
    // ResnetBlock2D (A.3); input may be a virtual channel concat [x1 | x2]
    int resnet(const std::string& p, int x1, int c1, int x2, int c2, int cout, int res, int tproj_t) {
        const int hw = res * res, cin = c1 + c2, M = UB * hw;
        const bool fq = u->fp8;          // GroupNorm+SiLU writes e4m3, both 3x3 convs contract in fp8
        int t1 = gn(x1, c1, x2, c2, hw, p + "norm1.weight", p + "norm1.bias", u->cfg.norm_eps, 1, fq);
        int t2 = conv3(t1, res, cin, cout, 1, 0, p + "conv1.weight", p + "conv1.bias", u->tproj_off.at(p), tproj_t, -1, fq);
        int t3 = gn(t2, cout, -1, 0, hw, p + "norm2.weight", p + "norm2.bias", u->cfg.norm_eps, 1, fq);
        int sc = x1;
        if (cin != cout) sc = gemm(x1, c1, x2, c2, M, cout, p + "conv_shortcut.weight", p + "conv_shortcut.bias", -1, 0);
        return conv3(t3, res, cout, cout, 1, 0, p + "conv2.weight", p + "conv2.bias", 0, -1, sc, fq);
    }
    // Transformer2DModel with one BasicTransformerBlock (A.4)
    int n_transformers = 0;
    int transformer(const std::string& p, int x, int C, int res) {
        const int hw = res * res, L = u->cfg.context_len;
        int M = UB * hw;
        const std::string t = p + "transformer_blocks.0.";
        const bool fq = u->fp8;
        int g = gn(x, C, -1, 0, hw, p + "norm.weight", p + "norm.bias", 1e-6f, 0, fq);
        int h0 = gemm(g, C, -1, 0, M, C, p + "proj_in.weight", p + "proj_in.bias", -1, 0, fq);
        int qkv, rs = -1, np = 0;
        if (!fq && (np = want_rowstats(pl.ops.back(), M, C, rs)) > 0) {       // norm1 folded into the projection
            qkv = gemm_ln(h0, rs, np, M, 3 * C, C, t + "attn1.qkv.weight", 0);
        } else {
            int n1 = ln(h0, M, C, t + "norm1.weight", t + "norm1.bias", fq);
            qkv = gemm(n1, C, -1, 0, M, 3 * C, t + "attn1.qkv.weight", "", -1, 0, fq);
        }
        // 64x64 level (head dim 40): the projection stores K and V head-major, [which][sample][head][token][40] behind the
        // token-major Q block, so that the self-attention's LDS-DMA pieces are contiguous (SD_ATTN_HEADMAJOR=0: off)
        static const bool hm_off = (getenv("SD_ATTN_HEADMAJOR") && atoi(getenv("SD_ATTN_HEADMAJOR")) == 0) ||
                                   getenv("SD_ATTN_NO_PIPE") || getenv("SD_ATTN_NO_DMA") || getenv("SD_GEMM_BIG");
        const bool hm = !hm_off && C / u->cfg.num_heads == 40 && C % 160 == 0 && hw % 128 == 0 && hw >= 256 &&
                        sd_gemm_tile_rows(M, 3 * C) == 128 && pl.ops.back().splitk == 1;
        int a1;
        if (hm) {
            pl.ops.back().hm = 1;
            pl.ops.back().HW = hw;
            a1 = attn(qkv, 0, C, qkv, (long)M * C, 2l * M * C, C, hw, hw, C);
            pl.ops.back().hm = 1;
        } else {
            a1 = attn(qkv, 0, 3 * C, qkv, C, 2 * C, 3 * C, hw, hw, C);
        }
        pl.ops.back().qps = 1;          // W_q of attn1 carries the scale (Packer::transformer)
        int h1 = gemm(a1, C, -1, 0, M, C, t + "attn1.to_out.0.weight", t + "attn1.to_out.0.bias", h0, 0);
        const int to_out_op = (int)pl.ops.size() - 1;
        if (prefix_rep > 1) {        // end of the prompt-independent prefix: both CFG halves continue from copies
            x = replicate(x, (size_t)M * C * 2, prefix_rep);
            h1 = replicate(h1, (size_t)M * C * 2, prefix_rep);
            UB *= prefix_rep;
            M = UB * hw;
            prefix_rep = 1;
        }
        // Prompt cross-attention.  The prompt is step-invariant, so per sample and head
        //   A_h = scale * W_q,h^T K_h^T  [C x 77]   and   B_h = V_h W_o,h^T  [77 x C]
        // are computed once per sampling run (sd_unet_set_context; 80 key slots per head, 3 of them padding).
        //  * SD_XATTN_FUSED (levels with >= n tokens, default 1024 = 64x64 and 32x32; 0 = never): ONE launch,
        //    Y = h1 + sum_h softmax_77(X A_h) B_h + b_o with the probabilities kept in registers (xattn.hip);
        //  * SD_XATTN_FOLD (levels with <= n tokens, default 1024): two GEMMs with per-sample weights,
        //    P = softmax_77(X A) in the GEMM epilogue and h2 = h1 + P B + b_o;
        //  * otherwise to_q GEMM, the 77-key flash-attention kernel and the to_out GEMM.
        // norm2 (round 5): folded into the first kernel of whichever form runs -- rstd from the row partials attn1.to_out's
        // epilogue delivers, gamma in the operand (A^T centred over the channel: the row mean drops out; the plain to_q weights
        // with the c1 correction), beta as a constant per key slot / output column.  SD_XATTN_LN=0: the separate LayerNorm launch.
        static const int fused_min_hw = getenv("SD_XATTN_FUSED") ? atoi(getenv("SD_XATTN_FUSED")) : 1024;
        static const int fold_max_hw = getenv("SD_XATTN_FOLD") ? atoi(getenv("SD_XATTN_FOLD")) : 1024;
        static const bool xln_off = getenv("SD_XATTN_LN") && atoi(getenv("SD_XATTN_LN")) == 0;
        const int NH = u->cfg.num_heads, NP = NH * 80;
        const int xmode = (fused_min_hw > 0 && hw >= fused_min_hw && sd_xattn_fused_applicable(hw, C, NH, L)) ? 0
                          : (hw <= fold_max_hw && hw % 128 == 0 && L <= 80) ? 1 : 2;
        int rs2 = -1, np2 = 0;
        const long rs2_rows = (long)pl.ops[to_out_op].M;     // (< M when the CFG pair was replicated after attn1.to_out ran)
        // (the set_context plan and the forward plan must agree on every operand it writes: the CFG-pair variant replicates the
        // rows after attn1.to_out of the FIRST transformer only -- the fused kernel reads its partials modulo their rows, the
        // GEMM consumers do not, so in the two GEMM forms that block keeps its LayerNorm in EVERY plan variant)
        const bool first_tf = n_transformers++ == 0;
        if (!fq && !xln_off && (xmode == 0 || (!first_tf && rs2_rows == M)) &&
            u->woff.count(t + (xmode == 2 ? "attn2.to_q.weight.ln" : "attn2.to_q.weight.T.ln"))) {
            if (first_tf) pl.ops[to_out_op].splitk = 1;      // (its row count differs between the plan variants: the split heuristic must not)
            np2 = want_rowstats(pl.ops[to_out_op], (int)rs2_rows, C, rs2);
        }
        int n2 = np2 > 0 ? h1 : ln(h1, M, C, t + "norm2.weight", t + "norm2.bias");
        // K|V of the prompt: projected once per sampling run by sd_unet_set_context
        int kv = ctx_tensor((size_t)UB * L * 2 * C * 2);
        pl.ctx_kv.push_back(kv);
        pl.ctx_w.push_back(W(t + "attn2.kv.weight"));
        pl.ctx_c.push_back(C);
        int h2;
        if (xmode == 0) {
            int at = ctx_tensor((size_t)UB * NP * C * 2), bw = ctx_tensor((size_t)UB * C * NP * 2);
            Plan::Fold fd{kv, at, bw, C, W(t + (np2 > 0 ? "attn2.to_q.weight.T.ln" : "attn2.to_q.weight.T")), W(t + "attn2.to_out.0.weight"), true};
            if (np2 > 0) { fd.c2 = ctx_tensor((size_t)UB * NP * 4); fd.lnu = W(t + "attn2.to_q.lnu"); }
            pl.ctx_fold.push_back(fd);
            Op o; o.kind = OP_XATTN; o.x1 = n2; o.r = h1; o.wt = at; o.x2 = bw; o.M = M; o.N = C; o.K = NP; o.rpb = hw;
            o.sm_valid = L; o.b = W(t + "attn2.to_out.0.bias"); o.heads = NH;
            if (np2 > 0) { o.lnrs = rs2; o.lnnp = np2; o.s1 = fd.c2; o.ldx_o = rs2_rows; }   // (s1: the c2 tensor; ldx_o: rows of the partials)
            o.out = tensor((size_t)M * C * 2); push(o); h2 = o.out;
        } else if (xmode == 1) {
            int at = ctx_tensor((size_t)UB * NP * C * 2), bw = ctx_tensor((size_t)UB * C * NP * 2);
            Plan::Fold fd{kv, at, bw, C, W(t + (np2 > 0 ? "attn2.to_q.weight.T.ln" : "attn2.to_q.weight.T")), W(t + "attn2.to_out.0.weight"), false};
            if (np2 > 0) {      // c1 = row sums of the ROUNDED centred operand (what is left of the mean term), c2 = the beta term
                fd.c2 = ctx_tensor((size_t)UB * NP * 4); fd.lnu = W(t + "attn2.to_q.lnu");
                fd.c1 = ctx_tensor((size_t)UB * NP * 4); fd.ones = W(t + "attn2.to_q.ones");
            }
            pl.ctx_fold.push_back(fd);
            int pr;
            { Op o; o.kind = OP_GEMM; o.x1 = n2; o.K1 = C; o.K = C; o.M = M; o.N = NP; o.epi = 2; o.sm_valid = L;
              o.wt = at; o.wbs = (long)NP * C; o.rpb = hw;
              if (np2 > 0) { o.lnrs = rs2; o.lnnp = np2; o.s1 = fd.c1; o.s2 = fd.c2; }          // (s1 / s2: the per-sample c1 / c2 tensors)
              o.out = tensor((size_t)M * NP * 2); push(o); pr = o.out; }
            { Op o; o.kind = OP_GEMM; o.x1 = pr; o.K1 = NP; o.K = NP; o.M = M; o.N = C; o.epi = 0;
              o.wt = bw; o.wbs = (long)C * NP; o.rpb = hw; o.b = W(t + "attn2.to_out.0.bias"); o.r = h1;
              o.out = tensor((size_t)M * C * 2); push(o); h2 = o.out; }
        } else {
            int q2 = np2 > 0 ? gemm_ln(h1, rs2, np2, M, C, C, t + "attn2.to_q.weight", 0)
                             : gemm(n2, C, -1, 0, M, C, t + "attn2.to_q.weight", "", -1, 0);
            int a2 = attn(q2, 0, C, kv, 0, C, 2 * C, hw, L, C);
            h2 = gemm(a2, C, -1, 0, M, C, t + "attn2.to_out.0.weight", t + "attn2.to_out.0.bias", h1, 0);
        }
        int ff;
        if (!fq && (np = want_rowstats(pl.ops.back(), M, C, rs)) > 0) {       // norm3 folded into the GEGLU projection
            ff = gemm_ln(h2, rs, np, M, 8 * C, C, t + "ff.geglu.weight", 1);
        } else {
            int n3 = ln(h2, M, C, t + "norm3.weight", t + "norm3.bias", fq);
            ff = gemm(n3, C, -1, 0, M, 8 * C, t + "ff.geglu.weight", t + "ff.geglu.bias", -1, 1, fq, fq ? t + "ff.net.0" : std::string());
        }
        // ff.net.2 + residual + proj_out + residual as ONE GEMM over [ff | h2] (Packer::ff_out_merge; SD_FF_MERGE=0: two)
        static const bool merge_off = getenv("SD_FF_MERGE") && atoi(getenv("SD_FF_MERGE")) == 0;
        int out;
        if (!fq && !merge_off) {
            out = gemm(ff, 4 * C, h2, C, M, C, p + "ff_out.weight", p + "ff_out.bias", x, 0);
        } else {
            int h3 = gemm(ff, 4 * C, -1, 0, M, C, t + "ff.net.2.weight", t + "ff.net.2.bias", h2, 0, fq);
            out = gemm(h3, C, -1, 0, M, C, p + "proj_out.weight", p + "proj_out.bias", x, 0);
        }
        // the block's output feeds the next resnet's GroupNorm (not at the small levels: their GroupNorms are single-launch or
        // fall back to their own pass, and the statistics epilogue would keep this GEMM on 128-row tiles)
        if (pl.ops.back().splitk == 1 && sd_gemm_tile_rows(M, C, pl.ops.back().K) == 128) want_stats(pl.ops.back(), M, C);
        return out;
    }

    // ---- AutoencoderKL decoder: latents/scale -> post_quant_conv -> decoder -> image ----
    int vae_resnet(const std::string& p, int x, int cin, int cout, int res) {
        const int hw = res * res, M = UB * hw;
        int t1 = gn(x, cin, -1, 0, hw, p + "norm1.weight", p + "norm1.bias", 1e-6f, 1);
        int t2 = conv3(t1, res, cin, cout, 1, 0, p + "conv1.weight", p + "conv1.bias", 0, -1, -1);
        int t3 = gn(t2, cout, -1, 0, hw, p + "norm2.weight", p + "norm2.bias", 1e-6f, 1);
        int sc = x;
        if (cin != cout) sc = gemm(x, cin, -1, 0, M, cout, p + "conv_shortcut.weight", p + "conv_shortcut.bias", -1, 0);
        return conv3(t3, res, cout, cout, 1, 0, p + "conv2.weight", p + "conv2.bias", 0, -1, sc);
    }
    // single-head attention with head dim C (512): too wide for the flash kernel's register tile, so it
    // is three GEMMs per image (S = Q K^T, row softmax, O = P V with V^T produced directly by a GEMM)
    int vae_attention(const std::string& p, int x, int C, int res) {
        const int hw = res * res, M = UB * hw;
        int g = gn(x, C, -1, 0, hw, p + "group_norm.weight", p + "group_norm.bias", 1e-6f, 0);
        int qk = gemm(g, C, -1, 0, M, 2 * C, p + "qk.weight", p + "qk.bias", -1, 0);
        const int vT = tensor((size_t)UB * C * hw * 2), S = tensor((size_t)hw * hw * 2), O = tensor((size_t)M * C * 2);
        for (int img = 0; img < UB; ++img) {
            {   // V^T[C, hw] = Wv[C, C] . g_img[hw, C]^T   (bias of V is added to O: rows of P sum to 1)
                Op o; o.kind = OP_GEMM; o.wx = W(p + "to_v.weight"); o.ldx_o = C; o.K1 = C; o.K = C; o.M = C; o.N = hw;
                o.wt = g; o.woff_el = (long)img * hw * C; o.ldw_o = C;
                o.out = vT; o.coff = (long)img * C * hw; o.ldc_o = hw;
                push(o);
            }
            {   // S[hw, hw] = Q_img . K_img^T
                Op o; o.kind = OP_GEMM; o.x1 = qk; o.xoff = (long)img * hw * 2 * C; o.ldx_o = 2 * C; o.K1 = C; o.K = C;
                o.M = hw; o.N = hw; o.wt = qk; o.woff_el = (long)img * hw * 2 * C + C; o.ldw_o = 2 * C;
                o.out = S; o.ldc_o = hw;
                push(o);
            }
            { Op o; o.kind = OP_SOFTMAX; o.x1 = S; o.out = S; o.M = hw; o.N = hw; o.scale = 1.0f / sqrtf((float)C); push(o); }
            {   // O_img[hw, C] = P . V + b_v
                Op o; o.kind = OP_GEMM; o.x1 = S; o.ldx_o = hw; o.K1 = hw; o.K = hw; o.M = hw; o.N = C;
                o.wt = vT; o.woff_el = (long)img * C * hw; o.ldw_o = hw; o.b = W(p + "to_v.bias");
                o.out = O; o.coff = (long)img * hw * C; o.ldc_o = C;
                push(o);
            }
        }
        return gemm(O, C, -1, 0, M, C, p + "to_out.0.weight", p + "to_out.0.bias", x, 0);
    }
    void build_vae() {
        const sd_unet_config& c = u->cfg;
        const int nl = c.num_levels, top = c.block_out_channels[nl - 1];
        int res = c.sample_size;
        int t_pq = tensor((size_t)UB * c.in_channels * res * res * 4);
        { Op o; o.kind = OP_PQCONV; o.x1 = T_LATENTS; o.out = t_pq; o.B = UB; o.HW = res * res;
          o.w = W("post_quant_conv.weight"); o.b = W("post_quant_conv.bias"); push(o); }
        int h;
        { Op o; o.kind = OP_CONV_IN; o.x1 = t_pq; o.B = UB; o.Hin = res; o.Win = res; o.Cin = c.in_channels; o.N = top;
          o.w = W("decoder.conv_in.weight"); o.b = W("decoder.conv_in.bias"); o.out = tensor((size_t)UB * res * res * top * 2);
          push(o); h = o.out; }
        pl.taps["conv_in"] = h;
        h = vae_resnet("decoder.mid_block.resnets.0.", h, top, top, res);
        h = vae_attention("decoder.mid_block.attentions.0.", h, top, res);
        h = vae_resnet("decoder.mid_block.resnets.1.", h, top, top, res);
        pl.taps["mid"] = h;
        int ch = top;
        for (int i = 0; i < nl; ++i) {
            const int co = c.block_out_channels[nl - 1 - i];
            const std::string bp = "decoder.up_blocks." + std::to_string(i) + ".";
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                h = vae_resnet(bp + "resnets." + std::to_string(j) + ".", h, ch, co, res);
                ch = co;
            }
            if (i < nl - 1) {
                h = conv3(h, res, co, co, 1, 1, bp + "upsamplers.0.conv.weight", bp + "upsamplers.0.conv.bias", 0, -1, -1);
                res *= 2;
            }
            pl.taps["up" + std::to_string(i)] = h;
        }
        int g = gn(h, ch, -1, 0, res * res, "decoder.conv_norm_out.weight", "decoder.conv_norm_out.bias", 1e-6f, 1);
        { Op o; o.kind = OP_CONV_OUT; o.x1 = g; o.out = T_EPS; o.B = UB; o.Hin = res; o.Win = res; o.Cin = ch; o.N = c.out_channels;
          o.w = W("decoder.conv_out.weight"); o.b = W("decoder.conv_out.bias"); push(o); }
    }

    // CLIPTextTransformer (transformers 4.48.0 modeling_clip.py): token + position embedding,
    // pre-LN layers with causal self-attention and a quick_gelu MLP, final LayerNorm -> last_hidden_state
    void build_clip() {
        const sd_clip_config& c = u->clip;
        const int L = c.max_positions, H = c.hidden_size, I = c.intermediate_size, M = UB * L;
        int t;
        { Op o; o.kind = OP_CLIP_EMBED; o.x1 = T_LATENTS; o.M = M; o.N = H; o.Nk = L;
          o.w = W("text_model.embeddings.token_embedding.weight"); o.g = W("text_model.embeddings.position_embedding.weight");
          o.out = tensor((size_t)M * H * 2); push(o); t = o.out; }
        for (int i = 0; i < c.num_layers; ++i) {
            const std::string p = clip_layer(i), a = p + "self_attn.";
            int n1 = ln(t, M, H, p + "layer_norm1.weight", p + "layer_norm1.bias");
            int qkv = gemm(n1, H, -1, 0, M, 3 * H, a + "qkv.weight", a + "qkv.bias", -1, 0);
            int at;
            { Op o; o.kind = OP_CLIP_ATTN; o.x1 = qkv; o.B = UB; o.Nq = L; o.N = H; o.heads = c.num_heads;
              o.out = tensor((size_t)M * H * 2); push(o); at = o.out; }
            t = gemm(at, H, -1, 0, M, H, a + "out_proj.weight", a + "out_proj.bias", t, 0);
            int n2 = ln(t, M, H, p + "layer_norm2.weight", p + "layer_norm2.bias");
            int f = gemm(n2, H, -1, 0, M, I, p + "mlp.fc1.weight", p + "mlp.fc1.bias", -1, 0);
            { Op o; o.kind = OP_QGELU; o.x1 = f; o.out = f; o.M = M; o.N = I; push(o); }
            t = gemm(f, I, -1, 0, M, H, p + "mlp.fc2.weight", p + "mlp.fc2.bias", t, 0);
            pl.taps["layer" + std::to_string(i)] = t;
        }
        int f = ln(t, M, H, "text_model.final_layer_norm.weight", "text_model.final_layer_norm.bias");
        { Op o; o.kind = OP_TO_F32; o.x1 = f; o.out = T_EPS; o.M = M; o.N = H; push(o); }
    }

    void build() {
        if (u->kind == 1) { build_vae(); return; }
        if (u->kind == 2) { build_clip(); return; }
        const sd_unet_config& c = u->cfg;
        const int nl = c.num_levels, c0 = c.block_out_channels[0], temb = 4 * c0;
        const int L = c.context_len;
        pl.ctx_bf16 = ctx_tensor((size_t)UB * L * c.cross_attention_dim * 2);
        // masked K / V expansions used by sd_unet_set_context for the folded cross-attention (sized for the widest level)
        pl.ctx_fold_scratch = ctx_tensor((size_t)3 * UB * c.num_heads * 80 * c.block_out_channels[nl - 1] * 2);
        // ---- time embedding (M = 1: the reference passes one scalar t per call) ----
        int t_sin = tensor((size_t)c0 * 4), t_h1 = tensor((size_t)temb * 4), t_emb = tensor((size_t)temb * 4);
        int t_proj = tensor((size_t)u->tproj_total * 4);
        { Op o; o.kind = OP_SINUSOID; o.out = t_sin; o.N = c0; push(o); }
        { Op o; o.kind = OP_GEMV; o.x1 = t_sin; o.out = t_h1; o.N = temb; o.K = c0; o.w = W("time_embedding.linear_1.weight"); o.b = W("time_embedding.linear_1.bias"); push(o); }
        { Op o; o.kind = OP_GEMV; o.x1 = t_h1; o.out = t_emb; o.N = temb; o.K = temb; o.silu_in = 1; o.w = W("time_embedding.linear_2.weight"); o.b = W("time_embedding.linear_2.bias"); push(o); }
        { Op o; o.kind = OP_GEMV; o.x1 = t_emb; o.out = t_proj; o.N = (int)u->tproj_total; o.K = temb; o.silu_in = 1; o.w = W("tproj.weight"); o.b = W("tproj.bias"); push(o); }
        // ---- conv_in ----
        int res = c.sample_size;
        int h;
        if (pl.rep > 1) { prefix_rep = pl.rep; UB /= pl.rep; }       // (restored by the first transformer block)
        { Op o; o.kind = OP_CONV_IN; o.x1 = T_LATENTS; o.B = UB; o.Hin = res; o.Win = res; o.Cin = c.in_channels; o.N = c0;
          o.w = W("conv_in.weight"); o.b = W("conv_in.bias"); o.out = tensor((size_t)UB * res * res * c0 * 2); push(o); h = o.out; }
        const int h_skip = prefix_rep > 1 ? replicate(h, (size_t)UB * res * res * c0 * 2, prefix_rep) : h;
        pl.taps["conv_in"] = h_skip;
        int ch = c0;
        std::vector<int> skips{h_skip}, skip_ch{c0};
        // ---- down ----
        for (int i = 0; i < nl; ++i) {
            const int co = c.block_out_channels[i];
            const std::string bp = "down_blocks." + std::to_string(i) + ".";
            wrapstack.push_back(Wrap{0, i, 0});
            for (int j = 0; j < c.layers_per_block; ++j) {
                wrapstack.push_back(Wrap{0, i, j});
                h = resnet(bp + "resnets." + std::to_string(j) + ".", h, ch, -1, 0, co, res, t_proj);
                ch = co;
                if (c.attn_levels[i]) h = transformer(bp + "attentions." + std::to_string(j) + ".", h, co, res);
                wrapstack.pop_back();
                skips.push_back(h); skip_ch.push_back(co);
            }
            if (i < nl - 1) {
                wrapstack.push_back(Wrap{0, i, c.layers_per_block});
                const std::string d = bp + "downsamplers.0.conv.";
                h = conv3(h, res, co, co, 2, 0, d + "weight", d + "bias", 0, -1, -1);
                wrapstack.pop_back();
                res /= 2;
                skips.push_back(h); skip_ch.push_back(co);
            }
            wrapstack.pop_back();
            pl.taps["down" + std::to_string(i)] = h;
        }
        // ---- mid ----
        wrapstack.push_back(Wrap{1, 0, 0});
        h = resnet("mid_block.resnets.0.", h, ch, -1, 0, ch, res, t_proj);
        h = transformer("mid_block.attentions.0.", h, ch, res);
        h = resnet("mid_block.resnets.1.", h, ch, -1, 0, ch, res, t_proj);
        wrapstack.pop_back();
        pl.taps["mid"] = h;
        // ---- up ----
        const int nres = c.layers_per_block + 1;
        for (int i = 0; i < nl; ++i) {
            const int lev = nl - 1 - i, co = c.block_out_channels[lev], rb = nl - 1 - i;
            const std::string bp = "up_blocks." + std::to_string(i) + ".";
            wrapstack.push_back(Wrap{2, rb, 0});
            for (int j = 0; j < nres; ++j) {
                const int s = skips.back(), sc = skip_ch.back();
                skips.pop_back(); skip_ch.pop_back();
                const int rl = nres - 1 - j;
                wrapstack.push_back(Wrap{2, rb, rl});
                h = resnet(bp + "resnets." + std::to_string(j) + ".", h, ch, s, sc, co, res, t_proj);
                ch = co;
                if (c.attn_levels[lev]) h = transformer(bp + "attentions." + std::to_string(j) + ".", h, co, res);
                wrapstack.pop_back();
            }
            if (i < nl - 1) {
                wrapstack.push_back(Wrap{2, rb, 0});
                const std::string up = bp + "upsamplers.0.conv.";
                h = conv3(h, res, co, co, 1, 1, up + "weight", up + "bias", 0, -1, -1);
                wrapstack.pop_back();
                res *= 2;
            }
            wrapstack.pop_back();
            pl.taps["up" + std::to_string(i)] = h;
        }
        // ---- out ----
        int g = gn(h, ch, -1, 0, res * res, "conv_norm_out.weight", "conv_norm_out.bias", c.norm_eps, 1);
        { Op o; o.kind = OP_CONV_OUT; o.x1 = g; o.out = T_EPS; o.B = UB; o.Hin = res; o.Win = res; o.Cin = ch; o.N = c.out_channels;
          o.w = W("conv_out.weight"); o.b = W("conv_out.bias"); push(o); }
    }
};

bool wrap_skipped(const Wrap& w, int branch) {
    const int cache_layer_id = branch % 3, cache_block_id = branch / 3;
    if (w.block_i > cache_block_id || w.type == 1) return true;
    if (w.block_i < cache_block_id) return false;
    return w.type == 0 ? w.layer_i >= cache_layer_id : w.layer_i > cache_layer_id;
}

void op_tensors(const Op& o, int ins[12], int& nin) {
    nin = 0;
    for (int t : {o.x1, o.x2, o.r, o.b2t, o.wt, o.s1, o.s2, o.lnrs, o.slab_t, o.slab_r, o.slab_b2t})
        if (t >= 0) ins[nin++] = t;
}

// A split-K conv / GEMM whose output is first read by a single-launch GroupNorm (the 8x8 and 16x16 levels: every resnet conv,
// the downsamplers, proj_out) hands its partial slabs to that GroupNorm instead of launching splitk_reduce_kernel: the
// reduce was a 42 MB pass at the launch floor (8-11 us) followed by a 6-10 us GroupNorm over 2.6 MB (27 pairs per forward at
// UNet batch 16).  Bit-identical to the two launches (same sums in the same order; SD_GN_SLAB=0: off).  Conditions: bf16,
// plain [M][N] output, no reader of the output between the two ops, both on the same side of the DeepCache boundary.
void fuse_deferred_reduce(sd_unet* u, Plan& pl) {
    if (getenv("SD_GN_SLAB") && atoi(getenv("SD_GN_SLAB")) == 0) return;       // (read when a plan is built: tests build both)
    const int nops = (int)pl.ops.size();
    auto skipped = [&](const Op& o) {
        if (pl.branch < 0) return false;
        for (int k = 0; k < o.nwrap; ++k)
            if (wrap_skipped(o.wraps[k], pl.branch)) return true;
        return false;
    };
    std::vector<int> producer(pl.tensors.size(), -1);
    for (int i = 0; i < nops; ++i)
        if (pl.ops[i].out >= 0) producer[pl.ops[i].out] = i;
    for (int g = 0; g < nops; ++g) {
        Op& G = pl.ops[g];
        if (G.kind != OP_GN || G.out_fp8 || G.x1 < 0 || !sd_groupnorm_slab_ok(G.B, G.HW, G.C1, G.C2, u->cfg.norm_num_groups)) continue;
        const int p = producer[G.x1];
        if (p < 0 || p >= g) continue;
        Op& P = pl.ops[p];
        if (P.kind != OP_CONV3 && !(P.kind == OP_GEMM && P.epi == 0)) continue;
        if (P.splitk <= 1 || P.splitk > 64 || P.aux < 0 || P.dt || P.out_fp8 || P.defer || P.subpix) continue;
        if (P.kind == OP_GEMM && (P.coff || P.ldc_o || P.hm || P.rs >= 0 || P.lnrs >= 0)) continue;
        if ((long)G.B * G.HW != P.M || G.C1 != P.N || skipped(P) != skipped(G)) continue;
        bool first_reader = true;
        for (int i = p + 1; i < g && first_reader; ++i) {
            int ins[12], nin;
            op_tensors(pl.ops[i], ins, nin);
            for (int k = 0; k < nin; ++k)
                if (ins[k] == P.out) first_reader = false;
        }
        if (!first_reader) continue;
        P.defer = 1;
        G.slab_t = P.aux; G.slab_k = P.splitk; G.slab_b = P.b; G.slab_r = P.r;
        G.slab_b2t = P.kind == OP_CONV3 ? P.b2t : -1; G.slab_b2idx = P.b2idx;
    }
}

void assign_memory(sd_unet* u, Plan& pl) {
    const int nops = (int)pl.ops.size();
    // DeepCache: which ops are skipped on skip steps, and which tensors they leave behind for running ops
    pl.skipped.assign(nops, 0);
    if (pl.branch >= 0) {
        for (int i = 0; i < nops; ++i)
            for (int k = 0; k < pl.ops[i].nwrap; ++k)
                if (wrap_skipped(pl.ops[i].wraps[k], pl.branch)) pl.skipped[i] = 1;
        std::vector<int> producer(pl.tensors.size(), -1);
        for (int i = 0; i < nops; ++i) {
            if (pl.ops[i].out >= 0) producer[pl.ops[i].out] = i;
            if (pl.ops[i].stats >= 0) producer[pl.ops[i].stats] = i;
            if (pl.ops[i].rs >= 0) producer[pl.ops[i].rs] = i;
        }
        for (int i = 0; i < nops; ++i) {
            if (pl.skipped[i]) continue;
            int ins[12], nin;
            op_tensors(pl.ops[i], ins, nin);
            for (int k = 0; k < nin; ++k) {
                const int p = producer[ins[k]];
                if (p >= 0 && pl.skipped[p]) pl.tensors[ins[k]].persistent = true;
            }
        }
    }
    if (u->debug_taps)
        for (auto& kv : pl.taps) pl.tensors[kv.second].persistent = true;
    // lifetimes over the full plan
    for (int i = 0; i < nops; ++i) {
        const Op& o = pl.ops[i];
        int ins[12], nin;
        op_tensors(o, ins, nin);
        for (int k = 0; k < nin; ++k) pl.tensors[ins[k]].last = std::max(pl.tensors[ins[k]].last, i);
        for (int t : {o.out, o.aux, o.stats, o.rs})
            if (t >= 0) {
                if (pl.tensors[t].def < 0) pl.tensors[t].def = i;
                pl.tensors[t].last = std::max(pl.tensors[t].last, i);
            }
    }
    // persistent region
    size_t off = 0;
    for (auto& t : pl.tensors)               // what sd_unet_set_context writes: first, so every variant agrees on it
        if (t.ctx) { t.off = off; off += t.bytes; }
    for (auto& t : pl.tensors)
        if (t.persistent && !t.ctx) { t.off = off; off += t.bytes; }
    const size_t arena0 = off;
    // arena: first-fit over live intervals
    struct Live { size_t off, bytes; int last; };
    std::vector<Live> live;
    size_t high = arena0;
    std::vector<std::vector<int>> def_at(nops);
    for (int t = 0; t < (int)pl.tensors.size(); ++t)
        if (!pl.tensors[t].persistent && pl.tensors[t].def >= 0) def_at[pl.tensors[t].def].push_back(t);
    for (int i = 0; i < nops; ++i) {
        for (int t : def_at[i]) {
            std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
            size_t cur = arena0;
            for (const Live& l : live) {
                if (l.off >= cur + pl.tensors[t].bytes) break;
                cur = std::max(cur, l.off + l.bytes);
            }
            pl.tensors[t].off = cur;
            live.push_back(Live{cur, pl.tensors[t].bytes, pl.tensors[t].last});
            high = std::max(high, cur + pl.tensors[t].bytes);
        }
        live.erase(std::remove_if(live.begin(), live.end(), [i](const Live& l) { return l.last <= i; }), live.end());
    }
    pl.total_bytes = high + 4096;
}

// CFG de-duplication applies to a forward whose UNet batch is exactly two copies of the latent batch (SD_CFG_DEDUP=0: off)
static int plan_rep(const sd_unet* u, int latent_batch, int unet_batch) {
    static const bool off = getenv("SD_CFG_DEDUP") && atoi(getenv("SD_CFG_DEDUP")) == 0;
    return (!off && u->kind == 0 && u->cfg.attn_levels[0] && latent_batch > 0 && unet_batch == 2 * latent_batch) ? 2 : 1;
}

int get_plan(sd_unet* u, int UB, int branch, Plan** out, int rep = 1) {
    SD_REQUIRE(u && u->finalized, "unet: parameters not finalized");
    SD_REQUIRE(UB > 0 && UB <= 4096, "unet: bad batch %d", UB);
    SD_REQUIRE(branch < 3 * u->cfg.num_levels, "unet: cache_branch_id %d out of range", branch);
    if (branch < 0) branch = -1;
    auto key = std::make_tuple(UB, branch, rep);
    auto it = u->plans.find(key);
    if (it == u->plans.end()) {
        Plan pl;
        pl.UB = UB;
        pl.branch = branch;
        pl.rep = rep;
        Builder b{u, pl, UB, {}};
        b.build();
        SD_REQUIRE(b.error.empty(), "unet: cannot build the plan for batch %d (cache branch %d): %s", UB, branch, b.error.c_str());
        fuse_deferred_reduce(u, pl);
        assign_memory(u, pl);
        it = u->plans.emplace(key, std::move(pl)).first;
    }
    *out = &it->second;
    return 0;
}

int run_op(sd_unet* u, const Plan& pl, const Op& o, char* ws, const float* latents, int latent_batch, float* eps_out,
           float timestep, hipStream_t stream) {
    auto T = [&](int id) -> char* { return id >= 0 ? ws + pl.tensors[id].off : nullptr; };
    const char* wb = u->dweights;
    switch (o.kind) {
        case OP_SINUSOID:
            return sd_launch_timestep_sinusoid(timestep, (float*)T(o.out), o.N, stream);
        case OP_GEMV:
            return sd_launch_gemv((const float*)T(o.x1), (const bf16_t*)(wb + o.w), (const float*)(wb + o.b),
                                  (float*)T(o.out), o.N, o.K, o.silu_in, stream);
        case OP_CONV_IN:
            return sd_launch_conv_in(o.x1 >= 0 ? (const float*)T(o.x1) : latents, o.x1 >= 0 ? o.B : latent_batch,
                                     (const float*)(wb + o.w), (const float*)(wb + o.b), (bf16_t*)T(o.out), o.B, o.Hin,
                                     o.Win, o.Cin, o.N, stream);
        case OP_PQCONV:
            return sd_launch_pqconv(latents, (const float*)(wb + o.w), (const float*)(wb + o.b), (float*)T(o.out), o.B,
                                    o.HW, timestep /* carries the latent scale for the VAE */, stream);
        case OP_SOFTMAX:
            return sd_launch_softmax_rows((bf16_t*)T(o.x1), o.M, o.N, o.scale, stream);
        case OP_GN: {
            GroupNormArgs a;
            a.x1 = (const bf16_t*)T(o.x1); a.C1 = o.C1; a.x2 = (const bf16_t*)T(o.x2); a.C2 = o.C2;
            a.gamma = (const float*)(wb + o.g); a.beta = (const float*)(wb + o.be);
            a.y = (bf16_t*)T(o.out); a.partial = (float*)T(o.aux);
            a.B = o.B; a.HW = o.HW; a.groups = u->cfg.norm_num_groups; a.nsplit = o.nsplit; a.eps = o.eps; a.silu = o.silu;
            a.out_fp8 = o.out_fp8; a.Cpad = o.Cpad; a.oscale = o.os;
            a.stats1 = (const float*)T(o.s1); a.stats2 = (const float*)T(o.s2);
            if (o.slab_t >= 0) {
                a.slab = (const float*)T(o.slab_t); a.splitk = o.slab_k; a.x1w = (bf16_t*)T(o.x1);
                a.sbias = o.slab_b != NOFF ? (const float*)(wb + o.slab_b) : nullptr;
                a.sbias2 = o.slab_b2t >= 0 ? (const float*)T(o.slab_b2t) + o.slab_b2idx : nullptr;
                a.sR = (const bf16_t*)T(o.slab_r); a.sldr = o.C1;
            }
            return sd_launch_groupnorm(a, stream);
        }
        case OP_CONV3: {
            GemmArgs a;
            a.X = (const bf16_t*)T(o.x1); a.W = (const bf16_t*)(wb + o.w); a.bias = (const float*)(wb + o.b);
            a.bias2 = o.b2t >= 0 ? (const float*)T(o.b2t) + o.b2idx : nullptr;
            a.R = (const bf16_t*)T(o.r); a.ldr = o.N; a.C = (bf16_t*)T(o.out); a.ldc = o.N;
            a.M = o.M; a.N = o.N; a.K = o.K; a.K1 = o.K;
            a.Hin = o.Hin; a.Win = o.Win; a.Cin = o.Cin; a.Hout = o.Hout; a.Wout = o.Wout; a.stride = o.stride; a.up = o.up;
            a.zero_page = g_zero_page; a.splitk = o.splitk; a.slab = (float*)T(o.aux); a.defer_reduce = o.defer;
            if (o.dt) { a.dt = 1; a.wscale = (const float*)(wb + o.wsc); a.xscale_inv = 1.0f / o.xs; }
            a.stats = (float*)T(o.stats);
            if (o.subpix) { a.subpix = 1; a.up = 0; a.w_batch_stride = (long)o.N * 4 * o.Cin; }
            return sd_launch_conv3x3(a, stream);
        }
        case OP_GEMM: {
            GemmArgs a;
            a.X = o.wx != NOFF ? (const bf16_t*)(wb + o.wx) : (const bf16_t*)T(o.x1) + o.xoff;
            a.ldx = o.ldx_o ? o.ldx_o : o.K1; a.X2 = (const bf16_t*)T(o.x2); a.ldx2 = o.K - o.K1; a.K1 = o.K1;
            a.W = o.wt >= 0 ? (const bf16_t*)T(o.wt) + o.woff_el : (const bf16_t*)(wb + o.w); a.ldw = o.ldw_o;
            a.bias = o.b != NOFF ? (const float*)(wb + o.b) : nullptr;
            a.R = (const bf16_t*)T(o.r); a.ldr = o.N; a.C = (bf16_t*)T(o.out) + o.coff;
            a.ldc = o.ldc_o ? o.ldc_o : (o.epi == 1 ? o.N / 2 : o.N);
            a.M = o.M; a.N = o.N; a.K = o.K; a.zero_page = g_zero_page; a.splitk = o.splitk; a.slab = (float*)T(o.aux);
            a.defer_reduce = o.defer;
            a.w_batch_stride = o.wbs; a.rows_per_batch = o.rpb; a.sm_valid = o.sm_valid;
            if (o.dt) { a.dt = 1; a.wscale = (const float*)(wb + o.wsc); a.xscale_inv = 1.0f / o.xs; }
            if (o.out_fp8) { a.out_fp8 = 1; a.oscale = o.os; a.ldc = o.Cpad; }
            a.stats = (float*)T(o.stats);
            a.rowstats = (float*)T(o.rs);
            if (o.hm) { a.hm_C = o.N / 3; a.hm_tok = o.HW; a.ldc = a.hm_C; a.KV = (bf16_t*)T(o.out) + (long)o.M * a.hm_C; }
            if (o.lnrs >= 0) { a.ln_rs = (const float*)T(o.lnrs); a.ln_np = o.lnnp; a.ln_c1 = (const float*)(wb + o.c1); a.ln_eps = 1e-5f; }
            if (o.lnrs >= 0 && o.s1 >= 0) { a.ln_c1 = (const float*)T(o.s1); a.bias = (const float*)T(o.s2); a.ln_per_sample = 1; }   // per-sample weights
            return sd_launch_gemm(a, o.epi, stream);
        }
        case OP_LN:
            if (o.out_fp8)
                return sd_launch_layernorm_fp8((const bf16_t*)T(o.x1), (const float*)(wb + o.g), (const float*)(wb + o.be),
                                               T(o.out), o.M, o.N, o.Cpad, o.eps, o.os, stream);
            return sd_launch_layernorm((const bf16_t*)T(o.x1), (const float*)(wb + o.g), (const float*)(wb + o.be),
                                       (bf16_t*)T(o.out), o.M, o.N, o.eps, stream);
        case OP_ATTN: {
            AttnArgs a;
            a.Q = (const bf16_t*)T(o.x1) + o.qoff; a.ldq = o.ldq;
            a.K = (const bf16_t*)T(o.x2) + o.koff; a.ldk = o.ldk;
            a.V = (const bf16_t*)T(o.x2) + o.voff; a.ldv = o.ldv;
            a.O = (bf16_t*)T(o.out); a.ldo = o.ldo;
            a.B = o.B; a.heads = o.heads; a.Nq = o.Nq; a.Nk = o.Nk; a.D = o.D;
            a.scale = 1.0f / sqrtf((float)o.D);
            a.q_prescaled = o.qps;
            a.consts = g_zero_page;
            a.kv_head_major = o.hm;
            return sd_launch_attention(a, stream);
        }
        case OP_XATTN: {
            XattnArgs a;
            a.X = (const bf16_t*)T(o.x1); a.R = (const bf16_t*)T(o.r); a.Y = (bf16_t*)T(o.out);
            a.At = (const bf16_t*)T(o.wt); a.Bw = (const bf16_t*)T(o.x2); a.bias = (const float*)(wb + o.b);
            a.M = o.M; a.C = o.N; a.rows_per_sample = o.rpb; a.L = o.sm_valid;
            a.rowstats = (float*)T(o.rs);
            if (o.lnrs >= 0) {      // norm2 folded in: X = the un-normalised residual stream
                a.ln_rs = (const float*)T(o.lnrs); a.ln_np = o.lnnp; a.ln_rows = o.ldx_o; a.ln_c2 = (const float*)T(o.s1); a.ln_eps = 1e-5f;
            }
            return sd_launch_xattn_fused(a, stream);
        }
        case OP_REPLICATE:
            return sd_launch_replicate(T(o.x1), T(o.out), (long)o.M * 16, o.N, stream);
        case OP_CLIP_EMBED:
            return sd_launch_clip_embed((const int*)latents, (const bf16_t*)(wb + o.w), (const bf16_t*)(wb + o.g),
                                        (bf16_t*)T(o.out), o.M, o.Nk, o.N, u->clip.vocab_size, stream);
        case OP_CLIP_ATTN:
            return sd_launch_clip_attention((const bf16_t*)T(o.x1), (bf16_t*)T(o.out), o.B, o.Nq, o.N, o.heads, stream);
        case OP_QGELU:
            return sd_launch_quick_gelu((bf16_t*)T(o.x1), (long)o.M * o.N, stream);
        case OP_TO_F32:
            return sd_launch_bf16_to_f32((const bf16_t*)T(o.x1), eps_out, (long)o.M * o.N, stream);
        case OP_CONV_OUT:
            return sd_launch_conv_out((const bf16_t*)T(o.x1), (const bf16_t*)(wb + o.w), (const float*)(wb + o.b), eps_out,
                                      o.B, o.Hin, o.Win, o.Cin, o.N, stream);
    }
    sd_set_error("unet: unknown op kind %d", o.kind);
    return -1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int sd_unet_create(const sd_unet_config* cfg, sd_unet** out) {
    SD_REQUIRE(cfg && out, "sd_unet_create: null argument");
    SD_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 8, "sd_unet_create: num_levels %d", cfg->num_levels);
    SD_REQUIRE(cfg->in_channels == 4 && cfg->out_channels >= 1 && cfg->out_channels <= 4, "sd_unet_create: in/out channels");
    for (int i = 0; i < cfg->num_levels; ++i) {
        const int c = cfg->block_out_channels[i];
        SD_REQUIRE(c % 64 == 0 && c % cfg->norm_num_groups == 0 && c / cfg->norm_num_groups >= 8,
                   "sd_unet_create: block_out_channels[%d]=%d must be a multiple of 64 with >= 8 channels per group", i, c);
        if (cfg->attn_levels[i]) {
            const int d = c / cfg->num_heads;
            SD_REQUIRE(c % cfg->num_heads == 0 && (d == 40 || d == 80 || d == 160),
                       "sd_unet_create: head dim %d at level %d not built (40/80/160)", d, i);
        }
    }
    const int dmid = cfg->block_out_channels[cfg->num_levels - 1] / cfg->num_heads;
    SD_REQUIRE(dmid == 40 || dmid == 80 || dmid == 160, "sd_unet_create: mid-block head dim %d not built", dmid);
    SD_REQUIRE(cfg->cross_attention_dim % 64 == 0, "sd_unet_create: cross_attention_dim must be a multiple of 64");
    SD_REQUIRE(cfg->sample_size % (1 << (cfg->num_levels - 1)) == 0, "sd_unet_create: sample_size not divisible");
    SD_REQUIRE(cfg->context_len >= 1, "sd_unet_create: context_len");
    SD_REQUIRE(cfg->weight_dtype == SD_DTYPE_BF16 || cfg->weight_dtype == SD_DTYPE_FP8_E4M3,
               "sd_unet_create: weight_dtype %d (0 = bf16, 1 = fp8 e4m3)", cfg->weight_dtype);
    SD_REQUIRE(cfg->fp8_act_scale_norm >= 0.f && cfg->fp8_act_scale_ff >= 0.f, "sd_unet_create: negative fp8 activation scale");
    sd_unet* u = new sd_unet();   // no device work here: parameter enumeration also runs on a CPU-only box
    u->cfg = *cfg;
    u->fp8 = cfg->weight_dtype == SD_DTYPE_FP8_E4M3;
    if (cfg->fp8_act_scale_norm > 0.f) u->s_norm = cfg->fp8_act_scale_norm;
    if (cfg->fp8_act_scale_ff > 0.f) u->s_ff = cfg->fp8_act_scale_ff;
    u->debug_taps = getenv("SD_DEBUG_TAPS") != nullptr;
    enumerate_params(u);
    *out = u;
    return 0;
}

// ---- AutoencoderKL decoder: `self.vae.decode(latents / scaling_factor)` of src/models.py:287-302 ----
extern "C" int sd_vae_create(const sd_unet_config* cfg, sd_unet** out) {
    SD_REQUIRE(cfg && out, "sd_vae_create: null argument");
    SD_REQUIRE(cfg->num_levels >= 1 && cfg->num_levels <= 8, "sd_vae_create: num_levels %d", cfg->num_levels);
    SD_REQUIRE(cfg->in_channels == 4 && cfg->out_channels >= 1 && cfg->out_channels <= 4, "sd_vae_create: in/out channels");
    for (int i = 0; i < cfg->num_levels; ++i) {
        const int c = cfg->block_out_channels[i], cpg = cfg->norm_num_groups ? c / cfg->norm_num_groups : 0;
        SD_REQUIRE(c % 64 == 0 && c % cfg->norm_num_groups == 0 && (cpg >= 8 || cpg == 4),
                   "sd_vae_create: block_out_channels[%d]=%d must be a multiple of 64 with 4 or >= 8 channels per group", i, c);
    }
    SD_REQUIRE(cfg->sample_size >= 8 && cfg->sample_size <= 64 && (cfg->sample_size * cfg->sample_size) % 64 == 0,
               "sd_vae_create: latent size %d (the mid-block attention handles 64 <= HW <= 4096 tokens)", cfg->sample_size);
    sd_unet* u = new sd_unet();
    u->kind = 1;
    u->cfg = *cfg;
    u->debug_taps = getenv("SD_DEBUG_TAPS") != nullptr;
    enumerate_params_vae(u);
    *out = u;
    return 0;
}

extern "C" int sd_vae_decode(sd_unet* u, void* stream, const float* latents, int batch, float latent_scale,
                             float* images_out, void* workspace, long long workspace_bytes) {
    SD_REQUIRE(u && u->kind == 1, "vae_decode: not a VAE handle");
    SD_REQUIRE(latents && images_out && workspace && batch > 0, "vae_decode: null argument");
    Plan* pl;
    int rc = get_plan(u, batch, -1, &pl);
    if (rc) return rc;
    SD_REQUIRE((long long)pl->total_bytes <= workspace_bytes, "vae_decode: workspace too small (%lld < %zu)", workspace_bytes,
               pl->total_bytes);
    SD_REQUIRE(((uintptr_t)workspace & 255) == 0, "vae_decode: workspace must be 256-byte aligned");
    for (size_t i = 0; i < pl->ops.size(); ++i)
        if ((rc = run_op(u, *pl, pl->ops[i], (char*)workspace, latents, batch, images_out, latent_scale, (hipStream_t)stream)))
            return rc;
    return 0;
}

// ---- CLIP text encoder: `self.text_encoder(text_input_ids)[0]` inside encode_prompt (src/models.py:139-155) ----
extern "C" int sd_clip_create(const sd_clip_config* cfg, sd_unet** out) {
    SD_REQUIRE(cfg && out, "sd_clip_create: null argument");
    SD_REQUIRE(cfg->hidden_size % 64 == 0 && cfg->intermediate_size % 64 == 0,
               "sd_clip_create: hidden %d / intermediate %d must be multiples of 64", cfg->hidden_size, cfg->intermediate_size);
    SD_REQUIRE(cfg->hidden_size <= 1536, "sd_clip_create: hidden size %d (LayerNorm kernel handles <= 1536)", cfg->hidden_size);
    SD_REQUIRE(cfg->num_heads > 0 && cfg->hidden_size % cfg->num_heads == 0, "sd_clip_create: heads");
    const int d = cfg->hidden_size / cfg->num_heads;
    SD_REQUIRE(d == 64 || d == 16, "sd_clip_create: head dim %d (64 and 16 are built)", d);
    SD_REQUIRE(cfg->max_positions >= 1 && cfg->max_positions <= 128, "sd_clip_create: max_positions %d", cfg->max_positions);
    SD_REQUIRE(cfg->vocab_size >= 1 && cfg->num_layers >= 1, "sd_clip_create: vocab/layers");
    SD_REQUIRE(fabsf(cfg->layer_norm_eps - 1e-5f) < 1e-9f, "sd_clip_create: layer_norm_eps %g (1e-5 is built)", cfg->layer_norm_eps);
    sd_unet* u = new sd_unet();
    u->kind = 2;
    memset(&u->cfg, 0, sizeof(u->cfg));
    u->cfg.num_levels = 1;
    u->clip = *cfg;
    enumerate_params_clip(u);
    *out = u;
    return 0;
}

extern "C" int sd_clip_encode(sd_unet* u, void* stream, const int* input_ids, int batch, float* hidden_out,
                              void* workspace, long long workspace_bytes) {
    SD_REQUIRE(u && u->kind == 2, "clip_encode: not a CLIP handle");
    SD_REQUIRE(input_ids && hidden_out && workspace && batch > 0, "clip_encode: null argument");
    Plan* pl;
    int rc = get_plan(u, batch, -1, &pl);
    if (rc) return rc;
    SD_REQUIRE((long long)pl->total_bytes <= workspace_bytes, "clip_encode: workspace too small (%lld < %zu)", workspace_bytes,
               pl->total_bytes);
    SD_REQUIRE(((uintptr_t)workspace & 255) == 0, "clip_encode: workspace must be 256-byte aligned");
    for (size_t i = 0; i < pl->ops.size(); ++i)
        if ((rc = run_op(u, *pl, pl->ops[i], (char*)workspace, (const float*)input_ids, batch, hidden_out, 0.f, (hipStream_t)stream)))
            return rc;
    return 0;
}

extern "C" void sd_unet_destroy(sd_unet* u) {
    if (!u) return;
    if (u->dweights) (void)hipFree(u->dweights);
    delete u;
}

extern "C" int sd_unet_num_params(const sd_unet* u) { return u ? (int)u->params.size() : -1; }

extern "C" int sd_unet_param_info(const sd_unet* u, int index, char* name, int name_cap, long long shape[4], int* ndim) {
    SD_REQUIRE(u && index >= 0 && index < (int)u->params.size(), "param_info: bad index %d", index);
    const ParamSpec& p = u->params[index];
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", p.name.c_str());
    for (int i = 0; i < 4; ++i) shape[i] = i < (int)p.shape.size() ? p.shape[i] : 1;
    if (ndim) *ndim = (int)p.shape.size();
    return 0;
}

extern "C" int sd_unet_load_param(sd_unet* u, const char* name, const float* host_data, long long numel) {
    SD_REQUIRE(u && name && host_data, "load_param: null argument");
    SD_REQUIRE(!u->finalized, "load_param: handle already finalized");
    auto it = u->pindex.find(name);
    SD_REQUIRE(it != u->pindex.end(), "load_param: unknown parameter '%s'", name);
    ParamSpec& p = u->params[it->second];
    SD_REQUIRE(p.numel() == numel, "load_param: '%s' expects %lld elements, got %lld", name, p.numel(), numel);
    p.data.assign(host_data, host_data + numel);
    p.loaded = true;
    return 0;
}

extern "C" int sd_unet_finalize(sd_unet* u) {
    SD_REQUIRE(u, "finalize: null handle");
    SD_REQUIRE(!u->finalized, "finalize: already finalized");
    for (auto& p : u->params) SD_REQUIRE(p.loaded, "finalize: parameter '%s' was never loaded", p.name.c_str());
    g_alloc_s = 0;
    const auto tp0 = std::chrono::steady_clock::now();
    {   // one reservation for the staging blob: growing it piecemeal re-copied it again and again (19 of the 22 s a UNet
        // finalize used to take).  In AGGREGATE the packed blob (2.26 GB for the bf16 UNet: bf16 weights, some of them twice --
        // sub-pixel upsamplers, *.ln copies, ff_out, attn2.to_q.T) stays below the fp32 parameters' bytes (3.44 GB), so this
        // rarely grows; if it has to, HostBlob::resize re-maps at twice the size.
        size_t total = 0;
        for (auto& p : u->params) total += p.data.size() * 4;
        SD_REQUIRE(u->hblob.reserve(total + (64u << 20)), "finalize: cannot map %zu bytes of host staging memory", total + (64u << 20));
    }
    try {
        if (pack_all(u)) return -1;      // host-only: repacks / quantises into the staging blob (runs without a GPU too)
    } catch (const std::bad_alloc&) {
        u->hblob.release();
        sd_set_error("finalize: out of host memory while packing the weights");
        return -2;
    }
    if (getenv("SD_PACK_TIMING"))
        fprintf(stderr, "libsdhip: pack %.2f s (of which staging-blob growth %.2f s), %zu bytes\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count(), g_alloc_s, u->hblob.size());
    if (ensure_zero_page()) return -2;
    SD_CHECK_HIP(hipMalloc((void**)&u->dweights, u->hblob.size()));
    SD_CHECK_HIP(hipMemcpy(u->dweights, u->hblob.data(), u->hblob.size(), hipMemcpyHostToDevice));
    u->hblob.release();
    for (auto& p : u->params) std::vector<float>().swap(p.data);
    u->finalized = true;
    return 0;
}

extern "C" long long sd_unet_debug_packed(const sd_unet* u, const char* key, void* host_out, long long nbytes) {
    SD_REQUIRE(u && key, "debug_packed: null argument");
    auto it = u->woff.find(key);
    SD_REQUIRE(it != u->woff.end(), "debug_packed: no packed item '%s'", key);
    SD_REQUIRE(!u->hblob.empty(), "debug_packed: the host staging blob is released once the weights are on the device");
    SD_REQUIRE(nbytes >= 0 && it->second + (size_t)nbytes <= u->hblob.size(), "debug_packed: '%s' + %lld bytes exceeds the blob", key, nbytes);
    if (host_out && nbytes) memcpy(host_out, u->hblob.data() + it->second, (size_t)nbytes);
    return (long long)it->second;
}

extern "C" long long sd_unet_workspace_bytes(sd_unet* u, int unet_batch, int cache_branch_id) {
    Plan* pl;
    if (get_plan(u, unet_batch, cache_branch_id, &pl)) return -1;
    size_t bytes = pl->total_bytes;
    if (unet_batch % 2 == 0 && plan_rep(u, unet_batch / 2, unet_batch) == 2) {      // the CFG-pair variant of the plan
        if (get_plan(u, unet_batch, cache_branch_id, &pl, 2)) return -1;
        bytes = std::max(bytes, pl->total_bytes);
    }
    return (long long)bytes;
}

extern "C" int sd_unet_set_context(sd_unet* u, void* stream, const float* ehs, int unet_batch, int cache_branch_id,
                                   void* workspace, long long workspace_bytes) {
    SD_REQUIRE(ehs && workspace, "set_context: null argument");
    Plan* plp;
    int rc = get_plan(u, unet_batch, cache_branch_id, &plp);
    if (rc) return rc;
    Plan& pl = *plp;
    SD_REQUIRE((long long)pl.total_bytes <= workspace_bytes, "set_context: workspace too small (%lld < %zu)",
               workspace_bytes, pl.total_bytes);
    SD_REQUIRE(((uintptr_t)workspace & 255) == 0, "set_context: workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    const int L = u->cfg.context_len, CD = u->cfg.cross_attention_dim, M = unet_batch * L;
    bf16_t* cb = (bf16_t*)(ws + pl.tensors[pl.ctx_bf16].off);
    if ((rc = sd_launch_f32_to_bf16(ehs, cb, (long)M * CD, (hipStream_t)stream))) return rc;
    for (size_t i = 0; i < pl.ctx_kv.size(); ++i) {
        GemmArgs a;
        a.X = cb; a.ldx = CD; a.K1 = CD; a.K = CD; a.M = M; a.N = 2 * pl.ctx_c[i];
        a.W = (const bf16_t*)(u->dweights + pl.ctx_w[i]);
        a.C = (bf16_t*)(ws + pl.tensors[pl.ctx_kv[i]].off); a.ldc = a.N;
        a.zero_page = g_zero_page;
        if ((rc = sd_launch_gemm(a, 0, (hipStream_t)stream))) return rc;
    }
    // folded prompt cross-attention: A^T = (scale K)_masked . W_q  and  B = W_o . V_masked^T per sample and layer
    const int NH = u->cfg.num_heads, NP = NH * 80;
    for (const Plan::Fold& f : pl.ctx_fold) {
        const int C = f.C, d = C / NH;
        const bf16_t* kvp = (const bf16_t*)(ws + pl.tensors[f.kv].off);
        bf16_t* kexp = (bf16_t*)(ws + pl.tensors[pl.ctx_fold_scratch].off);
        bf16_t* vexp = kexp + (size_t)unet_batch * NP * C;
        if ((rc = sd_launch_xattn_expand(kvp, kexp, unet_batch, L, C, NH, 0, 1.0f / sqrtf((float)d), (hipStream_t)stream))) return rc;
        if ((rc = sd_launch_xattn_expand(kvp, vexp, unet_batch, L, C, NH, C, 1.0f, (hipStream_t)stream))) return rc;
        // fused kernel: both operands are re-tiled ([32-column slice][rows][64 B]) so that its DMA pieces are contiguous;
        // the plain layouts land in the V expansion's scratch first (free after the second GEMM below has read it ... the
        // A^T GEMM output goes there BEFORE that GEMM runs, so it is re-tiled right away)
        bf16_t* at_dst = (bf16_t*)(ws + pl.tensors[f.at].off);
        bf16_t* tmp = (bf16_t*)(ws + pl.tensors[pl.ctx_fold_scratch].off) + (size_t)2 * unet_batch * NP * C;   // third scratch slab
        {   // A^T [UB*NP, C]: rows (sample, head, key), K-contiguous over the UNet channel -> W operand of GEMM 1
            GemmArgs a;
            a.X = kexp; a.ldx = C; a.K1 = C; a.K = C; a.M = unet_batch * NP; a.N = C;
            a.W = (const bf16_t*)(u->dweights + f.wqT);
            a.C = f.perm ? tmp : at_dst; a.ldc = C;
            a.zero_page = g_zero_page;
            if ((rc = sd_launch_gemm(a, 0, (hipStream_t)stream))) return rc;
            if (f.perm && (rc = sd_launch_retile32(tmp, at_dst, unet_batch, NP, C, NP, (hipStream_t)stream))) return rc;
        }
        if (f.c1 >= 0) {    // what is left of the mean term: c1[sample][slot] = sum over the channel of the ROUNDED centred row
            if ((rc = sd_launch_gemv((const float*)(u->dweights + f.ones), at_dst, nullptr, (float*)(ws + pl.tensors[f.c1].off),
                                     unet_batch * NP, C, 0, (hipStream_t)stream))) return rc;
        }
        if (f.c2 >= 0) {    // beta term of every key slot: c2[sample][slot] = (scale K_h[slot]) . u   (fp32 GEMV over the expanded rows)
            if ((rc = sd_launch_gemv((const float*)(u->dweights + f.lnu), kexp, nullptr, (float*)(ws + pl.tensors[f.c2].off),
                                     unet_batch * NP, C, 0, (hipStream_t)stream))) return rc;
        }
        {   // B^T [UB*NP, C] = V_masked . W_o^T (one GEMM, into the K expansion's scratch), then transposed per sample
            // to [C, NP]: K-contiguous over (head, key) -> W operand of GEMM 2
            GemmArgs a;
            a.X = vexp; a.ldx = C; a.K1 = C; a.K = C; a.M = unet_batch * NP; a.N = C;
            a.W = (const bf16_t*)(u->dweights + f.wo);
            a.C = kexp; a.ldc = C;
            a.zero_page = g_zero_page;
            if ((rc = sd_launch_gemm(a, 0, (hipStream_t)stream))) return rc;
            bf16_t* bw_dst = (bf16_t*)(ws + pl.tensors[f.bw].off);
            if ((rc = sd_launch_transpose_bf16(kexp, f.perm ? tmp : bw_dst, unet_batch, NP, C, (hipStream_t)stream, f.perm ? 1 : 0)))
                return rc;
            if (f.perm && (rc = sd_launch_retile32(tmp, bw_dst, unet_batch, C, NP, 32, (hipStream_t)stream))) return rc;
        }
    }
    return 0;
}

extern "C" int sd_unet_forward(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                               float timestep, float* eps_out, void* workspace, long long workspace_bytes,
                               int cache_mode, int cache_branch_id) {
    SD_REQUIRE(u && u->kind == 0, "forward: not a UNet handle");
    SD_REQUIRE(latents && eps_out && workspace, "forward: null argument");
    SD_REQUIRE(latent_batch > 0 && unet_batch % latent_batch == 0, "forward: unet batch %d not a multiple of latent batch %d",
               unet_batch, latent_batch);
    SD_REQUIRE(cache_mode >= 0 && cache_mode <= 2, "forward: cache_mode %d", cache_mode);
    SD_REQUIRE(cache_mode == SD_CACHE_OFF || cache_branch_id >= 0, "forward: DeepCache modes need cache_branch_id >= 0");
    Plan* pl;
    const int rep = plan_rep(u, latent_batch, unet_batch);
    u->last_rep = rep;
    int rc = get_plan(u, unet_batch, cache_branch_id, &pl, rep);
    if (rc) return rc;
    SD_REQUIRE((long long)pl->total_bytes <= workspace_bytes, "forward: workspace too small (%lld < %zu)", workspace_bytes,
               pl->total_bytes);
    SD_REQUIRE(((uintptr_t)workspace & 255) == 0, "forward: workspace must be 256-byte aligned");
    for (size_t i = 0; i < pl->ops.size(); ++i) {
        if (cache_mode == SD_CACHE_SKIP && pl->skipped[i]) continue;
        if ((rc = run_op(u, *pl, pl->ops[i], (char*)workspace, latents, latent_batch, eps_out, timestep, (hipStream_t)stream)))
            return rc;
    }
    return 0;
}

// ---- fp8 activation-scale calibration (SD_DTYPE_FP8_E4M3 handles) -------------------------------------------------------
// e4m3 is a floating-point format: a per-tensor scale buys no precision, it only positions the representable range
// (+-448 down to 2^-9) over the tensor's values.  The static defaults (8 for norm outputs, 2 for the GEGLU product) clip
// at |x| > 56 / 224; real checkpoints have layers beyond that.  Calibration runs ONE forward of the plan on the caller's
// inputs op by op; every op that writes an e4m3 activation tensor is run twice: first with a probe scale under which
// nothing can saturate, its largest |value| is read back from the e4m3 codes (a small reduction kernel + one host sync per
// tensor: ~125 per forward, calibration only), then with its final scale  448 / (margin * amax)  so that the ops
// downstream see correctly quantised inputs.  amax accumulates (max) over calls: calibrate on several timesteps / prompts,
// then every plan is rebuilt with the new scales.
static float e4m3_code_value(unsigned c) {
    c &= 0x7f;
    if (c == 0x7f) return 448.f;                       // NaN code: only a saturated probe could produce it
    const int e = (int)(c >> 3), m = (int)(c & 7);
    return e == 0 ? m * (1.f / 512.f) : (1.f + m / 8.f) * ldexpf(1.f, e - 7);
}

extern "C" int sd_unet_calibrate_fp8(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                                     float timestep, float margin, void* workspace, long long workspace_bytes) {
    SD_REQUIRE(u && u->kind == 0 && u->fp8, "calibrate_fp8: not an fp8 UNet handle");
    SD_REQUIRE(latents && workspace && latent_batch > 0 && unet_batch % latent_batch == 0, "calibrate_fp8: bad arguments");
    SD_REQUIRE(margin >= 1.f && margin <= 64.f, "calibrate_fp8: margin %g (1 .. 64: headroom over the observed amax)", margin);
    Plan* plp;
    const int rep = plan_rep(u, latent_batch, unet_batch);
    int rc = get_plan(u, unet_batch, -1, &plp, rep);
    if (rc) return rc;
    SD_REQUIRE((long long)plp->total_bytes <= workspace_bytes, "calibrate_fp8: workspace too small (%lld < %zu)", workspace_bytes,
               plp->total_bytes);
    SD_REQUIRE(((uintptr_t)workspace & 255) == 0, "calibrate_fp8: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    unsigned* dmax = (unsigned*)op_scratch(256);
    SD_REQUIRE(dmax, "calibrate_fp8: cannot allocate scratch");
    std::vector<float> eps((size_t)unet_batch * u->cfg.out_channels * u->cfg.sample_size * u->cfg.sample_size);
    float* deps = nullptr;
    SD_CHECK_HIP(hipMalloc((void**)&deps, eps.size() * 4));
    std::map<int, float> cur;                          // e4m3 tensor -> scale it was written with in THIS pass
    const Plan& pl = *plp;
    auto measure = [&](const Op& o, float probe, float* amax) -> int {
        Op t = o; t.os = probe;
        int r = run_op(u, pl, t, (char*)workspace, latents, latent_batch, deps, timestep, st);
        if (r) return r;
        SD_CHECK_HIP(hipMemsetAsync(dmax, 0, 4, st));
        const Tn& tn = pl.tensors[o.out];
        const long rows = o.kind == OP_GN ? (long)o.B * o.HW : (long)o.M;
        if ((r = sd_launch_amax_e4m3((char*)workspace + tn.off, rows * o.Cpad, dmax, st))) return r;
        unsigned code = 0;
        SD_CHECK_HIP(hipMemcpyAsync(&code, dmax, 4, hipMemcpyDeviceToHost, st));
        SD_CHECK_HIP(hipStreamSynchronize(st));
        *amax = code >= 0x7e ? -1.f : e4m3_code_value(code) / probe;       // -1: the probe itself saturated
        return 0;
    };
    for (size_t i = 0; i < pl.ops.size() && !rc; ++i) {
        Op o = pl.ops[i];
        if (o.dt) {                                    // consumer of an e4m3 tensor: the scale it was just written with
            auto it = cur.find(o.x1);
            if (it != cur.end()) o.xs = it->second;
        }
        if (o.out_fp8 && o.sname >= 0) {
            float amax = 0.f;
            rc = measure(o, 448.f / 4096.f, &amax);                        // |x| up to 4096, resolved down to ~0.15
            if (!rc && amax < 0.f) rc = measure(o, 448.f / 1048576.f, &amax);
            if (rc) break;
            if (amax < 0.f) amax = 1048576.f;
            float& seen = u->act_amax[o.sname];
            seen = std::max(seen, amax);
            // a power-of-two scale: the e4m3 grid is then an exact sub-grid of bf16 / fp32 values (reproducible emulation)
            const float want = 448.f / (margin * std::max(seen, 1e-6f));
            const float sc = ldexpf(1.f, (int)floorf(log2f(want)));
            u->act_scale[o.sname] = std::min(std::max(sc, ldexpf(1.f, -20)), ldexpf(1.f, 20));
            o.os = u->act_scale[o.sname];
            cur[o.out] = o.os;
        }
        rc = run_op(u, pl, o, (char*)workspace, latents, latent_batch, deps, timestep, st);
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(deps);
    if (rc) return rc;
    u->plans.clear();                                  // every plan variant is rebuilt with the calibrated scales
    return 0;
}

extern "C" int sd_unet_fp8_scale_count(const sd_unet* u) { return u ? (int)u->act_names.size() : -1; }

extern "C" int sd_unet_fp8_scale_info(const sd_unet* u, int index, char* name, int name_cap, float* scale, float* amax) {
    SD_REQUIRE(u && index >= 0 && index < (int)u->act_names.size(), "fp8_scale_info: bad index %d", index);
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", u->act_names[index].c_str());
    if (scale) *scale = u->act_scale[index];
    if (amax) *amax = u->act_amax[index];
    return 0;
}

extern "C" int sd_unet_set_fp8_scale(sd_unet* u, const char* name, float scale) {
    SD_REQUIRE(u && u->kind == 0 && u->fp8 && name, "set_fp8_scale: not an fp8 UNet handle");
    SD_REQUIRE(scale > 0.f && scale == scale, "set_fp8_scale: scale %g", scale);
    auto it = u->act_index.find(name);
    SD_REQUIRE(it != u->act_index.end(), "set_fp8_scale: no e4m3 activation tensor '%s' (build a plan first: sd_unet_workspace_bytes)", name);
    u->act_scale[it->second] = scale;
    u->plans.clear();
    return 0;
}

// algorithmic work of one op: flops for the contraction kernels, minimal HBM bytes for the rest
static void op_work(const Op& o, double* flops, double* bytes) {
    *flops = 0; *bytes = 0;
    switch (o.kind) {
        case OP_CONV3:
        case OP_GEMM: {
            const double K = o.Kalg ? o.Kalg : o.K, esz = o.dt ? 1.0 : 2.0;
            *flops = 2.0 * o.M * o.N * K;
            *bytes = esz * ((double)o.M * K + (double)o.N * K) + (o.out_fp8 ? 1.0 : 2.0) * (double)o.M * (o.epi == 1 ? o.N / 2 : o.N);
            break;
        }
        case OP_XATTN:   // to_q + Q K^T + P V + to_out of the block (to_k / to_v are hoisted out of the loop); X, R in, Y out
            *flops = 4.0 * o.M * (double)o.N * o.N + 4.0 * o.M * (double)o.sm_valid * o.N;
            *bytes = 3.0 * 2.0 * o.M * o.N;
            break;
        case OP_ATTN:
            *flops = 4.0 * o.B * o.heads * (double)o.Nq * o.Nk * o.D;
            *bytes = 2.0 * o.B * o.heads * o.D * (2.0 * o.Nq + 2.0 * o.Nk);
            break;
        case OP_GN:
            *bytes = 2.0 * 2.0 * o.B * (double)o.HW * (o.C1 + o.C2);   // read once + write once, bf16
            break;
        case OP_LN:
            *bytes = 2.0 * 2.0 * (double)o.M * o.N;
            break;
        case OP_REPLICATE:
            *bytes = 16.0 * o.M * (1.0 + o.N);
            break;
        default:
            break;
    }
}

// One forward with ONE event between consecutive launches (op j's time = e[j+1] - e[j]): a pair per launch put two markers
// between any two kernels and over-read every launch by ~10 us against the rocprofv3 kernel trace of the same forward; one
// marker leaves ~4-5 us (the launch latency a free-running stream hides under the previous kernel).
static int profiled_run(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch, float timestep,
                        float* eps_out, void* workspace, long long workspace_bytes, int cache_mode, int cache_branch_id,
                        Plan** plan, std::vector<std::pair<int, float>>* per_op) {
    SD_REQUIRE(latents && eps_out && workspace, "forward_profiled: null argument");
    SD_REQUIRE(latent_batch > 0 && unet_batch % latent_batch == 0, "forward_profiled: bad batch");
    Plan* pl;
    const int rep = plan_rep(u, latent_batch, unet_batch);
    u->last_rep = rep;
    int rc = get_plan(u, unet_batch, cache_branch_id, &pl, rep);
    if (rc) return rc;
    *plan = pl;
    SD_REQUIRE((long long)pl->total_bytes <= workspace_bytes, "forward_profiled: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev;
    std::vector<int> which;
    auto mark = [&]() -> int {
        hipEvent_t e;
        SD_CHECK_HIP(hipEventCreate(&e));
        SD_CHECK_HIP(hipEventRecord(e, st));
        ev.push_back(e);
        return 0;
    };
    if (mark()) return -2;
    for (size_t i = 0; i < pl->ops.size(); ++i) {
        if (cache_mode == SD_CACHE_SKIP && pl->skipped[i]) continue;
        rc = run_op(u, *pl, pl->ops[i], (char*)workspace, latents, latent_batch, eps_out, timestep, st);
        if (mark()) return -2;
        which.push_back((int)i);
        if (rc) break;
    }
    SD_CHECK_HIP(hipStreamSynchronize(st));
    for (size_t j = 0; j < which.size(); ++j) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[j], ev[j + 1]);
        per_op->push_back({which[j], ms});
    }
    for (auto e : ev) (void)hipEventDestroy(e);
    return rc;
}

extern "C" int sd_unet_forward_profiled(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                                        float timestep, float* eps_out, void* workspace, long long workspace_bytes,
                                        int cache_mode, int cache_branch_id, double kind_ms[SD_PROFILE_KINDS],
                                        long long kind_launches[SD_PROFILE_KINDS], double kind_flops[SD_PROFILE_KINDS],
                                        double kind_bytes[SD_PROFILE_KINDS]) {
    SD_REQUIRE(kind_ms && kind_launches && kind_flops && kind_bytes, "forward_profiled: null argument");
    Plan* pl = nullptr;
    std::vector<std::pair<int, float>> per_op;
    const int rc = profiled_run(u, stream, latents, latent_batch, unet_batch, timestep, eps_out, workspace, workspace_bytes,
                                cache_mode, cache_branch_id, &pl, &per_op);
    if (!pl) return rc;
    for (int k = 0; k < SD_PROFILE_KINDS; ++k) { kind_ms[k] = 0; kind_launches[k] = 0; kind_flops[k] = 0; kind_bytes[k] = 0; }
    for (auto& [i, ms] : per_op) {
        const Op& o = pl->ops[i];
        double fl, by;
        op_work(o, &fl, &by);
        // 4: the stride-1 convs on conv_halo_kernel's 9-tap mode (the roofline kernel); 21: the sub-pixel upsamplers that
        // sd_launch_conv3x3 puts on its 4-tap mode; 20: what runs on the implicit-GEMM kernel -- the stride-2 convs and the
        // sub-pixel upsamplers the 4-tap mode does not take (8x8 -> 16x16 at the bench batch).  The SAME predicate as the launch.
        bool halo4 = false;
        if (o.kind == OP_CONV3 && !o.dt && o.subpix) {
            GemmArgs a;
            a.M = o.M; a.N = o.N; a.K = o.K; a.K1 = o.K; a.Hin = o.Hin; a.Win = o.Win; a.Cin = o.Cin; a.Hout = o.Hout; a.Wout = o.Wout;
            a.stride = o.stride; a.up = 0; a.subpix = 1; a.splitk = o.splitk; a.slab = (float*)(o.aux >= 0 ? (void*)1 : nullptr);
            a.w_batch_stride = (long)o.N * 4 * o.Cin;
            halo4 = sd_conv_halo_subpix_applicable(a);
        }
        const int kd = o.kind == OP_XATTN ? 18 : o.kind == OP_REPLICATE ? 19 : halo4 ? 21 :
                       (o.kind == OP_CONV3 && !o.dt && (o.subpix || o.stride != 1)) ? 20 :
                       (o.dt ? (o.kind == OP_CONV3 ? 16 : 17) : o.kind);
        kind_ms[kd] += ms; kind_launches[kd] += 1; kind_flops[kd] += fl; kind_bytes[kd] += by;
    }
    return rc;
}

// The same measurement, one text line per launch: "index kind M N K ms GFLOP MB" (development: which shapes carry a group's
// time).  Returns the number of bytes written (without the terminator), < 0 on error.
extern "C" long long sd_unet_forward_op_times(sd_unet* u, void* stream, const float* latents, int latent_batch, int unet_batch,
                                              float timestep, float* eps_out, void* workspace, long long workspace_bytes,
                                              int cache_mode, int cache_branch_id, char* text, long long cap) {
    SD_REQUIRE(text && cap > 0, "forward_op_times: null argument");
    Plan* pl = nullptr;
    std::vector<std::pair<int, float>> per_op;
    const int rc = profiled_run(u, stream, latents, latent_batch, unet_batch, timestep, eps_out, workspace, workspace_bytes,
                                cache_mode, cache_branch_id, &pl, &per_op);
    if (rc) return rc;
    long long n = 0;
    for (auto& [i, ms] : per_op) {
        const Op& o = pl->ops[i];
        double fl, by;
        op_work(o, &fl, &by);
        // (a GroupNorm that finishes its producer's deferred split-K reduce reports the split factor in the K column)
        const int w = snprintf(text + n, (size_t)(cap - n), "%d %d %d %d %d %.5f %.3f %.3f\n", i, (int)o.kind, o.M, o.N,
                               o.kind == OP_GN ? o.slab_k : o.K, ms, fl * 1e-9, by * 1e-6);
        if (w < 0 || n + w >= cap) break;
        n += w;
    }
    return n;
}

extern "C" int sd_unet_debug_tensor(sd_unet* u, void* stream, const char* name, float* host_out, long long numel,
                                    void* workspace, int unet_batch, int cache_branch_id) {
    SD_REQUIRE(u && u->debug_taps, "debug_tensor: create the handle with SD_DEBUG_TAPS=1 in the environment");
    Plan* pl;
    int rc = get_plan(u, unet_batch, cache_branch_id, &pl, u->last_rep);
    if (rc) return rc;
    auto it = pl->taps.find(name);
    SD_REQUIRE(it != pl->taps.end(), "debug_tensor: unknown tap '%s'", name);
    const Tn& t = pl->tensors[it->second];
    SD_REQUIRE((size_t)numel * 2 <= t.bytes, "debug_tensor: '%s' holds at most %zu elements", name, t.bytes / 2);
    std::vector<unsigned short> tmp(numel);
    SD_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    SD_CHECK_HIP(hipMemcpy(tmp.data(), (char*)workspace + t.off, (size_t)numel * 2, hipMemcpyDeviceToHost));
    for (long long i = 0; i < numel; ++i) {
        unsigned v = (unsigned)tmp[i] << 16;
        memcpy(&host_out[i], &v, 4);
    }
    return 0;
}

extern "C" int sd_sched_step(void* stream, const float* eps, int cfg, float guidance, const float* x, const float* m1,
                             const float* m2, const float* m3, const float* noise, float* prev, float* y2, float* m_out,
                             const float coef[10], long long n) {
    SD_REQUIRE(coef, "sched_step: null coefficients");
    StepCoef c{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8], coef[9]};
    return sd_launch_sched_step(eps, cfg, guidance, x, m1, m2, m3, noise, prev, y2, m_out, c, (long)n, (hipStream_t)stream);
}

// ---- operator-level entry points -------------------------------------------------------------
extern "C" int sd_op_gemm(void* stream, const void* X, long long ldx, const void* X2, long long ldx2, int K1,
                          const void* W, const float* bias, const float* bias2, const void* R, long long ldr, void* C,
                          long long ldc, int M, int N, int K, int epi) {
    if (ensure_zero_page()) return -2;
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.X2 = (const bf16_t*)X2; a.ldx2 = ldx2; a.K1 = K1;
    a.W = (const bf16_t*)W; a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = ldr;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page;
    a.splitk = epi ? 1 : sd_gemm_splitk(M, N, K);
    if (a.splitk > 1) {
        a.slab = (float*)op_scratch((size_t)a.splitk * M * N * 4);
        SD_REQUIRE(a.slab, "sd_op_gemm: cannot allocate split-K scratch");
    }
    return sd_launch_gemm(a, epi, (hipStream_t)stream);
}

extern "C" int sd_op_gemm_batched(void* stream, const void* X, long long ldx, const void* W, long long w_batch_stride,
                                  int rows_per_batch, const float* bias, const void* R, long long ldr, void* C,
                                  long long ldc, int M, int N, int K, int epi, int sm_valid) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(epi == 0 || epi == 2, "sd_op_gemm_batched: epi %d (0 = std, 2 = softmax over 80-column groups)", epi);
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)W; a.bias = bias; a.R = (const bf16_t*)R; a.ldr = ldr;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page;
    a.w_batch_stride = w_batch_stride; a.rows_per_batch = rows_per_batch; a.sm_valid = sm_valid; a.splitk = 1;
    return sd_launch_gemm(a, epi, (hipStream_t)stream);
}

// the softmax-epilogue GEMM over per-sample weights with a LayerNorm folded in: X = the un-normalised rows, W = per-sample
// [N][K] operands scaled by gamma (centred or not), c1 / c2 = per-sample [N] vectors (row sums of the rounded W, beta term),
// rowstats [parts][M][2]: P = softmax_80col( rstd_m * (X W^T - mean_m c1) + c2 )
extern "C" int sd_op_gemm_batched_softmax_ln(void* stream, const void* X, long long ldx, const void* W, long long w_batch_stride,
                                             int rows_per_batch, void* C, long long ldc, int M, int N, int K, int sm_valid,
                                             const float* rowstats, int parts, const float* c1, const float* c2, float eps) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(rowstats && c1 && c2, "sd_op_gemm_batched_softmax_ln: null operand");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)W; a.bias = c2;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page;
    a.w_batch_stride = w_batch_stride; a.rows_per_batch = rows_per_batch; a.sm_valid = sm_valid; a.splitk = 1;
    a.ln_rs = rowstats; a.ln_np = parts; a.ln_c1 = c1; a.ln_eps = eps; a.ln_per_sample = 1;
    return sd_launch_gemm(a, 2, (hipStream_t)stream);
}

extern "C" int sd_op_conv3x3(void* stream, const void* X, const void* W, const float* bias, const float* bias2,
                             const void* R, void* Y, int B, int Hin, int Win, int Cin, int Cout, int stride, int upsample) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride %d", stride);
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = Cout;
    a.C = (bf16_t*)Y; a.ldc = Cout;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = stride; a.up = upsample ? 1 : 0;
    a.Hout = ((Hin << a.up) + 2 - 3) / stride + 1; a.Wout = ((Win << a.up) + 2 - 3) / stride + 1;
    a.M = B * a.Hout * a.Wout; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K; a.zero_page = g_zero_page;
    a.splitk = sd_conv3x3_splitk(a.M, a.N, Cin, Hin, Win, stride, a.up);
    if (a.splitk > 1) {
        a.slab = (float*)op_scratch((size_t)a.splitk * a.M * a.N * 4);
        SD_REQUIRE(a.slab, "sd_op_conv3x3: cannot allocate split-K scratch");
    }
    return sd_launch_conv3x3(a, (hipStream_t)stream);
}

// Timing ablations of the halo conv kernel (csrc/conv_halo.hip, template parameter DIAG; WRONG results by design, Y is
// scratch): ablate = 1 no LDS-DMA waits, 2 no LDS-DMA at all, 4 no tap barrier either, 8 no fragment reads either = the bare
// MFMA stream of the kernel's own tile -- the rate the matrix pipe sustains at the clock the chip holds under that load,
// which bench.py reports next to the nominal peak.  Stride-1 shapes the halo kernel takes, no split-K.
extern "C" int sd_op_conv3x3_ablate(void* stream, const void* X, const void* W, void* Y, int B, int Hin, int Win, int Cin, int Cout,
                                    int ablate) {
    if (ensure_zero_page()) return -2;
    const int abl = ablate & ~256;               // bit 8: the 4-wave layout (128 x 80 per wave; modes 0 and 8 only)
    SD_REQUIRE(abl == 0 || abl == 8 || (!(ablate & 256) && (abl == 1 || abl == 2 || abl == 4)), "conv3x3_ablate: mode %d", ablate);
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.C = (bf16_t*)Y; a.ldc = Cout; a.ldr = Cout;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = 1; a.up = 0; a.Hout = Hin; a.Wout = Win;
    a.M = B * Hin * Win; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K; a.zero_page = g_zero_page; a.splitk = 1; a.tune = ablate;
    SD_REQUIRE(Cin % 64 == 0 && Cout % 4 == 0 && sd_conv_halo_applicable(a), "conv3x3_ablate: not a halo-kernel shape");
    return sd_launch_conv3x3_halo(a, (hipStream_t)stream);
}

// nearest-2x upsample + 3x3 conv computed as four 2x2 convs on the low-res input (GemmArgs::subpix); W4 =
// [4 phases][Cout][Cin/64][4 taps][64] with the 3x3 taps that read the same low-res pixel summed (Packer::conv3_subpixel)
extern "C" int sd_op_conv3x3_upsample_subpixel(void* stream, const void* X, const void* W4, const float* bias, void* Y, int B,
                                               int Hin, int Win, int Cin, int Cout) {
    if (ensure_zero_page()) return -2;
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W4; a.bias = bias; a.C = (bf16_t*)Y; a.ldc = Cout;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = 1; a.up = 0; a.Hout = 2 * Hin; a.Wout = 2 * Win;
    a.M = 4 * B * Hin * Win; a.N = Cout; a.K = 4 * Cin; a.K1 = a.K; a.zero_page = g_zero_page; a.splitk = 1;
    a.subpix = 1; a.w_batch_stride = (long)Cout * 4 * Cin;
    return sd_launch_conv3x3(a, (hipStream_t)stream);
}

// ... -> GroupNorm(+SiLU) as the plan runs the pair (the up-blocks' Upsample2D feeds the next resnet's norm1): the conv's
// epilogue delivers the block statistics in the row order (sample, phase, low-res pixel).  Y = conv output [B, 2 Hin, 2 Win,
// Cout], Yn = normalised output.  Low-res pixels per sample must be a multiple of 128.
extern "C" int sd_op_conv3x3_upsample_subpixel_groupnorm(void* stream, const void* X, const void* W4, const float* bias, void* Y,
                                                         int B, int Hin, int Win, int Cin, int Cout, const float* gamma,
                                                         const float* beta, void* Yn, int groups, float eps, int silu) {
    if (ensure_zero_page()) return -2;
    const int HW = 4 * Hin * Win;
    SD_REQUIRE((Hin * Win) % 128 == 0 && !sd_groupnorm_uses_small(B, HW, Cout, 0, groups),
               "sd_op_conv3x3_upsample_subpixel_groupnorm: %dx%d -> x2, %d channels: no producer statistics at this size", Hin, Win, Cout);
    const size_t stats_bytes = (size_t)B * (HW / 64) * Cout * 2 * 4;
    char* scratch = (char*)op_scratch(stats_bytes + sd_groupnorm_scratch_bytes(B, HW, groups));
    SD_REQUIRE(scratch, "sd_op_conv3x3_upsample_subpixel_groupnorm: cannot allocate scratch");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W4; a.bias = bias; a.C = (bf16_t*)Y; a.ldc = Cout;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = 1; a.up = 0; a.Hout = 2 * Hin; a.Wout = 2 * Win;
    a.M = 4 * B * Hin * Win; a.N = Cout; a.K = 4 * Cin; a.K1 = a.K; a.zero_page = g_zero_page; a.splitk = 1;
    a.subpix = 1; a.w_batch_stride = (long)Cout * 4 * Cin;
    a.stats = (float*)scratch;
    if (int rc = sd_launch_conv3x3(a, (hipStream_t)stream)) return rc;
    GroupNormArgs g;
    g.x1 = (const bf16_t*)Y; g.C1 = Cout; g.gamma = gamma; g.beta = beta; g.y = (bf16_t*)Yn; g.B = B; g.HW = HW;
    g.groups = groups; g.eps = eps; g.silu = silu; g.nsplit = sd_groupnorm_nsplit(B, HW);
    g.partial = (float*)(scratch + stats_bytes);
    g.stats1 = (const float*)scratch;
    return sd_launch_groupnorm(g, (hipStream_t)stream);
}

extern "C" int sd_op_groupnorm(void* stream, const void* x1, int C1, const void* x2, int C2, const float* gamma,
                               const float* beta, void* y, int B, int HW, int groups, float eps, int silu) {
    GroupNormArgs a;
    a.x1 = (const bf16_t*)x1; a.C1 = C1; a.x2 = (const bf16_t*)x2; a.C2 = C2; a.gamma = gamma; a.beta = beta;
    a.y = (bf16_t*)y; a.B = B; a.HW = HW; a.groups = groups; a.eps = eps; a.silu = silu;
    a.nsplit = sd_groupnorm_nsplit(B, HW);
    a.partial = (float*)op_scratch(sd_groupnorm_scratch_bytes(B, HW, groups));
    SD_REQUIRE(a.partial, "sd_op_groupnorm: cannot allocate scratch");
    return sd_launch_groupnorm(a, (hipStream_t)stream);
}

// conv3x3 -> GroupNorm(+SiLU) as the plan runs the pair: the conv's epilogue delivers the per-64-row-block channel
// statistics, the GroupNorm skips its own statistics pass.  Y = conv output, Yn = normalised output.
extern "C" int sd_op_conv3x3_groupnorm(void* stream, const void* X, const void* W, const float* bias, const float* bias2,
                                       const void* R, void* Y, int B, int Hin, int Win, int Cin, int Cout,
                                       const float* gamma, const float* beta, void* Yn, int groups, float eps, int silu) {
    if (ensure_zero_page()) return -2;
    const int HW = Hin * Win;
    if (sd_groupnorm_uses_small(B, HW, Cout, 0, groups)) {
        // small images: the single-launch GroupNorm (no producer statistics); a split-K conv leaves its partial slabs to it
        // (GemmArgs::defer_reduce, fuse_deferred_reduce above) instead of launching splitk_reduce_kernel
        GemmArgs a;
        a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = Cout;
        a.C = (bf16_t*)Y; a.ldc = Cout; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = 1; a.up = 0; a.Hout = Hin; a.Wout = Win;
        a.M = B * HW; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K; a.zero_page = g_zero_page;
        a.splitk = sd_conv3x3_splitk(a.M, a.N, Cin, Hin, Win, 1, 0);
        const size_t slab_bytes = a.splitk > 1 ? (size_t)a.splitk * a.M * a.N * 4 : 0;
        char* scratch = (char*)op_scratch(slab_bytes + sd_groupnorm_scratch_bytes(B, HW, groups));
        SD_REQUIRE(scratch, "sd_op_conv3x3_groupnorm: cannot allocate scratch");
        const bool slab_off = getenv("SD_GN_SLAB") && atoi(getenv("SD_GN_SLAB")) == 0;     // per call: the test compares both
        if (a.splitk > 1) { a.slab = (float*)scratch; a.defer_reduce = (slab_off || !sd_groupnorm_slab_ok(B, HW, Cout, 0, groups)) ? 0 : 1; }
        if (int rc = sd_launch_conv3x3(a, (hipStream_t)stream)) return rc;
        GroupNormArgs g;
        g.x1 = (const bf16_t*)Y; g.C1 = Cout; g.gamma = gamma; g.beta = beta; g.y = (bf16_t*)Yn; g.B = B; g.HW = HW;
        g.groups = groups; g.eps = eps; g.silu = silu; g.nsplit = sd_groupnorm_nsplit(B, HW);
        g.partial = (float*)(scratch + slab_bytes);
        if (a.defer_reduce) {
            g.slab = a.slab; g.splitk = a.splitk; g.sbias = bias; g.sbias2 = bias2; g.sR = (const bf16_t*)R; g.sldr = Cout;
            g.x1w = (bf16_t*)Y;
        }
        return sd_launch_groupnorm(g, (hipStream_t)stream);
    }
    SD_REQUIRE(HW % 64 == 0, "sd_op_conv3x3_groupnorm: %dx%d pixels per sample: producer statistics come in 64-pixel blocks", Hin, Win);
    const size_t stats_bytes = (size_t)B * (HW / 64) * Cout * 2 * 4;
    char* scratch = (char*)op_scratch(stats_bytes + sd_groupnorm_scratch_bytes(B, HW, groups));
    SD_REQUIRE(scratch, "sd_op_conv3x3_groupnorm: cannot allocate scratch");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = Cout;
    a.C = (bf16_t*)Y; a.ldc = Cout; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = 1; a.up = 0; a.Hout = Hin; a.Wout = Win;
    a.M = B * HW; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K; a.zero_page = g_zero_page; a.splitk = 1;
    a.stats = (float*)scratch;
    if (int rc = sd_launch_conv3x3(a, (hipStream_t)stream)) return rc;
    GroupNormArgs g;
    g.x1 = (const bf16_t*)Y; g.C1 = Cout; g.gamma = gamma; g.beta = beta; g.y = (bf16_t*)Yn; g.B = B; g.HW = HW;
    g.groups = groups; g.eps = eps; g.silu = silu; g.nsplit = sd_groupnorm_nsplit(B, HW);
    g.partial = (float*)(scratch + stats_bytes);
    g.stats1 = (const float*)scratch;
    return sd_launch_groupnorm(g, (hipStream_t)stream);
}

extern "C" int sd_op_conv3x3_splitk(int M, int Cout, int Cin, int Hin, int Win, int stride, int upsample) {
    return sd_conv3x3_splitk(M, Cout, Cin, Hin, Win, stride, upsample ? 1 : 0);
}

extern "C" int sd_op_layernorm(void* stream, const void* x, const float* gamma, const float* beta, void* y, int rows,
                               int C, float eps) {
    return sd_launch_layernorm((const bf16_t*)x, gamma, beta, (bf16_t*)y, rows, C, eps, (hipStream_t)stream);
}

extern "C" int sd_op_attention(void* stream, const void* Q, long long ldq, const void* K, long long ldk, const void* V,
                               long long ldv, void* O, long long ldo, int B, int heads, int Nq, int Nk, int D, float scale) {
    AttnArgs a;
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
    a.O = (bf16_t*)O; a.ldo = ldo; a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale;
    if (ensure_zero_page()) return -2;
    a.consts = g_zero_page;
    return sd_launch_attention(a, (hipStream_t)stream);
}

// the CLIP text tower's causal self-attention on the fused projection output qkv [B * L][3 H] (q | k | v) -> out [B * L][H]
extern "C" int sd_op_clip_attention(void* stream, const void* qkv, void* out, int B, int L, int H, int heads) {
    return sd_launch_clip_attention((const bf16_t*)qkv, (bf16_t*)out, B, L, H, heads, (hipStream_t)stream);
}

// q|k|v projection the way the plan runs it at the 64x64 level: Q token-major [M][C], K and V head-major
// KV[2][M / tokens][C / 40][tokens][40] (GemmArgs::KV); W = [3 C][K] rows (q | k | v)
extern "C" int sd_op_gemm_qkv_headmajor(void* stream, const void* X, long long ldx, const void* W, void* Q, void* KV, int M,
                                        int C, int tokens, int K) {
    if (ensure_zero_page()) return -2;
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)W; a.C = (bf16_t*)Q; a.ldc = C; a.M = M; a.N = 3 * C;
    a.K = K; a.zero_page = g_zero_page; a.splitk = 1; a.KV = (bf16_t*)KV; a.hm_C = C; a.hm_tok = tokens;
    return sd_launch_gemm(a, 0, (hipStream_t)stream);
}

// self-attention with HEAD-MAJOR K / V ([B][heads][Nk][D] contiguous, as the plan's q|k|v projection stores them at the
// 64x64 level): d = 40, Nk a multiple of 64
extern "C" int sd_op_attention_headmajor(void* stream, const void* Q, long long ldq, const void* K, const void* V, void* O,
                                         long long ldo, int B, int heads, int Nq, int Nk, int D, float scale) {
    AttnArgs a;
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.kv_head_major = 1;
    a.O = (bf16_t*)O; a.ldo = ldo; a.B = B; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale;
    if (ensure_zero_page()) return -2;
    a.consts = g_zero_page;
    return sd_launch_attention(a, (hipStream_t)stream);
}

extern "C" int sd_op_conv_in(void* stream, const float* x, int Bsrc, const float* Wt, const float* bias, void* y, int B,
                             int H, int W, int Cin, int Cout) {
    return sd_launch_conv_in(x, Bsrc, Wt, bias, (bf16_t*)y, B, H, W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int sd_op_conv_out(void* stream, const void* x, const void* Wp, const float* bias, float* y, int B, int H,
                              int W, int Cin, int Cout) {
    return sd_launch_conv_out((const bf16_t*)x, (const bf16_t*)Wp, bias, y, B, H, W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int sd_op_time_embedding(void* stream, float t, const void* W1, const float* b1, const void* W2,
                                    const float* b2, float* scratch, float* temb, int dim_in, int dim) {
    int rc;
    if ((rc = sd_launch_timestep_sinusoid(t, scratch, dim_in, (hipStream_t)stream))) return rc;
    if ((rc = sd_launch_gemv(scratch, (const bf16_t*)W1, b1, scratch + dim_in, dim, dim_in, 0, (hipStream_t)stream))) return rc;
    return sd_launch_gemv(scratch + dim_in, (const bf16_t*)W2, b2, temb, dim, dim, 1, (hipStream_t)stream);
}

// ---- fp8-e4m3 operand path, operator level (parity tests of SD_DTYPE_FP8_E4M3) ---------------------------------
extern "C" int sd_op_gemm_fp8(void* stream, const void* X, long long ldx, const void* W, const float* wscale, float xscale,
                              const float* bias, const void* R, long long ldr, void* C, long long ldc, int M, int N, int K,
                              int epi, int out_fp8, float oscale) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(epi == 0 || epi == 1, "sd_op_gemm_fp8: epi %d (0 = std, 1 = GEGLU)", epi);
    SD_REQUIRE(xscale > 0.f && (!out_fp8 || (epi == 1 && oscale > 0.f)), "sd_op_gemm_fp8: bad scales / fp8 output needs the GEGLU epilogue");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)W; a.bias = bias; a.R = (const bf16_t*)R; a.ldr = ldr;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page;
    a.dt = 1; a.wscale = wscale; a.xscale_inv = 1.0f / xscale; a.out_fp8 = out_fp8; a.oscale = oscale;
    a.splitk = epi ? 1 : sd_gemm_splitk(M, N, K / 2, 128);
    if (a.splitk > 1) {
        a.slab = (float*)op_scratch((size_t)a.splitk * M * N * 4);
        SD_REQUIRE(a.slab, "sd_op_gemm_fp8: cannot allocate split-K scratch");
    }
    return sd_launch_gemm(a, epi, (hipStream_t)stream);
}

extern "C" int sd_op_conv3x3_fp8(void* stream, const void* X, const void* W, const float* wscale, float xscale,
                                 const float* bias, const float* bias2, const void* R, void* Y, int B, int Hin, int Win,
                                 int Cin, int Cout, int stride, int upsample) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(stride == 1 || stride == 2, "conv3x3 fp8: stride %d", stride);
    SD_REQUIRE(xscale > 0.f, "conv3x3 fp8: activation scale");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = Cout;
    a.C = (bf16_t*)Y; a.ldc = Cout;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.stride = stride; a.up = upsample ? 1 : 0;
    a.Hout = ((Hin << a.up) + 2 - 3) / stride + 1; a.Wout = ((Win << a.up) + 2 - 3) / stride + 1;
    a.M = B * a.Hout * a.Wout; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K; a.zero_page = g_zero_page;
    a.dt = 1; a.wscale = wscale; a.xscale_inv = 1.0f / xscale;
    a.splitk = sd_conv3x3_splitk(a.M, a.N, Cin, Hin, Win, stride, a.up, 1);
    if (a.splitk > 1) {
        a.slab = (float*)op_scratch((size_t)a.splitk * a.M * a.N * 4);
        SD_REQUIRE(a.slab, "sd_op_conv3x3_fp8: cannot allocate split-K scratch");
    }
    return sd_launch_conv3x3(a, (hipStream_t)stream);
}

extern "C" int sd_op_groupnorm_fp8(void* stream, const void* x1, int C1, const void* x2, int C2, const float* gamma,
                                   const float* beta, void* y, int B, int HW, int groups, float eps, int silu, int Cpad,
                                   float oscale) {
    GroupNormArgs a;
    a.x1 = (const bf16_t*)x1; a.C1 = C1; a.x2 = (const bf16_t*)x2; a.C2 = C2; a.gamma = gamma; a.beta = beta;
    a.y = (bf16_t*)y; a.B = B; a.HW = HW; a.groups = groups; a.eps = eps; a.silu = silu;
    a.out_fp8 = 1; a.Cpad = Cpad; a.oscale = oscale;
    a.nsplit = sd_groupnorm_nsplit(B, HW);
    a.partial = (float*)op_scratch(sd_groupnorm_scratch_bytes(B, HW, groups));
    SD_REQUIRE(a.partial, "sd_op_groupnorm_fp8: cannot allocate scratch");
    return sd_launch_groupnorm(a, (hipStream_t)stream);
}

extern "C" int sd_op_layernorm_fp8(void* stream, const void* x, const float* gamma, const float* beta, void* y, int rows,
                                   int C, int Cpad, float eps, float oscale) {
    return sd_launch_layernorm_fp8((const bf16_t*)x, gamma, beta, y, rows, C, Cpad, eps, oscale, (hipStream_t)stream);
}

extern "C" int sd_op_quantize_fp8(void* stream, const void* x_bf16, void* y_fp8, long long rows, int C, int Cpad, float scale) {
    return sd_launch_quantize_fp8((const bf16_t*)x_bf16, y_fp8, (long)rows, C, Cpad, scale, (hipStream_t)stream);
}

// ---- fused prompt cross-attention, operator level: Y = R + sum_h softmax_L(X A_h) B_h + b_o (xattn.hip) ----
// ---- LayerNorm folded into the consuming GEMM (the plan's norm1 -> q|k|v and norm3 -> GEGLU pairs) ----
// number of per-row partials a producer writes: kind 0 = GEMM with N output columns, kind 1 = fused cross-attention (M, C)
extern "C" int sd_op_ln_partials(int kind, int M, int N) { return kind == 0 ? (N + 159) / 160 * 2 : 2 * sd_xattn_slices(M, N); }
// producer: C = X W^T + bias + R, and rowstats[parts][M][2] = per-row (sum, sum of squares) partials of the stored C
extern "C" int sd_op_gemm_rowstats(void* stream, const void* X, long long ldx, const void* W, const float* bias, const void* R,
                                   long long ldr, void* C, long long ldc, int M, int N, int K, float* rowstats) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(rowstats, "sd_op_gemm_rowstats: null partials buffer");
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)W; a.bias = bias; a.R = (const bf16_t*)R; a.ldr = ldr;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page; a.splitk = 1; a.rowstats = rowstats;
    return sd_launch_gemm(a, 0, (hipStream_t)stream);
}
// The std-epilogue GEMM with EVERY side input / output the UNet plan combines on it (tests compare the lean kernel of
// gemm_lean.hip against the general one bit for bit through this entry, SD_GEMM_LEAN=0|1): second K segment, bias, bias2,
// residual, LayerNorm row partials and GroupNorm block statistics of the stored output (producer side), the LayerNorm fold
// (consumer side: ln_rs / ln_parts / ln_c1, bias = c2) and head-major K / V (hm_tokens > 0: N = 3 C, KV[2][M / tokens][C / 40][tokens][40]).
extern "C" int sd_op_gemm_plan(void* stream, const void* X, long long ldx, const void* X2, long long ldx2, int K1, const void* W,
                               const float* bias, const float* bias2, const void* R, long long ldr, void* C, long long ldc,
                               int M, int N, int K, float* rowstats, float* stats, const float* ln_rs, int ln_parts,
                               const float* ln_c1, float ln_eps, void* KV, int hm_tokens) {
    if (ensure_zero_page()) return -2;
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.X2 = (const bf16_t*)X2; a.ldx2 = ldx2; a.K1 = K1; a.W = (const bf16_t*)W;
    a.bias = bias; a.bias2 = bias2; a.R = (const bf16_t*)R; a.ldr = ldr; a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.zero_page = g_zero_page; a.splitk = 1; a.rowstats = rowstats; a.stats = stats;
    a.ln_rs = ln_rs; a.ln_np = ln_parts; a.ln_c1 = ln_c1; a.ln_eps = ln_eps;
    if (hm_tokens > 0) {
        SD_REQUIRE(N % 3 == 0 && KV, "sd_op_gemm_plan: head-major K / V needs N = 3 C and a KV buffer");
        a.KV = (bf16_t*)KV; a.hm_C = N / 3; a.hm_tok = hm_tokens;
    }
    return sd_launch_gemm(a, 0, (hipStream_t)stream);
}
// consumer: C = epi(LayerNorm(X) W^T + b) computed from the UN-normalised X: Wg = bf16(W * gamma), c1[n] = sum_k Wg[n][k],
// c2[n] = sum_k W[n][k] beta[k] + b[n]; mean / rstd of a row from its `parts` partials.  epi 0 = plain, 1 = GEGLU.
extern "C" int sd_op_gemm_ln(void* stream, const void* X, long long ldx, const void* Wg, const float* c1, const float* c2,
                             const float* rowstats, int parts, float eps, void* C, long long ldc, int M, int N, int K, int epi) {
    if (ensure_zero_page()) return -2;
    SD_REQUIRE(epi == 0 || epi == 1, "sd_op_gemm_ln: epi %d (0 = plain, 1 = GEGLU)", epi);
    GemmArgs a;
    a.X = (const bf16_t*)X; a.ldx = ldx; a.K1 = K; a.W = (const bf16_t*)Wg; a.bias = c2;
    a.C = (bf16_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero_page = g_zero_page; a.splitk = 1;
    a.ln_rs = rowstats; a.ln_np = parts; a.ln_c1 = c1; a.ln_eps = eps;
    return sd_launch_gemm(a, epi, (hipStream_t)stream);
}

extern "C" int sd_op_xattn_fused_rowstats(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                                          const float* bias, int M, int C, int rows_per_sample, int L, float* rowstats) {
    SD_REQUIRE(sd_xattn_fused_applicable(rows_per_sample, C, 8, L), "sd_op_xattn_fused_rowstats: shape not supported");
    XattnArgs a;
    a.X = (const bf16_t*)X; a.R = (const bf16_t*)R; a.Y = (bf16_t*)Y; a.At = (const bf16_t*)At; a.Bw = (const bf16_t*)Bw;
    a.bias = bias; a.M = M; a.C = C; a.rows_per_sample = rows_per_sample; a.L = L; a.rowstats = rowstats;
    return sd_launch_xattn_fused(a, (hipStream_t)stream);
}

extern "C" int sd_op_xattn_fused(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                                 const float* bias, int M, int C, int rows_per_sample, int L) {
    SD_REQUIRE(sd_xattn_fused_applicable(rows_per_sample, C, 8, L) || getenv("SD_XATTN_FUSED"),
               "sd_op_xattn_fused: shape not supported (8 heads x 80 key slots, 64 < L <= 80, C %% 32 == 0, tokens per sample %% 128 == 0)");
    XattnArgs a;
    a.X = (const bf16_t*)X; a.R = (const bf16_t*)R; a.Y = (bf16_t*)Y; a.At = (const bf16_t*)At; a.Bw = (const bf16_t*)Bw;
    a.bias = bias; a.M = M; a.C = C; a.rows_per_sample = rows_per_sample; a.L = L;
    return sd_launch_xattn_fused(a, (hipStream_t)stream);
}

// norm2 folded in (XattnArgs::ln_rs): X = the un-normalised rows, At = the CENTRED gamma-scaled operand, c2 [samples][640] fp32
extern "C" int sd_op_xattn_fused_ln(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                                    const float* bias, int M, int C, int rows_per_sample, int L, const float* ln_rowstats,
                                    int ln_parts, long long ln_rows, const float* c2, float eps, float* rowstats) {
    SD_REQUIRE(sd_xattn_fused_applicable(rows_per_sample, C, 8, L), "sd_op_xattn_fused_ln: shape not supported");
    SD_REQUIRE(ln_rowstats && c2, "sd_op_xattn_fused_ln: null operand");
    XattnArgs a;
    a.X = (const bf16_t*)X; a.R = (const bf16_t*)R; a.Y = (bf16_t*)Y; a.At = (const bf16_t*)At; a.Bw = (const bf16_t*)Bw;
    a.bias = bias; a.M = M; a.C = C; a.rows_per_sample = rows_per_sample; a.L = L; a.rowstats = rowstats;
    a.ln_rs = ln_rowstats; a.ln_np = ln_parts; a.ln_rows = ln_rows; a.ln_c2 = c2; a.ln_eps = eps;
    return sd_launch_xattn_fused(a, (hipStream_t)stream);
}

// diagnostic twin of sd_op_xattn_fused (tools/xattn_stamps.py): the kernel additionally stores 8 s_memtime stamps per
// workgroup to `stamps` (device memory, 8 * (M / 128) * slices 64-bit words, owned by the caller)
extern "C" int sd_op_xattn_fused_stamps(void* stream, const void* X, const void* R, void* Y, const void* At, const void* Bw,
                                        const float* bias, int M, int C, int rows_per_sample, int L, unsigned long long* stamps) {
    SD_REQUIRE(stamps, "sd_op_xattn_fused_stamps: null stamp buffer");
    SD_REQUIRE(sd_xattn_fused_applicable(rows_per_sample, C, 8, L), "sd_op_xattn_fused_stamps: shape not supported");
    XattnArgs a;
    a.X = (const bf16_t*)X; a.R = (const bf16_t*)R; a.Y = (bf16_t*)Y; a.At = (const bf16_t*)At; a.Bw = (const bf16_t*)Bw;
    a.bias = bias; a.M = M; a.C = C; a.rows_per_sample = rows_per_sample; a.L = L; a.stamps = stamps;
    return sd_launch_xattn_fused(a, (hipStream_t)stream);
}
