// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libsdhip.
// Wave = 64 lanes everywhere; bf16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits in memory

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

typedef __attribute__((ext_vector_type(8))) int i32x8;       // one fp8 MFMA A/B fragment: 32 K bytes (8 VGPRs)
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
// two 16-byte halves -> one 32-byte fragment (a register-sequence, no copies when the halves are loaded in place)
__device__ __forceinline__ i32x8 cat_u32x4(u32x4 lo, u32x4 hi) {
    const u32x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(i32x8, r);
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two f32 -> packed bf16x2 (lo in bits 0..15), round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}

// four f32 -> four OCP e4m3 bytes (byte 0 = a), saturating at +-448 (the conversion alone gives NaN above the
// finite range)
__device__ __forceinline__ unsigned pack4fp8(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    const int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; the IEEE division expands to ~10 instructions per element and made
// the GroupNorm + SiLU pass VALU-bound: ~210 instructions per 16-byte vector against ~17 us of HBM time per 64x64 instance)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-erf GELU (diffusers GEGLU uses F.gelu default).  erf via Abramowitz-Stegun 7.1.25
// (|abs err| <= 2.5e-5 -> |gelu err| <= 1.3e-5 |x|, 300x below the bf16 output rounding):
// 2 transcendentals + 8 VALU ops; the GEGLU epilogue evaluates 8192 of these per 128x128 tile.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.47047f, z, 1.0f));
    const float poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float half_erfc = 0.5f * poly * e;                 // 0.5 * erfc(|x|/sqrt2)
    const float phi = x < 0.f ? half_erfc : 1.0f - half_erfc; // Phi(x)
    return x * phi;
}

// value * GELU(gate) for two lanes of work at once, on packed fp32 instructions (v_pk_mul/fma_f32: two results per
// issue) and WITHOUT transcendentals (v_rcp / v_exp issue at a quarter of the fp32 rate; at K = 320 the GEGLU epilogue is as
// long as the tile's MFMA work):  Phi(x) = 0.5 + xc * P((xc / c)^2),  xc = clamp(x, -c, c),  c = 4.35,  P = degree-9
// least-squares fit at Chebyshev nodes of (Phi(x) - 0.5) / x on (0, c] (tools/fit_gelu.py).
// |Phi_hat - Phi| <= 1.25e-5 for all x incl. the saturated tails (1 - Phi(4.35) = 6.8e-6), i.e. |gelu err| <= 1.25e-5 |x|,
// the same bound as gelu_erf_f's A&S 7.1.25 (1.3e-5 |x|): 300x below the bf16 output rounding.
__device__ __forceinline__ f32x2_t geglu_pair(f32x2_t value, f32x2_t gate) {
    constexpr float C = 4.35f;
    const f32x2_t xc = {__builtin_amdgcn_fmed3f(gate[0], -C, C), __builtin_amdgcn_fmed3f(gate[1], -C, C)};
    const f32x2_t t = xc * (1.0f / C);
    const f32x2_t u = t * t;
    auto k = [](float c) { return f32x2_t{c, c}; };
    f32x2_t p = __builtin_elementwise_fma(u, k(-8.173780021e-01f), k(4.634953387e+00f));
    p = __builtin_elementwise_fma(p, u, k(-1.175370409e+01f));
    p = __builtin_elementwise_fma(p, u, k(1.781238736e+01f));
    p = __builtin_elementwise_fma(p, u, k(-1.828480159e+01f));
    p = __builtin_elementwise_fma(p, u, k(1.371556223e+01f));
    p = __builtin_elementwise_fma(p, u, k(-7.893326951e+00f));
    p = __builtin_elementwise_fma(p, u, k(3.560156552e+00f));
    p = __builtin_elementwise_fma(p, u, k(-1.257849752e+00f));
    p = __builtin_elementwise_fma(p, u, k(3.989407778e-01f));
    const f32x2_t phi = __builtin_elementwise_fma(xc, p, k(0.5f));
    return value * gate * phi;
}

// ---- GroupNorm statistics from the PRODUCER's epilogue -------------------------------------------------------------
// A wave of the GEMM / conv kernels owns 64 consecutive output rows x TN 16-column tiles (accumulator layout: lane
// (lrow = lane & 15, lq = lane >> 4) holds columns 16 a + 4 lq + (0..3) of row 16 b + lrow).  This writes, per 64-row block
// and channel, the sum and the sum of squares of the bf16-ROUNDED outputs (what GroupNorm will read) to
// stats[(block64 * C + channel) * 2 + {0, 1}]; the consuming GroupNorm reduces blocks and channels per group in a fixed
// order (deterministic), so the statistics pass over the tensor disappears.  Rows >= M are excluded, columns >= N skipped.
template <int SHR>
__device__ __forceinline__ float row_shr_add(float v) {      // v[i] += v[i - SHR] inside each row of 16 lanes (0 shifted in)
    const int sh = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + SHR, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, sh);
}
template <int TN, int TM>
__device__ __forceinline__ void tile_channel_stats(const f32x4 (&acc)[TN][TM], float* __restrict__ stats, long block64, int C,
                                                   int n_base, int N, int m_base, int M, int lane, int b0 = 0, int nb = TM) {
    const int lrow = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        float sv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            if (b < b0 || b >= b0 + nb) continue;            // (pixel tiles b0 .. b0 + nb - 1: the rows of this 64-row block)
            const bool live = m_base + b * 16 + lrow < M;
            const unsigned p0 = pack2bf(acc[a][b][0], acc[a][b][1]), p1 = pack2bf(acc[a][b][2], acc[a][b][3]);
            const float v[4] = {bflo(p0), bfhi(p0), bflo(p1), bfhi(p1)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = live ? v[j] : 0.f;
                sv[j] += x;
                qv[j] = __builtin_fmaf(x, x, qv[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sv[j] = row_shr_add<8>(row_shr_add<4>(row_shr_add<2>(row_shr_add<1>(sv[j]))));
            qv[j] = row_shr_add<8>(row_shr_add<4>(row_shr_add<2>(row_shr_add<1>(qv[j]))));
        }
        const int n = n_base + a * 16 + lq * 4;
        if (lrow == 15 && n < N) {
            float* o = stats + (block64 * C + n) * 2;
            *(f32x4*)o = f32x4{sv[0], qv[0], sv[1], qv[1]};
            *(f32x4*)(o + 4) = f32x4{sv[2], qv[2], sv[3], qv[3]};
        }
    }
}

// LayerNorm statistics of an output tile for the GEMM that consumes the rows un-normalised (GemmArgs::ln_rs): per row
// the sum and sum of squares of this wave's bf16-rounded outputs (TN 16-column tiles) -> part[m][2]; rows_of_wave = 16 TM.
template <int TN, int TM>
__device__ __forceinline__ void tile_row_stats(const f32x4 (&acc)[TN][TM], float* __restrict__ part, int n_base, int N,
                                               int m_base, int M, int lane) {
    const int lrow = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const bool live = n_base + a * 16 + lq * 4 < N;
            const unsigned p0 = pack2bf(acc[a][b][0], acc[a][b][1]), p1 = pack2bf(acc[a][b][2], acc[a][b][3]);
            const float v[4] = {bflo(p0), bfhi(p0), bflo(p1), bfhi(p1)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = live ? v[j] : 0.f;
                s += x;
                q = __builtin_fmaf(x, x, q);
            }
        }
        s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        const int m = m_base + b * 16 + lrow;
        if (lq == 0 && m < M) *(f32x2_t*)(part + (long)m * 2) = f32x2_t{s, q};
    }
}

// LayerNorm-fold arithmetic of the consuming GEMMs (GemmArgs::ln_rs), with EXPLICIT fused multiply-adds: two kernels (or
// two instantiations of one) must not differ by the compiler's choice of contraction -- a sample's result would then
// depend on which kernel the batch size selected (the lean and the general GEMM are bit-identical by construction).
__device__ __forceinline__ f32x2_t ln_mean_rstd(float s, float q, float inv_k, float eps) {
    const float mean = s * inv_k;
    const float var = __builtin_fmaf(q, inv_k, -(mean * mean));
    return f32x2_t{mean, __builtin_amdgcn_rsqf(fmaxf(var, 0.f) + eps)};
}
// SCALAR fused multiply-adds on purpose (and gemm_conv.hip / gemm_lean.hip are built with -fno-slp-vectorize so that hipcc
// does not re-pack them).  Written on f32x4 with __builtin_elementwise_fma the second multiply-add became
//     v_pk_fma_f32 d[0:1], t[0:1], mr[0:1], c2[0:1] op_sel:[0,1,0]       (both halves x rstd = the HI dword of the (mean, rstd) pair)
// and MI355X executes that form -- a packed fp32 instruction whose op_sel feeds the LO result lane from the HI dword of a
// source -- wrongly about once per 10^7 wave-instructions while the other wave of the SIMD is issuing MFMAs: the LO half
// of lanes 48-63 loses its product (d = c2 exactly).  Round 4 saw it as 16 rows x 1 column of the q|k|v projection equal to
// the bias, ~1 launch in 35; round 5 decoded > 100 events (always the lo half, always lanes 48-63, always this instruction,
// any tile / wave, also with nothing of the wave's own memory traffic in flight; 0 of 2800 launches with scalar FMAs) and
// measured that it is NOT an MFMA -> VALU distance (tools/probes/mfma_hazard.hip: hipcc's padding equals the hardware's
// need).  tools/probes/pk_fma_coexec.hip is the stand-alone form; asm_lint.py (PK_OPSEL) fails the build on the
// instruction form anywhere in the library; tools/lnfold_diag.py + SD_DIAG_LN_PACKED rebuild the failing variant.
__device__ __forceinline__ f32x4 ln_fold(f32x4 acc, f32x4 c1, float mean, float rstd, f32x4 c2) {   // rstd (acc - mean c1) + c2
#ifdef SD_DIAG_LN_PACKED     // (tools/build_variant.py: the round-4 form that failed; never in the product build)
    const f32x4 nm = {-mean, -mean, -mean, -mean}, rs = {rstd, rstd, rstd, rstd};
    return __builtin_elementwise_fma(__builtin_elementwise_fma(c1, nm, acc), rs, c2);
#else
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_fmaf(__builtin_fmaf(c1[j], -mean, acc[j]), rstd, c2[j]);
    return r;
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// host-side error plumbing (defined in unet.hip)
extern "C" const char* sd_last_error(void);
void sd_set_error(const char* fmt, ...);
#define SD_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            sd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                  \
        }                                                                               \
    } while (0)
#define SD_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            sd_set_error(__VA_ARGS__); \
            return -1;                 \
        }                              \
    } while (0)
