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

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
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
