// Probe: do MFMA and VALU work overlap on one SIMD (gfx950)?  (development aid)
//  mode 0: MFMA only      (14 x v_mfma_f32_32x32x16_bf16 per iteration, 2 independent accumulators)
//  mode 1: VALU only      (32 v_exp + 128 v_fma per iteration)
//  mode 2: both, separated (all MFMAs, then all VALU)      -- what the attention loop looks like per wave
//  mode 3: both, interleaved (1 MFMA : ~11 VALU)
// blocks of 256 threads; WPS waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f00 + lane * 3 + i); }
  f32x16 acc0 = {}, acc1 = {};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
      }
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float x = v[i];
          x = __builtin_fmaf(x, 0.999f, 0.001f); x = __builtin_fmaf(x, 1.001f, -0.001f);
          x = __builtin_fmaf(x, 0.998f, 0.002f); x = __builtin_fmaf(x, 1.002f, -0.002f);
          v[i] = __builtin_amdgcn_exp2f(x * 0.01f);
        }
    }
    if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 14; ++m) {
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {      // 32 exp + 128 fma over 14 gaps: ~2.3 groups per gap
          const int i = (m * 2 + q) % 16;
          float x = v[i];
          x = __builtin_fmaf(x, 0.999f, 0.001f); x = __builtin_fmaf(x, 1.001f, -0.001f);
          x = __builtin_fmaf(x, 0.998f, 0.002f); x = __builtin_fmaf(x, 1.002f, -0.002f);
          v[i] = __builtin_amdgcn_exp2f(x * 0.01f);
        }
        if (m < 4) {
          const int i = (28 + m) % 16;
          float x = v[i];
          x = __builtin_fmaf(x, 0.999f, 0.001f); x = __builtin_fmaf(x, 1.001f, -0.001f);
          x = __builtin_fmaf(x, 0.998f, 0.002f); x = __builtin_fmaf(x, 1.002f, -0.002f);
          v[i] = __builtin_amdgcn_exp2f(x * 0.01f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> float run(int blocks, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; ++wps) {           // waves per SIMD = blocks per CU (4 waves per block, 1 per SIMD)
    const int blocks = 256 * wps;
    float t0 = run<0>(blocks, iters, d), t1 = run<1>(blocks, iters, d), t2 = run<2>(blocks, iters, d), t3 = run<3>(blocks, iters, d);
    // cycles per wave-iteration of SIMD time at 2.1 GHz nominal: ms*1e-3*2.1e9 / (iters * wps)
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.1e9 / ((double)iters * wps); };
    printf("waves/SIMD %d: mfma %.0f  valu %.0f  separated %.0f  interleaved %.0f   (cycles per wave-iteration @2.1GHz; mfma+valu=%.0f)\n",
           wps, cyc(t0), cyc(t1), cyc(t2), cyc(t3), cyc(t0) + cyc(t1));
  }
  return 0;
}
