// Sustained rate of the two bf16 MFMA shapes on random operands (the clock a kernel gets under a matrix load depends on the
// energy per result: MI355X_MICROARCH.md "DVFS give-back").  Every wave keeps NACC independent accumulator tiles in flight;
// operands are random bf16 held in registers.  Development probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/_bin/mfma_rate && tools/probes/_bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void rate_kernel(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(tid * 8 + i) & 65535]; b[i] = ops[(tid * 8 + 4 + i) & 65535]; }
    float sum = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][15];
    }
    if (sum == 12345.678f) out[tid] = sum;      // keep the chain alive
}

template <int SHAPE, int NACC>
static void run(const char* tag, int waves_per_cu, const bf16x8* ops, float* out) {
    const int iters = 20000;
    const int blocks = 256, threads = waves_per_cu * 64;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, ops, out, iters);
    hipEventRecord(s);
    const int reps = 12;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, ops, out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double flop = 2.0 * (SHAPE == 16 ? 16.0 * 16 * 32 : 32.0 * 32 * 16) * NACC * (double)iters * blocks * waves_per_cu * reps;
    printf("%-28s %2d waves/CU, %2d accumulators: %8.1f TFLOP/s  (%.2f ms per launch)\n", tag, waves_per_cu, NACC, flop / (ms * 1e-3) / 1e12, ms / reps);
}

int main() {
    std::vector<unsigned short> h(65536 * 8);
    srand(7);
    for (auto& v : h) {      // random bf16 in [-2, 2): random mantissas and signs (the energy depends on the data)
        const float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
        unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
    }
    bf16x8* ops; float* out;
    hipMalloc((void**)&ops, h.size() * 2); hipMalloc((void**)&out, 1 << 22);
    hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    {   // sustained: ~2.5 s of back-to-back launches of the conv kernel's shape, rate per ~100 ms window (DVFS settles in 0.1 - 1 s)
        const int iters = 20000, reps = 40;
        hipEvent_t ev[26];
        for (auto& e : ev) hipEventCreate(&e);
        hipEventRecord(ev[0]);
        for (int w = 0; w < 25; ++w) {
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rate_kernel<16, 16>), dim3(256), dim3(512), 0, 0, ops, out, iters);
            hipEventRecord(ev[w + 1]);
        }
        hipEventSynchronize(ev[25]);
        printf("sustained v_mfma_f32_16x16x32_bf16, 8 waves/CU, 16 accumulators, TFLOP/s per window:");
        for (int w = 0; w < 25; ++w) {
            float ms; hipEventElapsedTime(&ms, ev[w], ev[w + 1]);
            printf(" %.0f", 2.0 * 16 * 16 * 32 * 16 * (double)iters * 256 * 8 * reps / (ms * 1e-3) / 1e12);
        }
        printf("\n");
    }
    for (int round = 0; round < 1; ++round) {
        run<16, 8>("v_mfma_f32_16x16x32_bf16", 4, ops, out);
        run<16, 8>("v_mfma_f32_16x16x32_bf16", 8, ops, out);
        run<32, 4>("v_mfma_f32_32x32x16_bf16", 4, ops, out);
        run<32, 4>("v_mfma_f32_32x32x16_bf16", 8, ops, out);
        run<16, 16>("v_mfma_f32_16x16x32_bf16", 8, ops, out);
        run<32, 8>("v_mfma_f32_32x32x16_bf16", 8, ops, out);
    }
    return 0;
}
