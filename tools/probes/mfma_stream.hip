// What the matrix pipe sustains on random bf16 operands: a bare stream of v_mfma_f32_16x16x32_bf16 against one of
// v_mfma_f32_32x32x16_bf16 (the same flops per cycle nominally), 2 waves per SIMD on every CU, ~100 ms each.
// Development probe: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_stream.hip -o /tmp/mfma_stream && /tmp/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512, 2) void stream(const bf16x8* ops, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[5];
    for (int i = 0; i < 4; ++i) a[i] = ops[(i * 64 + lane) % 4096];
    for (int i = 0; i < 5; ++i) b[i] = ops[((i + 4) * 64 + lane + 7 * threadIdx.x) % 4096];
    if (KIND == 0) {
        f32x4 acc[5][4];
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[5];
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)          // 10 MFMAs of 32x32x16 = the flops of 20 of 16x16x32
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i], a[r], acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& x : h) {   // bf16 normal-ish values: random sign, exponent around 1.0, random mantissa
        x = (unsigned short)(((rand() & 1) << 15) | ((124 + rand() % 6) << 7) | (rand() & 127));
    }
    bf16x8* d; float* out;
    hipMalloc(&d, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = 40000;
            // warm
            if (kind == 0) hipLaunchKernelGGL(stream<0>, dim3(256), dim3(512), 0, 0, d, out, 2000);
            else hipLaunchKernelGGL(stream<1>, dim3(256), dim3(512), 0, 0, d, out, 2000);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(stream<0>, dim3(256), dim3(512), 0, 0, d, out, iters);
            else hipLaunchKernelGGL(stream<1>, dim3(256), dim3(512), 0, 0, d, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 256.0 * 8 * iters * 20 * 16384.0;     // per wave and iteration: 20 x 16x16x32 (or 10 x 32x32x16)
            printf("%s: %.1f ms  %.0f TFLOP/s  (equivalent clock %.2f GHz of the 2.5 PFLOP/s @ 2.4 GHz peak)\n",
                   kind ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0 * 2.4);
        }
    return 0;
}
