// Small gfx950 kernels around the UNet body: timestep embedding (sinusoid + M=1 GEMV chain),
// conv_in (NCHW fp32 latents -> NHWC bf16, CFG batch duplication fused), conv_out (NHWC bf16 ->
// NCHW fp32 noise prediction) and the fused CFG-combine + scheduler.step update.
#include "common.h"
#include "kernels.h"

namespace {

// ---- timestep sinusoid: flip_sin_to_cos=True, freq_shift=0 -> [cos(t f_k) | sin(t f_k)] ----
__global__ void sinusoid_kernel(float t, float* __restrict__ out, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= dim) return;
    const int k = i < half ? i : i - half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float e = t * f;
    out[i] = i < half ? cosf(e) : sinf(e);
}

// ---- GEMV: one wave per output row --------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, const bf16_t* __restrict__ W,
                                                   const float* __restrict__ b, float* __restrict__ y, int N, int K,
                                                   int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* wr = W + (long)n * K;
    float acc = 0.f;
    for (int ch = lane; ch < K / 8; ch += 64) {
        const u32x4 w = *(const u32x4*)(wr + ch * 8);
        const f32x4 x0 = *(const f32x4*)(x + ch * 8), x1 = *(const f32x4*)(x + ch * 8 + 4);
        float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (silu_in) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = silu_f(xv[j]);
        }
        acc += bflo(w[0]) * xv[0] + bfhi(w[0]) * xv[1] + bflo(w[1]) * xv[2] + bfhi(w[1]) * xv[3] +
               bflo(w[2]) * xv[4] + bfhi(w[2]) * xv[5] + bflo(w[3]) * xv[6] + bfhi(w[3]) * xv[7];
    }
    acc = wave_sum(acc);
    if (lane == 0) y[n] = acc + (b ? b[n] : 0.f);
}

// ---- conv_in: 3x3, Cin = 4, one block per output row, thread = 2 output channels ------------
template <int CIN>
__global__ void conv_in_kernel(const float* __restrict__ x, int Bsrc, const float* __restrict__ Wt,
                               const float* __restrict__ bias, bf16_t* __restrict__ y, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* patch = (float*)smem;  // [CIN][3][W+2]
    const int yrow = blockIdx.x, b = blockIdx.y;
    const int bs = b % Bsrc;
    const int tid = threadIdx.x;
    const int PW = W + 2;
    for (int i = tid; i < CIN * 3 * PW; i += blockDim.x) {
        const int ci = i / (3 * PW), rem = i - ci * 3 * PW;
        const int dy = rem / PW, px = rem - dy * PW;
        const int iy = yrow + dy - 1, ix = px - 1;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((long)bs * CIN + ci) * H + iy) * W + ix];
        patch[i] = v;
    }
    __syncthreads();
    const int co = tid * 2;
    if (co >= Cout) return;
    float w0[CIN * 9], w1[CIN * 9];
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) {
        w0[k] = Wt[k * Cout + co];
        w1[k] = Wt[k * Cout + co + 1];
    }
    const float b0 = bias[co], b1 = bias[co + 1];
    bf16_t* yr = y + (((long)b * H + yrow) * W) * Cout + co;
    for (int ox = 0; ox < W; ++ox) {
        float a0 = b0, a1 = b1;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = patch[(ci * 3 + dy) * PW + ox + dx];
                    a0 += v * w0[ci * 9 + dy * 3 + dx];
                    a1 += v * w1[ci * 9 + dy * 3 + dx];
                }
        *(unsigned*)(yr + (long)ox * Cout) = pack2bf(a0, a1);
    }
}

// ---- conv_out: 3x3, Cout <= 4, one wave per output pixel ------------------------------------
__global__ __launch_bounds__(256) void conv_out_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wp,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B,
                                                       int H, int W, int Cin, int Cout) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long npix = (long)B * H * W;
    if (pix >= npix) return;
    const int b = (int)(pix / (H * W));
    const int rem = (int)(pix - (long)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const int nchunks = Cin / 8;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = lane; ch < nchunks; ch += 64) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const u32x4 v = *(const u32x4*)(x + (((long)b * H + iy) * W + ix) * Cin + ch * 8);
            const float f[8] = {bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1]),
                                bflo(v[2]), bfhi(v[2]), bflo(v[3]), bfhi(v[3])};
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co >= Cout) break;
                const u32x4 w = *(const u32x4*)(Wp + ((long)co * 9 + tap) * Cin + ch * 8);
                acc[co] += f[0] * bflo(w[0]) + f[1] * bfhi(w[0]) + f[2] * bflo(w[1]) + f[3] * bfhi(w[1]) +
                           f[4] * bflo(w[2]) + f[5] * bfhi(w[2]) + f[6] * bflo(w[3]) + f[7] * bfhi(w[3]);
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[co] = wave_sum(acc[co]);
    const float mine = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
    if (lane < Cout) y[(((long)b * Cout + lane) * H + oy) * W + ox] = mine + bias[lane];
}

// ---- conv_out on the matrix cores: D[cout (16 rows, <= 4 live)][16 pixels] += W[cout][k] . X[pixel][k], k = (tap, channel) ----
// The wave-per-pixel kernel above re-reads the 4 x 9 x Cin weights for every pixel (23 KB of L1 traffic per pixel at
// Cin = 320: 133 us for the 64x64 UNet output).  Here the weights sit in LDS once per workgroup, a wave owns T x 16 consecutive
// pixels (T MFMA tiles) and streams their 9 shifted input rows straight from global memory into the B operand
// (lane = (pixel, 8-channel group): 16 contiguous bytes each); out-of-image taps are zeroed after an unconditional load.
template <int T>
__global__ __launch_bounds__(256) void conv_out_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wp,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B,
                                                            int H, int W, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // W rows 0..3: [4][9 * Cin] bf16 (rows >= Cout zero)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = 9 * Cin;
    for (int i = tid; i < 4 * (K / 8); i += 256) {
        const int row = i / (K / 8);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < Cout) v = *(const u32x4*)(Wp + (long)i * 8);
        *(u32x4*)(smem + (long)i * 16) = v;
    }
    __syncthreads();
    const int l15 = lane & 15, kg = lane >> 4;
    const long npix = (long)B * H * W;
    const long pix0 = ((long)blockIdx.x * 4 + wave) * (16 * T);
    if (pix0 >= npix) return;
    int oy[T], ox[T];
    long base[T];
    bool valid[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const long pp = pix0 + t * 16 + l15;
        valid[t] = pp < npix;
        const long pc = valid[t] ? pp : npix - 1;
        const int b = (int)(pc / (H * W));
        const int rem = (int)(pc - (long)b * H * W);
        oy[t] = rem / W;
        ox[t] = rem - oy[t] * W;
        base[t] = (long)b * H * W;
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wrow = smem + ((long)(l15 & 3) * K + kg * 8) * 2;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const bf16_t* src[T];
        bool ok[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int iy = oy[t] + tap / 3 - 1, ix = ox[t] + tap % 3 - 1;
            ok[t] = valid[t] && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            src[t] = x + (base[t] + (long)cy * W + cx) * Cin + kg * 8;
        }
#pragma unroll 10
        for (int c0 = 0; c0 < Cin; c0 += 32) {
            bf16x8 wf = *(const bf16x8*)(wrow + (tap * Cin + c0) * 2);
            wf = l15 < 4 ? wf : zero8;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bf16x8 xf = *(const bf16x8*)(src[t] + c0);
                xf = ok[t] ? xf : zero8;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[t], 0, 0, 0);
            }
        }
    }
    if (kg == 0) {          // lanes 0..15: rows (output channels) 0..3 of pixel l15
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (!valid[t]) continue;
            const long b = base[t] / ((long)H * W);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < Cout) y[((b * Cout + j) * H + oy[t]) * W + ox[t]] = acc[t][j] + bias[j];
        }
    }
}

// ---- fused CFG + scheduler update ---------------------------------------------------------
__global__ void sched_step_kernel(const float* __restrict__ eps, int cfg, float guidance,
                                  const float* __restrict__ x, const float* __restrict__ m1,
                                  const float* __restrict__ m2, const float* __restrict__ m3,
                                  const float* __restrict__ noise, float* __restrict__ prev,
                                  float* __restrict__ y2, float* __restrict__ m_out, StepCoef c, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 e = ((const f32x4*)eps)[i];
    if (cfg) {
        const f32x4 et = ((const f32x4*)eps)[i + n4];
        e = e + guidance * (et - e);
    }
    const f32x4 xv = ((const f32x4*)x)[i];
    f32x4 p = c.px * xv + c.pe * e;
    if (m1) p += c.p1 * ((const f32x4*)m1)[i];
    if (m2) p += c.p2 * ((const f32x4*)m2)[i];
    if (m3) p += c.p3 * ((const f32x4*)m3)[i];
    if (noise) p += c.pn * ((const f32x4*)noise)[i];
    if (y2) ((f32x4*)y2)[i] = c.yx * xv + c.ye * e;
    if (m_out) ((f32x4*)m_out)[i] = c.mx * xv + c.me * e;
    ((f32x4*)prev)[i] = p;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = ((const f32x4*)src)[i];
    u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    ((u32x2*)dst)[i] = o;
}

__global__ void pqconv_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                              float* __restrict__ y, int B, int HW, float in_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int b = (int)(i / HW);
    const long p = i - (long)b * HW;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = x[((long)b * 4 + c) * HW + p] * in_scale;
#pragma unroll
    for (int co = 0; co < 4; ++co)
        y[((long)b * 4 + co) * HW + p] = bias[co] + W[co * 4 + 0] * v[0] + W[co * 4 + 1] * v[1] + W[co * 4 + 2] * v[2] + W[co * 4 + 3] * v[3];
}

}  // namespace

int sd_launch_pqconv(const float* x, const float* W, const float* bias, float* y, int B, int HW, float in_scale,
                     hipStream_t stream) {
    SD_REQUIRE(x && W && bias && y && B > 0 && HW > 0, "pqconv: bad operand");
    const long n = (long)B * HW;
    hipLaunchKernelGGL(pqconv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, W, bias, y, B, HW, in_scale);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_f32_to_bf16(const float* src, bf16_t* dst, long n, hipStream_t stream) {
    SD_REQUIRE(src && dst && n > 0 && n % 4 == 0, "f32_to_bf16: n=%ld must be a positive multiple of 4", n);
    const long n4 = n / 4;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, src, dst, n4);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_timestep_sinusoid(float t, float* out, int dim, hipStream_t stream) {
    SD_REQUIRE(out && dim > 0 && dim % 2 == 0, "sinusoid: bad dim %d", dim);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((dim + 255) / 256), dim3(256), 0, stream, t, out, dim);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_gemv(const float* x, const bf16_t* W, const float* b, float* y, int N, int K, int silu_in,
                   hipStream_t stream) {
    SD_REQUIRE(x && W && y, "gemv: null operand");
    SD_REQUIRE(K % 8 == 0 && N > 0, "gemv: K=%d must be a multiple of 8", K);
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, W, b, y, N, K, silu_in);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[b][h*80 + j][c] = (j < L and c in head h) ? scale * kv[b*L + j][col_off + c] : 0   (kv rows are [K | V], 2C wide)
__global__ void xattn_expand_kernel(const bf16_t* __restrict__ kv, bf16_t* __restrict__ out, int L, int C, int NH,
                                    int col_off, float scale) {
    const int row = blockIdx.x;                       // b * NH * 80 + h * 80 + j
    const int j = row % 80, h = (row / 80) % NH, b = row / (80 * NH);
    const int d = C / NH;
    const bf16_t* src = kv + ((long)b * L + (j < L ? j : 0)) * 2 * C + col_off;
    bf16_t* dst = out + (long)row * C;
    for (int c = threadIdx.x * 2; c < C; c += blockDim.x * 2) {
        unsigned v = 0;
        if (j < L && c / d == h) {                    // d is even: a channel pair never straddles heads
            const unsigned u = *(const unsigned*)(src + c);
            v = pack2bf(bflo(u) * scale, bfhi(u) * scale);
        }
        *(unsigned*)(dst + c) = v;
    }
}

// dst[b][c][r] = src[b][r][c]  (bf16, 32x32 tiles through LDS)
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int Cc, int perm16) {
    __shared__ bf16_t tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const bf16_t* s = src + (long)b * R * Cc;
    bf16_t* d = dst + (long)b * R * Cc;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Cc) tile[i][tx] = s[(long)(r0 + i) * Cc + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Cc && r0 + tx < R) {
            int rr = r0 + tx;
            if (perm16) rr = (rr & ~12) | ((rr & 4) << 1) | ((rr & 8) >> 1);
            d[(long)(c0 + i) * R + rr] = tile[tx][i];
        }
}

int sd_launch_transpose_bf16(const bf16_t* src, bf16_t* dst, int B, int R, int Cc, hipStream_t stream, int perm16) {
    SD_REQUIRE(src && dst && B > 0 && B <= 65535 && R > 0 && Cc > 0, "transpose: B=%d R=%d C=%d", B, R, Cc);
    SD_REQUIRE(!perm16 || R % 16 == 0, "transpose: the 16-group permutation needs R=%d to be a multiple of 16", R);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, B), dim3(256), 0, stream, src, dst, R, Cc,
                       perm16);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

// dst[b][R / RT][K / 32][RT][32] = src[b][R][K] (bf16): every (row tile, 32-column slice) becomes RT contiguous 64-byte
// rows, so that a 16-row x 64-byte LDS-DMA piece of the fused cross-attention kernel is ONE contiguous KiB (8 full cache
// lines) instead of 16 half lines at the row stride.  One 16-byte chunk per thread.
__global__ void retile32_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int K, int RT) {
    const long per = (long)R * K / 8;                            // 16-byte chunks per sample
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const int b = blockIdx.y;
    const int kc = (int)(i % (K / 8)), row = (int)(i / (K / 8));  // source chunk (row, 8 columns)
    const int ks = kc >> 2, cc = kc & 3;                          // 32-column slice, chunk inside it
    const long d = ((((long)(row / RT) * (K / 32) + ks) * RT + row % RT) * 4 + cc);
    const u32x4 v = *(const u32x4*)(src + ((long)b * R * K) + (long)row * K + kc * 8);
    *(u32x4*)(dst + ((long)b * R * K) + d * 8) = v;
}

// dst[r][i] = src[i] for r < rep: the prompt-independent prefix of a CFG forward is computed once per latent and handed
// to the conditional / unconditional halves (16-byte vectors; n16 = bytes / 16 of ONE copy)
__global__ void replicate_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16, int rep) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const u32x4 v = src[i];
    for (int r = 0; r < rep; ++r) dst[(long)r * n16 + i] = v;
}

int sd_launch_replicate(const void* src, void* dst, long bytes, int rep, hipStream_t stream) {
    SD_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && rep >= 1, "replicate: bytes=%ld rep=%d", bytes, rep);
    const long n16 = bytes / 16;
    hipLaunchKernelGGL(replicate_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst,
                       n16, rep);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_retile32(const bf16_t* src, bf16_t* dst, int B, int R, int K, int RT, hipStream_t stream) {
    SD_REQUIRE(src && dst && B > 0 && B <= 65535 && K % 32 == 0 && RT > 0 && R % RT == 0, "retile32: B=%d R=%d K=%d RT=%d", B, R, K, RT);
    const long per = (long)R * K / 8;
    hipLaunchKernelGGL(retile32_kernel, dim3((unsigned)((per + 255) / 256), B), dim3(256), 0, stream, src, dst, R, K, RT);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_xattn_expand(const bf16_t* kv, bf16_t* out, int B, int L, int C, int NH, int col_off, float scale,
                           hipStream_t stream) {
    SD_REQUIRE(kv && out && B > 0 && L > 0 && L <= 80 && NH > 0 && C % NH == 0 && (C / NH) % 2 == 0,
               "xattn_expand: B=%d L=%d C=%d heads=%d", B, L, C, NH);
    hipLaunchKernelGGL(xattn_expand_kernel, dim3(B * NH * 80), dim3(128), 0, stream, kv, out, L, C, NH, col_off, scale);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_conv_in(const float* x, int Bsrc, const float* Wt, const float* bias, bf16_t* y, int B, int H, int W,
                      int Cin, int Cout, hipStream_t stream) {
    SD_REQUIRE(x && Wt && bias && y, "conv_in: null operand");
    SD_REQUIRE(Cin == 4, "conv_in: Cin=%d (only 4 is built)", Cin);
    SD_REQUIRE(Cout % 2 == 0 && Cout / 2 <= 1024, "conv_in: Cout=%d", Cout);
    SD_REQUIRE(Bsrc > 0 && B > 0 && B % Bsrc == 0, "conv_in: batch %d not a multiple of source batch %d", B, Bsrc);
    const int threads = (Cout / 2 + 63) / 64 * 64;
    const size_t smem = (size_t)Cin * 3 * (W + 2) * sizeof(float);
    hipLaunchKernelGGL((conv_in_kernel<4>), dim3(H, B), dim3(threads), smem, stream, x, Bsrc, Wt, bias, y, H, W, Cout);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_conv_out(const bf16_t* x, const bf16_t* Wp, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, hipStream_t stream) {
    SD_REQUIRE(x && Wp && bias && y, "conv_out: null operand");
    SD_REQUIRE(Cin % 8 == 0 && Cout >= 1 && Cout <= 4, "conv_out: Cin=%d Cout=%d", Cin, Cout);
    const long npix = (long)B * H * W;
    static const bool no_mfma = getenv("SD_CONV_OUT_VALU") != nullptr;
    const size_t smem = (size_t)4 * 9 * Cin * 2;
    if (!no_mfma && Cin % 32 == 0 && smem <= 64 * 1024) {
        static const int tiles = getenv("SD_CONV_OUT_TILES") ? atoi(getenv("SD_CONV_OUT_TILES")) : 1;
        if (tiles == 2)
            hipLaunchKernelGGL(conv_out_mfma_kernel<2>, dim3((unsigned)((npix + 127) / 128)), dim3(256), smem, stream, x, Wp, bias, y,
                               B, H, W, Cin, Cout);
        else
            hipLaunchKernelGGL(conv_out_mfma_kernel<1>, dim3((unsigned)((npix + 63) / 64)), dim3(256), smem, stream, x, Wp, bias, y,
                               B, H, W, Cin, Cout);
        SD_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, stream, x, Wp, bias, y, B, H, W,
                       Cin, Cout);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_sched_step(const float* eps, int cfg, float guidance, const float* x, const float* m1,
                         const float* m2, const float* m3, const float* noise, float* prev, float* y2, float* m_out,
                         StepCoef c, long n, hipStream_t stream) {
    SD_REQUIRE(eps && x && prev, "sched_step: null operand");
    SD_REQUIRE(n > 0 && n % 4 == 0, "sched_step: n=%ld must be a positive multiple of 4", n);
    const long n4 = n / 4;
    hipLaunchKernelGGL(sched_step_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, eps, cfg, guidance,
                       x, m1, m2, m3, noise, prev, y2, m_out, c, n4);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
