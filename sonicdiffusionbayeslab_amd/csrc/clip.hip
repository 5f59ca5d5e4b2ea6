// Small kernels of the CLIP text encoder (`encode_prompt`, src/models.py:139-155).
// 77 tokens x 768 channels per prompt, once per batch and outside the timed loop: the projections and the MLP
// run on the shared bf16 MFMA GEMM; these are the pieces that have no other home.  Everything here is
// latency-, not throughput-, relevant (6.65 GMAC per prompt in total).
#include "common.h"
#include "kernels.h"

namespace {

// h[b, l, :] = token_embedding[ids[b, l], :] + position_embedding[l, :]   (fp32 sum, bf16 out)
__global__ void clip_embed_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ tok,
                                  const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int L, int H, int vocab) {
    const int row = blockIdx.x;                   // b * L + l
    const int l = row % L;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const bf16_t* t = tok + (long)id * H;
    const bf16_t* p = pos + (long)l * H;
    for (int c = threadIdx.x * 2; c < H; c += blockDim.x * 2) {
        const unsigned a = *(const unsigned*)(t + c), b = *(const unsigned*)(p + c);
        *(unsigned*)(out + (long)row * H + c) = pack2bf(bflo(a) + bflo(b), bfhi(a) + bfhi(b));
    }
}

// Causal self-attention of one (prompt, head): L <= 128 tokens, head dim D <= 64.  One thread per query row;
// K and V of the head are staged in LDS as fp32, scores never leave registers/LDS.  qkv is the fused
// projection output [B*L, 3*H] (q | k | v), out is [B*L, H].
template <int D>
__global__ void clip_attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int L, int H, float scale) {
    extern __shared__ float sm[];                 // K [L][D+1], V [L][D+1], P [blockDim][L]
    const int b = blockIdx.y, h = blockIdx.x;
    float* Ks = sm;
    float* Vs = sm + L * (D + 1);
    float* Ps = Vs + L * (D + 1);
    const long ld = 3L * H;
    for (int i = threadIdx.x; i < L * D; i += blockDim.x) {
        const int j = i / D, d = i - j * D;
        const bf16_t* row = qkv + ((long)b * L + j) * ld + h * D + d;
        Ks[j * (D + 1) + d] = bf2f(row[H]);
        Vs[j * (D + 1) + d] = bf2f(row[2 * H]);
    }
    __syncthreads();
    const int q = threadIdx.x;
    if (q >= L) return;
    float qv[D];
    const bf16_t* qrow = qkv + ((long)b * L + q) * ld + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) qv[d] = bf2f(qrow[d]) * scale;
    float* P = Ps + q * L;
    float m = -1e30f;
    for (int j = 0; j <= q; ++j) {                // causal mask: keys 0..q
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s += qv[d] * Ks[j * (D + 1) + d];
        P[j] = s;
        m = fmaxf(m, s);
    }
    float den = 0.f;
    for (int j = 0; j <= q; ++j) {
        const float e = __expf(P[j] - m);
        P[j] = e;
        den += e;
    }
    const float inv = 1.0f / den;
    float o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    for (int j = 0; j <= q; ++j) {
        const float pj = P[j];
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] += pj * Vs[j * (D + 1) + d];
    }
    bf16_t* orow = out + ((long)b * L + q) * H + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 2) *(unsigned*)(orow + d) = pack2bf(o[d] * inv, o[d + 1] * inv);
}

// quick_gelu in place: x * sigmoid(1.702 x)  (CLIP's hidden_act)
__global__ void quick_gelu_kernel(bf16_t* __restrict__ x, long n2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const unsigned v = *(const unsigned*)(x + 2 * i);
    const float a = bflo(v), b = bfhi(v);
    *(unsigned*)(x + 2 * i) = pack2bf(a / (1.0f + __expf(-1.702f * a)), b / (1.0f + __expf(-1.702f * b)));
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = bf2f(src[i]);
}

}  // namespace

int sd_launch_clip_embed(const int* ids, const bf16_t* tok, const bf16_t* pos, bf16_t* out, int rows, int L, int H,
                         int vocab, hipStream_t stream) {
    SD_REQUIRE(ids && tok && pos && out, "clip_embed: null operand");
    SD_REQUIRE(H % 2 == 0 && rows > 0 && L > 0, "clip_embed: rows=%d L=%d H=%d", rows, L, H);
    hipLaunchKernelGGL(clip_embed_kernel, dim3(rows), dim3(128), 0, stream, ids, tok, pos, out, L, H, vocab);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_clip_attention(const bf16_t* qkv, bf16_t* out, int B, int L, int H, int heads, hipStream_t stream) {
    SD_REQUIRE(qkv && out, "clip_attention: null operand");
    SD_REQUIRE(heads > 0 && H % heads == 0, "clip_attention: H=%d heads=%d", H, heads);
    const int D = H / heads;
    SD_REQUIRE(L >= 1 && L <= 128, "clip_attention: %d tokens (1..128 are built)", L);
    const int threads = (L + 63) / 64 * 64;
    const size_t smem = ((size_t)2 * L * (D + 1) + (size_t)threads * L) * sizeof(float);
    const float scale = 1.0f / sqrtf((float)D);
    if (D == 64) {
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)clip_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(clip_attn_kernel<64>, dim3(heads, B), dim3(threads), smem, stream, qkv, out, L, H, scale);
    } else if (D == 16) {   // the reduced configurations of the tests
        SD_CHECK_HIP(hipFuncSetAttribute((const void*)clip_attn_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(clip_attn_kernel<16>, dim3(heads, B), dim3(threads), smem, stream, qkv, out, L, H, scale);
    } else {
        SD_REQUIRE(false, "clip_attention: head dim %d (64 and 16 are built)", D);
    }
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_quick_gelu(bf16_t* x, long n, hipStream_t stream) {
    SD_REQUIRE(x && n > 0 && n % 2 == 0, "quick_gelu: n=%ld", n);
    const long n2 = n / 2;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, x, n2);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

int sd_launch_bf16_to_f32(const bf16_t* src, float* dst, long n, hipStream_t stream) {
    SD_REQUIRE(src && dst && n > 0, "bf16_to_f32: null operand");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}
