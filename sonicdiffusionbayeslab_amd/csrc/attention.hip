// Flash-style attention (self and cross) for gfx950, head dims 40 / 80 / 160 (SD-1.5: 8 heads
// at C = 320 / 640 / 1280), arbitrary key count (4096 ... 64 for self-attention, 77 for the
// prompt cross-attention), bf16 in / fp32 accumulate / bf16 out.
//
// Layout: Q, K, V, O are token-major [B, N, heads*D] views (row strides ldq/ldk/ldv/ldo), so
// the fused QKV projection output is consumed in place.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries.
//   S^T = K . Q^T   (v_mfma_f32_32x32x16_bf16, K rows as A operand, Q held in registers)
// puts the query on the lane and 16 keys in the accumulator registers, so the online softmax
// is lane-local (one cross-half exchange for the max and the final sum) and the bf16-packed
// probabilities are directly the B operand of
//   O^T += V^T . P^T
// with no LDS round trip.  K rows are read in the order pi(r) = r with bits 2,3 swapped so the
// accumulator-register -> k mapping of the second product meets V in natural key order; V^T
// fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose).
// K/V tiles (64 keys) are register-prefetched one tile ahead and staged in LDS with rows padded
// to an odd number of 16-byte slots (conflict-free ds_read_b128 for the K fragments).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

__device__ __forceinline__ int pi_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int D, bool USE_TR>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
    constexpr int DK = (D + 15) / 16 * 16;
    constexpr int KQ = DK / 16;
    constexpr int DVT = (D + 31) / 32;
    constexpr int CH = D / 8;
    constexpr int SLOTS = (DK > DVT * 32 ? DK : DVT * 32) / 8;
    constexpr int RS = (SLOTS | 1) * 16;         // LDS row stride in bytes (odd # of 16-B slots)
    constexpr int NPF = (64 * CH + 255) / 256;   // 16-B chunks each thread prefetches per tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + 64 * RS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 128 + wave * 32 + r;
    const bool qvalid = q < a.Nq;
    const float c = a.scale * 1.4426950408889634f;

    // ---- zero LDS once (pad columns stay zero for the whole kernel) ----
    for (int i = tid; i < 2 * 64 * RS / 16; i += 256) *(u32x4*)(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q][16kk + 8h .. +8) ----
    bf16x8 qf[KQ];
    {
        const bf16_t* qp = a.Q + ((long)b * a.Nq + (qvalid ? q : 0)) * a.ldq + head * D;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int col = kk * 16 + h * 8;
            if (qvalid && col < D) qf[kk] = *(const bf16x8*)(qp + col);
            else qf[kk] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    const bf16_t* kbase = a.K + (long)b * a.Nk * a.ldk + head * D;
    const bf16_t* vbase = a.V + (long)b * a.Nk * a.ldv + head * D;
    u32x4 kreg[NPF], vreg[NPF];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / CH, ch = idx - key * CH;
            const int gkey = t * 64 + key;
            if (idx < 64 * CH && gkey < a.Nk) {
                kreg[i] = *(const u32x4*)(kbase + (long)gkey * a.ldk + ch * 8);
                vreg[i] = *(const u32x4*)(vbase + (long)gkey * a.ldv + ch * 8);
            } else {
                kreg[i] = u32x4{0u, 0u, 0u, 0u};
                vreg[i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / CH, ch = idx - key * CH;
            if (idx < 64 * CH) {
                *(u32x4*)(Ks + key * RS + ch * 16) = kreg[i];
                *(u32x4*)(Vs + key * RS + ch * 16) = vreg[i];
            }
        }
    };

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int kfrag_off = pi_swap23(r) * RS + h * 16;
    // transposed V read: lane 4q4+p4 of each 16-lane group addresses key row q4, 4 columns at 4*p4
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int vtr_off = (8 * h + q4) * RS + (16 * g16 + 4 * p4) * 2;

    const int ntiles = (a.Nk + 63) / 64;
    prefetch(0);
    __syncthreads();  // LDS zero fill done
    for (int t = 0; t < ntiles; ++t) {
        commit();
        __syncthreads();
        if (t + 1 < ntiles) prefetch(t + 1);

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * 64 + sub * 32;
            if (key0 >= a.Nk) break;
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            const char* kp = Ks + sub * 32 * RS + kfrag_off;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(kp + kk * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s, 0, 0, 0);
            }
            if (key0 + 32 > a.Nk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (key0 + pi_swap23(row) >= a.Nk) s[i] = -1e30f;
                }
            }
            float tmax = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            if (!__all(m_new == m_run)) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                l_run *= alpha;
#pragma unroll
                for (int tt = 0; tt < DVT; ++tt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[tt][i] *= alpha;
                m_run = m_new;
            }
            const float mc = m_run * c;
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c, -mc));
                l_run += p[i];
            }
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 w = {pack2bf(p[8 * s2 + 0], p[8 * s2 + 1]), pack2bf(p[8 * s2 + 2], p[8 * s2 + 3]),
                           pack2bf(p[8 * s2 + 4], p[8 * s2 + 5]), pack2bf(p[8 * s2 + 6], p[8 * s2 + 7])};
                pf[s2] = __builtin_bit_cast(bf16x8, w);
            }
            const char* vp = Vs + sub * 32 * RS;
#pragma unroll
            for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 vf;
                    if (USE_TR) {
                        const char* ap = vp + s2 * 16 * RS + vtr_off + tt * 64;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) bf16x4*)(ap));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) bf16x4*)(ap + 4 * RS));
                        vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            vf[j] = *(const short*)(vp + (s2 * 16 + 8 * h + j) * RS + (32 * tt + r) * 2);
                    }
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o[tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // everyone done with this tile before it is overwritten
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qvalid) {
        bf16_t* op = a.O + ((long)b * a.Nq + q) * a.ldo + head * D;
#pragma unroll
        for (int tt = 0; tt < DVT; ++tt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dv = 32 * tt + 8 * g4 + 4 * h;
                if (dv < D) {
                    u32x2 w = {pack2bf(o[tt][4 * g4 + 0] * inv, o[tt][4 * g4 + 1] * inv),
                               pack2bf(o[tt][4 * g4 + 2] * inv, o[tt][4 * g4 + 3] * inv)};
                    *(u32x2*)(op + dv) = w;
                }
            }
        }
    }
}

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
    constexpr int DK = (D + 15) / 16 * 16;
    constexpr int DVT = (D + 31) / 32;
    constexpr int SLOTS = (DK > DVT * 32 ? DK : DVT * 32) / 8;
    constexpr int RS = (SLOTS | 1) * 16;
    const int smem = 2 * 64 * RS;
    static const bool no_tr = getenv("SD_ATTN_NO_TR") != nullptr;
    dim3 grid((a.Nq + 127) / 128, a.heads, a.B);
    if (no_tr) hipLaunchKernelGGL((attn_kernel<D, false>), grid, dim3(256), smem, stream, a);
    else hipLaunchKernelGGL((attn_kernel<D, true>), grid, dim3(256), smem, stream, a);
    SD_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int sd_launch_attention(const AttnArgs& a, hipStream_t stream) {
    SD_REQUIRE(a.Q && a.K && a.V && a.O, "attention: null operand");
    SD_REQUIRE(a.B > 0 && a.heads > 0 && a.Nq > 0 && a.Nk > 0, "attention: empty problem");
    SD_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0,
               "attention: row strides must keep 16-byte alignment");
    SD_REQUIRE(a.ldq >= (long)a.heads * a.D && a.ldk >= (long)a.heads * a.D && a.ldv >= (long)a.heads * a.D &&
                   a.ldo >= (long)a.heads * a.D, "attention: row stride smaller than heads*D");
    SD_REQUIRE(((uintptr_t)a.Q | (uintptr_t)a.K | (uintptr_t)a.V) % 16 == 0 && (uintptr_t)a.O % 8 == 0,
               "attention: operands must be 16-byte aligned");
    switch (a.D) {
        case 40: return launch_attn<40>(a, stream);
        case 80: return launch_attn<80>(a, stream);
        case 160: return launch_attn<160>(a, stream);
        default: sd_set_error("attention: head dim %d not supported (40, 80, 160)", a.D); return -1;
    }
}
