// What gfx950 REALLY needs between a bf16 MFMA and a vector-ALU instruction that touches its registers, measured with
// hand-placed wait states inside ONE asm statement (hipcc pads nothing inside asm):
//   WAR  a VALU instruction WRITES the MFMA's SrcC registers N wait states after the MFMA issued (dst != SrcC: the
//        "VGPR form" hipcc picks when it renames accumulators, e.g. in a peeled last K tile)
//   RAW  a VALU instruction READS the MFMA's D registers N wait states after it issued (v_mov_b32 and v_pk_add_f32 readers)
//   WAW  a VALU instruction WRITES the MFMA's D registers N wait states after it issued
// for v_mfma_f32_16x16x32_bf16 (4 passes) and v_mfma_f32_32x32x16_bf16 (8 passes), with 0..8 independent MFMAs queued in
// front of the probed one and with or without a second wave on the same SIMD streaming MFMAs (waves 4-7 of the workgroup).
// hipcc's own numbers (GCNHazardRecognizer, ROCm 7.2): WAR 3 / 7, RAW 7 / 11, WAW 8? -- the probe prints the smallest N at
// which no lane of any launch is wrong.  Round 5: the LayerNorm-fold wrong results of round 4 (16 rows x 1-2 columns,
// lanes 48-63 = the LAST pass of a 16x16 MFMA) sit exactly on the K-loop -> epilogue boundary where hipcc gives the WAR
// hazard its bare minimum (profiles/round5_notes.md).
// Development probe: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_hazard.hip -o /tmp/mfma_hazard && /tmp/mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define NOP_N "s_nop %[n]\n\t"
#define LONG_WAIT "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
// 16x16x32: C v[200:203], D v[204:207], A v[208:211], B v[212:215], captured v[216:219], scratch D v[220:251]
#define LOAD16                                                                                                         \
    "v_mov_b32 v200, %[c0]\n\tv_mov_b32 v201, %[c1]\n\tv_mov_b32 v202, %[c2]\n\tv_mov_b32 v203, %[c3]\n\t"            \
    "v_mov_b32 v204, %[sen]\n\tv_mov_b32 v205, %[sen]\n\tv_mov_b32 v206, %[sen]\n\tv_mov_b32 v207, %[sen]\n\t"        \
    "v_mov_b32 v216, %[sen]\n\tv_mov_b32 v217, %[sen]\n\tv_mov_b32 v218, %[sen]\n\tv_mov_b32 v219, %[sen]\n\t"        \
    "v_mov_b32 v198, 0\n\tv_mov_b32 v199, 0\n\t"                                                                        \
    "v_mov_b32 v208, %[a0]\n\tv_mov_b32 v209, %[a1]\n\tv_mov_b32 v210, %[a2]\n\tv_mov_b32 v211, %[a3]\n\t"            \
    "v_mov_b32 v212, %[b0]\n\tv_mov_b32 v213, %[b1]\n\tv_mov_b32 v214, %[b2]\n\tv_mov_b32 v215, %[b3]\n\t" LONG_WAIT
#define PRE1(r) "v_mfma_f32_16x16x32_bf16 v[" #r "], v[208:211], v[212:215], 0\n\t"
#define MFMA16 "v_mfma_f32_16x16x32_bf16 v[204:207], v[208:211], v[212:215], v[200:203]\n\t"
#define WAR16 "v_mov_b32 v203, %[sen]\n\tv_mov_b32 v202, %[sen]\n\tv_mov_b32 v201, %[sen]\n\tv_mov_b32 v200, %[sen]\n\t" LONG_WAIT \
    "v_mov_b32 v216, v204\n\tv_mov_b32 v217, v205\n\tv_mov_b32 v218, v206\n\tv_mov_b32 v219, v207\n\t"
#define RAW16 "v_mov_b32 v219, v207\n\tv_mov_b32 v218, v206\n\tv_mov_b32 v217, v205\n\tv_mov_b32 v216, v204\n\t" LONG_WAIT
#define RAWPK16 "v_pk_add_f32 v[218:219], v[206:207], v[198:199]\n\tv_pk_add_f32 v[216:217], v[204:205], v[198:199]\n\t" LONG_WAIT
#define WAW16 "v_mov_b32 v207, %[sen2]\n\tv_mov_b32 v206, %[sen2]\n\tv_mov_b32 v205, %[sen2]\n\tv_mov_b32 v204, %[sen2]\n\t" LONG_WAIT \
    "v_mov_b32 v216, v204\n\tv_mov_b32 v217, v205\n\tv_mov_b32 v218, v206\n\tv_mov_b32 v219, v207\n\t"
#define STORE_OUT "v_mov_b32 %[o0], v216\n\tv_mov_b32 %[o1], v217\n\tv_mov_b32 %[o2], v218\n\tv_mov_b32 %[o3], v219\n\t"

// 32x32x16: C v[160:175], D v[176:191]; the LAST pass covers rows 28..31 = registers 12..15 of lanes 32..63: the probe
// touches registers 15, 14, 13, 12 (in that order) and captures D registers 12..15.
#define LOAD32                                                                                                         \
    "v_mov_b32 v160, %[c0]\n\tv_mov_b32 v161, %[c1]\n\tv_mov_b32 v162, %[c2]\n\tv_mov_b32 v163, %[c3]\n\t"            \
    "v_mov_b32 v164, %[c0]\n\tv_mov_b32 v165, %[c1]\n\tv_mov_b32 v166, %[c2]\n\tv_mov_b32 v167, %[c3]\n\t"            \
    "v_mov_b32 v168, %[c0]\n\tv_mov_b32 v169, %[c1]\n\tv_mov_b32 v170, %[c2]\n\tv_mov_b32 v171, %[c3]\n\t"            \
    "v_mov_b32 v172, %[c0]\n\tv_mov_b32 v173, %[c1]\n\tv_mov_b32 v174, %[c2]\n\tv_mov_b32 v175, %[c3]\n\t"            \
    "v_mov_b32 v188, %[sen]\n\tv_mov_b32 v189, %[sen]\n\tv_mov_b32 v190, %[sen]\n\tv_mov_b32 v191, %[sen]\n\t"        \
    "v_mov_b32 v216, %[sen]\n\tv_mov_b32 v217, %[sen]\n\tv_mov_b32 v218, %[sen]\n\tv_mov_b32 v219, %[sen]\n\t"        \
    "v_mov_b32 v198, 0\n\tv_mov_b32 v199, 0\n\t"                                                                        \
    "v_mov_b32 v208, %[a0]\n\tv_mov_b32 v209, %[a1]\n\tv_mov_b32 v210, %[a2]\n\tv_mov_b32 v211, %[a3]\n\t"            \
    "v_mov_b32 v212, %[b0]\n\tv_mov_b32 v213, %[b1]\n\tv_mov_b32 v214, %[b2]\n\tv_mov_b32 v215, %[b3]\n\t" LONG_WAIT
#define PRE32(r) "v_mfma_f32_32x32x16_bf16 v[" #r "], v[208:211], v[212:215], 0\n\t"
#define MFMA32 "v_mfma_f32_32x32x16_bf16 v[176:191], v[208:211], v[212:215], v[160:175]\n\t"
#define WAR32 "v_mov_b32 v175, %[sen]\n\tv_mov_b32 v174, %[sen]\n\tv_mov_b32 v173, %[sen]\n\tv_mov_b32 v172, %[sen]\n\t" LONG_WAIT LONG_WAIT \
    "v_mov_b32 v216, v188\n\tv_mov_b32 v217, v189\n\tv_mov_b32 v218, v190\n\tv_mov_b32 v219, v191\n\t"
#define RAW32 "v_mov_b32 v219, v191\n\tv_mov_b32 v218, v190\n\tv_mov_b32 v217, v189\n\tv_mov_b32 v216, v188\n\t" LONG_WAIT LONG_WAIT
#define RAWPK32 "v_pk_add_f32 v[218:219], v[190:191], v[198:199]\n\tv_pk_add_f32 v[216:217], v[188:189], v[198:199]\n\t" LONG_WAIT LONG_WAIT
#define WAW32 "v_mov_b32 v191, %[sen2]\n\tv_mov_b32 v190, %[sen2]\n\tv_mov_b32 v189, %[sen2]\n\tv_mov_b32 v188, %[sen2]\n\t" LONG_WAIT LONG_WAIT \
    "v_mov_b32 v216, v188\n\tv_mov_b32 v217, v189\n\tv_mov_b32 v218, v190\n\tv_mov_b32 v219, v191\n\t"

#define CLOB                                                                                                             \
    "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174",  \
        "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188",      \
        "v189", "v190", "v191", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208",      \
        "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222",      \
        "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236",      \
        "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251"

#define OPS                                                                                                              \
    [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])                                               \
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]),    \
          [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [sen] "v"(sen), [sen2] "v"(sen2),  \
          [n] "n"(N > 0 ? N - 1 : 0)                                                                                         \
        : CLOB

enum { WAR = 0, RAW = 1, RAWPK = 2, WAW = 3 };

// KIND 0: 16x16x32, 1: 32x32x16.  PRE independent MFMAs of the same shape directly in front of the probed one.
template <int KIND, int MODE, int N, int PRE>
__device__ __forceinline__ void probe_once(const u32x4 a, const u32x4 b, const u32x4 c, unsigned sen, unsigned sen2, unsigned (&o)[4]) {
#define BODY(LOAD, P1, MF, TAIL)                                                                                         \
    if constexpr (PRE == 0) {                                                                                            \
        if constexpr (N == 0) asm volatile(LOAD MF TAIL STORE_OUT : OPS);                                                \
        else asm volatile(LOAD MF NOP_N TAIL STORE_OUT : OPS);                                                           \
    } else if constexpr (PRE == 2) {                                                                                     \
        if constexpr (N == 0) asm volatile(LOAD P1(220:223) P1(224:227) MF TAIL STORE_OUT : OPS);                        \
        else asm volatile(LOAD P1(220:223) P1(224:227) MF NOP_N TAIL STORE_OUT : OPS);                                   \
    } else {                                                                                                             \
        if constexpr (N == 0) asm volatile(LOAD P1(220:223) P1(224:227) P1(228:231) P1(232:235) P1(236:239) P1(240:243) MF TAIL STORE_OUT : OPS); \
        else asm volatile(LOAD P1(220:223) P1(224:227) P1(228:231) P1(232:235) P1(236:239) P1(240:243) MF NOP_N TAIL STORE_OUT : OPS); \
    }
    if constexpr (KIND == 0) {
        if constexpr (MODE == WAR) { BODY(LOAD16, PRE1, MFMA16, WAR16) }
        else if constexpr (MODE == RAW) { BODY(LOAD16, PRE1, MFMA16, RAW16) }
        else if constexpr (MODE == RAWPK) { BODY(LOAD16, PRE1, MFMA16, RAWPK16) }
        else { BODY(LOAD16, PRE1, MFMA16, WAW16) }
    } else {
        // (the 32x32 scratch destinations of the queued MFMAs are 16 registers wide: only two fit v[220:251])
#define P32(r) PRE32(r)
#define BODY32(TAIL)                                                                                                     \
    if constexpr (PRE == 0) {                                                                                            \
        if constexpr (N == 0) asm volatile(LOAD32 MFMA32 TAIL STORE_OUT : OPS);                                          \
        else asm volatile(LOAD32 MFMA32 NOP_N TAIL STORE_OUT : OPS);                                                     \
    } else {                                                                                                             \
        if constexpr (N == 0) asm volatile(LOAD32 P32(220:235) P32(236:251) MFMA32 TAIL STORE_OUT : OPS);                \
        else asm volatile(LOAD32 P32(220:235) P32(236:251) MFMA32 NOP_N TAIL STORE_OUT : OPS);                           \
    }
        if constexpr (MODE == WAR) { BODY32(WAR32) }
        else if constexpr (MODE == RAW) { BODY32(RAW32) }
        else if constexpr (MODE == RAWPK) { BODY32(RAWPK32) }
        else { BODY32(WAW32) }
    }
}

// ref == nullptr: write the captured registers to res (the reference run, N = 16 wait states);
// else count, per lane and captured register, the iterations whose capture differs from ref.
template <int KIND, int MODE, int N, int PRE>
__global__ __launch_bounds__(512) void probe(const u32x4* ops, const unsigned* ref, unsigned* res, int iters, int partner) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32x4 a = ops[lane], b = ops[64 + lane], c = ops[128 + lane];
    if (wave >= 4) {          // the SIMD partners of waves 0-3: a bare MFMA stream (or nothing)
        if (!partner) return;
        f32x4 acc[4] = {};
        for (int it = 0; it < iters * 6; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) res[0] = 1;      // (keeps the stream alive)
        return;
    }
    unsigned cnt[4] = {0, 0, 0, 0};
    const unsigned sen = 0x7fc01234u, sen2 = 0x42f60000u;      // a NaN; 123.0f
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < (it & 7); ++k) asm volatile("s_nop 3");     // a little jitter against the partner's stream
        unsigned o[4];
        probe_once<KIND, MODE, N, PRE>(a, b, c, sen, sen2, o);
        if (ref) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt[j] += o[j] != ref[(wave * 64 + lane) * 4 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt[j] = o[j];
        }
    }
    const int base = ((blockIdx.x * 4 + wave) * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) res[base + j] = cnt[j];
}

static const char* mode_name[] = {"WAR (VALU writes SrcC)", "RAW (v_mov reads D)", "RAW (v_pk_add_f32 reads D)", "WAW (VALU writes D)"};
static const char* kind_name[] = {"16x16x32", "32x32x16"};

struct Ctx { u32x4* ops; unsigned* ref; unsigned* res; int blocks; };

template <int KIND, int MODE, int N, int PRE>
void run_one(const Ctx& c, int partner, const std::vector<unsigned>& href, bool& any) {
    const int iters = 400;
    hipMemset(c.res, 0, c.blocks * 256 * 4 * 4);
    hipLaunchKernelGGL((probe<KIND, MODE, N, PRE>), dim3(c.blocks), dim3(512), 0, 0, c.ops, c.ref, c.res, iters, partner);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    std::vector<unsigned> h(c.blocks * 256 * 4);
    hipMemcpy(h.data(), c.res, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long per_reg[4] = {0, 0, 0, 0}, per_grp[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < h.size(); ++i) {
        per_reg[i & 3] += h[i];
        per_grp[((i >> 2) & 63) >> 4] += h[i];
    }
    const unsigned long tot = per_reg[0] + per_reg[1] + per_reg[2] + per_reg[3];
    any = tot != 0;
    printf("  N=%2d pre=%d partner=%d: %8lu wrong of %lu captures | by register (touched first..last) %lu %lu %lu %lu | by lane group 0-15 %lu, 16-31 %lu, 32-47 %lu, 48-63 %lu\n",
           N, PRE, partner, tot, (unsigned long)h.size() * iters, per_reg[3], per_reg[2], per_reg[1], per_reg[0], per_grp[0], per_grp[1], per_grp[2], per_grp[3]);
}

template <int KIND, int MODE, int PRE, int N>
void sweep(const Ctx& c, const std::vector<unsigned>& href) {
    for (int partner = 0; partner < 2; ++partner) {
        bool any;
        run_one<KIND, MODE, N, PRE>(c, partner, href, any);
    }
    if constexpr (N > 0) sweep<KIND, MODE, PRE, N - 1>(c, href);
}

template <int KIND, int MODE>
void all(const Ctx& c) {
    // reference: 16 wait states, nothing queued, no partner, one workgroup
    hipLaunchKernelGGL((probe<KIND, MODE, 16, 0>), dim3(1), dim3(512), 0, 0, c.ops, (const unsigned*)nullptr, c.ref, 1, 0);
    hipDeviceSynchronize();
    std::vector<unsigned> href(256 * 4);
    hipMemcpy(href.data(), c.ref, href.size() * 4, hipMemcpyDeviceToHost);
    printf("%s  %s   (reference capture lane 0: %08x %08x %08x %08x, lane 63: %08x %08x %08x %08x)\n", kind_name[KIND], mode_name[MODE],
           href[0], href[1], href[2], href[3], href[63 * 4], href[63 * 4 + 1], href[63 * 4 + 2], href[63 * 4 + 3]);
    constexpr int NMAX = KIND ? 14 : 10;
    sweep<KIND, MODE, 0, NMAX>(c, href);
    sweep<KIND, MODE, 2, NMAX>(c, href);
    if (KIND == 0) sweep<KIND, MODE, 6, NMAX>(c, href);
}

int main() {
    std::vector<unsigned> h(192 * 4);
    srand(3);
    auto bf = [](int v) { float f = (float)v; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (int i = 0; i < 128 * 4; ++i) h[i] = (unsigned)bf(rand() % 5 - 2) | ((unsigned)bf(rand() % 5 - 2) << 16);      // A, B: small integers
    for (int i = 128 * 4; i < 192 * 4; ++i) { float f = (float)(1000 + (i - 512)); memcpy(&h[i], &f, 4); }               // C: distinct integers
    Ctx c; c.blocks = 256;
    hipMalloc(&c.ops, h.size() * 4); hipMalloc(&c.ref, 256 * 4 * 4); hipMalloc(&c.res, c.blocks * 256 * 4 * 4);
    hipMemcpy(c.ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    all<0, WAR>(c); all<0, RAW>(c); all<0, RAWPK>(c); all<0, WAW>(c);
    all<1, WAR>(c); all<1, RAW>(c); all<1, RAWPK>(c); all<1, WAW>(c);
    return 0;
}
