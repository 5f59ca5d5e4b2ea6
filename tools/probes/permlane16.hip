// Probe: semantics of v_permlane16_swap_b32 on gfx950 (development aid).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[lane + 64] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, 512); k<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("vdst':"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]); printf("\n");
  printf("src' :"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]); printf("\n");
  return 0;
}
