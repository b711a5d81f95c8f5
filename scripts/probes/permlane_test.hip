// What v_permlane32_swap_b32 does on gfx950: a = lane, b = 100 + lane; prints (a, b) per lane afterwards, for the
// builtin (the compiler places the wait states the instruction needs behind a vector write) and for bare inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(int* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r.x;
  out[64 + threadIdx.x] = r.y;
  int c = threadIdx.x, d = 100 + threadIdx.x;
  asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(c), "+v"(d));
  out[128 + threadIdx.x] = c;
  out[192 + threadIdx.x] = d;
}
int main() {
  int* d;
  if (hipMalloc(&d, 256 * 4) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int i = 0; i < 64; i += 8) printf("lane %2d: builtin a=%3d b=%3d   asm+nops a=%3d b=%3d\n", i, h[i], h[64 + i], h[128 + i], h[192 + i]);
  return 0;
}
