#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* out, float* ref) {
  int t = threadIdx.x;
  float acc = 0.5f;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[t]), __builtin_bit_cast(bf16x2, b[t]), acc, false);
  out[t] = acc;
  // dependent chain of 4 on one accumulator (what the scan kernel does) vs the same with fmaf
  float c4 = 0.f, r4 = 0.f;
  for (int i = 0; i < 4; ++i) {
    const uint32_t x = a[(t + i) & 63], y = b[(t + 2 * i) & 63];
    c4 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x), __builtin_bit_cast(bf16x2, y), c4, false);
    r4 = fmaf(__uint_as_float(x << 16), __uint_as_float(y << 16), r4);
    r4 = fmaf(__uint_as_float(x & 0xFFFF0000u), __uint_as_float(y & 0xFFFF0000u), r4);
  }
  out[64 + t] = c4;
  ref[64 + t] = r4;
  float lo_a = __uint_as_float(a[t] << 16), hi_a = __uint_as_float(a[t] & 0xFFFF0000u);
  float lo_b = __uint_as_float(b[t] << 16), hi_b = __uint_as_float(b[t] & 0xFFFF0000u);
  ref[t] = fmaf(hi_a, hi_b, fmaf(lo_a, lo_b, 0.5f));
}
int main() {
  uint32_t ha[64], hb[64]; float ho[128], hr[128];
  for (int i = 0; i < 64; ++i) { float x = 0.01f * (i + 1), y = -0.02f * (i - 20), z = 0.3f + i, w = 1.0f / (i + 1);
    uint32_t xb, yb, zb, wb; std::memcpy(&xb, &x, 4); std::memcpy(&yb, &y, 4); std::memcpy(&zb, &z, 4); std::memcpy(&wb, &w, 4);
    ha[i] = (xb >> 16) | (yb & 0xFFFF0000u); hb[i] = (zb >> 16) | (wb & 0xFFFF0000u); }
  uint32_t *a, *b; float *o, *r;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 512); hipMalloc(&r, 512);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, b, o, r);
  hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("%d dot2 %.9g ref %.9g\n", i, ho[i], hr[i]);
  double me = 0; for (int i = 0; i < 64; ++i) me = fmax(me, fabs(ho[i] - hr[i]) / fabs(hr[i]));
  printf("max rel err %.3g\n", me);
  double m4 = 0; for (int i = 64; i < 128; ++i) m4 = fmax(m4, fabs(ho[i] - hr[i]) / fmax(1e-6, fabs(hr[i])));
  for (int i = 64; i < 68; ++i) printf("chain %d dot2 %.9g ref %.9g\n", i - 64, ho[i], hr[i]);
  printf("chain max rel err %.3g\n", m4);
  return 0;
}
