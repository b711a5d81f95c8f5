// Device-side helpers shared by every kernel of the DEWI hot path (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dewi {

constexpr int kWave = 64;  // CDNA wavefront

// ---------------------------------------------------------------------------------------------
// Order-preserving fp32 -> u32 map.  Larger key == larger float.  NaN maps to the top, which is
// where NumPy's sort/partition put it (the reference's argpartition therefore ranks NaN rows
// first).  -0.0 and +0.0 map to the same key (NumPy compares them equal).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_f32(float f) {
  if (f != f) return 0xFFFFFFFFu;
  f = f + 0.0f;  // -0 -> +0
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}
// 64-bit candidate key: better candidate == larger key; ties on the score go to the LOWER row.
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
  return (static_cast<uint64_t>(ord_f32(score)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - row);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return 0xFFFFFFFFu - static_cast<uint32_t>(key); }
__device__ __forceinline__ float key_score(uint64_t key) { return unord_f32(static_cast<uint32_t>(key >> 32)); }
constexpr uint64_t kKeyEmpty = 0ull;             // below every real key (real keys have hi word >= 1)
constexpr uint64_t kKeyInactive = ~0ull;         // list position beyond the requested capacity

// ---------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP crossbar (no LDS traffic).  After the six steps lane 63 holds
// the reduction of all 64 lanes; the order of operations is fixed, so fp32 sums are bit-for-bit
// reproducible from launch to launch.
//   0xB1  quad_perm [1,0,3,2]      0x4E  quad_perm [2,3,0,1]
//   0x124 row_ror:4                0x128 row_ror:8
//   0x142 row_bcast:15 (rows 1,3)  0x143 row_bcast:31 (rows 2,3)
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#define DEWI_STEP(CTRL, MASK) \
  v = v + __int_as_float(dpp_i32<CTRL, MASK>(0, __float_as_int(v)));
  DEWI_STEP(0xB1, 0xF)
  DEWI_STEP(0x4E, 0xF)
  DEWI_STEP(0x124, 0xF)
  DEWI_STEP(0x128, 0xF)
  DEWI_STEP(0x142, 0xA)
  DEWI_STEP(0x143, 0xC)
#undef DEWI_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------
// ||q|| of the cosine query preparation (reference backends.py:420-424), defined so that EVERY kernel
// — whatever its lane layout and summation order — arrives at the same fp32 norm and therefore at
// the same prepared query: squares and their sum in float64 (a product of two fp32 values is exact
// there and the sum's rounding error, ~1e-13 relative, is far below fp32 resolution), then ONE
// rounding to fp32 after the square root.  `ss` is this lane's partial sum of squares; the xor
// butterfly leaves bit-identical totals in all 64 lanes (each step adds the same two values).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_query_norm(double ss) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, kWave);
  return static_cast<float>(__dsqrt_rn(ss));
}
// Sum of a float64 over the wave, bit-identical in all 64 lanes (xor butterfly as above).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ double square_f64(float v) { return static_cast<double>(v) * static_cast<double>(v); }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#define DEWI_STEP(CTRL, MASK) v = v + static_cast<uint32_t>(dpp_i32<CTRL, MASK>(0, static_cast<int>(v)));
  DEWI_STEP(0xB1, 0xF)
  DEWI_STEP(0x4E, 0xF)
  DEWI_STEP(0x124, 0xF)
  DEWI_STEP(0x128, 0xF)
  DEWI_STEP(0x142, 0xA)
  DEWI_STEP(0x143, 0xC)
#undef DEWI_STEP
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define DEWI_STEP(CTRL, MASK)                                                        \
  {                                                                                  \
    uint32_t o = static_cast<uint32_t>(dpp_i32<CTRL, MASK>(-1, static_cast<int>(v))); \
    v = o < v ? o : v;                                                               \
  }
  DEWI_STEP(0xB1, 0xF)
  DEWI_STEP(0x4E, 0xF)
  DEWI_STEP(0x124, 0xF)
  DEWI_STEP(0x128, 0xF)
  DEWI_STEP(0x142, 0xA)
  DEWI_STEP(0x143, 0xC)
#undef DEWI_STEP
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define DEWI_STEP(CTRL, MASK)                                                       \
  {                                                                                 \
    uint32_t o = static_cast<uint32_t>(dpp_i32<CTRL, MASK>(0, static_cast<int>(v))); \
    v = o > v ? o : v;                                                              \
  }
  DEWI_STEP(0xB1, 0xF)
  DEWI_STEP(0x4E, 0xF)
  DEWI_STEP(0x124, 0xF)
  DEWI_STEP(0x128, 0xF)
  DEWI_STEP(0x142, 0xA)
  DEWI_STEP(0x143, 0xC)
#undef DEWI_STEP
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// Maximum of a 64-bit key over the wave (same value in every lane): high word first, then the
// low word among the lanes that hold the maximum high word.
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
  const uint32_t hi = static_cast<uint32_t>(v >> 32);
  const uint32_t mhi = wave_max_u32(hi);
  const uint32_t lo = hi == mhi ? static_cast<uint32_t>(v) : 0u;
  const uint32_t mlo = wave_max_u32(lo);
  return (static_cast<uint64_t>(mhi) << 32) | mlo;
}

// Minimum of a 64-bit key over the wave: minimum high word first, then the minimum low word
// among the lanes that hold it.  Returns the same value in every lane.
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
  const uint32_t hi = static_cast<uint32_t>(v >> 32);
  const uint32_t mhi = wave_min_u32(hi);
  const uint32_t lo = hi == mhi ? static_cast<uint32_t>(v) : 0xFFFFFFFFu;
  const uint32_t mlo = wave_min_u32(lo);
  return (static_cast<uint64_t>(mhi) << 32) | mlo;
}

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

}  // namespace dewi
