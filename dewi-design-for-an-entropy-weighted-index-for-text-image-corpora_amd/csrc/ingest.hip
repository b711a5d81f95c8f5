// Bulk ingest kernels for gfx950: row normalisation, bf16 rounding, payload SoA.
//
// Replaces the per-row work of ExactIndex.add (reference src/dewi/backends.py:403-406:
// `emb = emb.astype(float32); emb = emb / np.linalg.norm(emb)`) and the per-candidate payload
// reads of ExactIndex.search (:450-458).  All three are one-pass, HBM-bound elementwise kernels
// (bytes = what they read + what they write).
#include "common.hpp"
#include "launch.hpp"

namespace dewi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wavefront per row; the row is read twice (second read hits L1/L2: a 768-float row is 3 KiB).
template <int VEC>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int64_t n_rows, int dim) {
  const int lane = lane_id();
  const int64_t gwave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t row = gwave; row < n_rows; row += n_waves) {
    const float* s = src + row * dim;
    float* d = dst + row * dim;
    // ||row|| as the query kernels compute ||q|| (common.hpp wave_query_norm): squares and their sum in float64, ONE rounding
    // to fp32 after the square root — within half an ulp of the true norm, where the reference's np.linalg.norm
    // (backends.py:399-401: an fp32 BLAS dot product in an unspecified order) is within ~1.5; the stored rows then agree with
    // the reference's to 1 ulp on the golden inputs (2 ulp bound; tests/test_hip_index_api.py compares the saved matrices)
    double ss = 0.0;
    if constexpr (VEC == 4) {
      const f32x4* sv = reinterpret_cast<const f32x4*>(s);
      for (int u = lane; u < dim / 4; u += kWave) {
        const f32x4 v = sv[u];
        ss += square_f64(v.x) + square_f64(v.y) + square_f64(v.z) + square_f64(v.w);
      }
    } else {
      for (int j = lane; j < dim; j += kWave) ss += square_f64(s[j]);
    }
    const float norm = wave_query_norm(ss);
    // no zero-norm guard, like the reference: 0/0 -> NaN
    if constexpr (VEC == 4) {
      const f32x4* sv = reinterpret_cast<const f32x4*>(s);
      f32x4* dv = reinterpret_cast<f32x4*>(d);
      for (int u = lane; u < dim / 4; u += kWave) {
        f32x4 v = sv[u];
        v.x = __fdiv_rn(v.x, norm);
        v.y = __fdiv_rn(v.y, norm);
        v.z = __fdiv_rn(v.z, norm);
        v.w = __fdiv_rn(v.w, norm);
        dv[u] = v;
      }
    } else {
      for (int j = lane; j < dim; j += kWave) d[j] = __fdiv_rn(s[j], norm);
    }
  }
}

hipError_t launch_normalize_rows(const float* d_src, float* d_dst, int64_t n_rows, int dim, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  const bool vec = dim % 4 == 0 && (reinterpret_cast<uintptr_t>(d_src) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(d_dst) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(normalize_rows_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_src,
                       d_dst, n_rows, dim);
  else
    hipLaunchKernelGGL(normalize_rows_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_src,
                       d_dst, n_rows, dim);
  return hipGetLastError();
}

// Row-wise cosine of two N x d matrices: the post-embedding arithmetic of CrossModalDependency
// (reference src/dewi/signals/cross_modal.py:69, 124-139: F.cosine_similarity of the text and image
// embeddings of the same document = the I_hat signal).  torch semantics: each vector is divided by
// max(||x||, eps) with eps = 1e-8.  One wavefront per row; HBM-bound, 2*n*d*4 bytes read.
template <int VEC>
__global__ __launch_bounds__(256) void row_cosine_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ out, int64_t n_rows, int dim, float eps) {
  const int lane = lane_id();
  const int64_t gwave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t row = gwave; row < n_rows; row += n_waves) {
    const float* a = A + row * dim;
    const float* b = B + row * dim;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    if constexpr (VEC == 4) {
      const f32x4* av = reinterpret_cast<const f32x4*>(a);
      const f32x4* bv = reinterpret_cast<const f32x4*>(b);
      for (int u = lane; u < dim / 4; u += kWave) {
        const f32x4 x = __builtin_nontemporal_load(av + u), y = __builtin_nontemporal_load(bv + u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ab = __builtin_fmaf(x[i], y[i], ab);
          aa = __builtin_fmaf(x[i], x[i], aa);
          bb = __builtin_fmaf(y[i], y[i], bb);
        }
      }
    } else {
      for (int j = lane; j < dim; j += kWave) {
        ab = __builtin_fmaf(a[j], b[j], ab);
        aa = __builtin_fmaf(a[j], a[j], aa);
        bb = __builtin_fmaf(b[j], b[j], bb);
      }
    }
    ab = wave_sum_f32(ab);
    aa = wave_sum_f32(aa);
    bb = wave_sum_f32(bb);
    if (lane == 0) {
      const float na = __builtin_fmaxf(__fsqrt_rn(aa), eps), nb = __builtin_fmaxf(__fsqrt_rn(bb), eps);
      out[row] = __fdiv_rn(ab, __fmul_rn(na, nb));
    }
  }
}

// dim == 512 fast path (CLIP embeddings, config C5): a row of each matrix is exactly two 1 KiB wave
// loads, so the R rows of a step are fetched with 4R independent non-temporal loads before anything
// is reduced.  Launch shape as for the corpus scan: one 8-wave workgroup per CU.  Measured at
// 1 M x 512 x 2 matrices: 0.60 ms = 6.8 TB/s with R = 2 (generic kernel: 0.69 ms; R = 1 or 4 and
// 2-16 workgroups per CU: 0.61-0.66 ms).
template <int R>
__global__ __launch_bounds__(512) void row_cosine_512_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             float* __restrict__ out, int64_t n_rows, float eps) {
  const int lane = lane_id();
  const int64_t gwave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int64_t n_groups = (n_rows + R - 1) / R;
  for (int64_t g = gwave; g < n_groups; g += n_waves) {
    f32x4 x[R][2], y[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t row = g * R + r;
      row = row < n_rows ? row : n_rows - 1;   // the tail group re-reads the last row; its result is not stored
      const f32x4* av = reinterpret_cast<const f32x4*>(A + row * 512);
      const f32x4* bv = reinterpret_cast<const f32x4*>(B + row * 512);
      x[r][0] = __builtin_nontemporal_load(av + lane);
      x[r][1] = __builtin_nontemporal_load(av + 64 + lane);
      y[r][0] = __builtin_nontemporal_load(bv + lane);
      y[r][1] = __builtin_nontemporal_load(bv + 64 + lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // same accumulation order as row_cosine_kernel<4>
          ab = __builtin_fmaf(x[r][u][i], y[r][u][i], ab);
          aa = __builtin_fmaf(x[r][u][i], x[r][u][i], aa);
          bb = __builtin_fmaf(y[r][u][i], y[r][u][i], bb);
        }
      }
      ab = wave_sum_f32(ab);
      aa = wave_sum_f32(aa);
      bb = wave_sum_f32(bb);
      const int64_t row = g * R + r;
      if (lane == 0 && row < n_rows) {
        const float na = __builtin_fmaxf(__fsqrt_rn(aa), eps), nb = __builtin_fmaxf(__fsqrt_rn(bb), eps);
        out[row] = __fdiv_rn(ab, __fmul_rn(na, nb));
      }
    }
  }
}

hipError_t launch_row_cosine(const float* d_a, const float* d_b, float* d_out, int64_t n_rows, int dim, float eps,
                             hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  const bool vec = dim % 4 == 0 && (reinterpret_cast<uintptr_t>(d_a) % 16 == 0) && (reinterpret_cast<uintptr_t>(d_b) % 16 == 0);
  if (vec && dim == 512) {
    static int compute_units = 0;
    if (compute_units == 0) {
      int dev = 0, cu = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
        cu = 256;
      compute_units = cu;
    }
    constexpr int R = 2;
    const int64_t groups = (n_rows + R - 1) / R;
    const int64_t want = (groups + 7) / 8;   // workgroups of 8 waves
    const int blocks = static_cast<int>(want < compute_units ? want : compute_units);
    hipLaunchKernelGGL(row_cosine_512_kernel<R>, dim3(blocks), dim3(512), 0, stream, d_a, d_b, d_out, n_rows, eps);
    return hipGetLastError();
  }
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (vec)
    hipLaunchKernelGGL(row_cosine_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_a, d_b, d_out,
                       n_rows, dim, eps);
  else
    hipLaunchKernelGGL(row_cosine_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_a, d_b, d_out,
                       n_rows, dim, eps);
  return hipGetLastError();
}

// fp32 -> bf16, round to nearest even; NaN stays NaN (quiet bit forced).
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  if (f != f) return static_cast<uint16_t>((u >> 16) | 0x0040u);
  return static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                          int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f32_to_bf16_rne(src[i]);
}

hipError_t launch_f32_to_bf16(const float* d_src, uint16_t* d_dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, d_src, d_dst, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void payload_soa_kernel(const double* __restrict__ dewi,
                                                          const double* __restrict__ ht,
                                                          const double* __restrict__ hi, float* __restrict__ dewi32,
                                                          float* __restrict__ ent32, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    dewi32[i] = static_cast<float>(dewi[i]);
    ent32[i] = static_cast<float>(__dmul_rn(__dadd_rn(ht[i], hi[i]), 0.5));
  }
}

hipError_t launch_payload_soa(const double* dewi, const double* ht, const double* hi, float* dewi32, float* ent32,
                              int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(payload_soa_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, dewi, ht, hi,
                     dewi32, ent32, n);
  return hipGetLastError();
}

}  // namespace dewi
