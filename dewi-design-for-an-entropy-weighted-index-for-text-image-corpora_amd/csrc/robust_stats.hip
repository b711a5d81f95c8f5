// Robust statistics (median / MAD) and the DEWI score for gfx950 (MI355X).
//
// Replaces scorer.RobustStats.fit (reference src/dewi/scorer.py:18-26) and
// RobustStats.z + DewiScorer._components/score/score_conditional (:28-31, 49-89).
//
// fit: an exact order statistic needs no sort.  Each fp32 value is mapped to an order-preserving
// 32-bit key and the two middle ranks are located by a 3-pass MSB-first radix select (11+11+10
// bits): a pass histograms the keys that still match the decided prefix (LDS histogram per
// workgroup, flushed to a global histogram with atomics), then a one-workgroup "pick" walks the
// global histogram to the bin holding the rank.  Every pass streams the column once, so fit reads
// 2 (median, MAD) x 3 x n_signals x n x 4 bytes — HBM/L2-bound and launch-latency dominated at
// n = 1M.  Selection is order-independent, so the result is exactly NumPy's.
//
// score: one elementwise float64 kernel, 7 loads + 1 store per document, written so that every
// intermediate rounds as in the reference (no FMA contraction: this file is built with
// -ffp-contract=off and uses explicit __dmul_rn/__dadd_rn).
#include "common.hpp"
#include "launch.hpp"

namespace dewi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBins = 2048;
constexpr int kFitThreads = 256;

struct FitState {       // per problem p = 2*signal + {0: lower middle, 1: upper middle}
  uint32_t prefix;      // key bits decided so far (right-aligned)
  uint32_t rank;        // 0-based ascending rank still to locate inside the prefix
};

struct FitWorkspace {
  // laid out by robust_fit_workspace_bytes(); all arrays are per phase (0 median, 1 MAD)
  uint32_t* hist;       // [2][3][P][kBins]
  FitState* state;      // [2][P]
  uint32_t* nan_count;  // [2][n_signals]
};

static FitWorkspace carve(void* ws, int n_signals) {
  const int P = 2 * n_signals;
  FitWorkspace w;
  char* p = static_cast<char*>(ws);
  w.hist = reinterpret_cast<uint32_t*>(p);
  p += sizeof(uint32_t) * 2 * 3 * P * kBins;
  w.state = reinterpret_cast<FitState*>(p);
  p += sizeof(FitState) * 2 * P;
  w.nan_count = reinterpret_cast<uint32_t*>(p);
  return w;
}

static size_t hist_path_bytes(int n_signals) {
  const size_t P = 2 * static_cast<size_t>(n_signals);
  const size_t b = sizeof(uint32_t) * 2 * 3 * P * kBins + sizeof(FitState) * 2 * P + sizeof(uint32_t) * 2 * n_signals + 256;
  return (b + 255) / 256 * 256;
}

size_t robust_fit_workspace_bytes(int n_signals) {
  // [histogram path (also the sharded fit's pieces) | two-launch path of robust_fit_fast.hip]
  return hist_path_bytes(n_signals) + robust_fit_fast_bytes(n_signals);
}

template <int PASS>
__device__ __forceinline__ bool digit_of(uint32_t key, uint32_t prefix, uint32_t& digit) {
  if constexpr (PASS == 0) {
    digit = key >> 21;
    return true;
  } else if constexpr (PASS == 1) {
    digit = (key >> 10) & 0x7FFu;
    return (key >> 21) == prefix;
  } else {
    digit = key & 0x3FFu;
    return (key >> 10) == prefix;
  }
}

// grid (blocks_per_signal, n_signals).  MAD == true: keys are ord(|x - med[s]|), fp32 subtraction.
// Launch shape: few fat workgroups (kHistThreads threads, about one workgroup per CU over all signals),
// 16-byte loads.  A workgroup pays for zeroing and flushing its 2 x 2048 LDS bins once, and in the
// second pass nearly every bin is non-zero: with 1 715 workgroups of 256 threads the flush alone was
// 7 M global atomics per pass (27 us); 224 workgroups of 1 024 threads flush 0.9 M.
constexpr int kHistThreads = 1024;

template <int PASS, bool MAD>
__global__ __launch_bounds__(kHistThreads) void fit_hist_kernel(const float* __restrict__ S, int64_t n, int64_t ld,
                                                                const float* __restrict__ med,
                                                                const FitState* __restrict__ state,
                                                                uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ nan_count) {
  // PASS 0: both problems of a signal (lower / upper middle rank) see the same keys, and real signals
  // crowd into a few dozen of the 2048 top-bit bins: ONE histogram, kept in four copies selected by
  // lane so that the lanes of a wave serialise on a bin four times less, written to both problems at
  // the flush.  Later passes: one histogram per problem (the prefixes may differ), bins well spread.
  __shared__ uint32_t lh[4][kBins];
  __shared__ uint32_t lnan;
  const int s = static_cast<int>(blockIdx.y);
  const int tid = static_cast<int>(threadIdx.x);
  for (int i = tid; i < 4 * kBins; i += kHistThreads) (&lh[0][0])[i] = 0;
  if (tid == 0) lnan = 0;
  __syncthreads();
  const float* col = S + static_cast<int64_t>(s) * ld;
  const float m = MAD ? med[s] : 0.f;
  uint32_t pre0 = 0, pre1 = 0;
  if constexpr (PASS > 0) {
    pre0 = state[2 * s].prefix;
    pre1 = state[2 * s + 1].prefix;
  }
  uint32_t nans = 0;
  auto take = [&](float x) {
    if constexpr (MAD) x = __builtin_fabsf(__fsub_rn(x, m));
    if constexpr (PASS == 0) nans += (x != x) ? 1u : 0u;
    const uint32_t key = ord_f32(x);
    uint32_t d;
    if constexpr (PASS == 0) {
      digit_of<0>(key, 0u, d);
      atomicAdd(&lh[tid & 3][d], 1u);
    } else {
      if (digit_of<PASS>(key, pre0, d)) atomicAdd(&lh[0][d], 1u);
      if (digit_of<PASS>(key, pre1, d)) atomicAdd(&lh[1][d], 1u);
    }
  };
  // head (to 16-byte alignment), 16-byte body, tail — every element exactly once
  const int64_t mis = (reinterpret_cast<uintptr_t>(col) & 15) / 4;
  int64_t head = mis ? 4 - mis : 0;
  head = head < n ? head : n;
  const int64_t n4 = (n - head) / 4;
  const int64_t gtid = static_cast<int64_t>(blockIdx.x) * kHistThreads + tid;
  const int64_t gstride = static_cast<int64_t>(gridDim.x) * kHistThreads;
  if (gtid < head) take(col[gtid]);
  const f32x4* body = reinterpret_cast<const f32x4*>(col + head);
  for (int64_t i = gtid; i < n4; i += gstride) {
    const f32x4 v = body[i];
    take(v.x);
    take(v.y);
    take(v.z);
    take(v.w);
  }
  const int64_t tail0 = head + 4 * n4;
  if (tail0 + gtid < n) take(col[tail0 + gtid]);   // fewer than 4 elements
  if constexpr (PASS == 0) {
    if (nans) atomicAdd(&lnan, nans);
  }
  __syncthreads();
  uint32_t* gh = hist + static_cast<int64_t>(2 * s) * kBins;
  if constexpr (PASS == 0) {
    for (int i = tid; i < kBins; i += kHistThreads) {
      const uint32_t v = lh[0][i] + lh[1][i] + lh[2][i] + lh[3][i];
      if (v) {
        atomicAdd(&gh[i], v);
        atomicAdd(&gh[kBins + i], v);
      }
    }
  } else {
    for (int i = tid; i < 2 * kBins; i += kHistThreads) {
      const uint32_t v = (&lh[0][0])[i];
      if (v) atomicAdd(&gh[i], v);
    }
  }
  if constexpr (PASS == 0) {
    if (tid == 0 && lnan) atomicAdd(&nan_count[s], lnan);
  }
}

// One workgroup per problem: find the bin holding `rank`, extend the prefix, rebase the rank.
template <int PASS>
__global__ __launch_bounds__(kFitThreads) void fit_pick_kernel(const uint32_t* __restrict__ hist,
                                                               FitState* __restrict__ state, int64_t n) {
  constexpr int BITS = PASS == 2 ? 10 : 11;
  constexpr int PER = kBins / kFitThreads;  // 8 bins per thread
  __shared__ uint32_t wave_tot[kFitThreads / kWave];
  const int p = static_cast<int>(blockIdx.x);
  const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
  const uint32_t* h = hist + static_cast<int64_t>(p) * kBins;
  uint32_t rank;
  uint32_t prefix = 0;
  if constexpr (PASS == 0) {
    // lower middle (n-1)/2, upper middle n/2: identical when n is odd
    rank = static_cast<uint32_t>((p & 1) ? (n / 2) : ((n - 1) / 2));
  } else {
    rank = state[p].rank;
    prefix = state[p].prefix;
  }
  uint32_t v[PER], local = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    v[j] = h[tid * PER + j];
    local += v[j];
  }
  uint32_t incl = local;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t before = incl - local;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (rank >= before && rank < before + v[j]) {
      FitState st;
      st.prefix = (prefix << BITS) | static_cast<uint32_t>(tid * PER + j);
      st.rank = rank - before;
      state[p] = st;
    }
    before += v[j];
  }
}

// med[s] / mad[s] from the two located keys.  NumPy: even n -> mean of the two middles computed in
// fp32 ((a+b)/2); any NaN in the column -> NaN.
__global__ void fit_finish_kernel(const FitState* __restrict__ state, const uint32_t* __restrict__ nan_count,
                                  int n_signals, int64_t n, float* __restrict__ out) {
  const int s = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n_signals) return;
  const float a = unord_f32(state[2 * s].prefix);
  const float b = unord_f32(state[2 * s + 1].prefix);
  float r = __fmul_rn(__fadd_rn(a, b), 0.5f);
  if (n & 1) r = a;
  if (nan_count[s]) r = __builtin_nanf("");
  out[s] = r;
}

// ---- the select split at its histogram boundaries (also the pieces of the sharded fit: each rank
// histograms its own rows, the caller sums the histogram region over ranks, every rank picks with the
// global row count).
static uint32_t* hist_of(const FitWorkspace& w, int n_signals, int phase, int pass) {
  const int64_t P = 2 * n_signals;
  return w.hist + (static_cast<int64_t>(phase) * 3 + pass) * P * kBins;
}

void robust_fit_region(int n_signals, int phase, int pass, int which, size_t* offset_bytes, size_t* count_u32) {
  const size_t P = 2 * static_cast<size_t>(n_signals);
  if (which == 0) {
    *offset_bytes = sizeof(uint32_t) * (static_cast<size_t>(phase) * 3 + pass) * P * kBins;
    *count_u32 = P * kBins;
  } else {
    *offset_bytes = sizeof(uint32_t) * 2 * 3 * P * kBins + sizeof(FitState) * 2 * P +
                    sizeof(uint32_t) * static_cast<size_t>(phase) * n_signals;
    *count_u32 = static_cast<size_t>(n_signals);
  }
}

hipError_t launch_fit_begin(void* d_ws, int n_signals, hipStream_t stream) {
  return hipMemsetAsync(d_ws, 0, hist_path_bytes(n_signals), stream);
}

hipError_t launch_fit_hist(const float* S, int64_t n, int64_t ld, int n_signals, int phase, int pass, const float* med,
                           void* d_ws, hipStream_t stream) {
  FitWorkspace w = carve(d_ws, n_signals);
  const int P = 2 * n_signals;
  uint32_t* hist = hist_of(w, n_signals, phase, pass);
  FitState* state = w.state + phase * P;
  uint32_t* nanc = w.nan_count + phase * n_signals;
  int64_t bx = (n + kHistThreads * 16 - 1) / (kHistThreads * 16);
  if (bx < 1) bx = 1;   // an empty shard still launches (and contributes an all-zero histogram)
  const int64_t per_signal = (256 + n_signals - 1) / n_signals;   // ~one workgroup per CU over all signals
  if (bx > per_signal) bx = per_signal;
  const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(n_signals));
#define DEWI_HIST(PASS, MAD) \
  hipLaunchKernelGGL((fit_hist_kernel<PASS, MAD>), grid, dim3(kHistThreads), 0, stream, S, n, ld, med, state, hist, nanc)
  if (phase == 0) {
    if (pass == 0) DEWI_HIST(0, false); else if (pass == 1) DEWI_HIST(1, false); else DEWI_HIST(2, false);
  } else {
    if (pass == 0) DEWI_HIST(0, true); else if (pass == 1) DEWI_HIST(1, true); else DEWI_HIST(2, true);
  }
#undef DEWI_HIST
  return hipGetLastError();
}

hipError_t launch_fit_pick(int64_t n_total, int n_signals, int phase, int pass, void* d_ws, hipStream_t stream) {
  FitWorkspace w = carve(d_ws, n_signals);
  const int P = 2 * n_signals;
  const uint32_t* hist = hist_of(w, n_signals, phase, pass);
  FitState* state = w.state + phase * P;
  if (pass == 0)
    hipLaunchKernelGGL((fit_pick_kernel<0>), dim3(P), dim3(kFitThreads), 0, stream, hist, state, n_total);
  else if (pass == 1)
    hipLaunchKernelGGL((fit_pick_kernel<1>), dim3(P), dim3(kFitThreads), 0, stream, hist, state, n_total);
  else
    hipLaunchKernelGGL((fit_pick_kernel<2>), dim3(P), dim3(kFitThreads), 0, stream, hist, state, n_total);
  return hipGetLastError();
}

hipError_t launch_fit_finish(int64_t n_total, int n_signals, int phase, void* d_ws, float* d_out, hipStream_t stream) {
  FitWorkspace w = carve(d_ws, n_signals);
  hipLaunchKernelGGL(fit_finish_kernel, dim3((n_signals + 63) / 64), dim3(64), 0, stream, w.state + phase * 2 * n_signals,
                     w.nan_count + phase * n_signals, n_signals, n_total, d_out);
  return hipGetLastError();
}

hipError_t launch_robust_fit(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                             void* d_ws, hipStream_t stream) {
  if (robust_fit_fast_supported(n, n_signals))   // one device, columns the compact buffers hold: two launches
    return launch_robust_fit_fast(d_S, n, ld, n_signals, d_med, d_mad, static_cast<char*>(d_ws) + hist_path_bytes(n_signals),
                                  stream);
  hipError_t e = launch_fit_begin(d_ws, n_signals, stream);
  if (e != hipSuccess) return e;
  for (int phase = 0; phase < 2; ++phase) {
    for (int pass = 0; pass < 3; ++pass) {
      if ((e = launch_fit_hist(d_S, n, ld, n_signals, phase, pass, d_med, d_ws, stream)) != hipSuccess) return e;
      if ((e = launch_fit_pick(n, n_signals, phase, pass, d_ws, stream)) != hipSuccess) return e;
    }
    if ((e = launch_fit_finish(n, n_signals, phase, d_ws, phase == 0 ? d_med : d_mad, stream)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// score
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void score_kernel(const T* __restrict__ S, int64_t n, int64_t ld, ScoreParams sp,
                                                    const float* __restrict__ d_med, const float* __restrict__ d_mad,
                                                    double* __restrict__ out, float* __restrict__ out32) {
  if (d_med != nullptr) {
    // statistics straight from the fit kernels' device output (dewi_score_f64_dev): what the host layer does with
    // them between fit and score — widen to float64, `mad or 1e-8` (scorer.py:24; -0.0 is falsy too, NaN is not),
    // 1.4826 * mad rounded once (scorer.py:31) — done here, so nothing has to visit the host in between
#pragma unroll
    for (int s = 0; s < DEWI_NUM_SIGNALS; ++s) {
      const float m = d_mad[s];
      sp.med[s] = static_cast<double>(d_med[s]);
      sp.scale[s] = __dmul_rn(1.4826, m == 0.f ? 1e-8 : static_cast<double>(m));
    }
  }
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    double z[DEWI_NUM_SIGNALS];
#pragma unroll
    for (int s = 0; s < DEWI_NUM_SIGNALS; ++s) {
      const double x = static_cast<double>(S[static_cast<int64_t>(s) * ld + i]);
      z[s] = __ddiv_rn(__dsub_rn(x, sp.med[s]), sp.scale[s]);            // scorer.py:28-31
    }
    const double Ht = __dmul_rn(0.5, __dadd_rn(z[0], z[1]));               // scorer.py:53
    const double Hi = __dmul_rn(0.5, __dadd_rn(z[2], z[3]));               // scorer.py:54
    const double I = z[4], R = z[5], Nz = z[6];
    double U;
    if (sp.mode == DEWI_MODE_STANDARD) {                                   // scorer.py:67-73
      U = __dadd_rn(__dmul_rn(sp.w[0], Ht), __dmul_rn(sp.w[1], Hi));
      U = __dsub_rn(U, __dmul_rn(sp.w[2], I));
    } else {                                                               // scorer.py:80-87
      U = __dadd_rn(__dmul_rn(sp.w[0], __dsub_rn(Ht, I)), __dmul_rn(sp.w[1], __dsub_rn(Hi, I)));
    }
    U = __dsub_rn(U, __dmul_rn(sp.w[3], R));
    U = __dsub_rn(U, __dmul_rn(sp.w[4], Nz));
    // np.clip(U, -delta, delta): minimum(maximum(U, lo), hi); NaN propagates
    if (U == U) {
      U = U < -sp.delta ? -sp.delta : U;
      U = U > sp.delta ? sp.delta : U;
    }
    const double r = __ddiv_rn(1.0, __dadd_rn(1.0, exp(-U)));              // scorer.py:60-62
    if (out) out[i] = r;
    if (out32) out32[i] = static_cast<float>(r);
  }
}

hipError_t launch_score(const void* d_S, int is_f64, int64_t n, int64_t ld, const ScoreParams& sp, const float* d_med,
                        const float* d_mad, double* d_out, float* d_out32, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (is_f64)
    hipLaunchKernelGGL(score_kernel<double>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                       static_cast<const double*>(d_S), n, ld, sp, d_med, d_mad, d_out, d_out32);
  else
    hipLaunchKernelGGL(score_kernel<float>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                       static_cast<const float*>(d_S), n, ld, sp, d_med, d_mad, d_out, d_out32);
  return hipGetLastError();
}

}  // namespace dewi
