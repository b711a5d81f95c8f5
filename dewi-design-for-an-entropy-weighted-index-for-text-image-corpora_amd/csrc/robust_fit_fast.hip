// Exact median / MAD of fp32 columns in TWO launches (one per phase) for gfx950 (MI355X).
//
// Replaces scorer.RobustStats.fit (reference src/dewi/scorer.py:18-26) on one device; the histogram
// kernels of robust_stats.hip remain the pieces of the fit over doc-id shards (and the path for
// columns too long for this one).  That path streams every column six times through fifteen launches
// (7 x 1M: 74-99 us, launch-gap bound).  Here a phase is ONE kernel:
//
//   1. every workgroup of a column derives the same BRACKET [lo, hi] of keys from a sample of 4096 values
//      (256 runs of 16 consecutive values: 256 cache lines): the sample's ranks m/2 -/+ 208 (6.5 sigma of a sample median's rank), widened to
//      22-bit key prefixes (LDS histogram passes) — a guess, verified below, never trusted;
//   2. ONE pass over the column: count the keys below lo, equal to lo, equal to hi, the NaNs, and
//      COLLECT the keys strictly inside the bracket (~10 % of the column) into 16 BUCKETS by the top
//      bits of their offset from lo — through LDS, then write-through (sc1) into the column's bucket
//      buffers;
//   3. the last workgroup of a column to arrive (ticket counter) checks that both middle ranks fall
//      inside [lo, hi], finds the bucket that holds them from the 16 bucket counts and selects exactly
//      among that bucket's ~6 K keys.  If the bracket missed, or a buffer overflowed — adversarial data
//      only — that workgroup selects over the whole column by itself: always exact, merely slower.
//
// Algorithmic bytes: one median pass + one MAD pass = 2 x n_signals x n x 4 (56 MB at
// C5) — which is now what is streamed.  Order statistics are order-independent, so the result is exactly
// NumPy's: even n -> fp32 (a+b)/2, MAD keys from the fp32 subtraction |x - med|, any NaN -> NaN, -0 == +0.
//
// Inter-workgroup hand-off (MI355X_MICROARCH.md, "inter-workgroup visibility"): every byte of the bucket
// buffers is stored sc1 (write-through, relaxed agent-scope atomic stores), every storing wave drains its
// stores (vmcnt(0)) before the workgroup barrier behind which ONE lane takes the ticket with an agent-scope
// atomic add; the workgroup whose add came last reads the buffers with sc1 loads only.  No L2 write-back
// or L1 invalidate is needed then (a release + acquire fence pair cost ~15 us per launch here).
#include "common.hpp"
#include "launch.hpp"

namespace dewi {

namespace fastfit {

#ifndef DEWI_FIT_THREADS
#define DEWI_FIT_THREADS 1024
#endif
constexpr int kT = DEWI_FIT_THREADS;   // threads per workgroup (256-thread workgroups, four per CU, measured slower: 27 us tails)
constexpr int kSample = 4096;          // sample keys per column (4 per thread; 2 * kSample == 1 << 13)
constexpr uint32_t kDelta = 208;       // bracket half-width in sample ranks: 6.5 sigma (sigma = sqrt(m)/2 = 32)
constexpr int kBins = 2048;
constexpr int kBuckets = 16;           // buckets of the bracket (top 4 bits of a key's offset from lo)
constexpr int kBucketLds = 8192 / kBuckets / (1024 / kT) ;   // keys per bucket a workgroup stages in LDS (expected ~1/3 of it at 1M rows)
constexpr int kBatch = 8;              // independent loads a thread keeps in flight
constexpr int kRegKeys = 8192 / kT;    // keys per thread of the tail's register-resident bucket (8192 keys in all)
constexpr int kPer = 2048 / kT;        // histogram bins a thread owns in a pick
constexpr int kCopies = kT >= 1024 ? 4 : 2;   // copies 0 and 1 double as the two refinement histograms
constexpr int64_t kSmallN = 64 * 1024; // at or below: one workgroup per column selects exactly, no bracket

struct Counters {                      // per (phase, column); zeroed by the launcher
  uint32_t lt, eqlo, eqhi, nan, overflow, ticket, fallback, pad;
  uint32_t bucket[kBuckets];           // keys collected per bucket
  uint32_t stamp[8];                   // diagnostics (DEWI_FIT_STAMPS): 10 ns ticks of the last workgroup's steps
};

typedef float f32x4q __attribute__((ext_vector_type(4)));

template <bool MAD>
__device__ __forceinline__ uint32_t key_of(float x, float m, uint32_t& is_nan) {
  if constexpr (MAD) x = __builtin_fabsf(__fsub_rn(x, m));     // scorer.py:23: np.abs(arr - med), fp32
  is_nan = (x != x) ? 1u : 0u;
  return ord_f32(x);
}

struct Lds {
  uint32_t keys[kSample];              // the sample
  uint32_t hist[kCopies][kBins];       // histogram (lane-selected copies in the contended pass);
                                       // later the 16 x 512 staging buffers of the buckets
  uint32_t wave_tot[kT / kWave];
  uint32_t res[4];                     // pick results
  uint32_t misc[8];
  uint32_t bcount[kBuckets];
  uint32_t bbase[kBuckets];
  uint32_t red[4];
};

// Ascending picks over h[0..kBins): for each of NR ranks the bin holding it and the number of keys before that bin.
// Thread t owns the kPer consecutive bins starting at t * kPer.  Every thread returns the same values; h is intact.
template <int NR>
__device__ __forceinline__ void pick_asc_n(Lds& sh, const uint32_t* h, const uint32_t (&rank)[NR], uint32_t (&bin)[NR],
                                           uint32_t (&before)[NR]) {
  const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
  uint32_t v[kPer], local = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    v[j] = h[tid * kPer + j];
    local += v[j];
  }
  uint32_t incl = local;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += o;
  }
  if (lane == 63) sh.wave_tot[wave] = incl;
  __syncthreads();
  uint32_t b = incl - local;
  b += wave_sum_u32(lane < wave ? sh.wave_tot[lane] : 0u);    // lane w of every wave reads wave w's total
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      if (rank[q] >= b && rank[q] < b + v[j]) {
        sh.res[2 * q] = static_cast<uint32_t>(tid * kPer + j);
        sh.res[2 * q + 1] = b;
      }
    }
    b += v[j];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    bin[q] = sh.res[2 * q];
    before[q] = sh.res[2 * q + 1];
  }
  __syncthreads();
}
__device__ __forceinline__ void pick_asc(Lds& sh, const uint32_t* h, uint32_t rank, uint32_t& bin, uint32_t& before) {
  const uint32_t r[1] = {rank};
  uint32_t bn[1], bf[1];
  pick_asc_n<1>(sh, h, r, bn, bf);
  bin = bn[0];
  before = bf[0];
}
__device__ __forceinline__ void pick_asc2(Lds& sh, const uint32_t* h, uint32_t rank_a, uint32_t rank_b, uint32_t& bin_a,
                                          uint32_t& before_a, uint32_t& bin_b, uint32_t& before_b) {
  const uint32_t r[2] = {rank_a, rank_b};
  uint32_t bn[2], bf[2];
  pick_asc_n<2>(sh, h, r, bn, bf);
  bin_a = bn[0];
  before_a = bf[0];
  bin_b = bn[1];
  before_b = bf[1];
}

// Exact ascending order statistic among `count` keys keyfn(i) < 2^total_bits: the key at 0-based `rank`, the
// number of keys below it and equal to it.  MSB-first radix select, 11 bits per pass (32 bits: 11 + 11 + 10), one
// workgroup.  Callers that know their keys span a narrow range pass them as OFFSETS from its lower end with
// total_bits = the width of the range: the histogram of a pass then spreads over its bins instead of piling
// every key into one or two of them (64 lanes of an LDS atomic on one address run one after the other).
template <class KeyFn>
__device__ uint32_t block_select_asc(Lds& sh, KeyFn keyfn, int64_t count, uint32_t rank, int total_bits, uint32_t& below,
                                     uint32_t& eq) {
  const int tid = static_cast<int>(threadIdx.x);
  uint32_t* hist = sh.hist[0];
  uint32_t prefix = 0;
  below = 0;
  eq = 0;
  int remaining = total_bits;
  bool first = true;
  while (remaining > 0) {
    const int bits = remaining >= 22 ? 11 : (remaining > 11 ? remaining - 11 : remaining);   // 32 -> 11, 11, 10
    const int shift = remaining - bits;
    for (int b = tid; b < kBins; b += kT) hist[b] = 0;
    __syncthreads();
    // eight independent loads in flight per thread before the first atomic
    for (int64_t base = 0; base < count; base += kBatch * kT) {
      uint32_t key[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int64_t i = base + u * kT + tid;
        key[u] = i < count ? keyfn(i) : 0u;
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int64_t i = base + u * kT + tid;
        if (i < count && (first || (key[u] >> (shift + bits)) == prefix))
          atomicAdd(&hist[(key[u] >> shift) & ((1u << bits) - 1u)], 1u);
      }
    }
    __syncthreads();
    uint32_t bin, before;
    pick_asc(sh, hist, rank, bin, before);
    if (shift == 0) eq = hist[bin];
    __syncthreads();
    prefix = (prefix << bits) | bin;
    below += before;
    rank -= before;
    remaining = shift;
    first = false;
  }
  return prefix;
}

// Smallest key above `key0` among `count` keys (0xFFFFFFFF if none); with `any` set: the smallest key.
template <class KeyFn>
__device__ uint32_t block_min_above(Lds& sh, KeyFn keyfn, int64_t count, uint32_t key0, bool any = false) {
  const int tid = static_cast<int>(threadIdx.x);
  if (tid == 0) sh.misc[2] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t best = 0xFFFFFFFFu;
  for (int64_t base = 0; base < count; base += kBatch * kT) {
    uint32_t key[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int64_t i = base + u * kT + tid;
      key[u] = i < count ? keyfn(i) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      if ((any || key[u] > key0) && key[u] < best) best = key[u];
  }
  best = wave_min_u32(best);
  if ((tid & 63) == 0) atomicMin(&sh.misc[2], best);
  __syncthreads();
  const uint32_t r = sh.misc[2];
  __syncthreads();
  return r;
}

// The same two primitives over keys that already sit in registers: thread t holds key[u] = the (u * kT + t)-th of
// `count` <= kRegKeys * kT = 8192 keys (the tail's bucket: loaded once, then two or three passes without memory traffic).
__device__ uint32_t block_select_regs(Lds& sh, const uint32_t (&key)[kRegKeys], uint32_t count, uint32_t rank, int total_bits,
                                      uint32_t& below, uint32_t& eq) {
  const int tid = static_cast<int>(threadIdx.x);
  uint32_t* hist = sh.hist[0];
  uint32_t prefix = 0;
  below = 0;
  eq = 0;
  int remaining = total_bits;
  bool first = true;
  while (remaining > 0) {
    const int bits = remaining >= 22 ? 11 : (remaining > 11 ? remaining - 11 : remaining);
    const int shift = remaining - bits;
    for (int b = tid; b < kBins; b += kT) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRegKeys; ++u) {
      if (static_cast<uint32_t>(u * kT + tid) < count && (first || (key[u] >> (shift + bits)) == prefix))
        atomicAdd(&hist[(key[u] >> shift) & ((1u << bits) - 1u)], 1u);
    }
    __syncthreads();
    uint32_t bin, before;
    pick_asc(sh, hist, rank, bin, before);
    if (shift == 0) eq = hist[bin];
    __syncthreads();
    prefix = (prefix << bits) | bin;
    below += before;
    rank -= before;
    remaining = shift;
    first = false;
  }
  return prefix;
}

__device__ uint32_t block_min_above_regs(Lds& sh, const uint32_t (&key)[kRegKeys], uint32_t count, uint32_t key0, bool any) {
  const int tid = static_cast<int>(threadIdx.x);
  if (tid == 0) sh.misc[2] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t best = 0xFFFFFFFFu;
#pragma unroll
  for (int u = 0; u < kRegKeys; ++u)
    if (static_cast<uint32_t>(u * kT + tid) < count && (any || key[u] > key0) && key[u] < best) best = key[u];
  best = wave_min_u32(best);
  if ((tid & 63) == 0) atomicMin(&sh.misc[2], best);
  __syncthreads();
  const uint32_t r = sh.misc[2];
  __syncthreads();
  return r;
}

// The two middle keys of `count` keys (lower middle (count-1)/2, upper middle count/2).
template <class KeyFn>
__device__ void block_middles(Lds& sh, KeyFn keyfn, int64_t count, uint32_t& k0, uint32_t& k1) {
  uint32_t below, eq;
  const uint32_t r0 = static_cast<uint32_t>((count - 1) / 2), r1 = static_cast<uint32_t>(count / 2);
  k0 = block_select_asc(sh, keyfn, count, r0, 32, below, eq);
  k1 = (r1 == r0 || r1 < below + eq) ? k0 : block_min_above(sh, keyfn, count, k0);
}

__device__ __forceinline__ float finish_value(uint32_t k0, uint32_t k1, int64_t n, uint32_t nans) {
  const float a = unord_f32(k0), b = unord_f32(k1);
  float r = (n & 1) ? a : __fmul_rn(__fadd_rn(a, b), 0.5f);   // NumPy: even n -> fp32 mean of the two middles
  if (nans) r = __builtin_nanf("");                            // any NaN in the column -> NaN
  return r;
}

// grid (slices, n_signals); slices == 1 whenever n <= kSmallN.  bucket_cap: keys a column's bucket buffer holds.
template <bool MAD>
__global__ __launch_bounds__(kT) void fit_fast_kernel(const float* __restrict__ S, int64_t n, int64_t ld,
                                                      const float* __restrict__ med, Counters* __restrict__ ctr,
                                                      uint32_t* compact, int64_t bucket_cap, float* __restrict__ out) {
  __shared__ Lds sh;
  const int tid = static_cast<int>(threadIdx.x);
  const int s = static_cast<int>(blockIdx.y);
  const float* col = S + static_cast<int64_t>(s) * ld;
  const float m = MAD ? med[s] : 0.f;
  auto col_key = [&](int64_t i) {
    uint32_t nanflag;
    return key_of<MAD>(col[i], m, nanflag);
  };
  auto whole_column = [&]() {            // exact, by this workgroup alone: small columns, and the fallback
    uint32_t nans = 0;
    for (int64_t base = 0; base < n; base += kBatch * kT) {
      float x[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int64_t i = base + u * kT + tid;
        x[u] = i < n ? col[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        uint32_t f;
        key_of<MAD>(x[u], m, f);
        nans += f;
      }
    }
    if (tid == 0) sh.misc[3] = 0;
    __syncthreads();
    if (nans) atomicAdd(&sh.misc[3], nans);
    __syncthreads();
    const uint32_t total_nans = sh.misc[3];
    __syncthreads();
    uint32_t k0, k1;
    block_middles(sh, col_key, n, k0, k1);
    if (tid == 0) out[s] = finish_value(k0, k1, n, total_nans);
  };
  if (n <= kSmallN) {
    whole_column();
    return;
  }
#ifdef DEWI_FIT_STAMPS
  const uint64_t st0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- 1. bracket from a sample of kSample values taken as 256 RUNS of 16 consecutive values (one 64-byte line each,
  // run r around the middle of the r-th of 256 equal parts); every workgroup of the column computes the same one.
  // Round 3 took 4096 single values n / 4096 apart: 4096 lines of 64 B per workgroup and column for 16 KB of values —
  // the memory side moved 41 MB per launch for 28 MB of columns (PMC; profiles/hbm_traffic.json reported 55 MB because it
  // doubles FETCH_SIZE, which is right for 16-byte streaming loads and wrong for these dword gathers).  Runs cost 256 lines.
  // The bracket is a guess either way — verified below, never trusted: data arranged against the runs (or locally
  // correlated enough to fool them) takes the exact whole-column select.  The sample's loads go out FIRST (loads return
  // in issue order: behind the slice's loads below the bracket would wait for the whole slice).
  uint32_t skey[kSample / kT];
#pragma unroll
  for (int j = 0; j < kSample / kT; ++j) {
    const int64_t i = tid + j * kT;
    const int64_t centre = ((2 * (i >> 4) + 1) * n) >> 9;          // / (2 * 256 runs)
    const int64_t pos = (centre & ~static_cast<int64_t>(15)) + (i & 15);
    skey[j] = col_key(pos < n ? pos : n - 1);
  }
  // this workgroup's slice: head (to 16-byte alignment), 16-byte body split over the slices, tail — every element
  // exactly once.  The first batch of the slice's loads is issued NOW: the data does not depend on the bracket,
  // and the bracket's LDS passes below cover the HBM round trip.
  const int64_t slices = gridDim.x, slice = blockIdx.x;
  const int64_t mis = (reinterpret_cast<uintptr_t>(col) & 15) / 4;
  int64_t head = mis ? 4 - mis : 0;
  head = head < n ? head : n;
  const int64_t n4 = (n - head) / 4;
  const f32x4q* body = reinterpret_cast<const f32x4q*>(col + head);
  const int64_t b0 = n4 * slice / slices, b1 = n4 * (slice + 1) / slices;
  f32x4q pre[kBatch];
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const int64_t i = b0 + u * kT + tid;
    if (i < b1) pre[u] = __builtin_nontemporal_load(body + i);
  }

#pragma unroll
  for (int j = 0; j < kSample / kT; ++j) sh.keys[tid + j * kT] = skey[j];
  for (int b = tid; b < kCopies * kBins; b += kT) (&sh.hist[0][0])[b] = 0;
  __syncthreads();
  // top 11 bits: real signals crowd into a few dozen of these bins, so several copies selected by lane
#pragma unroll
  for (int j = 0; j < kSample / kT; ++j) atomicAdd(&sh.hist[tid % kCopies][sh.keys[tid + j * kT] >> 21], 1u);
  __syncthreads();
  for (int b = tid; b < kBins; b += kT) {
    uint32_t t = sh.hist[0][b];
#pragma unroll
    for (int cp = 1; cp < kCopies; ++cp) t += sh.hist[cp][b];
    sh.hist[0][b] = t;
  }
  __syncthreads();
  uint32_t bin_lo, before_lo, bin_hi, before_hi;
  pick_asc2(sh, sh.hist[0], kSample / 2 - kDelta, kSample / 2 + kDelta, bin_lo, before_lo, bin_hi, before_hi);
  // next 11 bits of the two sample keys, both refinements in one pass (copies 0 and 1 of the histogram)
  for (int b = tid; b < 2 * kBins; b += kT) (&sh.hist[0][0])[b] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSample / kT; ++j) {
    const uint32_t key = sh.keys[tid + j * kT];
    if ((key >> 21) == bin_lo) atomicAdd(&sh.hist[0][(key >> 10) & 0x7FFu], 1u);
    if ((key >> 21) == bin_hi) atomicAdd(&sh.hist[1][(key >> 10) & 0x7FFu], 1u);
  }
  __syncthreads();
  uint32_t sub_lo, sub_hi, unused;
  pick_asc(sh, sh.hist[0], kSample / 2 - kDelta - before_lo, sub_lo, unused);
  pick_asc(sh, sh.hist[1], kSample / 2 + kDelta - before_hi, sub_hi, unused);
  const uint32_t lo = (bin_lo << 21) | (sub_lo << 10);              // lower edge of the 22-bit bin
  const uint32_t hi = (bin_hi << 21) | (sub_hi << 10) | 0x3FFu;     // upper edge
  // keys strictly inside (lo, hi) have offsets key - lo - 1 in [0, span - 1); bucket = the top 4 of the span's bits
  const uint32_t span = hi - lo;
  const int span_bits = span > 1 ? 32 - __builtin_clz(span - 1) : 1;
  const int bshift = span_bits > 4 ? span_bits - 4 : 0;
  if (tid < kBuckets) sh.bcount[tid] = 0;
  __syncthreads();                                                   // the histogram copies are free: bucket staging
  uint32_t* const stage = &sh.hist[0][0];                            // [kBuckets][kBucketLds]
#ifdef DEWI_FIT_STAMPS
  const uint64_t st1 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- 2. one pass over this workgroup's slice
  uint32_t c_lt = 0, c_eqlo = 0, c_eqhi = 0, c_nan = 0;
  auto take = [&](float x) {
    uint32_t f;
    const uint32_t key = key_of<MAD>(x, m, f);
    c_nan += f;
    if (key < lo) {
      ++c_lt;
    } else if (key <= hi) {
      if (key == lo) {
        ++c_eqlo;
      } else if (key == hi) {
        ++c_eqhi;
      } else {
        const uint32_t b = (key - lo - 1u) >> bshift;
        const uint32_t p = atomicAdd(&sh.bcount[b], 1u);
        if (p < kBucketLds) stage[b * kBucketLds + p] = key;
      }
    }
  };
  if (slice == 0 && tid < head) take(col[tid]);
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {                            // the prefetched first batch
    const int64_t i = b0 + u * kT + tid;
    if (i < b1) {
      take(pre[u].x);
      take(pre[u].y);
      take(pre[u].z);
      take(pre[u].w);
    }
  }
  for (int64_t base = b0 + kBatch * kT; base < b1; base += kBatch * kT) {     // longer slices: batch by batch
    f32x4q v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int64_t i = base + u * kT + tid;
      if (i < b1) v[u] = __builtin_nontemporal_load(body + i);
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int64_t i = base + u * kT + tid;
      if (i < b1) {
        take(v[u].x);
        take(v[u].y);
        take(v[u].z);
        take(v[u].w);
      }
    }
  }
  const int64_t tail0 = head + 4 * n4;
  if (slice == slices - 1 && tail0 + tid < n) take(col[tail0 + tid]);
#ifdef DEWI_FIT_STAMPS
  const uint64_t st2 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- 3. publish: four counters per workgroup (reduced in LDS first), the staged keys into the column's buckets
  Counters* c = ctr + s;
  if (tid < 4) sh.red[tid] = 0;
  __syncthreads();
  c_lt = wave_sum_u32(c_lt);
  c_eqlo = wave_sum_u32(c_eqlo);
  c_eqhi = wave_sum_u32(c_eqhi);
  c_nan = wave_sum_u32(c_nan);
  if ((tid & 63) == 0) {
    if (c_lt) atomicAdd(&sh.red[0], c_lt);
    if (c_eqlo) atomicAdd(&sh.red[1], c_eqlo);
    if (c_eqhi) atomicAdd(&sh.red[2], c_eqhi);
    if (c_nan) atomicAdd(&sh.red[3], c_nan);
  }
  __syncthreads();
  if (tid < 4) {
    uint32_t* dst4 = tid == 0 ? &c->lt : (tid == 1 ? &c->eqlo : (tid == 2 ? &c->eqhi : &c->nan));
    if (sh.red[tid]) atomicAdd(dst4, sh.red[tid]);
  } else if (tid >= 64 && tid < 64 + kBuckets) {
    const int b = tid - 64;
    const uint32_t mine = sh.bcount[b];
    uint32_t base = 0;
    bool over = mine > kBucketLds;
    if (!over && mine) {
      base = atomicAdd(&c->bucket[b], mine);
      over = static_cast<int64_t>(base) + mine > bucket_cap;
    }
    if (over) atomicAdd(&c->overflow, 1u);
    sh.bbase[b] = over ? 0xFFFFFFFFu : base;
  }
  __syncthreads();
  uint32_t* const cbase = compact + static_cast<int64_t>(s) * kBuckets * bucket_cap;
  for (int b = 0; b < kBuckets; ++b) {
    const uint32_t base = sh.bbase[b];
    if (base == 0xFFFFFFFFu) continue;
    const uint32_t mine = sh.bcount[b];
    uint32_t* dstb = cbase + static_cast<int64_t>(b) * bucket_cap + base;
    for (uint32_t i = tid; i < mine; i += kT)   // write-through (sc1) store: see the hand-off note at the top
      __hip_atomic_store(dstb + i, stage[b * kBucketLds + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains its stores
  __syncthreads();
  if (tid == 0) sh.misc[5] = atomicAdd(&c->ticket, 1u);
  __syncthreads();
#ifdef DEWI_FIT_STAMPS
  const uint64_t st3 = __builtin_amdgcn_s_memrealtime();
#endif
  if (sh.misc[5] != static_cast<uint32_t>(slices - 1)) return;

  // ---- 4. last workgroup of the column (its ticket add returned last, behind the barrier above): verify, select.
  // Several of these workgroups share a CU, which the measured sc1-only hand-offs of the guide do not cover: one
  // lane also invalidates this CU's L1 (agent-scope acquire, ~1.7 us) before anybody reads.
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tid < 5) {
    const uint32_t* src = tid == 0 ? &c->lt : (tid == 1 ? &c->eqlo : (tid == 2 ? &c->eqhi : (tid == 3 ? &c->nan : &c->overflow)));
    sh.misc[tid] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (tid >= 64 && tid < 64 + kBuckets) {
    sh.bcount[tid - 64] = __hip_atomic_load(&c->bucket[tid - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const uint32_t t_lt = sh.misc[0], t_eqlo = sh.misc[1], t_eqhi = sh.misc[2], t_nan = sh.misc[3], t_over = sh.misc[4];
  uint32_t bc[kBuckets];
  uint64_t t_in = 0;
#pragma unroll
  for (int b = 0; b < kBuckets; ++b) {
    bc[b] = sh.bcount[b];
    t_in += bc[b];
  }
  __syncthreads();
  const uint64_t r0 = static_cast<uint64_t>((n - 1) / 2), r1 = static_cast<uint64_t>(n / 2);
  // ascending layout of the column's keys: [below lo | == lo | bucket 0 .. 15 (collected) | == hi | above hi]
  // (lo == hi: every key of the bracket was counted as "== lo"; the other groups are empty)
  const uint64_t a0 = t_lt, a1 = a0 + t_eqlo, a2 = a1 + t_in, a3 = a2 + t_eqhi;
  const bool bracket_ok = t_over == 0 && r0 >= a0 && r1 < a3;
  if (!bracket_ok) {
    if (tid == 0) c->fallback = 1u;                   // diagnostics: this column took the fallback
    whole_column();                                   // exact whatever the data: merely one workgroup's speed
    return;
  }
  // key at rank `rank_in` of bucket b; also how many keys of that bucket are below / equal to it (so that the
  // next rank can often be answered without another pass)
  uint32_t breg[kRegKeys];                          // the bucket of r0 as offsets from its smallest key, if it fits
  bool in_regs = false;
  auto bucket_key = [&](int b, uint32_t rank_in, uint32_t& below, uint32_t& eq) {
    const uint32_t* src = cbase + static_cast<int64_t>(b) * bucket_cap;
    const uint32_t origin = lo + 1u + (static_cast<uint32_t>(b) << bshift);      // smallest key of the bucket
    const int bits = bshift > 0 ? bshift : 1;
    if (bc[b] <= static_cast<uint32_t>(kRegKeys * kT)) {
#pragma unroll
      for (int u = 0; u < kRegKeys; ++u) {
        const uint32_t i = static_cast<uint32_t>(u * kT + tid);
        breg[u] = i < bc[b] ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - origin : 0u;
      }
      in_regs = true;
      return origin + block_select_regs(sh, breg, bc[b], rank_in, bits, below, eq);
    }
    auto keyfn = [&](int64_t i) { return __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - origin; };
    return origin + block_select_asc(sh, keyfn, bc[b], rank_in, bits, below, eq);
  };
  uint32_t k0 = 0, k1 = 0;
  if (r0 < a1) {
    k0 = lo;
  } else if (r0 >= a2) {
    k0 = hi;
  }
  int b0k = -1;
  uint32_t below0 = 0, eq0 = 0, rin0 = 0;
  if (r0 >= a1 && r0 < a2) {
    uint64_t acc = a1;
    for (int b = 0; b < kBuckets; ++b) {          // uniform: every thread walks the same counts
      if (r0 < acc + bc[b]) {
        b0k = b;
        rin0 = static_cast<uint32_t>(r0 - acc);
        break;
      }
      acc += bc[b];
    }
    k0 = bucket_key(b0k, rin0, below0, eq0);
  }
  if (r1 == r0) {
    k1 = k0;
  } else if (r1 < a1) {
    k1 = lo;
  } else if (r1 >= a2) {
    k1 = hi;
  } else if (b0k >= 0 && rin0 + 1 < bc[b0k]) {      // the next rank is in the same bucket
    if (rin0 + 1 < below0 + eq0) {
      k1 = k0;
    } else if (in_regs) {
      const uint32_t origin = lo + 1u + (static_cast<uint32_t>(b0k) << bshift);
      k1 = origin + block_min_above_regs(sh, breg, bc[b0k], k0 - origin, false);
    } else {
      const uint32_t* src = cbase + static_cast<int64_t>(b0k) * bucket_cap;
      auto keyfn = [&](int64_t i) { return __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      k1 = block_min_above(sh, keyfn, bc[b0k], k0);
    }
  } else {                                          // first key of the next non-empty bucket (r1 < a2: there is one)
    int b = b0k + 1;                                // b0k == -1 (r0 was "== lo"): start at bucket 0
    while (b < kBuckets - 1 && bc[b] == 0) ++b;
    const uint32_t* src = cbase + static_cast<int64_t>(b) * bucket_cap;
    auto keyfn = [&](int64_t i) { return __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    k1 = block_min_above(sh, keyfn, bc[b], 0u, true);
  }
  if (tid == 0) out[s] = finish_value(k0, k1, n, t_nan);
#ifdef DEWI_FIT_STAMPS
  if (tid == 0) {
    const uint64_t st4 = __builtin_amdgcn_s_memrealtime();
    c->stamp[0] = static_cast<uint32_t>(st1 - st0);   // bracket
    c->stamp[1] = static_cast<uint32_t>(st2 - st1);   // stream
    c->stamp[2] = static_cast<uint32_t>(st3 - st2);   // publish + ticket
    c->stamp[3] = static_cast<uint32_t>(st4 - st3);   // tail
  }
#endif
}

}  // namespace fastfit

static size_t fast_counters_bytes(int n_signals) {
  return (sizeof(fastfit::Counters) * 2 * static_cast<size_t>(n_signals) + 255) / 256 * 256;
}

size_t robust_fit_fast_bytes(int n_signals) {
  return fast_counters_bytes(n_signals) + sizeof(uint32_t) * static_cast<size_t>(n_signals) * kFitFastCap + 256;
}

bool robust_fit_fast_supported(int64_t n, int n_signals) {
  // expected keys inside the bracket: 2 * kDelta / kSample = 10 % of the column, 1/16 of them per bucket; keep 2x
  // head-room in a bucket's share of the compact buffer
  const int64_t per_bucket = n * 2 * fastfit::kDelta / fastfit::kSample / fastfit::kBuckets;
  return n_signals >= 1 && n_signals <= 1024 && 2 * per_bucket <= kFitFastCap / fastfit::kBuckets;
}

hipError_t launch_robust_fit_fast(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                                  void* d_fast_ws, hipStream_t stream) {
  char* p = static_cast<char*>(d_fast_ws);
  fastfit::Counters* ctr = reinterpret_cast<fastfit::Counters*>(p);
  uint32_t* compact = reinterpret_cast<uint32_t*>(p + fast_counters_bytes(n_signals));
  hipError_t e = hipMemsetAsync(ctr, 0, sizeof(fastfit::Counters) * 2 * n_signals, stream);
  if (e != hipSuccess) return e;
  int64_t slices = 1;
  if (n > fastfit::kSmallN) {
    slices = (256 * (1024 / fastfit::kT)) / n_signals;    // 1024 threads per CU over all columns
    const int64_t by_size = n / 4096;                     // at least 4 K elements per workgroup
    if (slices > by_size) slices = by_size;
    if (slices < 1) slices = 1;
  }
  const int64_t bucket_cap = kFitFastCap / fastfit::kBuckets;
  const dim3 grid(static_cast<unsigned>(slices), static_cast<unsigned>(n_signals));
  hipLaunchKernelGGL(fastfit::fit_fast_kernel<false>, grid, dim3(fastfit::kT), 0, stream, d_S, n, ld,
                     static_cast<const float*>(nullptr), ctr, compact, bucket_cap, d_med);
  hipLaunchKernelGGL(fastfit::fit_fast_kernel<true>, grid, dim3(fastfit::kT), 0, stream, d_S, n, ld,
                     static_cast<const float*>(d_med), ctr + n_signals, compact, bucket_cap, d_mad);
  return hipGetLastError();
}

}  // namespace dewi
