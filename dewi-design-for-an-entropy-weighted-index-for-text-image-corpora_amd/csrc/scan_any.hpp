// Row scan for ANY embedding width up to 1024 16-byte units per row (fp32: 4096 columns, bf16: 8192), gfx950.
//
// Same contract as knn_scan.hip (steps 1-3 of ExactIndex.search, reference src/dewi/backends.py:420-444 — one BLAS call
// for every dim there, `np.dot(self._embeddings, query)` :431-433): normalise the query, score every row, keep the best c
// keys per wave in registers.  knn_scan.hip / knn_scan_bf16.hip keep the widths they were tuned on (dim = 256 U);
// everything else used to fall to scan_generic_*, which re-read the queries from global memory inside the inner loop and
// kept one short row in flight per wave (0.24-0.44 of HBM at 3 GB, profiles/r04/dims_before).  Two kernels replace it:
//
//   scan_rows_any        rows of 33 .. 64 U units.  One row per wave step, lane l holds the units l + 64 u of the row and
//                        the matching query fragments IN REGISTERS; units past the end of the row are predicated off (EXEC
//                        mask: no bytes move for them, the registers read as zero).  R rows are issued back to back so that
//                        a wave keeps ~3-4 KiB in flight whatever the width.
//   scan_short_rows_any  rows of 1 .. 32 units: P = 2^log2p lanes share a row, a wave-instruction loads 64 / P CONSECUTIVE
//                        rows (one contiguous span), R such loads are in flight; the P partial sums are folded on the DPP
//                        crossbar (quad_perm, row_half_mirror, row_mirror, row_bcast15) and the lane that ends up with a row's
//                        score offers it to the wave's list — a ballot finds the few rows worth an offer.
//
// Each in two forms: rows that are whole units (fp32: dim % 4 == 0, bf16: dim % 8 == 0; PH = false) and rows that are not
// (PH = true, see unit_keep_mask below) — so every width up to 1024 units streams the corpus with aligned 16-byte loads and
// register-resident queries; scan_generic_* is left with rows wider than that.
//
// Roofline: HBM.  Algorithmic bytes per launch = n_rows * dim * sizeof(elem) + n_queries * dim * 4.
#pragma once
#include "scan_common.hpp"

namespace dewi {

// ---------------------------------------------------------------------------------------------
// The query values that face one 16-byte unit of a row
// ---------------------------------------------------------------------------------------------
template <int ELEM, int SPACE>
struct UnitFrag;

template <int SPACE>
struct UnitFrag<0, SPACE> {   // fp32 rows: 4 columns per unit
  f32x4 q;
  __device__ __forceinline__ void load(const float* __restrict__ qrow, int unit, bool active) {
    q = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) q = reinterpret_cast<const f32x4*>(qrow)[unit];
  }
  // rows that are not whole units: the unit's first column is `first` (negative before the row, >= dim behind it: zeros there)
  __device__ __forceinline__ void load_shifted(const float* __restrict__ qrow, int first, int dim) {
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = first + e;
      t[e] = static_cast<unsigned>(idx) < static_cast<unsigned>(dim) ? qrow[idx] : 0.f;
    }
    q = f32x4{t[0], t[1], t[2], t[3]};
  }
  __device__ __forceinline__ double sumsq() const { return square_f64(q.x) + square_f64(q.y) + square_f64(q.z) + square_f64(q.w); }
  __device__ __forceinline__ void scale(float norm) {
    q.x = __fdiv_rn(q.x, norm);
    q.y = __fdiv_rn(q.y, norm);
    q.z = __fdiv_rn(q.z, norm);
    q.w = __fdiv_rn(q.w, norm);
  }
  __device__ __forceinline__ void finish() {}
  __device__ __forceinline__ float dot(u32x4 e, float acc) const { return accum4<SPACE>(__builtin_bit_cast(f32x4, e), q, acc); }
};

template <int SPACE>
struct UnitFrag<1, SPACE> {   // bf16 rows: 8 columns per unit; the prepared query is rounded to bf16 (knn_scan_bf16.hip)
  float f[8];
  uint32_t p[4];              // cosine: packed pairs for v_dot2c_f32_bf16
  __device__ __forceinline__ void load(const float* __restrict__ qrow, int unit, bool active) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      a = reinterpret_cast<const f32x4*>(qrow)[2 * unit];
      b = reinterpret_cast<const f32x4*>(qrow)[2 * unit + 1];
    }
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  __device__ __forceinline__ void load_shifted(const float* __restrict__ qrow, int first, int dim) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = first + e;
      f[e] = static_cast<unsigned>(idx) < static_cast<unsigned>(dim) ? qrow[idx] : 0.f;
    }
  }
  __device__ __forceinline__ double sumsq() const {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += square_f64(f[i]);
    return s;
  }
  __device__ __forceinline__ void scale(float norm) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = __fdiv_rn(f[i], norm);
  }
  __device__ __forceinline__ void finish() {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = round_to_bf16(f[i]);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = (__float_as_uint(f[2 * i]) >> 16) | (__float_as_uint(f[2 * i + 1]) & 0xFFFF0000u);
    }
  }
  __device__ __forceinline__ float dot(u32x4 e, float acc) const {
    if constexpr (SPACE == DEWI_SPACE_COSINE) return dot8_packed(e, p, acc);
    else return dot8<SPACE>(e, f, acc);
  }
};

// Rows that are NOT whole 16-byte units (fp32: dim % 4 != 0, bf16: dim % 8 != 0; or a shard whose first row does not start on a
// unit) — the PH = true forms of both kernels.  The loads stay aligned 16-byte units; a row then starts `head` bytes into its first unit
// and that offset repeats every G = 16 / gcd(16, row bytes) rows.  A wave (long rows) or a lane group (short rows) only ever takes rows
// of ONE residue mod G, so its query fragments are loaded once, shifted by its own head, zero outside the row; the first and the last
// unit of a row also hold columns of the neighbouring rows (read again by the wave that owns them: <= 2 units per row of extra traffic)
// and those are cleared with a bit mask before the arithmetic — a NaN / Inf neighbour must not reach this row's sum.
// The bits to keep of the unit whose first column is `first`:
template <int ELEM>
__device__ __forceinline__ u32x4 unit_keep_mask(int first, int dim) {
  uint32_t m[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if constexpr (ELEM == 0) {
      m[d] = static_cast<unsigned>(first + d) < static_cast<unsigned>(dim) ? 0xFFFFFFFFu : 0u;
    } else {
      m[d] = (static_cast<unsigned>(first + 2 * d) < static_cast<unsigned>(dim) ? 0x0000FFFFu : 0u) |
             (static_cast<unsigned>(first + 2 * d + 1) < static_cast<unsigned>(dim) ? 0xFFFF0000u : 0u);
    }
  }
  return u32x4{m[0], m[1], m[2], m[3]};
}
__device__ __forceinline__ u32x4 and_u4(u32x4 a, u32x4 b) { return u32x4{a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w}; }
// the raw query's float64 sum of squares, lane-strided: the same per-lane partial sums whatever the wave's head is
__device__ __forceinline__ double strided_sumsq(const float* __restrict__ qrow, int dim, int lane) {
  double ss = 0.0;
  for (int i = lane; i < dim; i += kWave) ss += square_f64(qrow[i]);
  return ss;
}

// Rows a wave issues before its first reduction.  Measured on MI355X at 3 GB (scripts/sweep_any.sh, profiles/r04): rows
// that are not whole KiB want ~5-6 KiB in flight per wave — more than the 3 KiB of the dim = 768 kernel, because the
// reduction and the list offer come round more often per byte and a wave shares its first and last cache line with its
// neighbours.  TB/s by (row bytes, rows in flight): 640 B: 8 rows 6.47, 12 rows 6.92; 768 B: 8 -> 6.99, 12 -> 6.87;
// 1040 B: 3 -> 5.81, 4 -> 6.45; 1280 B: 3 -> 6.67, 4 -> 7.07; 1536 B: 2 -> 6.33, 3 -> 7.07, 4 -> 6.95; 2064 B: 1 -> 4.94,
// 2 -> 6.69; 2800 B: 1 -> 6.25, 2 -> 6.97.  So up to three choices per units-per-lane count, picked by the row's width.
#ifndef DEWI_ANY_RSHORT
#define DEWI_ANY_RSHORT 8
#endif
constexpr int kAnyLevels = 3;
constexpr int any_level(int units) {            // 0: the narrowest rows of a units-per-lane count ... 2: the widest
  const int u = (units + 63) / 64;
  if (u == 1) return units <= 40 ? 0 : 1;                        // 528-640 B : 656-1024 B
  if (u == 2) return units <= 72 ? 0 : (units <= 92 ? 1 : 2);    // 1040-1152 B : 1168-1472 B : 1488-2048 B (1408 B: 4 rows 6.98 TB/s,
                                                                 // 3 rows 6.86; 1424 B: 6.97 / 6.92; 1536 B: 6.95 / 7.07)
  if (u == 4) return units <= 224 ? 0 : 1;                       // 3088-3584 B : 3600-4096 B
  return 1;
}
constexpr int any_rows(int u, int nq, int level) {
#ifdef DEWI_ANY_NQ4_OLD
  if (nq >= 4) {                     // four queries behind every row: the arithmetic hides more of the latency
    if (u == 1) return level == 0 ? 6 : 4;
    if (u <= 3) return 2;
    return 1;
  }
#endif
  if (u == 1) return level == 0 ? 12 : 8;
  if (u == 2) return level == 0 ? 6 : (level == 1 ? 4 : 3);
  if (u == 3) return 2;
  if (u == 4) return level == 0 ? 2 : 1;
  return 1;
}
constexpr int any_nq_max_odd(int u_pad) { return u_pad <= 4 ? 4 : 1; }   // rows that are not whole units (PH kernels)
// most queries one corpus pass serves besides 1 (launch_any_long instantiates exactly these)
constexpr int any_nq_max(int elem_bytes, int u_pad) {
  if (elem_bytes == 2) return u_pad <= 4 ? 4 : (u_pad <= 8 ? 2 : 1);
  return u_pad <= 8 ? 4 : 2;
}
constexpr int kAnyShortRows = DEWI_ANY_RSHORT;   // loads (of 64 / P rows each) a wave of the short-row kernel keeps in flight

// ---------------------------------------------------------------------------------------------
// 33 .. 64 U units per row
// ---------------------------------------------------------------------------------------------
// `units`: 16-byte units per row; PH (rows that are not whole units): COLUMNS per row instead.
template <int ELEM, int U, int R, int NQ, int SPACE, int S, bool PH = false>
__device__ __forceinline__ void scan_rows_any_body(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                   const float* __restrict__ Q, int n_candidates,
                                                   uint64_t* __restrict__ keys, int64_t keys_per_query, MergeShared& merge_buf) {
  constexpr int kCols = ELEM ? 8 : 4;
  constexpr int kElemBytes = ELEM ? 2 : 4;
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  // (wave-uniform on purpose: row numbers and row addresses then live in scalar registers — without it the compiler keeps one
  // 64-bit vector address per row in flight and updates each of them every iteration)
  const int wave_in_block = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int dim = PH ? units : units * kCols;
  // PH: this wave takes the rows ph, ph + G, ph + 2 G, ... — all of them start head_el columns into their first unit
  int row_step = 1, ph = 0, off0 = 0, head_el = 0, row_bytes = 0;
  int64_t wave_pos = gwave, wave_cnt = n_waves;      // this wave's place among the waves that share its rows
  if constexpr (PH) {
    off0 = static_cast<int>(reinterpret_cast<uintptr_t>(E) & 15u);
    E = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(E) - off0);   // (pointer arithmetic: stays a global pointer)
    row_bytes = dim * kElemBytes;
    const int tz = __builtin_ctz(static_cast<unsigned>(row_bytes) | 16u);   // row_step = 16 / gcd(16, row bytes)
    row_step = 16 >> tz;
    ph = static_cast<int>(gwave & (row_step - 1));
    wave_pos = gwave >> (4 - tz);
    wave_cnt = (n_waves - ph + row_step - 1) >> (4 - tz);                   // (n_waves >= 8 >= row_step: never 0)
    const int head = (off0 + ph * row_bytes) & 15;
    head_el = head / kElemBytes;
    units = (head + row_bytes + 15) >> 4;                                   // units THIS wave's rows touch
  }

  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) act[u] = lane + 64 * u < units;
  [[maybe_unused]] u32x4 keep_first, keep_last;        // PH: the row's own columns in its first / last unit
  [[maybe_unused]] int u_last = 0;
  if constexpr (PH) {
    u_last = (units - 1) >> 6;
    keep_first = unit_keep_mask<ELEM>(lane * kCols - head_el, dim);
    keep_last = unit_keep_mask<ELEM>((lane + 64 * u_last) * kCols - head_el, dim);
  }

  // Query fragments, normalised here for cosine (reference backends.py:420-424) with the float64-summed norm every
  // kernel uses (common.hpp wave_query_norm): a query has one prepared form whatever kernel serves it.
  UnitFrag<ELEM, SPACE> qf[NQ][U];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float* qrow = Q + static_cast<int64_t>(qi) * dim;
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (PH) {
        qf[qi][u].load_shifted(qrow, (lane + 64 * u) * kCols - head_el, dim);
      } else {
        qf[qi][u].load(qrow, lane + 64 * u, act[u]);
        ss += qf[qi][u].sumsq();
      }
    }
    if constexpr (PH && SPACE == DEWI_SPACE_COSINE) ss = strided_sumsq(qrow, dim, lane);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      const float norm = wave_query_norm(ss);
      if (norm > 0.f) {
#pragma unroll
        for (int u = 0; u < U; ++u) qf[qi][u].scale(norm);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) qf[qi][u].finish();
  }

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  // i: index among this wave's rows (not PH: the row itself).  Consecutive rows of a wave are a whole number of units apart.
  const int64_t row_units = PH ? (static_cast<int64_t>(row_step) * row_bytes) >> 4 : units;
  auto row_of = [&](int64_t i) { return PH ? ph + row_step * i : i; };
  auto first_unit = [&](int64_t i) { return PH ? (off0 + row_of(i) * row_bytes) >> 4 : i * units; };
  auto fetch = [&](u32x4(&v)[U], const u32x4* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (act[u]) v[u] = load_u4<true>(p + 64 * u);
    }
  };
  auto consume = [&](u32x4(&v)[U], int64_t row) {
    if constexpr (PH) {
      v[0] = and_u4(v[0], keep_first);
#pragma unroll
      for (int u = 1; u < U; ++u)
        if (u == u_last) v[u] = and_u4(v[u], keep_last);
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) acc = qf[qi][u].dot(v[u], acc);
      float s = wave_sum_f32(acc);
      if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
      if constexpr (DENSE) {
        if (lane == 0) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
      } else {
        lst[qi].offer(s, static_cast<uint32_t>(row), lane);
      }
    }
  };

  // rows of this wave's residue: ph + row_step * i, i in [0, n_mine)   (not PH: every row, i = the row)
  const int64_t n_mine = PH ? (n_rows > ph ? (n_rows - ph + row_step - 1) / row_step : 0) : n_rows;
  const int64_t n_groups = n_mine / R;
  for (int64_t g = wave_pos; g < n_groups; g += wave_cnt) {
    u32x4 v[R][U];
    const u32x4* p = E + first_unit(g * R) + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) fetch(v[r], p + r * row_units);
#pragma unroll
    for (int r = 0; r < R; ++r) consume(v[r], row_of(g * R + r));
  }
  for (int64_t i = n_groups * R + wave_pos; i < n_mine; i += wave_cnt) {   // fewer than R rows left
    u32x4 v[U];
    fetch(v, E + first_unit(i) + lane);
    consume(v, row_of(i));
  }

  if constexpr (S == 1) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      block_merge_store(lst[qi], merge_buf, keys + qi * keys_per_query + static_cast<int64_t>(blockIdx.x) * n_candidates,
                        n_candidates, lane, wave_in_block);
  } else if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      lst[qi].store(keys + qi * keys_per_query + gwave * n_candidates, n_candidates, lane);
  }
}

template <int ELEM, int U, int R, int NQ, int SPACE, int S, bool PH = false>
__global__ __launch_bounds__(kScanThreads) void scan_rows_any(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                              const float* __restrict__ Q, int n_candidates,
                                                              uint64_t* __restrict__ keys, int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_rows_any_body<ELEM, U, R, NQ, SPACE, S, PH>(E, n_rows, units, Q, n_candidates, keys, keys_per_query, merge_buf);
}

// Rows that are not whole units, ONE query, up to two units per lane: a wave on CONSECUTIVE rows (scan_rows_odd_contig).
// The per-residue form above gives neighbouring rows to different waves, so every cache line two rows share is requested by
// two waves; with nontemporal loads more than half of the second requests went back to HBM (FETCH_SIZE x 2 at dim 301:
// 1.063 x the corpus; plain loads kept them cached but streamed slower).  Narrow rows can afford the other layout: the wave
// keeps one register set of query fragments and masks PER RESIDUE (kOddPeriod x U x 8 registers) and takes R consecutive rows
// per step, R a multiple of the period, so that row i of a step always has residue i mod period — a compile-time choice of
// the register set.  Same loads, same lanes, same sums per row as the per-residue form (a row's offset inside its first unit
// is a property of its address): results are bit-equal, shards included.
constexpr int kOddPeriod = 4;   // serves offsets that repeat every 2 or 4 rows (every fp32 width; bf16 rows with a period of 8 keep the per-residue form)
template <int ELEM, int U, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_odd_contig(const u32x4* __restrict__ E, int64_t n_rows, int dim,
                                                                     const float* __restrict__ Q, int n_candidates,
                                                                     uint64_t* __restrict__ keys, int64_t keys_per_query) {
  static_assert(R % kOddPeriod == 0, "a step's rows must cover whole periods");
  constexpr int G = kOddPeriod;
  constexpr int kCols = ELEM ? 8 : 4;
  constexpr int kElemBytes = ELEM ? 2 : 4;
  constexpr bool DENSE = S == 0;
  __shared__ MergeShared merge_buf;
  const int lane = lane_id();
  const int wave_in_block = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int off0 = static_cast<int>(reinterpret_cast<uintptr_t>(E) & 15u);
  E = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(E) - off0);
  const int row_bytes = dim * kElemBytes;

  // per residue g (rows g, g + G, ...): the units a row touches, the row's own columns in each of them, the query fragments
  bool act[G][U];
  u32x4 keep[G][U];
  UnitFrag<ELEM, SPACE> qf[G][U];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int head = (off0 + g * row_bytes) & 15;
    const int head_el = head / kElemBytes;
    const int units = (head + row_bytes + 15) >> 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      act[g][u] = lane + 64 * u < units;
      keep[g][u] = unit_keep_mask<ELEM>((lane + 64 * u) * kCols - head_el, dim);
      qf[g][u].load_shifted(Q, (lane + 64 * u) * kCols - head_el, dim);
    }
  }
  if constexpr (SPACE == DEWI_SPACE_COSINE) {
    const float norm = wave_query_norm(strided_sumsq(Q, dim, lane));
    if (norm > 0.f) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int u = 0; u < U; ++u) qf[g][u].scale(norm);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int u = 0; u < U; ++u) qf[g][u].finish();
  }

  WaveList<DENSE ? 1 : S> lst;
  if constexpr (!DENSE) lst.init(n_candidates, lane);

  auto fetch = [&](u32x4(&v)[U], int64_t row, const bool(&a)[U]) {
    const u32x4* p = E + ((off0 + row * row_bytes) >> 4) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (a[u]) v[u] = load_u4<true>(p + 64 * u);
    }
  };
  auto consume = [&](const u32x4(&v)[U], int64_t row, const u32x4(&k)[U], const UnitFrag<ELEM, SPACE>(&q)[U]) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = q[u].dot(and_u4(v[u], k[u]), acc);
    float s = wave_sum_f32(acc);
    if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
    if constexpr (DENSE) {
      if (lane == 0) keys[row] = make_key(s, static_cast<uint32_t>(row));
    } else {
      lst.offer(s, static_cast<uint32_t>(row), lane);
    }
  };

  const int64_t n_groups = n_rows / R;
  for (int64_t grp = gwave; grp < n_groups; grp += n_waves) {
    u32x4 v[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r) fetch(v[r], grp * R + r, act[r % G]);
#pragma unroll
    for (int r = 0; r < R; ++r) consume(v[r], grp * R + r, keep[r % G], qf[r % G]);
  }
  for (int64_t row = n_groups * R + gwave; row < n_rows; row += n_waves) {   // fewer than R rows left: residue at run time
    const int g_row = static_cast<int>(row & (G - 1));                       // (n_groups * R is a multiple of G)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g_row == g) {
        u32x4 v[U];
        fetch(v, row, act[g]);
        consume(v, row, keep[g], qf[g]);
      }
    }
  }

  if constexpr (S == 1) {
    block_merge_store(lst, merge_buf, keys + static_cast<int64_t>(blockIdx.x) * n_candidates, n_candidates, lane, wave_in_block);
  } else if constexpr (!DENSE) {
    lst.store(keys + gwave * n_candidates, n_candidates, lane);
  }
}
// rows per step of that kernel (multiples of kOddPeriod), from the rows-in-flight choice of the per-residue form
constexpr int odd_contig_rows(int u, int rows_any) { return u == 1 ? (rows_any >= 12 ? 12 : 8) : (rows_any >= 6 ? 8 : 4); }

// REPAIR form: every flagged query of a batch, one corpus pass each, in one launch (knn_scan.hip scan_rows_f32_flagged)
template <int ELEM, int U, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_any_flagged(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                      const float* __restrict__ Q, int n_candidates,
                                                                      uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                                      const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  const int dim = units * (ELEM ? 8 : 4);
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_rows_any_body<ELEM, U, R, 1, SPACE, S>(E, n_rows, units, Q + static_cast<int64_t>(q) * dim, n_candidates,
                                                keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// 1 .. 32 units per row: P = 2^log2p lanes per row
// ---------------------------------------------------------------------------------------------
// Sum over each aligned group of 2^log2p lanes, in a fixed order.  Valid in (at least) the LAST lane of every group:
// up to 16 lanes every lane of the group holds the sum; the 32-lane step (row_bcast15) leaves it in DPP rows 1 and 3.
__device__ __forceinline__ float group_sum_f32(float v, int log2p) {
#define DEWI_STEP(CTRL, MASK) v = v + __int_as_float(dpp_i32<CTRL, MASK>(0, __float_as_int(v)));
  if (log2p >= 1) { DEWI_STEP(0xB1, 0xF) }    // quad_perm [1,0,3,2]
  if (log2p >= 2) { DEWI_STEP(0x4E, 0xF) }    // quad_perm [2,3,0,1]
  if (log2p >= 3) { DEWI_STEP(0x141, 0xF) }   // row_half_mirror: the other quad of the 8
  if (log2p >= 4) { DEWI_STEP(0x140, 0xF) }   // row_mirror: the other 8 of the 16
  if (log2p >= 5) { DEWI_STEP(0x142, 0xA) }   // row_bcast15 into rows 1, 3: the other 16 of the 32
#undef DEWI_STEP
  return v;
}

// `units`: 16-byte units per row; PH (rows that are not whole units): COLUMNS per row instead, and the planner has made sure that
// the rows of one load are a multiple of the period G of the rows' offsets — a lane's rows all have the residue sub mod G.
template <int ELEM, int R, int NQ, int SPACE, int S, bool PH = false>
__device__ __forceinline__ void scan_short_rows_any_body(const u32x4* __restrict__ E, int64_t n_rows, int units, int log2p,
                                                         const float* __restrict__ Q, int n_candidates,
                                                         uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                         MergeShared& merge_buf) {
  constexpr int kCols = ELEM ? 8 : 4;
  constexpr int kElemBytes = ELEM ? 2 : 4;
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  const int wave_in_block = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);   // (see scan_rows_any_body)
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int dim = PH ? units : units * kCols;
  const int group = 1 << log2p;           // lanes per row
  const int rows_per_load = kWave >> log2p;
  const int sub = lane >> log2p;          // which row of the load
  const int pos = lane & (group - 1);     // which unit of the row
  int off0 = 0, row_bytes = 0, head_el = 0;
  [[maybe_unused]] u32x4 keep;            // PH: the row's own columns in this lane's unit
  if constexpr (PH) {
    off0 = static_cast<int>(reinterpret_cast<uintptr_t>(E) & 15u);
    E = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(E) - off0);
    row_bytes = dim * kElemBytes;
    const int row_step = 16 >> __builtin_ctz(static_cast<unsigned>(row_bytes) | 16u);
    const int head = (off0 + (sub & (row_step - 1)) * row_bytes) & 15;
    head_el = head / kElemBytes;
    units = (head + row_bytes + 15) >> 4;                  // units THIS lane's rows touch
    keep = unit_keep_mask<ELEM>(pos * kCols - head_el, dim);
  }
  const bool act = pos < units;
  const bool holder = pos == group - 1;   // the lane group_sum_f32 leaves the row's score in

  UnitFrag<ELEM, SPACE> qf[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float* qrow = Q + static_cast<int64_t>(qi) * dim;
    if constexpr (PH) qf[qi].load_shifted(qrow, pos * kCols - head_el, dim);
    else qf[qi].load(qrow, pos, act);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      // every lane group holds a copy of the query: the first one alone feeds the norm (PH: a lane-strided sum of the raw query)
      const float norm = wave_query_norm(PH ? strided_sumsq(qrow, dim, lane) : (sub == 0 ? qf[qi].sumsq() : 0.0));
      if (norm > 0.f) qf[qi].scale(norm);
    }
    qf[qi].finish();
  }

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  const int64_t rows_per_step = static_cast<int64_t>(rows_per_load) * R;
  const int64_t n_steps = (n_rows + rows_per_step - 1) / rows_per_step;
  const int64_t load_units = PH ? (static_cast<int64_t>(rows_per_load) * row_bytes) >> 4 : static_cast<int64_t>(rows_per_load) * units;
  for (int64_t st = gwave; st < n_steps; st += n_waves) {
    const int64_t row_first = st * rows_per_step + sub;
    u32x4 v[R];
    // (the loads of a step are a whole number of units apart: rows_per_load rows)
    const u32x4* p = E + (PH ? (off0 + row_first * row_bytes) >> 4 : row_first * units) + pos;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row_first + static_cast<int64_t>(r) * rows_per_load;
      v[r] = u32x4{0u, 0u, 0u, 0u};
      if (act && row < n_rows) v[r] = load_u4<true>(p + r * load_units);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row_first + static_cast<int64_t>(r) * rows_per_load;
      const bool mine = holder && row < n_rows;
      if constexpr (PH) v[r] = and_u4(v[r], keep);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float s = group_sum_f32(qf[qi].dot(v[r], 0.f), log2p);
        if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
        if constexpr (DENSE) {
          if (mine) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
        } else {
          // rows worth an offer: not below the list's current worst (NaN rows pass, as in WaveList::offer)
          unsigned long long m = __ballot(mine && !(s < lst[qi].thr_s));
          while (m != 0ull) {
            const int src = __ffsll(m) - 1;
            m &= m - 1ull;
            const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), src));
            const uint32_t rw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(row)), src));
            lst[qi].offer(sc, rw, lane);
          }
        }
      }
    }
  }

  if constexpr (S == 1) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      block_merge_store(lst[qi], merge_buf, keys + qi * keys_per_query + static_cast<int64_t>(blockIdx.x) * n_candidates,
                        n_candidates, lane, wave_in_block);
  } else if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      lst[qi].store(keys + qi * keys_per_query + gwave * n_candidates, n_candidates, lane);
  }
}

template <int ELEM, int R, int NQ, int SPACE, int S, bool PH = false>
__global__ __launch_bounds__(kScanThreads) void scan_short_rows_any(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                    int log2p, const float* __restrict__ Q, int n_candidates,
                                                                    uint64_t* __restrict__ keys, int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_short_rows_any_body<ELEM, R, NQ, SPACE, S, PH>(E, n_rows, units, log2p, Q, n_candidates, keys, keys_per_query, merge_buf);
}

template <int ELEM, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_short_rows_any_flagged(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                            int log2p, const float* __restrict__ Q,
                                                                            int n_candidates, uint64_t* __restrict__ keys,
                                                                            int64_t keys_per_query,
                                                                            const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  const int dim = units * (ELEM ? 8 : 4);
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_short_rows_any_body<ELEM, R, 1, SPACE, S>(E, n_rows, units, log2p, Q + static_cast<int64_t>(q) * dim, n_candidates,
                                                   keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// dispatch (one translation unit per element type instantiates it: knn_scan_any_f32.hip, knn_scan_any_bf16.hip)
// ---------------------------------------------------------------------------------------------
template <int ELEM, int NQ, int SPACE, int S, bool PH>
static hipError_t launch_any_long(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int c, uint64_t* keys,
                                  hipStream_t stream) {
  const int width = PH ? plan.row_cols : plan.units;
#define DEWI_ANY_LAUNCH(UU, RR)                                                                                          \
  hipLaunchKernelGGL((scan_rows_any<ELEM, UU, RR, NQ, SPACE, S, PH>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, \
                     width, Q, c, keys, plan.keys_per_query);                                                           \
  return hipGetLastError();
#define DEWI_ANY_CASE(UU)                                                                                       \
  case UU:                                                                                                      \
    if constexpr (any_rows(UU, NQ, 0) != any_rows(UU, NQ, 1)) {                                                 \
      if (plan.level == 0) { DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 0)) }                                         \
    }                                                                                                           \
    if constexpr (any_rows(UU, NQ, 2) != any_rows(UU, NQ, 1)) {                                                 \
      if (plan.level == 2) { DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 2)) }                                         \
    }                                                                                                           \
    DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 1))
  // Queries per pass by row width (the fragments of NQ queries must fit the registers next to the rows in flight): an fp32
  // unit keeps 4 registers per query, a bf16 unit 4 (cosine, packed pairs) or 8 (l2) — any_nq_max() is the same table
  // (PH, rows that are not whole units: any_nq_max_odd — four queries up to 4 units per lane, one beyond)
  constexpr bool kNarrow = PH ? (NQ == 1 || NQ == 4) : (ELEM == 0 ? (NQ == 1 || NQ == 4) : true);   // units per lane 1 .. 4 (bf16) / 1 .. 8 (fp32)
  constexpr bool kMiddle = PH ? NQ == 1 : (ELEM == 0 ? (NQ == 1 || NQ == 4) : (NQ == 1 || NQ == 2));   // 5 .. 8
  constexpr bool kWide = PH ? NQ == 1 : (ELEM == 0 ? (NQ == 1 || NQ == 2) : NQ == 1);    // 10 .. 16
  if constexpr (kNarrow && NQ != 2) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(1)
      DEWI_ANY_CASE(2)
      DEWI_ANY_CASE(3)
      DEWI_ANY_CASE(4)
      default: break;
    }
  }
  if constexpr (kMiddle) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(5)
      DEWI_ANY_CASE(6)
      DEWI_ANY_CASE(8)
      default: break;
    }
  }
  if constexpr (kWide) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(10)
      DEWI_ANY_CASE(12)
      DEWI_ANY_CASE(16)
      default: break;
    }
  }
#undef DEWI_ANY_LAUNCH
#undef DEWI_ANY_CASE
  return hipErrorInvalidValue;
}

template <int ELEM, int NQ, int SPACE, int S, bool PH>
static hipError_t launch_any_kind(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int c, uint64_t* keys,
                                  hipStream_t stream) {
  if (plan.kind == kScanAnyShort) {
    if constexpr (NQ == 2) {
      return hipErrorInvalidValue;
    } else {
      hipLaunchKernelGGL((scan_short_rows_any<ELEM, kAnyShortRows, NQ, SPACE, S, PH>), dim3(plan.blocks), dim3(kScanThreads), 0, stream,
                         E, n_rows, PH ? plan.row_cols : plan.units, plan.log2p, Q, c, keys, plan.keys_per_query);
      return hipGetLastError();
    }
  }
  return launch_any_long<ELEM, NQ, SPACE, S, PH>(plan, E, n_rows, Q, c, keys, stream);
}

// PH = false: rows of whole units (knn_scan_any_*.hip); PH = true: the others (knn_scan_odd_*.hip)
template <int ELEM, bool PH = false>
static hipError_t launch_scan_any_impl(const ScanPlan& plan, const void* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                                       int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream) {
  const u32x4* E = static_cast<const u32x4*>(d_E);
  const float* Q = d_q_raw + static_cast<int64_t>(q0) * dim;
  uint64_t* keys = d_keys + static_cast<int64_t>(q0) * plan.keys_per_query;
#define DEWI_ANY_S(NQ, SPACE)                                                                                        \
  switch (plan.slots) {                                                                                              \
    case 0: return launch_any_kind<ELEM, NQ, SPACE, 0, PH>(plan, E, n_rows, Q, n_candidates, keys, stream);          \
    case 1: return launch_any_kind<ELEM, NQ, SPACE, 1, PH>(plan, E, n_rows, Q, n_candidates, keys, stream);          \
    default: return launch_any_kind<ELEM, NQ, SPACE, kMaxSlots, PH>(plan, E, n_rows, Q, n_candidates, keys, stream); \
  }
#define DEWI_ANY_Q(NQ)                 \
  if (space == DEWI_SPACE_COSINE) {    \
    DEWI_ANY_S(NQ, DEWI_SPACE_COSINE)  \
  } else {                             \
    DEWI_ANY_S(NQ, DEWI_SPACE_L2)      \
  }
  if constexpr (PH) {
    if (nq == 1 && plan.odd_contig) {   // narrow rows, one query: a wave on consecutive rows (scan_rows_odd_contig)
#define DEWI_ODD_CONTIG(UU, RR, SPACE)                                                                                              \
  switch (plan.slots) {                                                                                                             \
    case 0: hipLaunchKernelGGL((scan_rows_odd_contig<ELEM, UU, RR, SPACE, 0>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,  \
                               n_rows, dim, Q, n_candidates, keys, plan.keys_per_query); break;                                      \
    case 1: hipLaunchKernelGGL((scan_rows_odd_contig<ELEM, UU, RR, SPACE, 1>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,  \
                               n_rows, dim, Q, n_candidates, keys, plan.keys_per_query); break;                                      \
    default: hipLaunchKernelGGL((scan_rows_odd_contig<ELEM, UU, RR, SPACE, kMaxSlots>), dim3(plan.blocks), dim3(kScanThreads), 0,    \
                                stream, E, n_rows, dim, Q, n_candidates, keys, plan.keys_per_query); break;                          \
  }                                                                                                                                 \
  return hipGetLastError();
#define DEWI_ODD_CONTIG_R(UU, RR)                                       \
  if (plan.u_pad == UU && plan.odd_contig_rows == RR) {                  \
    if (space == DEWI_SPACE_COSINE) { DEWI_ODD_CONTIG(UU, RR, DEWI_SPACE_COSINE) } \
    else { DEWI_ODD_CONTIG(UU, RR, DEWI_SPACE_L2) }                      \
  }
      DEWI_ODD_CONTIG_R(1, 12)
      DEWI_ODD_CONTIG_R(1, 8)
      DEWI_ODD_CONTIG_R(2, 8)
      DEWI_ODD_CONTIG_R(2, 4)
#undef DEWI_ODD_CONTIG_R
#undef DEWI_ODD_CONTIG
      return hipErrorInvalidValue;
    }
  }
  if (nq == 1) {
    DEWI_ANY_Q(1)
  } else if (nq == 2) {
    if constexpr (!PH) { DEWI_ANY_Q(2) }
  } else if (nq == 4) {
    if constexpr (PH && ELEM == 1) {
      // bf16 l2 keeps 8 fp32 registers per query and unit next to the masks: four queries spill — one pass per query there
      if (space != DEWI_SPACE_COSINE) {
        for (int i = 0; i < 4; ++i) {
          const hipError_t e = launch_scan_any_impl<ELEM, PH>(plan, d_E, n_rows, dim, d_q_raw, q0 + i, 1, n_candidates, space, d_keys, stream);
          if (e != hipSuccess) return e;
        }
        return hipSuccess;
      }
      DEWI_ANY_S(4, DEWI_SPACE_COSINE)
    } else {
      DEWI_ANY_Q(4)
    }
  }
#undef DEWI_ANY_Q
#undef DEWI_ANY_S
  return hipErrorInvalidValue;
}

// flagged (repair) launches: the middle rows-in-flight level, rows of at most 512 units (see scan_flagged_supported)
template <int ELEM, int SPACE, int S>
static hipError_t launch_any_flagged_s(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int n_queries, int c,
                                       uint64_t* keys, const uint32_t* flags, hipStream_t stream) {
  if (plan.kind == kScanAnyShort) {
    hipLaunchKernelGGL((scan_short_rows_any_flagged<ELEM, kAnyShortRows, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream,
                       E, n_rows, plan.units, plan.log2p, Q, c, keys, plan.keys_per_query, flags, n_queries);
    return hipGetLastError();
  }
#define DEWI_ANY_FLAGGED(UU)                                                                                                     \
  case UU:                                                                                                                       \
    hipLaunchKernelGGL((scan_rows_any_flagged<ELEM, UU, any_rows(UU, 1, 1), SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0,  \
                       stream, E, n_rows, plan.units, Q, c, keys, plan.keys_per_query, flags, n_queries);                         \
    return hipGetLastError();
  switch (plan.u_pad) {
    DEWI_ANY_FLAGGED(1)
    DEWI_ANY_FLAGGED(2)
    DEWI_ANY_FLAGGED(3)
    DEWI_ANY_FLAGGED(4)
    DEWI_ANY_FLAGGED(5)
    DEWI_ANY_FLAGGED(6)
    DEWI_ANY_FLAGGED(8)
    default: break;
  }
#undef DEWI_ANY_FLAGGED
  return hipErrorInvalidValue;
}

template <int ELEM>
static hipError_t launch_scan_any_flagged_impl(const ScanPlan& plan, const void* d_E, int64_t n_rows, const float* d_q_raw,
                                               int n_queries, int n_candidates, int space, uint64_t* d_keys,
                                               const uint32_t* d_flags, hipStream_t stream) {
  const u32x4* E = static_cast<const u32x4*>(d_E);
#define DEWI_ANY_FS(SPACE)                                                                                                         \
  switch (plan.slots) {                                                                                                            \
    case 0: return launch_any_flagged_s<ELEM, SPACE, 0>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
    case 1: return launch_any_flagged_s<ELEM, SPACE, 1>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
    default: return launch_any_flagged_s<ELEM, SPACE, kMaxSlots>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
  }
  if (space == DEWI_SPACE_COSINE) {
    DEWI_ANY_FS(DEWI_SPACE_COSINE)
  } else {
    DEWI_ANY_FS(DEWI_SPACE_L2)
  }
#undef DEWI_ANY_FS
}

}  // namespace dewi
