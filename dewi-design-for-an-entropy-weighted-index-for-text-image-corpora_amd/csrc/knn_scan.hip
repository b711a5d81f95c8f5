// Corpus scan: similarity of every embedding row against a small group of queries, fused with a
// streaming top-c selection, for gfx950 (MI355X).
//
// Replaces steps 1-3 of ExactIndex.search (reference src/dewi/backends.py:420-444):
//   q <- q/||q||;  scores = E @ q  (or -sum((E-q)^2));  argpartition(scores, -c)[-c:]
// The N-long score vector is never written: each wavefront keeps the best c keys it has seen in
// registers (one 64-bit key per lane per slot) and only those leave the kernel.
//
// Roofline: HBM.  Algorithmic bytes per launch = n_rows * dim * sizeof(elem) (the corpus, read
// exactly once) + n_queries*dim*4; nothing else is proportional to n_rows.
//
// Data movement (fast path, dim = 256*U floats): one wavefront owns one row at a time; lane l
// issues U global_load_dwordx4 at byte offsets 16*l + 1024*u of the row, so every wave-instruction
// reads 1 KiB contiguous and a row is U such instructions.  R rows (R*U loads) are issued
// back-to-back before the first FMA, which keeps R*U KiB per wave in flight; consecutive waves of
// the whole grid take consecutive R-row groups, so one sweep of the grid reads one contiguous
// span of the matrix.  Query fragments live in registers for the whole kernel (the same 4*U
// floats per lane are needed for every row), so there is no LDS traffic at all; the cross-lane
// sum runs on the DPP crossbar.
#include "scan_any.hpp"

namespace dewi {

// ---------------------------------------------------------------------------------------------
// Fast path: dim == 256*U, one row per wavefront step, R rows per iteration, NQ queries per pass.
// ---------------------------------------------------------------------------------------------
// S = key slots per lane: 1 (c <= 64, block-merged sorted output, one list per workgroup),
// 4 (c <= 256, one unsorted list per wave) or 0 (dense: one key per row).
template <int U, int R, int NQ, int SPACE, int S, bool NT>
__device__ __forceinline__ void scan_rows_f32_body(const float* __restrict__ E, int64_t n_rows,
                                                   const float* __restrict__ Q, int n_candidates,
                                                   uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                   MergeShared& merge_buf) {
  constexpr int D4 = 64 * U;  // float4 units per row
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);

  // Query fragments; cosine queries are normalised here (reference backends.py:420-424:
  // divide by the norm unless it is zero).  Every wave repeats the same arithmetic, so all
  // waves hold identical bits.
  f32x4 qf[NQ][U];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const f32x4* qp = reinterpret_cast<const f32x4*>(Q) + static_cast<int64_t>(qi) * D4 + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) qf[qi][u] = qp[u * 64];
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      double ss = 0.0;   // float64 sum of squares: the same norm in every kernel (common.hpp, wave_query_norm)
#pragma unroll
      for (int u = 0; u < U; ++u)
        ss += square_f64(qf[qi][u].x) + square_f64(qf[qi][u].y) + square_f64(qf[qi][u].z) + square_f64(qf[qi][u].w);
      const float norm = wave_query_norm(ss);
      if (norm > 0.f) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          qf[qi][u].x = __fdiv_rn(qf[qi][u].x, norm);
          qf[qi][u].y = __fdiv_rn(qf[qi][u].y, norm);
          qf[qi][u].z = __fdiv_rn(qf[qi][u].z, norm);
          qf[qi][u].w = __fdiv_rn(qf[qi][u].w, norm);
        }
      }
    }
  }

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  const f32x4* Ev = reinterpret_cast<const f32x4*>(E);
  auto consume = [&](const f32x4(&v)[U], int64_t row) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) acc = accum4<SPACE>(v[u], qf[qi][u], acc);
      float s = wave_sum_f32(acc);
      if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
      if constexpr (DENSE) {
        if (lane == 0) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
      } else {
        lst[qi].offer(s, static_cast<uint32_t>(row), lane);
      }
    }
  };

  // Full R-row groups.  Deliberately NOT software-pipelined and with a small R: on MI355X the scan
  // is fastest with only ~24 KiB of loads in flight per CU (8 waves x one 3 KiB row; 1M x 768:
  // R=1 0.434 ms, R=2 0.449, R=8 0.459, prefetching the next group 0.441; R=2 with the two rows n_waves
  // apart, so that the chip still sweeps one window: 0.437 vs 0.429) — more outstanding requests lower
  // the achieved HBM rate instead of raising it.
  const int64_t n_groups = n_rows / R;
  for (int64_t g = gwave; g < n_groups; g += n_waves) {
    const int64_t row0 = g * R;
    f32x4 v[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f32x4* p = Ev + (row0 + r) * D4 + lane;
#pragma unroll
      for (int u = 0; u < U; ++u) v[r][u] = load_x4<NT>(p + u * 64);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) consume(v[r], row0 + r);
  }
  // Remainder rows (fewer than R), one per wave.
  for (int64_t row = n_groups * R + gwave; row < n_rows; row += n_waves) {
    f32x4 v[U];
    const f32x4* p = Ev + row * D4 + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_x4<NT>(p + u * 64);
    consume(v, row);
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

template <int U, int R, int NQ, int SPACE, int S, bool NT>
__global__ __launch_bounds__(kScanThreads) void scan_rows_f32(const float* __restrict__ E, int64_t n_rows,
                                                              const float* __restrict__ Q, int n_candidates,
                                                              uint64_t* __restrict__ keys,
                                                              int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_rows_f32_body<U, R, NQ, SPACE, S, NT>(E, n_rows, Q, n_candidates, keys, keys_per_query, merge_buf);
}

// REPAIR form (abi.cpp batch_repair): one launch answers every query of a batch whose flag is set — the queries a
// matrix-core pass refused — one corpus pass each, and returns at once when none is (the usual case: one ~3 us dispatch).
template <int U, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_f32_flagged(const float* __restrict__ E, int64_t n_rows,
                                                                      const float* __restrict__ Q, int n_candidates,
                                                                      uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                                      const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_rows_f32_body<U, R, 1, SPACE, S, true>(E, n_rows, Q + static_cast<int64_t>(q) * (256 * U), n_candidates,
                                                keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// Generic path: any dim (since round 4 only rows wider than 1024 units reach it: scan_any.hpp serves the rest).  G (power
// of two <= 64) lanes share a row, a wave covers 64/G rows per step; queries come pre-normalised from global memory
// (L1/L2 resident).  VEC = 4 when rows can be read as float4 (dim % 4 == 0, 16-byte aligned base), else scalar loads.
// ---------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<1> { using type = float; };
template <>
struct VecT<4> { using type = f32x4; };

template <int VEC, int NQ, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_generic_f32(const float* __restrict__ E, int64_t n_rows, int dim,
                                                                 const float* __restrict__ Qn, int group,
                                                                 int n_candidates, uint64_t* __restrict__ keys,
                                                                 int64_t keys_per_query) {
  using V = typename VecT<VEC>::type;
  constexpr bool DENSE = S == 0;
  __shared__ MergeShared merge_buf;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int rows_per_step = kWave / group;
  const int sub = lane / group;  // which row of the step
  const int lg = lane % group;   // position inside the row group
  const int units = dim / VEC;

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  const int64_t n_steps = (n_rows + rows_per_step - 1) / rows_per_step;
  for (int64_t st = gwave; st < n_steps; st += n_waves) {
    const int64_t row = st * rows_per_step + sub;
    float acc[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) acc[qi] = 0.f;
    if (row < n_rows) {
      const V* ep = reinterpret_cast<const V*>(E + row * dim);
      for (int u = lg; u < units; u += group) {
        const V e = ep[u];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          const V q = reinterpret_cast<const V*>(Qn + static_cast<int64_t>(qi) * dim)[u];
          if constexpr (VEC == 4) {
            acc[qi] = accum4<SPACE>(e, q, acc[qi]);
          } else {
            if constexpr (SPACE == DEWI_SPACE_COSINE) {
              acc[qi] = __builtin_fmaf(e, q, acc[qi]);
            } else {
              const float d = e - q;
              acc[qi] = __builtin_fmaf(d, d, acc[qi]);
            }
          }
        }
      }
    }
    // butterfly inside each group of `group` lanes
    for (int off = group >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) acc[qi] += __shfl_xor(acc[qi], off, kWave);
    }
    for (int r = 0; r < rows_per_step; ++r) {
      const int64_t rr = st * rows_per_step + r;
      if (rr >= n_rows) break;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float s = __shfl(acc[qi], r * group, kWave);
        if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
        if constexpr (DENSE) {
          if (lane == 0) keys[qi * keys_per_query + rr] = make_key(s, static_cast<uint32_t>(rr));
        } else {
          lst[qi].offer(s, static_cast<uint32_t>(rr), lane);
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

// ---------------------------------------------------------------------------------------------
// Query preparation for the generic path: one wave per query, cosine -> q / ||q|| unless 0.
// ---------------------------------------------------------------------------------------------
// qn2 (optional): ||prepared query||^2 per output row — float64-summed, one rounding to fp32 — for the l2 score of the
// matrix-core passes, 2<e,q> - ||e||^2 - ||q||^2.
__global__ __launch_bounds__(kWave) void prepare_queries_f32(const float* __restrict__ Q, float* __restrict__ Qn,
                                                             int dim, int space, int to_bf16, int n_real,
                                                             float* __restrict__ qn2) {
  const int lane = lane_id();
  const float* q = Q + static_cast<int64_t>(blockIdx.x) * dim;
  float* o = Qn + static_cast<int64_t>(blockIdx.x) * dim;
  if (static_cast<int>(blockIdx.x) >= n_real) {   // padding row of a 32-query block
    for (int j = lane; j < dim; j += kWave) o[j] = 0.f;
    if (qn2 != nullptr && lane == 0) qn2[blockIdx.x] = 0.f;
    return;
  }
  auto round_bf16 = [](float v) {   // nearest-even bf16, kept as the fp32 value it represents
    if (v != v) return v;
    const uint32_t u = __float_as_uint(v);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
  };
  // dim % 4 == 0 and at most 2048: the row stays in registers (all loads in flight together, see prepare_queries_bf16)
  if ((dim & 3) == 0 && dim <= 2048) {
    const int n4 = dim >> 2;
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + u * kWave;
      v[u] = j < n4 ? reinterpret_cast<const f32x4*>(q)[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float norm = 1.f;
    bool scale = false;
    if (space == DEWI_SPACE_COSINE) {
      double ss = 0.0;   // float64: the same norm as every other kernel's (common.hpp, wave_query_norm)
#pragma unroll
      for (int u = 0; u < 8; ++u) ss += square_f64(v[u].x) + square_f64(v[u].y) + square_f64(v[u].z) + square_f64(v[u].w);
      norm = wave_query_norm(ss);
      scale = norm > 0.f;
    }
    double out2 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + u * kWave;
      if (j < n4) {
        f32x4 r;
        r.x = scale ? __fdiv_rn(v[u].x, norm) : v[u].x;
        r.y = scale ? __fdiv_rn(v[u].y, norm) : v[u].y;
        r.z = scale ? __fdiv_rn(v[u].z, norm) : v[u].z;
        r.w = scale ? __fdiv_rn(v[u].w, norm) : v[u].w;
        if (to_bf16) {
          r.x = round_bf16(r.x);
          r.y = round_bf16(r.y);
          r.z = round_bf16(r.z);
          r.w = round_bf16(r.w);
        }
        out2 += square_f64(r.x) + square_f64(r.y) + square_f64(r.z) + square_f64(r.w);
        reinterpret_cast<f32x4*>(o)[j] = r;
      }
    }
    if (qn2 != nullptr) {
      out2 = wave_sum_f64(out2);
      if (lane == 0) qn2[blockIdx.x] = static_cast<float>(out2);
    }
    return;
  }
  float norm = 1.f;
  bool scale = false;
  if (space == DEWI_SPACE_COSINE) {
    double ss = 0.0;
    for (int j = lane; j < dim; j += kWave) ss += square_f64(q[j]);
    norm = wave_query_norm(ss);
    scale = norm > 0.f;
  }
  double out2 = 0.0;
  for (int j = lane; j < dim; j += kWave) {
    const float v = scale ? __fdiv_rn(q[j], norm) : q[j];
    const float r = to_bf16 ? round_bf16(v) : v;
    out2 += square_f64(r);
    o[j] = r;
  }
  if (qn2 != nullptr) {
    out2 = wave_sum_f64(out2);
    if (lane == 0) qn2[blockIdx.x] = static_cast<float>(out2);
  }
}

hipError_t launch_prepare_queries(const float* d_q, float* d_qn, int n_queries, int dim, int space, int to_bf16,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(prepare_queries_f32, dim3(n_queries), dim3(kWave), 0, stream, d_q, d_qn, dim, space, to_bf16, n_queries,
                     static_cast<float*>(nullptr));
  return hipGetLastError();
}

hipError_t launch_prepare_queries_padded(const float* d_q, float* d_qn, int n_queries, int n_rows_out, int dim, int space,
                                         float* d_qn2, hipStream_t stream) {
  hipLaunchKernelGGL(prepare_queries_f32, dim3(n_rows_out), dim3(kWave), 0, stream, d_q, d_qn, dim, space, 0, n_queries, d_qn2);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Planner + dispatch
// ---------------------------------------------------------------------------------------------
static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

ScanPlan plan_scan(int64_t n_rows, int dim, int elem_bytes, int n_candidates, int compute_units,
                   const Tuning& tuning) {
  ScanPlan p{};
  const int u = dim / 256;
  p.dense = n_candidates > kMaxListCandidates;
  if (elem_bytes == 2) {  // bf16 corpus: rows taken in pairs, 16-byte units of 8 columns
    p.fast = (dim % 256 == 0) && (u >= 1 && u <= 4);
    p.vec = (dim % 8 == 0) ? 8 : 1;
    p.rows_per_iter = 2;
  } else {
    p.fast = (dim % 256 == 0) && (u == 1 || u == 2 || u == 3 || u == 4 || u == 6);
    p.vec = (dim % 4 == 0) ? 4 : 1;
    p.rows_per_iter = u == 1 ? 4 : (u == 2 ? 2 : 1);  // ~3-4 KiB in flight per wave
    // four queries per corpus pass quadruple the arithmetic behind every row: two rows in flight per wave
    // (1 M x 768, batch 4: R=1 0.562 ms, R=2 0.468, R=4 0.468)
    p.rows_per_iter_batch = p.rows_per_iter < 2 ? 2 : p.rows_per_iter;
    if (p.fast && tuning.rows_per_iter > 0) p.rows_per_iter = p.rows_per_iter_batch = tuning.rows_per_iter;
  }
  const int units = (dim + p.vec - 1) / p.vec;
  p.group = p.fast ? kWave : (next_pow2(units) > kWave ? kWave : next_pow2(units));
  if (!p.fast) p.rows_per_iter = kWave / p.group;
  p.kind = p.fast ? kScanFast : kScanGeneric;
  p.nq_max = 4;
  p.units = p.u_pad = p.log2p = 0;
  p.level = 1;
  // rows of whole 16-byte units outside the tuned set: the any-width kernels (scan_any.hpp), queries in registers
  const int cols_per_unit = elem_bytes == 2 ? 8 : 4;
  if (!p.fast && dim % cols_per_unit == 0) {
    const int n_units = dim / cols_per_unit;
    if (n_units <= 32) {
      p.kind = kScanAnyShort;
      p.units = n_units;
      while ((1 << p.log2p) < n_units) ++p.log2p;
      p.rows_per_iter = p.rows_per_iter_batch = (kWave >> p.log2p) * kAnyShortRows;
    } else if (n_units <= 64 * 16) {
      static const int pads[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
      const int need = (n_units + 63) / 64;
      p.kind = kScanAnyLong;
      p.units = n_units;
      for (int v : pads)
        if (v >= need) { p.u_pad = v; break; }
      p.nq_max = any_nq_max(elem_bytes, p.u_pad);
      // (sweeps: dewi_tuning_set rows_per_iter = 1, 2, 3 picks the level instead of the width)
      p.level = tuning.rows_per_iter > 0 ? (tuning.rows_per_iter - 1) % kAnyLevels : any_level(n_units);
      p.rows_per_iter = any_rows(p.u_pad, 1, p.level);
      p.rows_per_iter_batch = any_rows(p.u_pad, p.nq_max, p.level);
    }
  }
  // rows that are NOT whole units (round 4; they used to take scan_generic_*): the same two kernels in their PH form —
  // aligned 16-byte loads, each wave / lane group on the rows of one residue mod G = 16 / gcd(16, row bytes)
  p.odd_rows = false;
  p.odd_contig = false;
  p.odd_contig_rows = 0;
  p.row_cols = dim;
  if (!p.fast && dim % cols_per_unit != 0) {
    const int row_bytes = dim * elem_bytes;
    const int most_units = (row_bytes + (16 - elem_bytes) + 15) / 16;   // a row that starts 16 - elem_bytes into its first unit
    int period = 16, tz = 0;
    while (tz < 4 && (row_bytes >> tz) % 2 == 0) ++tz;
    period >>= tz;
    int log2p = 0;
    while ((1 << log2p) < most_units) ++log2p;
    if (most_units <= 32 && (kWave >> log2p) % period == 0) {
      p.kind = kScanAnyShort;
      p.odd_rows = true;
      p.units = most_units;
      p.log2p = log2p;
      p.rows_per_iter = p.rows_per_iter_batch = (kWave >> p.log2p) * kAnyShortRows;
    } else if (most_units <= 64 * 16) {
      static const int pads[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
      const int need = (most_units + 63) / 64;
      p.kind = kScanAnyLong;
      p.odd_rows = true;
      p.units = most_units;
      for (int v : pads)
        if (v >= need) { p.u_pad = v; break; }
      p.nq_max = any_nq_max_odd(p.u_pad);
      // (rows in flight by the bytes a row HAS — (row_bytes + 15) / 16 units — not by the units it can touch — and one choice
      // further towards more rows: a wave's rows lie row_step rows apart here.  3 GB, TB/s by level: dim 161 6.18 / 6.01 / 6.01,
      // 301 6.56 / 6.48 / 6.03, 387 6.64 / 6.62 / 6.65, 1001 6.79 / 6.72 / 6.72; profiles/r04/odd_rows/sweep_levels.txt)
      int level = any_level((row_bytes + 15) / 16);
      if (level > 0) --level;
      p.level = tuning.rows_per_iter > 0 ? (tuning.rows_per_iter - 1) % kAnyLevels : level;
      p.rows_per_iter = any_rows(p.u_pad, 1, p.level);
      p.rows_per_iter_batch = any_rows(p.u_pad, p.nq_max, p.level);
      // narrow rows whose offsets repeat every 2 or 4 rows: one query runs with a wave on consecutive rows (scan_any.hpp)
      if (p.u_pad <= 2 && period <= kOddPeriod && tuning.rows_per_iter == 0) {
        p.odd_contig = true;
        p.odd_contig_rows = odd_contig_rows(p.u_pad, p.rows_per_iter);
        p.rows_per_iter = p.odd_contig_rows;
      }
    }
  }
  p.raw_queries = p.kind != kScanGeneric;
  p.nontemporal = tuning.nontemporal < 0 ? true : tuning.nontemporal != 0;
  // One 8-wave workgroup per CU (8 waves x R*U KiB in flight each): 8 waves per CU measured
  // fastest on MI355X (1M x 768, R=8: 0.443 ms vs 0.448 ms at 32 waves per CU with R=4), and one
  // list per CU keeps the select kernel's input at compute_units lists.  Never more waves than
  // there are row groups to give each wave at least four iterations of work.
  int64_t max_blocks = static_cast<int64_t>(compute_units);
  if (tuning.scan_blocks > 0) max_blocks = tuning.scan_blocks;
  const int64_t rows_per_wave_iter = p.rows_per_iter;
  const int64_t groups = (n_rows + rows_per_wave_iter - 1) / rows_per_wave_iter;
  int64_t blocks = (groups + 4 * (kScanThreads / kWave) - 1) / (4 * (kScanThreads / kWave));
  if (blocks < 1) blocks = 1;
  if (blocks > max_blocks) blocks = max_blocks;
  p.blocks = static_cast<int>(blocks);
  p.waves = p.blocks * (kScanThreads / kWave);
  p.nq_per_launch = 4;
  p.slots = p.dense ? 0 : (n_candidates <= kWave ? 1 : kMaxSlots);
  p.n_lists = p.dense ? 0 : (p.slots == 1 ? p.blocks : p.waves);
  p.keys_per_query = p.dense ? n_rows : static_cast<int64_t>(p.n_lists) * n_candidates;
  return p;
}

template <int U, int R, int NQ, int SPACE, int S>
static void launch_fast_nt(const ScanPlan& plan, const float* E, int64_t n_rows, const float* Q, int c,
                           uint64_t* keys, hipStream_t stream) {
  if (plan.nontemporal)
    hipLaunchKernelGGL((scan_rows_f32<U, R, NQ, SPACE, S, true>), dim3(plan.blocks), dim3(kScanThreads), 0,
                       stream, E, n_rows, Q, c, keys, plan.keys_per_query);
  else
    hipLaunchKernelGGL((scan_rows_f32<U, R, NQ, SPACE, S, false>), dim3(plan.blocks), dim3(kScanThreads), 0,
                       stream, E, n_rows, Q, c, keys, plan.keys_per_query);
}

template <int U, int NQ, int SPACE, int S>
static bool launch_fast_r(const ScanPlan& plan, const float* E, int64_t n_rows, const float* Q, int c,
                          uint64_t* keys, hipStream_t stream) {
  // R only changes how rows are grouped per wave; results do not depend on it.  Combinations that
  // are not instantiated (register budget, build time) step down to the next smaller R.
  const int r = NQ >= 4 ? plan.rows_per_iter_batch : plan.rows_per_iter;
  if (r >= 8) {
    if constexpr (U <= 3 && NQ == 1 && S == 1) {
      launch_fast_nt<U, 8, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
      return true;
    }
  }
  if (r >= 4) {
    if constexpr (U <= 3 && NQ == 1) {
      launch_fast_nt<U, 4, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
      return true;
    }
  }
  if (r >= 2) {
    if constexpr (U <= 3) {
      launch_fast_nt<U, 2, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
      return true;
    }
  }
  launch_fast_nt<U, 1, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
  return true;
}

template <int NQ, int SPACE, int S>
static bool launch_fast_u(const ScanPlan& plan, int dim, const float* E, int64_t n_rows, const float* Q, int c,
                          uint64_t* keys, hipStream_t stream) {
  switch (dim / 256) {
    case 1: return launch_fast_r<1, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
    case 2: return launch_fast_r<2, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
    case 3: return launch_fast_r<3, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
    case 4: return launch_fast_r<4, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
    case 6: return launch_fast_r<6, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
    default: return false;
  }
}

template <int NQ, int SPACE, int S>
static void launch_generic(const ScanPlan& plan, int dim, const float* E, int64_t n_rows, const float* Qn, int c,
                           uint64_t* keys, hipStream_t stream) {
  if (plan.vec == 4)
    hipLaunchKernelGGL((scan_generic_f32<4, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, dim, Qn, plan.group, c, keys, plan.keys_per_query);
  else
    hipLaunchKernelGGL((scan_generic_f32<1, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, dim, Qn, plan.group, c, keys, plan.keys_per_query);
}

template <int NQ, int SPACE, int S>
static hipError_t launch_scan_impl(const ScanPlan& plan, const float* E, int64_t n_rows, int dim, const float* Qraw,
                                   const float* Qn, int c, uint64_t* keys, hipStream_t stream) {
  if (plan.fast) {
    if (!launch_fast_u<NQ, SPACE, S>(plan, dim, E, n_rows, Qraw, c, keys, stream)) return hipErrorInvalidValue;
  } else {
    launch_generic<NQ, SPACE, S>(plan, dim, E, n_rows, Qn, c, keys, stream);
  }
  return hipGetLastError();
}

template <int U, int SPACE>
static hipError_t launch_flagged_s(const ScanPlan& plan, const float* E, int64_t n_rows, const float* Q, int n_queries, int c,
                                   uint64_t* keys, const uint32_t* flags, hipStream_t stream) {
  constexpr int R = U == 1 ? 4 : (U == 2 ? 2 : 1);
  switch (plan.slots) {
    case 0:
      hipLaunchKernelGGL((scan_rows_f32_flagged<U, R, SPACE, 0>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, Q, c,
                         keys, plan.keys_per_query, flags, n_queries);
      break;
    case 1:
      hipLaunchKernelGGL((scan_rows_f32_flagged<U, R, SPACE, 1>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, Q, c,
                         keys, plan.keys_per_query, flags, n_queries);
      break;
    default:
      hipLaunchKernelGGL((scan_rows_f32_flagged<U, R, SPACE, kMaxSlots>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows,
                         Q, c, keys, plan.keys_per_query, flags, n_queries);
      break;
  }
  return hipGetLastError();
}

bool scan_flagged_supported(const ScanPlan& plan, int elem_bytes) {
  (void)elem_bytes;
  if (plan.kind == kScanFast) return true;
  // any-width kernels: the widths the matrix-core passes run at outside the dim = 256 U set — rows of whole units up to 1536
  // columns (the depth-split pass with a partial last chunk), 1280 / 2048, bf16 also 3072 / 4096 (512 units per row)
  // (rows that are not whole units never meet a matrix-core pass: those take whole 16-byte units)
  return !plan.odd_rows && (plan.kind == kScanAnyShort || (plan.kind == kScanAnyLong && plan.u_pad <= 8));
}

hipError_t launch_scan_flagged_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                   int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                   hipStream_t stream) {
  if (plan.kind == kScanAnyLong || plan.kind == kScanAnyShort)
    return launch_scan_any_flagged_f32(plan, d_E, n_rows, dim, d_q_raw, n_queries, n_candidates, space, d_keys, d_flags, stream);
  if (plan.kind != kScanFast) return hipErrorInvalidValue;
#define DEWI_FLAGGED(UU)                                                                                                 \
  case UU:                                                                                                               \
    return space == DEWI_SPACE_COSINE                                                                                    \
               ? launch_flagged_s<UU, DEWI_SPACE_COSINE>(plan, d_E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream) \
               : launch_flagged_s<UU, DEWI_SPACE_L2>(plan, d_E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream);
  switch (dim / 256) {
    DEWI_FLAGGED(1)
    DEWI_FLAGGED(2)
    DEWI_FLAGGED(3)
    DEWI_FLAGGED(4)
    DEWI_FLAGGED(6)
    default: break;
  }
#undef DEWI_FLAGGED
  return hipErrorInvalidValue;
}

hipError_t launch_scan_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                           const float* d_q_norm, int q0, int nq, int n_candidates, int space, uint64_t* d_keys,
                           hipStream_t stream) {
  if (plan.odd_rows) return launch_scan_odd_f32(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
  if (plan.kind == kScanAnyLong || plan.kind == kScanAnyShort)
    return launch_scan_any_f32(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
  const float* qr = d_q_raw + static_cast<int64_t>(q0) * dim;
  const float* qn = d_q_norm ? d_q_norm + static_cast<int64_t>(q0) * dim : nullptr;
  uint64_t* keys = d_keys + static_cast<int64_t>(q0) * plan.keys_per_query;
#define DEWI_DISPATCH_S(NQ, SPACE)                                                                              \
  switch (plan.slots) {                                                                                          \
    case 0: return launch_scan_impl<NQ, SPACE, 0>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);   \
    case 1: return launch_scan_impl<NQ, SPACE, 1>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);   \
    default: return launch_scan_impl<NQ, SPACE, kMaxSlots>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream); \
  }
#define DEWI_DISPATCH(NQ)                                  \
  if (space == DEWI_SPACE_COSINE) {                        \
    DEWI_DISPATCH_S(NQ, DEWI_SPACE_COSINE)                 \
  } else {                                                 \
    DEWI_DISPATCH_S(NQ, DEWI_SPACE_L2)                     \
  }
  if (nq == 1) {
    DEWI_DISPATCH(1)
  } else if (nq == 4) {
    DEWI_DISPATCH(4)
  } else if (nq == 8 && plan.fast && plan.slots == 1) {   // eight queries per corpus pass: row-per-wave kernel, c <= 64 only
    if (space == DEWI_SPACE_COSINE)
      return launch_scan_impl<8, DEWI_SPACE_COSINE, 1>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);
    return launch_scan_impl<8, DEWI_SPACE_L2, 1>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);
  }
#undef DEWI_DISPATCH
#undef DEWI_DISPATCH_S
  return hipErrorInvalidValue;
}

}  // namespace dewi
