// Candidate selection, DEWI blend and final top-k for gfx950 (MI355X).
//
// Replaces steps 3-9 of ExactIndex.search (reference src/dewi/backends.py:439-481):
//   top = argpartition(scores, -c)[-c:]                    -> block radix-select over the scan's keys
//   adj = (1-eta)*scores[top] + eta*dewi[top] (+pref*ent)  -> fp32, two rounded products + one add
//   argpartition(adj, -k)[-k:], argsort(-adj)              -> bitonic sort in LDS, first k
// One workgroup per query.  The work is O(keys) with keys << corpus bytes; this kernel is
// latency-bound, not bandwidth-bound, and is kept to a handful of passes over L2-resident keys.
#include <stdlib.h>

#include "select_common.hpp"

namespace dewi {

__device__ __forceinline__ int pow2_at_least(int v) {
  int p = 2;
  while (p < v) p <<= 1;
  return p;
}

__device__ __forceinline__ float blend(const RerankParams& rp, float sim, float dewi, float ent) {
  // A10: the reference's HNSW / FAISS backends blend a similarity derived from the library's distance
  // (backends.py:229-231 `1 - dist`; :338-341 `1.0 / (1.0 + dist)`), not the raw score.  The library's
  // distance is 1 - <e,q> (hnswlib cosine, fp32) or the squared L2 distance (= -score in l2 space).
  if (rp.transform != DEWI_SIM_RAW) {
    const float dist = rp.space == DEWI_SPACE_L2 ? -sim : __fsub_rn(1.f, sim);
    sim = rp.transform == DEWI_SIM_ONE_MINUS_DIST ? __fsub_rn(1.f, dist) : __fdiv_rn(1.f, __fadd_rn(1.f, dist));
  }
  // reference backends.py:461-465: (1-eta)*s and eta*dewi are rounded separately, then added.
  float adj = __fadd_rn(__fmul_rn(rp.w_sim, sim), __fmul_rn(rp.w_dewi, dewi));
  if (rp.use_ent) adj = __fadd_rn(adj, __fmul_rn(rp.w_ent, ent));
  return adj;
}

// After sh.sel[0..n_sel) holds the candidates sorted by (sim desc, row asc) and dewi/ent of entry t
// are available through `fetch(t, &dewi, &ent, &id)`: blend, sort, emit the first k.
template <class Fetch>
__device__ void rerank_and_emit(SelectShared& sh, int n_sel, int k, const RerankParams& rp, Fetch fetch,
                                int64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int p2 = pow2_at_least(n_sel);
  int64_t my_id = -1;  // id of candidate t == tid, kept in a register for the one-candidate-per-thread case
  for (int t = tid; t < p2; t += nt) {
    uint64_t k2 = kKeyEmpty;
    if (t < n_sel) {
      float dewi, ent;
      int64_t id;
      fetch(t, dewi, ent, id);
      if (t == tid) my_id = id;
      const float adj = blend(rp, key_score(sh.sel[t]), dewi, ent);
      // ties on the adjusted score: the candidate that ranked higher on similarity first
      k2 = (static_cast<uint64_t>(ord_f32(adj)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - static_cast<uint32_t>(t));
    }
    sh.sel2[t] = k2;
  }
  if (n_sel <= kRankSortMax) {
    // few candidates: each thread ranks its own adjusted key and writes its output slot directly
    __syncthreads();
    const int S = rank_split(n_sel, nt);
    if (S == 1) {
      for (int t = tid; t < n_sel; t += nt) {
        const uint64_t mine = sh.sel2[t];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < n_sel; ++j) rank += sh.sel2[j] > mine ? 1 : 0;  // adjusted keys are unique (low word = t)
        if (rank < k) {
          int64_t id = my_id;
          if (t != tid) {  // more candidates than threads: fetch again
            float dewi, ent;
            fetch(t, dewi, ent, id);
          }
          out_ids[rank] = id;
          out_scores[rank] = unord_f32(static_cast<uint32_t>(mine >> 32));
        }
      }
    } else {
      // S lanes share a candidate: each counts a slice of the comparisons (see rank_sort_desc)
      const int part = tid & (S - 1);
      for (int t = tid / S; t < n_sel; t += nt / S) {
        const uint64_t mine = sh.sel2[t];
        int rank = 0;
        const int iters = (n_sel + S - 1) / S;   // uniform trip count, see rank_sort_desc
#pragma unroll 8
        for (int i = 0; i < iters; ++i) {
          const int j = part + i * S;
          rank += sh.sel2[j < p2 ? j : p2 - 1] > mine ? 1 : 0;   // sel2[n_sel .. p2) holds empty keys (never greater)
        }
        for (int m = 1; m < S; m <<= 1) rank += __shfl_xor(rank, m, kWave);
        if (part == 0 && rank < k) {
          float dewi, ent;
          int64_t id;
          fetch(t, dewi, ent, id);
          out_ids[rank] = id;
          out_scores[rank] = unord_f32(static_cast<uint32_t>(mine >> 32));
        }
      }
    }
    return;
  }
  bitonic_sort_desc<false>(sh.sel2, nullptr, p2);
  for (int j = tid; j < k && j < n_sel; j += nt) {
    const uint64_t k2 = sh.sel2[j];
    const int t = static_cast<int>(0xFFFFFFFFu - static_cast<uint32_t>(k2));
    float dewi, ent;
    int64_t id;
    fetch(t, dewi, ent, id);
    out_ids[j] = id;
    out_scores[j] = unord_f32(static_cast<uint32_t>(k2 >> 32));
  }
}

// ---------------------------------------------------------------------------------------------
// Single-device (or per-shard) select: keys from the scan -> final results or candidate records.
// ---------------------------------------------------------------------------------------------
// Exact route: 8-bit MSB radix select for the n_candidates-th largest key, compaction of everything at
// or above it into sh.sel, bitonic sort.  Returns the number of valid keys in sh.sel.
template <class Keys>
__device__ int exact_top_candidates(const Keys& keys, int n_candidates, SelectShared& sh) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const uint64_t thr = block_kth_largest(keys, static_cast<uint32_t>(n_candidates), sh);
  if (tid == 0) sh.count = 0;
  const int p2 = pow2_at_least(n_candidates);
  for (int t = tid; t < p2; t += nt) sh.sel[t] = kKeyEmpty;
  __syncthreads();
  keys.for_each(tid, nt, [&](uint64_t key) {
    if (key >= thr) {
      const uint32_t pos = atomicAdd(&sh.count, 1u);
      if (pos < static_cast<uint32_t>(p2)) sh.sel[pos] = key;
    }
  });
  __syncthreads();
  const int n_sel = static_cast<int>(sh.count < static_cast<uint32_t>(n_candidates) ? sh.count : n_candidates);
  if (n_sel <= kRankSortMax && sh.count <= static_cast<uint32_t>(kRankSortMax)) {
    // exactly n_sel keys were gathered: one ranking pass instead of log^2 bitonic stages
    for (int t = tid; t < n_sel; t += nt) sh.sel2[t] = sh.sel[t];
    __syncthreads();
    rank_sort_desc(sh.sel2, sh.sel, n_sel);
    return n_sel;
  }
  bitonic_sort_desc<false>(sh.sel, nullptr, p2);
  return n_sel;
}

// Same for a dense LDS array of unique keys (survivors of the batched matrix-core scan), using the
// prefix-skipping 11-bit select.
__device__ int top_candidates_lds(const uint64_t* keys, int n, int n_candidates, SelectShared& sh, WideRadixShared& ws) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const uint64_t thr = block_kth_largest_lds(keys, n, static_cast<uint32_t>(n_candidates), ws);
  if (tid == 0) sh.count = 0;
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const uint64_t key = keys[i];
    if (key >= thr) {
      const uint32_t pos = atomicAdd(&sh.count, 1u);   // exactly min(n, n_candidates) keys pass
      sh.sel2[pos] = key;
    }
  }
  __syncthreads();
  const int n_sel = static_cast<int>(sh.count);
  if (n_sel <= kRankSortMax) {
    if (tid < kWave) sh.sel2[n_sel + tid] = kKeyEmpty;   // padding for the unmasked ranking loop
    __syncthreads();
    rank_sort_desc<true>(sh.sel2, sh.sel, n_sel);
    return n_sel;
  }
  const int p2 = pow2_at_least(n_sel);
  for (int t = tid; t < p2; t += nt) sh.sel[t] = t < n_sel ? sh.sel2[t] : kKeyEmpty;
  bitonic_sort_desc<false>(sh.sel, nullptr, p2);
  return n_sel;
}

// EXACT REFINEMENT of approximate survivor scores (RefineParams).  Two producers:
//   l2      the depth-split pass over an fp32 corpus: a = 2<e,q> - ||e||^2 - ||q||^2, within
//           M(a) = margin * (3 ||q||^2 + 2 d) * 1.01,   d = max(0, -a)
//           of the row's exact score (the kernel's bound is margin * (||e||^2 + ||q||^2), and ||e|| <= ||q|| + sqrt(d_exact)
//           gives ||e||^2 + ||q||^2 <= 3 ||q||^2 + 2 d_exact; the 1 % covers d_exact vs d);
//   cosine  the 256-query bf16 pass over the bf16 SHADOW of an fp32 corpus: a = <bf16(e), bf16(q)>, within the constant
//           M = margin (launch.hpp shadow_margin) of the fp32 row kernel's <e, q>.
// `keys` (dense LDS array, n unique keys) carry those scores.  Steps:
//   1. a_c = the c-th largest approximate score (radix select as before); L = a_c - M(a_c) is a lower bound of the c-th
//      best EXACT score (c rows have exact >= a - M(a) >= L, M not shrinking as a falls);
//   2. every key with a + M(a) >= L is a candidate (c' >= c of them) — no row of the exact top c can be missing;
//   3. one wave per candidate re-scores it with THE ROW KERNEL'S ARITHMETIC (scan_rows_f32<U, ...>: lane l takes the
//      16-byte units l + 64 u, u ascending; cosine: the query normalised by the float64-summed norm, fmaf(e, q, acc);
//      l2: d = e - q, fmaf(d, d, acc), negated; wave_sum_f32), so the score — and with it the ranking — is bit for bit
//      what the one-query search computes for that row (dim = 256 U, U <= 3);
//   4. the exact keys are sorted, the best c stay in sh.sel.
// Returns the number of valid keys in sh.sel, or -2 when more candidates qualify than sh.sel2 holds (the query is
// refused and the caller re-runs it on the row kernels).  The dense array doubles as scratch once the candidates are
// compacted.
// Part 1 (two sources): find the cut and compact the candidates into sh.sel2; returns their number.
__device__ __forceinline__ float refine_bound(const RefineParams& rf, float q2, float a) {
  const float d = a < 0.f ? -a : 0.f;              // M(a); NaN scores (NaN rows rank first) get an infinite bound
  return a == a ? (rf.space == DEWI_SPACE_L2 ? rf.margin * (3.f * q2 + 2.f * d) * 1.01f : rf.margin) : __builtin_inff();
}
__device__ __forceinline__ float refine_low(const RefineParams& rf, float q2, uint64_t thr) {
  if (thr == 1ull) return -__builtin_inff();       // fewer than c keys: every key is a candidate
  const float a_c = key_score(thr);
  const float low = a_c - refine_bound(rf, q2, a_c);
  return low == low ? low : -__builtin_inff();
}
// wave-aggregated append of `key` (if pass) to sh.sel2: one LDS atomic per wave and call
__device__ __forceinline__ void refine_append(SelectShared& sh, bool pass, uint64_t key, int lane) {
  const unsigned long long m = __ballot(pass);
  if (m != 0ull) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&sh.count, static_cast<uint32_t>(__popcll(m)));
    base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
    const uint32_t pos = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
    if (pass && pos < static_cast<uint32_t>(kMaxSortCandidates)) sh.sel2[pos] = key;
  }
}

__device__ __forceinline__ int refine_rescore(uint64_t* keys, int n_candidates, SelectShared& sh, const RefineParams& rf, int q);

// survivors staged in LDS (`keys`, n of them)
__device__ __forceinline__ int refine_top_candidates(uint64_t* keys, int n, int n_candidates, SelectShared& sh,
                                                     WideRadixShared& ws, const RefineParams& rf, int q) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63;
  const uint64_t thr = block_kth_largest_lds(keys, n, static_cast<uint32_t>(n_candidates), ws);
  const float q2 = rf.space == DEWI_SPACE_L2 ? rf.qn2[q] : 0.f;
  const float low = refine_low(rf, q2, thr);
  if (tid == 0) sh.count = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += nt) {
    const int i = i0 + tid;
    const uint64_t key = i < n ? keys[i] : kKeyEmpty;
    const float a = key_score(key);
    refine_append(sh, key != kKeyEmpty && !(a + refine_bound(rf, q2, a) < low), key, lane);
  }
  __syncthreads();
  return refine_rescore(keys, n_candidates, sh, rf, q);
}

// more survivors than the staging holds (dense data, large c): the same two steps over the segments in global memory —
// the 8-bit radix select re-reads them once per pass (slower, L2-resident), nothing is refused.  `scratch`: >= 2048 keys.
__device__ __forceinline__ int refine_from_segments(const SegmentKeys& kv, uint64_t* scratch, int n_candidates, SelectShared& sh,
                                                    const RefineParams& rf, int q) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63;
  const uint64_t thr = block_kth_largest(kv, static_cast<uint32_t>(n_candidates), sh);
  const float q2 = rf.space == DEWI_SPACE_L2 ? rf.qn2[q] : 0.f;
  const float low = refine_low(rf, q2, thr);
  if (tid == 0) sh.count = 0;
  __syncthreads();
  // every thread walks the same number of (segment, slot) steps so that the ballots inside refine_append are wave-wide
  constexpr int kGroup = 4;
  const int sub = tid % kGroup;
  for (int seg0 = 0; seg0 < kv.n_seg; seg0 += nt / kGroup) {
    const int seg = seg0 + tid / kGroup;
    uint32_t cnt = seg < kv.n_seg ? kv.count[seg * kv.count_stride] : 0u;
    cnt = cnt < static_cast<uint32_t>(kv.cap) ? cnt : static_cast<uint32_t>(kv.cap);
    const uint32_t steps = static_cast<uint32_t>(kv.cap + kGroup - 1) / kGroup;
    for (uint32_t st = 0; st < steps; ++st) {
      const uint32_t j = st * kGroup + sub;
      uint64_t key = kKeyEmpty;
      if (j < cnt) {
        const uint64_t v = kv.p[seg * kv.seg_stride + j];
        key = kv.raw ? make_key(__uint_as_float(static_cast<uint32_t>(v)), static_cast<uint32_t>(v >> 32)) : v;
      }
      if (__builtin_amdgcn_ballot_w64(j < cnt) == 0ull) break;          // this wave's segments are exhausted
      const float a = key_score(key);
      refine_append(sh, key != kKeyEmpty && !(a + refine_bound(rf, q2, a) < low), key, lane);
    }
  }
  __syncthreads();
  return refine_rescore(scratch, n_candidates, sh, rf, q);
}

// ONE query whose approximate scores come from the bf16 ROW kernel over the bf16 shadow of an fp32 corpus (cosine): `n_lists`
// (<= 256, one per workgroup of the scan) lists of `list_len` keys, each sorted descending, list_len >= c.
//   a0 = the c-th largest of 64 lane maxima of the list heads: a lower bound of a_c, the c-th best approximate score;
//   a row of the exact top c has exact >= a_c - M (c rows do), so approximate >= a_c - 2M >= a0 - 2M =: low — every key at or
//   above `low` is a candidate (a superset of refine_top_candidates' set: the band hangs below a0 instead of a_c);
//   a list holds its workgroup's best list_len rows: if ALL of them are candidates the workgroup may have dropped one, and the
//   query is refused (-2: adversarial corpora — thousands of near-copies of one row; the caller repeats it on the fp32 scan).
// Four threads per list walk it together (sorted: once nothing of a wave passes, nothing later does).
__device__ __forceinline__ int refine_from_sorted_lists(const uint64_t* __restrict__ keys, int n_lists, int list_len, int n_candidates,
                                                        SelectShared& sh, uint64_t* scratch, const RefineParams& rf, int q) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63;
  if (tid == 0) {
    sh.count = 0;
    sh.total = 0;
    sh.bound = 1ull;
  }
  __syncthreads();
  if (tid < kWave) {
    uint64_t lane_max = kKeyEmpty;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = tid + kWave * u;
      const uint64_t m = l < n_lists ? keys[static_cast<int64_t>(l) * list_len] : kKeyEmpty;
      lane_max = m > lane_max ? m : lane_max;
    }
    const uint32_t mh = static_cast<uint32_t>(lane_max >> 32), ml = static_cast<uint32_t>(lane_max);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kWave; ++j) {
      const uint64_t o = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mh), j))) << 32) |
                         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ml), j));
      rank += (o > lane_max || (o == lane_max && j < tid)) ? 1 : 0;
    }
    // (fewer than c non-empty lanes: no bound, every key is a candidate)
    if (rank == n_candidates - 1 && lane_max != kKeyEmpty) sh.bound = lane_max;
  }
  __syncthreads();
  const uint64_t bound = sh.bound;
  const float low = bound == 1ull ? -__builtin_inff() : key_score(bound) - 2.f * rf.margin;
  constexpr int kGroup = 4;
  const int sub = tid % kGroup;
  const int steps = (list_len + kGroup - 1) / kGroup;
  for (int l0 = 0; l0 < n_lists; l0 += nt / kGroup) {
    const int l = l0 + tid / kGroup;
    for (int st = 0; st < steps; ++st) {
      const int j = st * kGroup + sub;
      const uint64_t key = (l < n_lists && j < list_len) ? keys[static_cast<int64_t>(l) * list_len + j] : kKeyEmpty;
      const float a = key_score(key);
      const bool pass = key != kKeyEmpty && !(a < low);      // NaN scores pass (NaN rows rank first)
      if (pass && j == list_len - 1) sh.total = 1;            // the whole list is inside the band
      if (__builtin_amdgcn_ballot_w64(pass) == 0ull) break;
      refine_append(sh, pass, key, lane);
    }
  }
  __syncthreads();
  if (sh.total) return -2;
  return refine_rescore(scratch, n_candidates, sh, rf, q);
}

// Part 2: sh.count candidates in sh.sel2 -> exact scores -> the best n_candidates, sorted, in sh.sel.
template <int kMaxUnits, int kBatch>
__device__ __forceinline__ int refine_rescore_units(uint64_t* keys, int n_candidates, SelectShared& sh, const RefineParams& rf, int q) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
  const bool l2 = rf.space == DEWI_SPACE_L2;
  const int n_cand = static_cast<int>(sh.count);
  if (n_cand > kMaxSortCandidates) return -2;
  // exact scores, one wave per candidate, kBatch candidates per wave and round with all their loads in flight together
  // (the candidates of a query are a few dozen to a few hundred 3 KiB rows: latency, not bandwidth)
  // <4, 4>: l2 over an fp32 corpus on the matrix cores (dim 256 / 512 / 768) and the bf16 shadow up to dim 1024; <6, 2>: the shadow at
  // dim 1536 (six units per lane: two candidates per wave and round keep the row fragments inside the register budget)
  // lane l takes the 16-byte units l + 64 u of the row that lie inside it (round 4: any dim % 4 == 0 from 132 columns on — the lane
  // layout of scan_rows_any; the whole-KiB widths are the case where every unit exists: scan_rows_f32<U = dim / 256>)
  const int n4 = rf.dim >> 2;                      // 16-byte units per row
  const int units = (n4 + 63) >> 6;                // units per lane, the last one possibly partial
  bool have[kMaxUnits];
#pragma unroll
  for (int u = 0; u < kMaxUnits; ++u) have[u] = lane + 64 * u < n4;
  typedef float f32x4r __attribute__((ext_vector_type(4)));
  const f32x4r* qp = reinterpret_cast<const f32x4r*>(rf.Q + static_cast<int64_t>(q) * rf.dim) + lane;
  f32x4r qv[kMaxUnits];
#pragma unroll
  for (int u = 0; u < kMaxUnits; ++u) {
    qv[u] = f32x4r{0.f, 0.f, 0.f, 0.f};
    if (u < units && have[u]) qv[u] = qp[u * 64];
  }
  if (!l2) {   // cosine: the prepared query of scan_rows_f32 — float64 sum of squares in its order, one norm, __fdiv_rn
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < kMaxUnits; ++u)
      if (u < units) ss += square_f64(qv[u].x) + square_f64(qv[u].y) + square_f64(qv[u].z) + square_f64(qv[u].w);
    const float norm = wave_query_norm(ss);
    if (norm > 0.f) {
#pragma unroll
      for (int u = 0; u < kMaxUnits; ++u) {
        qv[u].x = __fdiv_rn(qv[u].x, norm);
        qv[u].y = __fdiv_rn(qv[u].y, norm);
        qv[u].z = __fdiv_rn(qv[u].z, norm);
        qv[u].w = __fdiv_rn(qv[u].w, norm);
      }
    }
  }
  for (int t0 = wave * kBatch; t0 < n_cand; t0 += n_waves * kBatch) {
    f32x4r e[kBatch][kMaxUnits];
    uint32_t rows[kBatch];
#pragma unroll
    for (int bb = 0; bb < kBatch; ++bb) {
      const int t = t0 + bb < n_cand ? t0 + bb : t0;
      rows[bb] = key_row(sh.sel2[t]);
      const f32x4r* ev = reinterpret_cast<const f32x4r*>(rf.E + static_cast<int64_t>(rows[bb]) * rf.dim) + lane;
#pragma unroll
      for (int u = 0; u < kMaxUnits; ++u) {
        e[bb][u] = f32x4r{0.f, 0.f, 0.f, 0.f};      // (a unit past the row's end: e = q = 0, fmaf(0, 0, acc) == acc — scan_rows_any's predicated lanes)
        if (u < units && have[u]) e[bb][u] = ev[u * 64];
      }
    }
#pragma unroll
    for (int bb = 0; bb < kBatch; ++bb) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < kMaxUnits; ++u) {         // u ascending, x y z w: scan_rows_f32's accum4.  Unrolled with a
        if (u < units) {                            // uniform guard: a run-time index would put e[][] into scratch memory
          if (l2) {
            float d;
            d = e[bb][u].x - qv[u].x; acc = __builtin_fmaf(d, d, acc);
            d = e[bb][u].y - qv[u].y; acc = __builtin_fmaf(d, d, acc);
            d = e[bb][u].z - qv[u].z; acc = __builtin_fmaf(d, d, acc);
            d = e[bb][u].w - qv[u].w; acc = __builtin_fmaf(d, d, acc);
          } else {                                  // accum4<cosine>
            acc = __builtin_fmaf(e[bb][u].x, qv[u].x, acc);
            acc = __builtin_fmaf(e[bb][u].y, qv[u].y, acc);
            acc = __builtin_fmaf(e[bb][u].z, qv[u].z, acc);
            acc = __builtin_fmaf(e[bb][u].w, qv[u].w, acc);
          }
        }
      }
      const float sum = wave_sum_f32(acc);
      const float sc = l2 ? -sum : sum;
      if (lane == 0 && t0 + bb < n_cand) keys[t0 + bb] = make_key(sc, rows[bb]);   // the dense array is scratch now
    }
  }
  __syncthreads();
  if (n_cand <= 1024) {          // one ranking pass (n^2 / threads compares, one barrier) beats the 45-66 barriers of the bitonic network here
    rank_sort_desc(keys, sh.sel, n_cand);
  } else {
    const int p2 = pow2_at_least(n_cand);
    for (int t = tid; t < p2; t += nt) sh.sel[t] = t < n_cand ? keys[t] : kKeyEmpty;
    bitonic_sort_desc<false>(sh.sel, nullptr, p2);
  }
  return n_cand < n_candidates ? n_cand : n_candidates;
}
__device__ __forceinline__ int refine_rescore(uint64_t* keys, int n_candidates, SelectShared& sh, const RefineParams& rf, int q) {
  if (rf.dim > 1024) return refine_rescore_units<6, 2>(keys, n_candidates, sh, rf, q);
  return refine_rescore_units<4, 4>(keys, n_candidates, sh, rf, q);
}

// Gathers the best n_candidates keys of one query into sh.sel, sorted descending; returns how many
// are valid.  Two routes:
//  * sorted lists (the scan's block-merged output: `sorted_lists` lists of n_candidates keys, each
//    sorted descending): the c-th largest LIST MAXIMUM is a lower bound of the c-th largest key
//    overall (the c largest maxima are c distinct keys), and only the c lists with the largest
//    maxima can hold a key at or above it.  So: rank the maxima (one barrier), let the owners of
//    those c lists walk them from the top until they drop below the bound (~1-2 reads each), and
//    rank-sort the c..2c survivors (one barrier).
//  * anything else: exact 8-bit MSB radix select over all keys, then compaction and bitonic sort.
__device__ int gather_top_candidates(const uint64_t* __restrict__ keys, int64_t keys_per_query, int sorted_lists,
                                     int n_candidates, SelectShared& sh) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  if (sorted_lists > 0 && sorted_lists <= 4 * kWave && n_candidates <= kWave) {
    // Up to 256 lists (one per CU — the usual case) and c <= 64.  ANY lower bound of the c-th largest
    // key works; it only has to be cheap and tight.  One wave: lane l loads the maxima of lists l,
    // l+64, l+128, l+192 and keeps the largest; the c-th largest of those 64 LANE maxima (a subset
    // of all maxima, so a valid bound, and at most a few ranks below the exact one) is found by
    // ranking across the wave with 64 readlanes.  ~4 K cycles; ranking all 256 maxima against each
    // other (65 K 64-bit compares on one CU) cost 11.6 K, pulling the c largest out one DPP
    // wave-maximum at a time 14 K.
    if (tid == 0) {
      sh.n_contrib = 0;
      sh.count = 0;
    }
    __syncthreads();
    if (tid < kWave) {
      uint64_t m[4];
      uint64_t lane_max = kKeyEmpty;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = tid + kWave * u;
        m[u] = l < sorted_lists ? keys[static_cast<int64_t>(l) * n_candidates] : kKeyEmpty;
        lane_max = m[u] > lane_max ? m[u] : lane_max;
      }
      const uint32_t mh = static_cast<uint32_t>(lane_max >> 32), ml = static_cast<uint32_t>(lane_max);
      int rank = 0;
#pragma unroll
      for (int j = 0; j < kWave; ++j) {
        const uint64_t o = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mh), j))) << 32) |
                           static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ml), j));
        rank += (o > lane_max || (o == lane_max && j < tid)) ? 1 : 0;
      }
      // the lane ranked c-th (0-based c-1) holds the bound — unless it is empty (fewer than c lists)
      const unsigned long long who = __ballot(rank == n_candidates - 1);
      uint64_t bound = 1ull;
      if (who != 0ull) {
        const int src = __ffsll(who) - 1;
        bound = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mh), src))) << 32) |
                static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ml), src));
        if (bound == kKeyEmpty) bound = 1ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (m[u] != kKeyEmpty && m[u] >= bound) sh.val[atomicAdd(&sh.n_contrib, 1u)] = static_cast<uint32_t>(tid + kWave * u);
      }
      if (tid == 0) sh.bound = bound;
    }
    __syncthreads();
    const uint64_t bound = sh.bound;
    for (int t = tid; t < static_cast<int>(sh.n_contrib); t += nt) {
      const uint64_t* lst = keys + static_cast<int64_t>(sh.val[t]) * n_candidates;
      for (int j = 0; j < n_candidates; ++j) {
        const uint64_t key = lst[j];
        if (key == kKeyEmpty || key < bound) break;
        const uint32_t pos = atomicAdd(&sh.count, 1u);
        if (pos < static_cast<uint32_t>(kMaxSortCandidates)) sh.sel2[pos] = key;
      }
    }
    __syncthreads();
    const int survivors = static_cast<int>(sh.count);
    if (survivors <= kRankSortMax) {
      rank_sort_desc(sh.sel2, sh.sel, survivors);
      return survivors < n_candidates ? survivors : n_candidates;
    }
    if (survivors <= kMaxSortCandidates) {
      const int p2 = pow2_at_least(survivors);
      for (int t = tid; t < p2; t += nt) sh.sel[t] = t < survivors ? sh.sel2[t] : kKeyEmpty;
      bitonic_sort_desc<false>(sh.sel, nullptr, p2);
      return survivors < n_candidates ? survivors : n_candidates;
    }
    __syncthreads();  // more survivors than LDS holds (degenerate input): exact select at the bottom
  } else
  if (sorted_lists > 0 && sorted_lists <= kMaxSortCandidates) {
    // list maxima (entry 0); entry 1 is loaded in the same round trip because the owner of a
    // contributing list almost always needs it a moment later
    uint64_t second = kKeyEmpty;
    for (int t = tid; t < sorted_lists; t += nt) {
      const uint64_t* lst = keys + static_cast<int64_t>(t) * n_candidates;
      sh.sel2[t] = lst[0];
      if (t == tid && n_candidates > 1) second = lst[1];
    }
    if (tid == 0) {
      sh.count = 0;
      sh.total = 0;       // "a list maximum ranked c-th" flag
      sh.pick_digit = 0;  // index of that list
    }
    __syncthreads();
    const bool all_lists = sorted_lists < n_candidates;
    // rank of every maximum = number of maxima that beat it.  The L x L comparisons are split over
    // all threads: `parts` threads share one list, each counting over a slice of the maxima.
    for (int t = tid; t < sorted_lists; t += nt) sh.val[t] = 0;
    __syncthreads();
    const int parts = nt / sorted_lists > 1 ? nt / sorted_lists : 1;
    const int slice = (sorted_lists + parts - 1) / parts;
    for (int idx = tid; idx < sorted_lists * parts; idx += nt) {
      const int t = idx / parts, part = idx % parts;
      const uint64_t mine = sh.sel2[t];
      const int j0 = part * slice, j1 = j0 + slice < sorted_lists ? j0 + slice : sorted_lists;
      uint32_t cnt = 0;
      for (int j = j0; j < j1; ++j) {
        const uint64_t o = sh.sel2[j];
        cnt += (o > mine || (o == mine && j < t)) ? 1u : 0u;
      }
      if (cnt) atomicAdd(&sh.val[t], cnt);
    }
    __syncthreads();
    for (int t = tid; t < sorted_lists; t += nt) {
      if (sh.val[t] == static_cast<uint32_t>(n_candidates - 1)) {
        sh.pick_digit = static_cast<uint32_t>(t);
        sh.total = 1;
      }
    }
    __syncthreads();
    uint64_t bound = 1ull;  // every non-empty key
    if (!all_lists && sh.total != 0) bound = sh.sel2[sh.pick_digit];
    if (bound == kKeyEmpty) bound = 1ull;
    __syncthreads();        // maxima in sel2 are dead from here on; sel2 receives the survivors
    for (int t = tid; t < sorted_lists; t += nt) {
      if (!all_lists && sh.val[t] >= static_cast<uint32_t>(n_candidates)) continue;
      const uint64_t* lst = keys + static_cast<int64_t>(t) * n_candidates;
      for (int j = 0; j < n_candidates; ++j) {
        const uint64_t key = (j == 1 && t == tid) ? second : lst[j];
        if (key == kKeyEmpty || key < bound) break;
        const uint32_t pos = atomicAdd(&sh.count, 1u);
        if (pos < static_cast<uint32_t>(kMaxSortCandidates)) sh.sel2[pos] = key;
      }
    }
    __syncthreads();
    const int survivors = static_cast<int>(sh.count);
    if (survivors <= kRankSortMax) {
      rank_sort_desc(sh.sel2, sh.sel, survivors);
      return survivors < n_candidates ? survivors : n_candidates;
    }
    if (survivors <= kMaxSortCandidates) {
      const int p2 = pow2_at_least(survivors);
      for (int t = tid; t < p2; t += nt) sh.sel[t] = t < survivors ? sh.sel2[t] : kKeyEmpty;
      bitonic_sort_desc<false>(sh.sel, nullptr, p2);
      return survivors < n_candidates ? survivors : n_candidates;
    }
    // more survivors than LDS holds (degenerate input): exact select below
  }
  return exact_top_candidates(ArrayKeys{keys, keys_per_query}, n_candidates, sh);
}

__global__ __launch_bounds__(kSelectThreads) void select_rerank_kernel(
    const uint64_t* __restrict__ keys_all, int64_t keys_per_query, int sorted_lists, int n_candidates, int k,
    RerankParams rp, const float* __restrict__ dewi32, const float* __restrict__ ent32, int64_t id_offset,
    int64_t* __restrict__ out_ids, float* __restrict__ out_scores, dewi_candidate* __restrict__ out_cand,
    const uint32_t* __restrict__ counts, SegmentLayout seg, RefineParams refine, QueryFlags flags) {
  __shared__ SelectShared sh;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int q = static_cast<int>(blockIdx.x);
  const uint64_t* keys = keys_all + static_cast<int64_t>(q) * keys_per_query;
  int n_sel;
  if (flags.mode == 2) {           // repair launch: only the queries the approximate pass refused (block-uniform)
    if (flags.p[q] == 0u) return;
  } else if (flags.mode == 1 && tid == 0) {
    flags.p[q] = 0u;               // (the same thread raises it in refuse(): program order)
  }
  auto refuse = [&]() {
    if (flags.mode == 1) {         // the repair launches behind this one answer the query on the exact row kernels
      if (tid == 0) flags.p[q] = 1u;
      return;
    }
    // no flags: ids -1 / records -2 mark a query the approximate pass could not answer
    if (out_cand != nullptr) {   // shard mode: every record of this query carries the marker id -2
      for (int j = tid; j < n_candidates; j += nt) {
        dewi_candidate rec;
        rec.sim = __builtin_nanf("");
        rec.dewi = 0.f;
        rec.ent = 0.f;
        rec.id = -2;
        out_cand[static_cast<int64_t>(q) * n_candidates + j] = rec;
      }
      return;
    }
    for (int j = tid; j < k; j += nt) {
      out_ids[static_cast<int64_t>(q) * k + j] = -1;
      out_scores[static_cast<int64_t>(q) * k + j] = __builtin_nanf("");
    }
  };
  if (counts != nullptr) {
    // Output of the batched matrix-core scan: `seg.n_seg` per-workgroup segments.  A count above the
    // segment capacity means that buffer overflowed and this query was NOT answered: the caller
    // re-runs it on the exact small-batch path (ids = -1 is the documented marker).
    SegmentKeys kv{keys_all + static_cast<int64_t>(q) * seg.cap, counts + q * seg.count_query_stride, seg.seg_stride, seg.count_stride, seg.n_seg,
                   seg.cap, seg.raw != 0};
    extern __shared__ __attribute__((aligned(16))) uint64_t dyn_keys[];
    __shared__ WideRadixShared ws;
    __shared__ uint32_t seg_base[kSelectThreads], seg_cnt[kSelectThreads];
    const uint32_t cap = static_cast<uint32_t>(seg.cap);
    if (tid == 0) sh.total = 0;
    __syncthreads();
    if (seg.n_seg <= nt && seg.raw != 0) {
      // One thread per half-segment: its count gives the overflow verdict and, through a workgroup
      // prefix sum, the place of its records in the dense LDS array — no position atomics, and the
      // counts are read once.
      const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
      uint32_t c = 0;
      if (tid < seg.n_seg) {
        c = kv.count[tid * kv.count_stride];
        if (c > cap) sh.total = 1;
        c = c < cap ? c : cap;
      }
      uint32_t incl = c;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
      }
      if (lane == 63) ws.wave_tot[wave] = incl;
      __syncthreads();
      if (sh.total) {
        refuse();
        return;
      }
      uint32_t before = 0, total = 0;
      for (int w = 0; w < n_waves; ++w) {
        const uint32_t wt = ws.wave_tot[w];
        before += w < wave ? wt : 0u;
        total += wt;
      }
      seg_base[tid] = before + incl - c;
      seg_cnt[tid] = c;
      __syncthreads();
      if (total <= static_cast<uint32_t>(seg.lds_keys)) {
        // Two lanes per half-segment, 8 predicated loads each issued back to back (a per-lane trip
        // count would make hipcc wait after every load); segments with more than 16 records loop.
        constexpr int kGroup = 2, kBatch = 8;
        const int sub = tid % kGroup;
        for (int sg = tid / kGroup; sg < seg.n_seg; sg += nt / kGroup) {
          const uint32_t n_rec = seg_cnt[sg], b = seg_base[sg];
          const uint64_t* sp = kv.p + sg * kv.seg_stride;
          for (uint32_t j0 = 0; j0 < n_rec; j0 += kGroup * kBatch) {
            uint64_t v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
              const uint32_t j = j0 + sub + u * kGroup;
              v[u] = j < n_rec ? sp[j] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
              const uint32_t j = j0 + sub + u * kGroup;
              if (j < n_rec)
                dyn_keys[b + j] = make_key(__uint_as_float(static_cast<uint32_t>(v[u])), static_cast<uint32_t>(v[u] >> 32));
            }
          }
        }
        __syncthreads();
        if (refine.E != nullptr) {
          n_sel = refine_top_candidates(dyn_keys, static_cast<int>(total), n_candidates, sh, ws, refine, q);
          if (n_sel == -2) {      // more candidates inside the error band than the sort holds: unanswered, re-run exactly
            refuse();
            return;
          }
        } else {
          n_sel = top_candidates_lds(dyn_keys, static_cast<int>(total), n_candidates, sh, ws);
        }
      } else if (refine.E != nullptr) {
        n_sel = refine_from_segments(kv, dyn_keys, n_candidates, sh, refine, q);   // more survivors than the staging holds
        if (n_sel == -2) {
          refuse();
          return;
        }
      } else {
        n_sel = exact_top_candidates(kv, n_candidates, sh);   // more survivors than LDS holds
      }
    } else {
      for (int s = tid; s < seg.n_seg; s += nt)
        if (kv.count[s * kv.count_stride] > cap) sh.total = 1;
      __syncthreads();
      if (sh.total) {
        refuse();
        return;
      }
      __syncthreads();
      if (refine.E != nullptr) {
        // more segments than threads (or ordered keys) with the exact refinement on: the scores are approximate here too —
        // widened cut and exact re-scoring straight from the segments, never the approximate keys as they stand
        n_sel = refine_from_segments(kv, dyn_keys, n_candidates, sh, refine, q);
        if (n_sel == -2) {
          refuse();
          return;
        }
      } else {
        // Gather the valid records of the half-segments into LDS once; the radix-select passes then run
        // over a dense LDS array instead of re-walking global memory five times.
        if (tid == 0) sh.count = 0;
        __syncthreads();
        kv.for_each(tid, nt, [&](uint64_t key) {
          const uint32_t pos = atomicAdd(&sh.count, 1u);
          if (pos < static_cast<uint32_t>(seg.lds_keys)) dyn_keys[pos] = key;
        });
        __syncthreads();
        const uint32_t total = sh.count;
        __syncthreads();
        if (total <= static_cast<uint32_t>(seg.lds_keys))
          n_sel = exact_top_candidates(ArrayKeys{dyn_keys, static_cast<int64_t>(total)}, n_candidates, sh);
        else
          n_sel = exact_top_candidates(kv, n_candidates, sh);
      }
    }
  } else if (refine.E != nullptr && refine.list_len > 0) {
    extern __shared__ __attribute__((aligned(16))) uint64_t dyn_scratch[];     // kMaxSortCandidates keys (launcher)
    n_sel = refine_from_sorted_lists(keys, sorted_lists, refine.list_len, n_candidates, sh, dyn_scratch, refine, q);
    if (n_sel == -2) {
      refuse();
      return;
    }
  } else {
    n_sel = gather_top_candidates(keys, keys_per_query, sorted_lists, n_candidates, sh);
  }
  if (out_cand != nullptr) {
    dewi_candidate* oc = out_cand + static_cast<int64_t>(q) * n_candidates;
    for (int t = tid; t < n_candidates; t += nt) {
      dewi_candidate rec;
      if (t < n_sel) {
        const uint32_t row = key_row(sh.sel[t]);
        rec.sim = key_score(sh.sel[t]);
        rec.dewi = dewi32[row];
        rec.ent = ent32[row];
        rec.id = static_cast<int32_t>(static_cast<int64_t>(row) + id_offset);
      } else {
        rec.sim = -__builtin_inff();
        rec.dewi = 0.f;
        rec.ent = 0.f;
        rec.id = -1;
      }
      oc[t] = rec;
    }
    return;
  }
  auto fetch = [&](int t, float& dewi, float& ent, int64_t& id) {
    const uint32_t row = key_row(sh.sel[t]);
    dewi = dewi32[row];
    ent = ent32[row];
    id = static_cast<int64_t>(row) + id_offset;
  };
  rerank_and_emit(sh, n_sel, k, rp, fetch, out_ids + static_cast<int64_t>(q) * k,
                  out_scores + static_cast<int64_t>(q) * k);
}

// ---------------------------------------------------------------------------------------------
// Multi-shard merge: n_lists sorted candidate lists per query -> global top-c -> blend -> top-k.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSelectThreads) void merge_rerank_kernel(const dewi_candidate* __restrict__ lists,
                                                                      int n_lists, int n_queries, int list_len,
                                                                      int n_candidates, int k, RerankParams rp,
                                                                      int64_t* __restrict__ out_ids,
                                                                      float* __restrict__ out_scores) {
  __shared__ SelectShared sh;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int q = static_cast<int>(blockIdx.x);
  const int m = n_lists * list_len;
  const int p2 = pow2_at_least(m);
  if (tid == 0) {
    sh.count = 0;
    sh.total = 0;
  }
  __syncthreads();
  const bool small = m <= kRankSortMax;   // the usual case (8 shards x 2k records): one ranking pass
  for (int t = tid; t < p2; t += nt) {
    uint64_t key = kKeyEmpty;
    uint32_t src = 0;
    if (t < m) {
      const int l = t / list_len, j = t % list_len;
      const int64_t at = (static_cast<int64_t>(l) * n_queries + q) * list_len + j;
      const dewi_candidate rec = lists[at];
      if (rec.id == -2) sh.total = 1;   // that shard's batched path overflowed: the query is unanswered
      if (rec.id >= 0) {
        key = make_key(rec.sim, static_cast<uint32_t>(rec.id));
        src = static_cast<uint32_t>(t);
        atomicAdd(&sh.count, 1u);
      }
    }
    if (small) {
      sh.sel2[t] = key;
    } else {
      sh.sel[t] = key;
      sh.val[t] = src;
    }
  }
  __syncthreads();
  if (sh.total) {   // same marker as the single-device batched path: ids -1, the caller re-runs the query
    for (int j = tid; j < k; j += nt) {
      out_ids[static_cast<int64_t>(q) * k + j] = -1;
      out_scores[static_cast<int64_t>(q) * k + j] = __builtin_nanf("");
    }
    return;
  }
  if (small) {
    for (int t = tid; t < m; t += nt) {
      const uint64_t mine = sh.sel2[t];
      int rank = 0;
      for (int j = 0; j < m; ++j) {
        const uint64_t o = sh.sel2[j];
        rank += (o > mine || (o == mine && j < t)) ? 1 : 0;
      }
      sh.sel[rank] = mine;
      sh.val[rank] = static_cast<uint32_t>(t);
    }
    __syncthreads();
  } else {
    bitonic_sort_desc<true>(sh.sel, sh.val, p2);
  }
  const int n_sel = static_cast<int>(sh.count < static_cast<uint32_t>(n_candidates) ? sh.count : n_candidates);
  auto fetch = [&](int t, float& dewi, float& ent, int64_t& id) {
    const int s = static_cast<int>(sh.val[t]);
    const int l = s / list_len, j = s % list_len;
    const dewi_candidate rec = lists[(static_cast<int64_t>(l) * n_queries + q) * list_len + j];
    dewi = rec.dewi;
    ent = rec.ent;
    id = rec.id;
  };
  rerank_and_emit(sh, n_sel, k, rp, fetch, out_ids + static_cast<int64_t>(q) * k,
                  out_scores + static_cast<int64_t>(q) * k);
}

// ---------------------------------------------------------------------------------------------
// Large candidate counts (c > kMaxSortCandidates, i.e. k > 1024): same steps with the candidate
// arrays in global memory instead of LDS.  One workgroup per query; g1/g2 are [n_queries][p2]
// scratch arrays (p2 = power of two >= c).  Not a fast path — it exists so that every k the
// reference accepts (up to k == N) is answered, on one device and over doc-id shards alike.
// ---------------------------------------------------------------------------------------------
__device__ void bitonic_sort_desc_global(uint64_t* key, int p2) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  for (int size = 2; size <= p2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();  // all waves of the workgroup share one CU and its L1: block-level visibility
      for (int t = tid; t < (p2 >> 1); t += nt) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool first_half = (lo & size) == 0;
        const uint64_t a = key[lo], b = key[hi];
        if (first_half ? (a < b) : (a > b)) {
          key[lo] = b;
          key[hi] = a;
        }
      }
    }
  }
  __syncthreads();
}

// out_cand == nullptr: final (ids, scores), id_offset added.  out_cand != nullptr: the shard's n_out records
// (the first n_sel real, the rest padding), exactly as select_rerank_kernel writes them.
__global__ __launch_bounds__(kSelectThreads) void select_rerank_large_kernel(
    const uint64_t* __restrict__ keys_all, int64_t keys_per_query, int n_candidates, int p2, int k, RerankParams rp,
    const float* __restrict__ dewi32, const float* __restrict__ ent32, int64_t id_offset, uint64_t* __restrict__ g1_all,
    uint64_t* __restrict__ g2_all, int64_t* __restrict__ out_ids, float* __restrict__ out_scores,
    dewi_candidate* __restrict__ out_cand, int n_out) {
  __shared__ SelectShared sh;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int q = static_cast<int>(blockIdx.x);
  const ArrayKeys keys{keys_all + static_cast<int64_t>(q) * keys_per_query, keys_per_query};
  uint64_t* g1 = g1_all + static_cast<int64_t>(q) * p2;
  uint64_t* g2 = g2_all + static_cast<int64_t>(q) * p2;
  const uint64_t thr = block_kth_largest(keys, static_cast<uint32_t>(n_candidates), sh);
  if (tid == 0) sh.count = 0;
  for (int t = tid; t < p2; t += nt) g1[t] = kKeyEmpty;
  __syncthreads();
  keys.for_each(tid, nt, [&](uint64_t key) {
    if (key >= thr) {
      const uint32_t pos = atomicAdd(&sh.count, 1u);
      if (pos < static_cast<uint32_t>(p2)) g1[pos] = key;
    }
  });
  __syncthreads();
  const int n_sel = static_cast<int>(sh.count < static_cast<uint32_t>(n_candidates) ? sh.count : n_candidates);
  bitonic_sort_desc_global(g1, p2);                       // (sim desc, row asc)
  if (out_cand != nullptr) {
    dewi_candidate* oc = out_cand + static_cast<int64_t>(q) * n_out;
    for (int t = tid; t < n_out; t += nt) {
      dewi_candidate rec;
      if (t < n_sel) {
        const uint64_t key = g1[t];
        const uint32_t row = key_row(key);
        rec.sim = key_score(key);
        rec.dewi = dewi32[row];
        rec.ent = ent32[row];
        rec.id = static_cast<int32_t>(static_cast<int64_t>(row) + id_offset);
      } else {
        rec.sim = -__builtin_inff();
        rec.dewi = 0.f;
        rec.ent = 0.f;
        rec.id = -1;
      }
      oc[t] = rec;
    }
    return;
  }
  for (int t = tid; t < p2; t += nt) {
    uint64_t k2 = kKeyEmpty;
    if (t < n_sel) {
      const uint32_t row = key_row(g1[t]);
      const float adj = blend(rp, key_score(g1[t]), dewi32[row], ent32[row]);
      k2 = (static_cast<uint64_t>(ord_f32(adj)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - static_cast<uint32_t>(t));
    }
    g2[t] = k2;
  }
  bitonic_sort_desc_global(g2, p2);
  for (int j = tid; j < k && j < n_sel; j += nt) {
    const uint64_t k2 = g2[j];
    const uint32_t t = 0xFFFFFFFFu - static_cast<uint32_t>(k2);
    out_ids[static_cast<int64_t>(q) * k + j] = static_cast<int64_t>(key_row(g1[t])) + id_offset;
    out_scores[static_cast<int64_t>(q) * k + j] = unord_f32(static_cast<uint32_t>(k2 >> 32));
  }
}

hipError_t launch_select_rerank_large(const uint64_t* d_keys, int64_t keys_per_query, int n_queries, int n_candidates,
                                      int p2, int k, const RerankParams& rp, const float* d_dewi32,
                                      const float* d_ent32, int64_t id_offset, uint64_t* d_g1, uint64_t* d_g2,
                                      int64_t* d_out_ids, float* d_out_scores, dewi_candidate* d_out_cand, int n_out,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(select_rerank_large_kernel, dim3(n_queries), dim3(kSelectThreads), 0, stream, d_keys,
                     keys_per_query, n_candidates, p2, k, rp, d_dewi32, d_ent32, id_offset, d_g1, d_g2, d_out_ids,
                     d_out_scores, d_out_cand, n_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Multi-shard merge with more than kMaxSortCandidates records per query (k > 128 at eight shards): the
// shard lists are SORTED (sim desc, id asc; padding at the tail), so a record's global rank is its position
// in its own list plus, for every other list, the number of records there that beat it — one binary search
// per (record, other list), no sort of the concatenation.  Records ranked below n_candidates drop out; the
// survivors land in g1 (keys, in rank order) and src (where the record lives).  Then the blend, a bitonic
// sort of the adjusted keys in global memory, and the first k.  Scratch per query: g1 [p2] u64, g2 [p2] u64,
// src [p2] u32 with p2 = power of two >= n_candidates.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t record_key(const dewi_candidate& rec) {
  return rec.id >= 0 ? make_key(rec.sim, static_cast<uint32_t>(rec.id)) : kKeyEmpty;
}

__global__ __launch_bounds__(kSelectThreads) void merge_rerank_large_kernel(
    const dewi_candidate* __restrict__ lists, int n_lists, int n_queries, int list_len, int n_candidates, int p2, int k,
    RerankParams rp, uint64_t* __restrict__ g1_all, uint64_t* __restrict__ g2_all, uint32_t* __restrict__ src_all,
    int64_t* __restrict__ out_ids, float* __restrict__ out_scores) {
  __shared__ uint32_t n_valid, refused;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int q = static_cast<int>(blockIdx.x);
  uint64_t* g1 = g1_all + static_cast<int64_t>(q) * p2;
  uint64_t* g2 = g2_all + static_cast<int64_t>(q) * p2;
  uint32_t* src = src_all + static_cast<int64_t>(q) * p2;
  auto list_of = [&](int l) { return lists + (static_cast<int64_t>(l) * n_queries + q) * list_len; };
  if (tid == 0) {
    n_valid = 0;
    refused = 0;
  }
  for (int t = tid; t < p2; t += nt) g1[t] = kKeyEmpty;
  __syncthreads();
  const int64_t m = static_cast<int64_t>(n_lists) * list_len;
  for (int64_t t = tid; t < m; t += nt) {
    const int l = static_cast<int>(t / list_len), j = static_cast<int>(t % list_len);
    const dewi_candidate rec = list_of(l)[j];
    if (rec.id == -2) refused = 1;   // that shard's batched path overflowed: the query is unanswered
    if (rec.id < 0) continue;
    atomicAdd(&n_valid, 1u);
    const uint64_t mine = record_key(rec);
    int64_t rank = j;                // every record ahead of it in its own list beats it
    for (int o = 0; o < n_lists && rank < n_candidates; ++o) {
      if (o == l) continue;
      const dewi_candidate* other = list_of(o);
      int lo = 0, hi = list_len;     // first position of `other` that does NOT beat `mine`
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t ok = record_key(other[mid]);
        const bool beats = ok > mine || (ok == mine && o < l);
        if (beats) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    if (rank < n_candidates) {
      g1[rank] = mine;
      src[rank] = static_cast<uint32_t>(t);
    }
  }
  __syncthreads();
  if (refused) {   // same marker as the single-device batched path: ids -1, the caller re-runs the query
    for (int j = tid; j < k; j += nt) {
      out_ids[static_cast<int64_t>(q) * k + j] = -1;
      out_scores[static_cast<int64_t>(q) * k + j] = __builtin_nanf("");
    }
    return;
  }
  const int n_sel = static_cast<int>(n_valid < static_cast<uint32_t>(n_candidates) ? n_valid : n_candidates);
  auto record_at = [&](int t) {
    const uint32_t s = src[t];
    return list_of(static_cast<int>(s / list_len))[s % list_len];
  };
  for (int t = tid; t < p2; t += nt) {
    uint64_t k2 = kKeyEmpty;
    if (t < n_sel) {
      const dewi_candidate rec = record_at(t);
      const float adj = blend(rp, rec.sim, rec.dewi, rec.ent);
      k2 = (static_cast<uint64_t>(ord_f32(adj)) << 32) | static_cast<uint64_t>(0xFFFFFFFFu - static_cast<uint32_t>(t));
    }
    g2[t] = k2;
  }
  bitonic_sort_desc_global(g2, p2);
  for (int j = tid; j < k && j < n_sel; j += nt) {
    const uint64_t k2 = g2[j];
    const int t = static_cast<int>(0xFFFFFFFFu - static_cast<uint32_t>(k2));
    out_ids[static_cast<int64_t>(q) * k + j] = record_at(t).id;
    out_scores[static_cast<int64_t>(q) * k + j] = unord_f32(static_cast<uint32_t>(k2 >> 32));
  }
}

size_t merge_large_workspace_bytes(int n_queries, int n_candidates) {
  int p2 = 2;
  while (p2 < n_candidates) p2 <<= 1;
  return static_cast<size_t>(n_queries) * p2 * (8 + 8 + 4);
}

hipError_t launch_merge_rerank_large(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len,
                                     int n_candidates, int k, const RerankParams& rp, void* d_ws, int64_t* d_out_ids,
                                     float* d_out_scores, hipStream_t stream) {
  int p2 = 2;
  while (p2 < n_candidates) p2 <<= 1;
  uint64_t* g1 = static_cast<uint64_t*>(d_ws);
  uint64_t* g2 = g1 + static_cast<size_t>(n_queries) * p2;
  uint32_t* src = reinterpret_cast<uint32_t*>(g2 + static_cast<size_t>(n_queries) * p2);
  hipLaunchKernelGGL(merge_rerank_large_kernel, dim3(n_queries), dim3(kSelectThreads), 0, stream, d_lists, n_lists,
                     n_queries, list_len, n_candidates, p2, k, rp, g1, g2, src, d_out_ids, d_out_scores);
  return hipGetLastError();
}

hipError_t launch_select_rerank(const uint64_t* d_keys, int64_t keys_per_query, int sorted_lists, int n_queries,
                                int n_candidates, int k, const RerankParams& rp, const float* d_dewi32,
                                const float* d_ent32, int64_t id_offset, int64_t* d_out_ids, float* d_out_scores,
                                dewi_candidate* d_out_cand, const uint32_t* d_counts, const SegmentLayout& seg,
                                hipStream_t stream, const RefineParams& refine, const QueryFlags& flags) {
  int threads = kSelectThreads;
  const bool refine_lists = refine.E != nullptr && refine.list_len > 0;   // one query through the shadow on the row kernel
  if (refine_lists) {
    if (d_counts != nullptr || sorted_lists <= 0 || sorted_lists > 4 * kWave || refine.list_len > kWave ||
        n_candidates > refine.list_len || refine.space != DEWI_SPACE_COSINE)
      return hipErrorInvalidValue;
  } else if (sorted_lists > 0) {
    threads = sorted_lists <= 4 * kWave ? 256 : kSelectThreads;  // <= 256 lists: one wave finds the bound
  } else if (keys_per_query <= 4096 && n_candidates <= 128 && refine.E == nullptr) {
    threads = 256;   // (refine mode re-scores its candidates one wave each: sixteen waves)
  }
  // the repair's select returns at once for every query but the refused ones (rare): launch it small — 4 waves per query
  // instead of 16 cost the usual batch less to dispatch, and a refused query's select is not where its time goes
  if (flags.mode == 2) threads = 256;
  // DEWI_SELECT_THREADS (tests only, 256 .. 1024): fewer threads than survivor segments puts a matrix-core batch on the
  // kernel's more-segments-than-threads route
  static const int threads_override = [] { const char* e = getenv("DEWI_SELECT_THREADS"); return e ? atoi(e) : 0; }();
  if (d_counts != nullptr && threads_override >= 256 && threads_override <= kSelectThreads && threads_override % kWave == 0)
    threads = threads_override;
  size_t dyn = refine_lists ? static_cast<size_t>(kMaxSortCandidates) * 8 : 0;     // scratch of the exact re-scoring
  if (d_counts != nullptr && seg.lds_keys > 0) {
    // the staging limit (seg.lds_keys, which tests shrink) is one thing, the allocation another: the exact re-scoring uses
    // the dynamic array as scratch for up to kMaxSortCandidates keys whatever was staged
    const int alloc_keys = (refine.E != nullptr && seg.lds_keys < kMaxSortCandidates) ? kMaxSortCandidates : seg.lds_keys;
    dyn = static_cast<size_t>(alloc_keys) * 8;
    static PerDeviceOnce attr_once;
    const hipError_t e = attr_once.run([] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&select_rerank_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    });
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(select_rerank_kernel, dim3(n_queries), dim3(threads), dyn, stream, d_keys, keys_per_query,
                     sorted_lists, n_candidates, k, rp, d_dewi32, d_ent32, id_offset, d_out_ids, d_out_scores,
                     d_out_cand, d_counts, seg, refine, flags);
  return hipGetLastError();
}

hipError_t launch_merge_rerank(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len,
                               int n_candidates, int k, const RerankParams& rp, int64_t* d_out_ids,
                               float* d_out_scores, hipStream_t stream) {
  const int threads = n_lists * list_len <= 512 ? 256 : kSelectThreads;
  hipLaunchKernelGGL(merge_rerank_kernel, dim3(n_queries), dim3(threads), 0, stream, d_lists, n_lists, n_queries,
                     list_len, n_candidates, k, rp, d_out_ids, d_out_scores);
  return hipGetLastError();
}

}  // namespace dewi
