// Corpus scan over a bf16 embedding matrix (config C3's storage format) for a small group of
// queries, gfx950 (MI355X).  Same contract as knn_scan.hip: steps 1-3 of ExactIndex.search
// (reference src/dewi/backends.py:420-444) with the corpus rounded to bf16 after normalisation and
// the normalised query rounded to bf16; every product is exact in fp32 and sums are fp32.
//
// Roofline: HBM.  Algorithmic bytes per launch = n_rows * dim * 2 + n_queries * dim * 4.
//
// Data movement (fast path, dim = 256*H): a bf16 row is H/2 KiB, so rows are taken in PAIRS — one
// pair is H contiguous KiB = H global_load_dwordx4 wave-instructions of 1 KiB each.  Lane l of
// load j holds 8 consecutive columns of row (64j+l >= 32H); for odd H the middle load straddles
// the two rows at lane 32.  Each lane keeps the 8 query values that match its columns, per load,
// in registers; two DPP wave reductions give the two row scores.  One pair (3 KiB at dim 768) is
// in flight per wave, 8 waves per CU — the shape that measured best on the fp32 kernel.
#include "scan_common.hpp"

#ifndef DEWI_BF16_PAIRS
#define DEWI_BF16_PAIRS 2
#endif

namespace dewi {

template <int H, int NQ, int SPACE, int S, bool NT>
__device__ __forceinline__ void scan_rows_bf16_body(const uint16_t* __restrict__ E, int64_t n_rows,
                                                    const float* __restrict__ Q, int n_candidates,
                                                    uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                    MergeShared& merge_buf) {
  constexpr int D = 256 * H;          // columns
  constexpr int UPR = 32 * H;         // 16-byte units per row
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);

  // which row of the pair, and which columns, this lane sees in load j
  bool second[H];
  float qf[NQ][H][8];
#pragma unroll
  for (int j = 0; j < H; ++j) second[j] = (64 * j + lane) >= UPR;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    double ss = 0.0;   // float64 sum of squares: the same norm in every kernel (common.hpp, wave_query_norm)
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int unit = 64 * j + lane - (second[j] ? UPR : 0);
      const f32x4* qp = reinterpret_cast<const f32x4*>(Q + static_cast<int64_t>(qi) * D) + 2 * unit;
      const f32x4 a = qp[0], b = qp[1];
      qf[qi][j][0] = a.x; qf[qi][j][1] = a.y; qf[qi][j][2] = a.z; qf[qi][j][3] = a.w;
      qf[qi][j][4] = b.x; qf[qi][j][5] = b.y; qf[qi][j][6] = b.z; qf[qi][j][7] = b.w;
      if (!second[j]) {  // the first-row units cover every column exactly once
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += square_f64(qf[qi][j][i]);
      }
    }
    float norm = 1.f;
    bool scale = false;
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      norm = wave_query_norm(ss);
      scale = norm > 0.f;  // reference backends.py:422-424
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = scale ? __fdiv_rn(qf[qi][j][i], norm) : qf[qi][j][i];
        qf[qi][j][i] = round_to_bf16(v);
      }
    }
  }

  // cosine: keep the query as packed bf16 pairs (half the registers, feeds v_dot2c directly)
  uint32_t qp[NQ][H][4];
  if constexpr (SPACE == DEWI_SPACE_COSINE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
      for (int j = 0; j < H; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          qp[qi][j][i] = (__float_as_uint(qf[qi][j][2 * i]) >> 16) | (__float_as_uint(qf[qi][j][2 * i + 1]) & 0xFFFF0000u);
      }
    }
  }
  auto dot = [&](u32x4 e, int qi, int j, float acc) {
    if constexpr (SPACE == DEWI_SPACE_COSINE) return dot8_packed(e, qp[qi][j], acc);
    else return dot8<SPACE>(e, qf[qi][j], acc);
  };

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  auto finish_row = [&](int qi, float acc, int64_t row) {
    float s = wave_sum_f32(acc);
    if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
    if constexpr (DENSE) {
      if (lane == 0) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
    } else {
      lst[qi].offer(s, static_cast<uint32_t>(row), lane);
    }
  };
  auto consume_pair = [&](const u32x4(&v)[H], int64_t row0, bool has_second) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (2 * j + 1 < H) {            // whole load in the first row
          a0 = dot(v[j], qi, j, a0);
        } else if (2 * j >= H) {        // whole load in the second row
          a1 = dot(v[j], qi, j, a1);
        } else {                        // odd H: lanes 0-31 first row, lanes 32-63 second row
          const float t = dot(v[j], qi, j, 0.f);
          a0 += second[j] ? 0.f : t;
          a1 += second[j] ? t : 0.f;
        }
      }
      finish_row(qi, a0, row0);
      if (has_second) finish_row(qi, a1, row0 + 1);
    }
  };

  const u32x4* Ev = reinterpret_cast<const u32x4*>(E);
  const int64_t n_pairs = n_rows >> 1;
  int64_t p = gwave;
  // DEWI_BF16_PAIRS row pairs per trip: that many x H independent 1 KiB loads are in flight before the
  // first reduction starts (a bf16 row pair carries half the bytes of an fp32 row per reduction, so one
  // pair per trip left the memory pipe waiting on the DPP chains: 230 us per 1 M x 768 pass; two: 220).
  constexpr int P = DEWI_BF16_PAIRS;
  if constexpr (P > 1) {
    for (; p + (P - 1) * n_waves < n_pairs; p += P * n_waves) {
      u32x4 v[P][H];
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const u32x4* base = Ev + (p + u * n_waves) * (2 * UPR) + lane;
#pragma unroll
        for (int j = 0; j < H; ++j) v[u][j] = load_u4<NT>(base + 64 * j);
      }
#pragma unroll
      for (int u = 0; u < P; ++u) consume_pair(v[u], 2 * (p + u * n_waves), true);
    }
  }
  for (; p < n_pairs; p += n_waves) {
    const u32x4* base = Ev + p * (2 * UPR) + lane;
    u32x4 v[H];
#pragma unroll
    for (int j = 0; j < H; ++j) v[j] = load_u4<NT>(base + 64 * j);
    consume_pair(v, 2 * p, true);
  }
  if ((n_rows & 1) && (n_pairs % n_waves) == gwave) {  // last, unpaired row: first-row units only
    const u32x4* base = Ev + n_pairs * (2 * UPR) + lane;
    u32x4 v[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
      v[j] = zero;
      if (!second[j]) v[j] = load_u4<NT>(base + 64 * j);
    }
    consume_pair(v, 2 * n_pairs, false);
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

template <int H, int NQ, int SPACE, int S, bool NT>
__global__ __launch_bounds__(kScanThreads) void scan_rows_bf16(const uint16_t* __restrict__ E, int64_t n_rows,
                                                               const float* __restrict__ Q, int n_candidates,
                                                               uint64_t* __restrict__ keys,
                                                               int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_rows_bf16_body<H, NQ, SPACE, S, NT>(E, n_rows, Q, n_candidates, keys, keys_per_query, merge_buf);
}

// REPAIR form: every flagged query of a batch, one corpus pass each, in one launch (see scan_rows_f32_flagged)
template <int H, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_bf16_flagged(const uint16_t* __restrict__ E, int64_t n_rows,
                                                                       const float* __restrict__ Q, int n_candidates,
                                                                       uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                                       const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_rows_bf16_body<H, 1, SPACE, S, true>(E, n_rows, Q + static_cast<int64_t>(q) * (256 * H), n_candidates,
                                              keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// Generic path: any dim.  G lanes per row; queries arrive normalised and bf16-rounded as fp32.
// VEC = 8 (16-byte loads) when dim % 8 == 0, else scalar.
// ---------------------------------------------------------------------------------------------
template <int VEC, int NQ, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_generic_bf16(const uint16_t* __restrict__ E, int64_t n_rows, int dim,
                                                                  const float* __restrict__ Qn, int group,
                                                                  int n_candidates, uint64_t* __restrict__ keys,
                                                                  int64_t keys_per_query) {
  constexpr bool DENSE = S == 0;
  __shared__ MergeShared merge_buf;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int rows_per_step = kWave / group;
  const int sub = lane / group, lg = lane % group;
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
      const uint16_t* ep = E + row * dim;
      for (int u = lg; u < units; u += group) {
        if constexpr (VEC == 8) {
          const u32x4 e = reinterpret_cast<const u32x4*>(ep)[u];
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            const f32x4* qp = reinterpret_cast<const f32x4*>(Qn + static_cast<int64_t>(qi) * dim) + 2 * u;
            const f32x4 a = qp[0], b = qp[1];
            const float q8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            acc[qi] = dot8<SPACE>(e, q8, acc[qi]);
          }
        } else {
          const float e = __uint_as_float(static_cast<uint32_t>(ep[u]) << 16);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            const float q = Qn[static_cast<int64_t>(qi) * dim + u];
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
// dispatch
// ---------------------------------------------------------------------------------------------
template <int H, int NQ, int SPACE, int S>
static void launch_fast_h(const ScanPlan& plan, const uint16_t* E, int64_t n_rows, const float* Q, int c, uint64_t* keys,
                          hipStream_t stream) {
  if (plan.nontemporal)
    hipLaunchKernelGGL((scan_rows_bf16<H, NQ, SPACE, S, true>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, Q, c, keys, plan.keys_per_query);
  else
    hipLaunchKernelGGL((scan_rows_bf16<H, NQ, SPACE, S, false>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, Q, c, keys, plan.keys_per_query);
}

template <int NQ, int SPACE, int S>
static hipError_t launch_bf16_impl(const ScanPlan& plan, const uint16_t* E, int64_t n_rows, int dim, const float* Qraw,
                                   const float* Qn, int c, uint64_t* keys, hipStream_t stream) {
  if (plan.fast) {
    switch (dim / 256) {
      case 1: launch_fast_h<1, NQ, SPACE, S>(plan, E, n_rows, Qraw, c, keys, stream); break;
      case 2: launch_fast_h<2, NQ, SPACE, S>(plan, E, n_rows, Qraw, c, keys, stream); break;
      case 3: launch_fast_h<3, NQ, SPACE, S>(plan, E, n_rows, Qraw, c, keys, stream); break;
      case 4: launch_fast_h<4, NQ, SPACE, S>(plan, E, n_rows, Qraw, c, keys, stream); break;
      default: return hipErrorInvalidValue;
    }
  } else if (plan.vec == 8) {
    hipLaunchKernelGGL((scan_generic_bf16<8, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, dim, Qn, plan.group, c, keys, plan.keys_per_query);
  } else {
    hipLaunchKernelGGL((scan_generic_bf16<1, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E,
                       n_rows, dim, Qn, plan.group, c, keys, plan.keys_per_query);
  }
  return hipGetLastError();
}

template <int H, int SPACE>
static hipError_t launch_flagged_bf16_s(const ScanPlan& plan, const uint16_t* E, int64_t n_rows, const float* Q, int n_queries, int c,
                                        uint64_t* keys, const uint32_t* flags, hipStream_t stream) {
  switch (plan.slots) {
    case 0:
      hipLaunchKernelGGL((scan_rows_bf16_flagged<H, SPACE, 0>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, Q, c, keys,
                         plan.keys_per_query, flags, n_queries);
      break;
    case 1:
      hipLaunchKernelGGL((scan_rows_bf16_flagged<H, SPACE, 1>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, Q, c, keys,
                         plan.keys_per_query, flags, n_queries);
      break;
    default:
      hipLaunchKernelGGL((scan_rows_bf16_flagged<H, SPACE, kMaxSlots>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, Q,
                         c, keys, plan.keys_per_query, flags, n_queries);
      break;
  }
  return hipGetLastError();
}

hipError_t launch_scan_flagged_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                    int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                    hipStream_t stream) {
  if (plan.kind == kScanAnyLong || plan.kind == kScanAnyShort)
    return launch_scan_any_flagged_bf16(plan, d_E, n_rows, dim, d_q_raw, n_queries, n_candidates, space, d_keys, d_flags, stream);
  if (plan.kind != kScanFast) return hipErrorInvalidValue;
#define DEWI_FLAGGED(HH)                                                                                                      \
  case HH:                                                                                                                    \
    return space == DEWI_SPACE_COSINE                                                                                         \
               ? launch_flagged_bf16_s<HH, DEWI_SPACE_COSINE>(plan, d_E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream) \
               : launch_flagged_bf16_s<HH, DEWI_SPACE_L2>(plan, d_E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream);
  switch (dim / 256) {
    DEWI_FLAGGED(1)
    DEWI_FLAGGED(2)
    DEWI_FLAGGED(3)
    DEWI_FLAGGED(4)
    default: break;
  }
#undef DEWI_FLAGGED
  return hipErrorInvalidValue;
}

hipError_t launch_scan_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                            const float* d_q_norm, int q0, int nq, int n_candidates, int space, uint64_t* d_keys,
                            hipStream_t stream) {
  if (plan.odd_rows) return launch_scan_odd_bf16(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
  if (plan.kind == kScanAnyLong || plan.kind == kScanAnyShort)
    return launch_scan_any_bf16(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
  const float* qr = d_q_raw + static_cast<int64_t>(q0) * dim;
  const float* qn = d_q_norm ? d_q_norm + static_cast<int64_t>(q0) * dim : nullptr;
  uint64_t* keys = d_keys + static_cast<int64_t>(q0) * plan.keys_per_query;
#define DEWI_BF16_S(NQ, SPACE)                                                                                  \
  switch (plan.slots) {                                                                                         \
    case 0: return launch_bf16_impl<NQ, SPACE, 0>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);  \
    case 1: return launch_bf16_impl<NQ, SPACE, 1>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream);  \
    default: return launch_bf16_impl<NQ, SPACE, kMaxSlots>(plan, d_E, n_rows, dim, qr, qn, n_candidates, keys, stream); \
  }
#define DEWI_BF16(NQ)                    \
  if (space == DEWI_SPACE_COSINE) {      \
    DEWI_BF16_S(NQ, DEWI_SPACE_COSINE)   \
  } else {                               \
    DEWI_BF16_S(NQ, DEWI_SPACE_L2)       \
  }
  if (nq == 1) {
    DEWI_BF16(1)
  } else if (nq == 4) {
    DEWI_BF16(4)
  }
#undef DEWI_BF16
#undef DEWI_BF16_S
  return hipErrorInvalidValue;
}

}  // namespace dewi
