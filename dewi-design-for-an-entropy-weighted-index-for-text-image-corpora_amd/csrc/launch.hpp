// Host-side launch declarations shared by abi.cpp and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/dewi_hip.h"

namespace dewi {

// HIP keeps function attributes (hipFuncSetAttribute) per device: run `f` once per device this
// process launches on, under a lock, so that two host threads — or two devices — cannot race on
// a plain "done" flag.
struct PerDeviceOnce {
  std::mutex mu;
  uint64_t done[4] = {0, 0, 0, 0};   // one bit per device ordinal (256 devices)
  template <class F>
  hipError_t run(F&& f) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    const bool tracked = dev >= 0 && dev < 256;
    if (tracked && ((done[dev >> 6] >> (dev & 63)) & 1ull)) return hipSuccess;
    e = f();
    if (e == hipSuccess && tracked) done[dev >> 6] |= 1ull << (dev & 63);
    return e;
  }
};

// Per-wave candidate lists hold at most kMaxListCandidates keys (4 register slots x 64 lanes);
// larger candidate counts take the dense path (one key per row, selected afterwards).
constexpr int kMfmaMinQueries = 2;     // bf16 corpus: batches of at least this many queries take the matrix-core path
                                       // (1 M x 768, k = 10: 2 / 3 / 4 queries 0.45 / 0.68 / 0.35 ms on the small-batch kernels, 0.32 ms batched)
constexpr int kMaxListCandidates = 256;
// The select / re-rank kernel sorts its candidates in LDS.
constexpr int kMaxSortCandidates = 2048;
#ifndef DEWI_SCAN_THREADS
#define DEWI_SCAN_THREADS 512
#endif
constexpr int kScanThreads = DEWI_SCAN_THREADS;   // 8 waves per workgroup (256 only for tuning experiments)
constexpr int kSelectThreads = 1024;

// which row kernel serves a (dim, element type): the tuned dim = 256 U kernels, the any-width kernels of scan_any.hpp
// (rows of whole 16-byte units, and — ScanPlan::odd_rows — rows that are not), or the scalar-capable generic kernel (rows
// wider than 1024 units)
enum ScanKind { kScanFast = 0, kScanAnyLong = 1, kScanAnyShort = 2, kScanGeneric = 3 };

struct ScanPlan {
  int blocks;         // workgroups of kScanThreads
  int waves;          // blocks * (kScanThreads / 64)
  int rows_per_iter;  // rows each wave loads before it reduces (fast path)
  int rows_per_iter_batch;  // the same for passes that serve four queries at once
  bool fast;          // dim == 256*U (fp32: U in 1,2,3,4,6; bf16: 1..4): row-per-wave / row-pair-per-wave kernel
  bool dense;         // one key per row instead of per-wave lists
  int slots;          // key registers per lane per query: 1 (c <= 64), 4 (c <= 256), 0 (dense)
  int n_lists;        // candidate lists the scan emits per query: workgroups (slots == 1, each sorted
                      // descending) or wavefronts (slots == 4, unsorted); 0 when dense
  int group;          // generic path: lanes per row (power of two, <= 64)
  int vec;            // generic path: elements per 16-byte load (4 fp32 / 8 bf16) or 1 for scalar loads
  int nq_per_launch;  // queries handled by one corpus pass
  int kind;           // ScanKind
  int units;          // any-width kernels: 16-byte units per row
  int u_pad;          // kScanAnyLong: units per lane the instantiated kernel holds (>= ceil(units / 64))
  int log2p;          // kScanAnyShort: log2 of the lanes that share a row
  int level;          // kScanAnyLong: which rows-in-flight choice of its units-per-lane count (scan_any.hpp any_level / any_rows)
  bool odd_rows;      // any-width kernels: rows are NOT whole 16-byte units (fp32: dim % 4, bf16: dim % 8) — the PH = true kernels
  int row_cols;       // odd_rows: columns per row (what those kernels take instead of `units`; `units` is then the most a row touches)
  bool odd_contig;    // odd_rows, kScanAnyLong, at most two units per lane, offsets repeating every 2 or 4 rows: ONE query takes
                      // scan_rows_odd_contig (a wave on consecutive rows), odd_contig_rows rows per step
  int odd_contig_rows;
  int nq_max;         // most queries one corpus pass of the row kernel serves besides 1 (4; 2 or 1 for wide rows: scan_any.hpp any_nq_max)
  bool raw_queries;   // the kernel normalises the raw queries itself (everything but kScanGeneric)
  int64_t keys_per_query;  // number of uint64 keys the scan emits per query
  bool nontemporal;
};

struct Tuning {
  int scan_blocks;
  int rows_per_iter;
  int nontemporal;  // -1 planner default, 0 off, 1 on
  int mfma;         // batched bf16 matrix-core path: 0 off, anything else on
};

// Measurement hooks (abi.cpp; dewi_timing_enable / dewi_timing_read): bracket the dominant corpus-pass kernel
// of a call with hipEvents on `stream` when this call is sampled.  No-ops when timing is off.
void timing_begin(hipStream_t stream);
void timing_end(hipStream_t stream);

ScanPlan plan_scan(int64_t n_rows, int dim, int elem_bytes, int n_candidates, int compute_units,
                   const Tuning& tuning);

// ---- knn_scan.hip -------------------------------------------------------------------------
// Normalises queries (cosine) into d_qn [n_queries][dim]; used by the generic scan path.
// to_bf16: additionally round the normalised query to bf16 (stored as the fp32 value it represents).
hipError_t launch_prepare_queries(const float* d_q, float* d_qn, int n_queries, int dim, int space, int to_bf16,
                                  hipStream_t stream);
// The same (fp32, not rounded) into n_rows_out >= n_queries rows; rows behind the real queries are zero.
// d_qn2 (may be null): ||prepared query||^2 per output row.
hipError_t launch_prepare_queries_padded(const float* d_q, float* d_qn, int n_queries, int n_rows_out, int dim, int space,
                                         float* d_qn2, hipStream_t stream);
// One corpus pass for queries [q0, q0+nq): emits plan.keys_per_query keys per query into
// d_keys + q * plan.keys_per_query.
hipError_t launch_scan_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                           const float* d_q_norm, int q0, int nq, int n_candidates, int space, uint64_t* d_keys,
                           hipStream_t stream);

// ---- knn_scan_any_f32.hip / knn_scan_any_bf16.hip: plan.kind == kScanAnyLong / kScanAnyShort (raw queries)
hipError_t launch_scan_any_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                               int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream);
hipError_t launch_scan_any_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                                int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream);

// ---- knn_scan_odd_f32.hip / knn_scan_odd_bf16.hip: the same kernels for rows that are not whole units (plan.odd_rows)
hipError_t launch_scan_odd_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                               int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream);
hipError_t launch_scan_odd_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                                int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream);

// REPAIR launches (abi.cpp batch_repair): ONE launch scans the corpus once for every query q of [0, n_queries) whose
// d_flags[q] != 0 (raw queries, keys at d_keys + q * plan.keys_per_query) and returns at once when no flag is set.
// scan_flagged_supported: the row-kernel plans these launches exist for (every shape a matrix-core pass runs at).
bool scan_flagged_supported(const ScanPlan& plan, int elem_bytes);
hipError_t launch_scan_flagged_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                   int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                   hipStream_t stream);
hipError_t launch_scan_flagged_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                    int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                    hipStream_t stream);
hipError_t launch_scan_any_flagged_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                       int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                       hipStream_t stream);
hipError_t launch_scan_any_flagged_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                        int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                        hipStream_t stream);

// ---- knn_scan_bf16.hip: the same pass over a bf16 corpus
hipError_t launch_scan_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                            const float* d_q_norm, int q0, int nq, int n_candidates, int space, uint64_t* d_keys,
                            hipStream_t stream);

// ---- knn_mfma_bf16.hip: batched (up to 256 queries per corpus pass) bf16 path on the matrix cores
struct MfmaLayout {
  int groups;              // passes of up to 256 queries
  int q_pad;               // groups * 256
  int64_t n_tiles;         // 32-row tiles
  int64_t n_sample_tiles;  // every tile_stride-th tile
  int tile_stride;         // 32; 16 / 8 when the scores only pre-select (bf16 shadow of an fp32 corpus) and c is large
  int64_t sample_stride;   // group maxima per query the sample pass emits (sample workgroups * 32)
  int n_blocks;            // workgroups of the filter pass
  int n_seg;               // candidate half-segments per query (2 per workgroup)
  int seg_cap;             // records per half-segment
  size_t qb_off, thr_off, cnt_off, dense_off, cand_off, total;
};
// Normalised (cosine, unless the norm is 0) bf16 queries, [n_rows_out][dim]; rows >= n_queries are zero.
hipError_t launch_prepare_queries_bf16(const float* d_Q, uint16_t* d_out, int n_queries, int n_rows_out, int dim, int space,
                                       float* d_qn2, hipStream_t stream);
bool mfma_path_supported(int64_t n_rows, int dim, int n_queries, int n_candidates, int space);
// preselect: the pass runs over the bf16 shadow of an fp32 corpus (thresholds two error bounds lower: a finer sample keeps
// the survivors of a query inside what the select kernel stages)
MfmaLayout plan_mfma(int64_t n_rows, int dim, int n_queries, int n_candidates, int compute_units, bool preselect = false);
// Fills cand keys [groups][n_seg][256][seg_cap] and counts [groups][256][n_seg] in the workspace.
// thr_bias: subtracted from every query's threshold in the filter pass (0: exact scores; 2 * shadow_margin(dim): the
// scores pre-select for an exact re-scoring of an fp32 corpus)
hipError_t launch_mfma_bf16(const MfmaLayout& m, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_Q,
                            int n_queries, int n_candidates, int space, char* ws, int compute_units,
                            hipStream_t stream, float thr_bias = 0.f);
float shadow_margin(int dim);

// ---- knn_mfma_f32.hip: batched (32 queries per corpus pass) fp32 path on the matrix cores
constexpr int kMfmaF32MinQueries = 5;  // fp32 corpus: batches of at least this many queries take the matrix-core path
                                       // (1 M x 768: 4 queries per pass 0.47 ms on the row-per-wave kernel, 8 queries 0.68 ms)
struct MfmaF32Layout {
  int groups;              // passes of up to 32 queries
  int q_pad;               // groups * 32
  int64_t n_tiles;         // 32-row tiles
  int64_t tile_stride;     // the sample pass takes every tile_stride-th tile
  int64_t n_sample_tiles;
  int sample_blocks;       // workgroups of the sample pass
  int64_t sample_stride;   // group maxima per query (sample_blocks * 32)
  int n_blocks;            // workgroups of the filter pass = survivor segments per query
  int n_seg;
  int seg_cap;             // records per (workgroup, query) segment
  size_t qn_off, qn2_off, thr_off, cnt_off, dense_off, cand_off, total;   // qn2: ||q||^2 per query (l2 space)
};
// elem_type 0: fp32 corpus (>= kMfmaF32MinQueries queries); 1: bf16 corpus (>= kMfmaMinQueries queries: the same
// depth-split kernel on 32x32x16 bf16 MFMAs — batches of up to 32 queries at the tile-delivery rate, and the
// dimensions the 256-query kernel of knn_mfma_bf16.hip cannot hold in registers: 1024, 1536)
bool mfma_f32_path_supported(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int space);
// l2 over an fp32 corpus: |matrix-core score - (-||e - q||^2)| <= depth_l2_margin(dim) * (||e||^2 + ||q||^2)
float depth_l2_margin(int dim);
MfmaF32Layout plan_mfma_f32(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int compute_units,
                            bool preselect = false);
// thr_bias (cosine): subtracted from every threshold of the filter pass (pre-selection over a bf16 shadow: 2 * shadow_margin)
hipError_t launch_mfma_f32(const MfmaF32Layout& m, int elem_type, const void* d_E, int64_t n_rows, int dim, const float* d_Q,
                           int n_queries, int n_candidates, int space, char* ws, hipStream_t stream, float thr_bias = 0.f);

// Per-query threshold = the n_candidates-th largest of each query's sample values (knn_mfma_bf16.hip).
hipError_t launch_sample_threshold(const float* dense, int64_t n_sample, int64_t stride, int n_candidates, float* thr,
                                   int n_queries, hipStream_t stream);

// ---- select_rerank.hip --------------------------------------------------------------------
struct RerankParams {
  float w_sim;   // fp32(1 - eta)
  float w_dewi;  // fp32(eta)
  float w_ent;   // fp32(entropy_pref)
  int use_ent;   // entropy_pref != 0
  int transform; // DEWI_SIM_* : how the raw score becomes the similarity that is blended (A10)
  int space;     // DEWI_SPACE_* of the raw score (the transforms are defined on the library's distance)
};
// keys [n_queries][keys_per_query] -> top n_candidates by key -> either final (ids, scores) or
// sorted candidate records.
// sorted_lists > 0: the keys are `sorted_lists` lists of n_candidates keys, each sorted descending.
// d_counts (may be NULL): the keys are per-workgroup SEGMENTS (batched matrix-core scan): segment s of
// query q holds min(d_counts[s*count_stride + q*count_query_stride], cap) keys at d_keys + s*seg_stride + q*cap; a count
// above `cap` marks an overflowed buffer: that query's ids are set to -1, scores to NaN.
struct SegmentLayout {
  int n_seg;
  int cap;
  int raw;               // 1: entries are raw records (row << 32 | fp32 score bits), not ordered keys
  int64_t seg_stride;    // keys between consecutive segments (= queries_per_pass * cap)
  int64_t count_stride;  // counts between consecutive segments of one query
  int lds_keys;          // records the select kernel may stage in dynamic LDS (0: read segments in place)
  int64_t count_query_stride;  // counts between consecutive queries: 1 with count_stride = queries_per_pass (segment-major,
                               // depth-split pass) or n_seg with count_stride = 1 (query-major: a query's counts are
                               // contiguous and the select kernel reads them coalesced — the 256-query pass)
};
// Exact refinement of approximate survivor scores (l2 on the matrix cores, fp32 corpus): E != NULL switches it on.  The
// select kernel widens the candidate cut by the error bound and re-scores the candidates with the row kernels' arithmetic.
struct RefineParams {
  const float* E;        // corpus rows [n_rows][dim] fp32 (NULL: off)
  const float* Q;        // this launch's RAW queries [n_queries][dim] fp32 (cosine: normalised here as the row kernels do)
  const float* qn2;      // l2: ||q||^2 per query
  int dim;               // any dim % 4 == 0 from 132 to 1536 columns whose one-query search takes scan_rows_f32 / scan_rows_any
  float margin;          // l2: depth_l2_margin(dim), the bound per unit of ||e||^2 + ||q||^2; cosine: the bound itself
  int space;             // DEWI_SPACE_*
  int list_len;          // > 0: the keys are `sorted_lists` sorted lists of THIS length from a row kernel over the bf16 shadow
                         // (one query; n_candidates is then only the cut c <= list_len), see refine_from_sorted_lists
};
// one query through the bf16 shadow on the bf16 ROW kernel: length of the per-workgroup lists it is asked for (the cut's c plus
// room for the rows inside the error band: ~0.2 per workgroup at 1 M gaussian rows; lists of 32 instead of 40 at c = 20 were
// no faster: 0.2306 vs 0.2282 ms scan on two boxes), 0 = this route does not serve that cut
inline int shadow_list_len(int n_candidates) {
  if (n_candidates > 32) return 0;
  return n_candidates <= 16 ? 32 : 2 * n_candidates;
}
// Per-query refusal flags of a batch (one u32 per query, in the caller's workspace).  mode 1: the launch WRITES them — 1
// for a query it could not answer (survivor segment overflowed, error band wider than the sort), 0 otherwise — and leaves
// the outputs of a refused query untouched; mode 2: the launch ANSWERS only flagged queries (the repair's select over the row
// kernels' keys) and returns at once for the others.  p == NULL: no flags (a refused query is marked id -1 / -2 in its outputs).
struct QueryFlags {
  uint32_t* p;
  int mode;
};
hipError_t launch_select_rerank(const uint64_t* d_keys, int64_t keys_per_query, int sorted_lists, int n_queries,
                                int n_candidates, int k, const RerankParams& rp, const float* d_dewi32, const float* d_ent32,
                                int64_t id_offset, int64_t* d_out_ids, float* d_out_scores,
                                dewi_candidate* d_out_cand, const uint32_t* d_counts, const SegmentLayout& seg,
                                hipStream_t stream, const RefineParams& refine = RefineParams{nullptr, nullptr, nullptr, 0, 0.f, 0},
                                const QueryFlags& flags = QueryFlags{nullptr, 0});
// c > kMaxSortCandidates: dense keys [n_queries][keys_per_query] in, scratch g1/g2 [n_queries][p2].
// d_out_cand != NULL: n_out records per query (the shard's candidates) instead of final results.
hipError_t launch_select_rerank_large(const uint64_t* d_keys, int64_t keys_per_query, int n_queries, int n_candidates,
                                      int p2, int k, const RerankParams& rp, const float* d_dewi32,
                                      const float* d_ent32, int64_t id_offset, uint64_t* d_g1, uint64_t* d_g2,
                                      int64_t* d_out_ids, float* d_out_scores, dewi_candidate* d_out_cand, int n_out,
                                      hipStream_t stream);
// n_lists * list_len > kMaxSortCandidates: rank merge of the sorted shard lists through global scratch.
size_t merge_large_workspace_bytes(int n_queries, int n_candidates);
hipError_t launch_merge_rerank_large(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len,
                                     int n_candidates, int k, const RerankParams& rp, void* d_ws, int64_t* d_out_ids,
                                     float* d_out_scores, hipStream_t stream);
hipError_t launch_merge_rerank(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len,
                               int n_candidates, int k, const RerankParams& rp, int64_t* d_out_ids,
                               float* d_out_scores, hipStream_t stream);

// ---- ingest.hip ---------------------------------------------------------------------------
hipError_t launch_normalize_rows(const float* d_src, float* d_dst, int64_t n_rows, int dim, hipStream_t stream);
hipError_t launch_row_cosine(const float* d_a, const float* d_b, float* d_out, int64_t n_rows, int dim, float eps,
                             hipStream_t stream);
hipError_t launch_f32_to_bf16(const float* d_src, uint16_t* d_dst, int64_t n, hipStream_t stream);
hipError_t launch_payload_soa(const double* dewi, const double* ht, const double* hi, float* dewi32, float* ent32,
                              int64_t n, hipStream_t stream);

// ---- robust_stats.hip ---------------------------------------------------------------------
size_t robust_fit_workspace_bytes(int n_signals);
// robust_fit_fast.hip: the two-launch fit of one device's columns (bracket from a sample, one pass, exact select
// among the collected keys).  Its workspace region follows the histogram path's inside the caller's workspace.
constexpr int64_t kFitFastCap = 1024 * 1024;   // keys the compact buffer of a column holds (4 MiB)
size_t robust_fit_fast_bytes(int n_signals);
bool robust_fit_fast_supported(int64_t n, int n_signals);
hipError_t launch_robust_fit_fast(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                                  void* d_fast_ws, hipStream_t stream);
void robust_fit_region(int n_signals, int phase, int pass, int which, size_t* offset_bytes, size_t* count_u32);
hipError_t launch_fit_begin(void* d_ws, int n_signals, hipStream_t stream);
hipError_t launch_fit_hist(const float* S, int64_t n, int64_t ld, int n_signals, int phase, int pass, const float* med,
                           void* d_ws, hipStream_t stream);
hipError_t launch_fit_pick(int64_t n_total, int n_signals, int phase, int pass, void* d_ws, hipStream_t stream);
hipError_t launch_fit_finish(int64_t n_total, int n_signals, int phase, void* d_ws, float* d_out, hipStream_t stream);
hipError_t launch_robust_fit(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                             void* d_ws, hipStream_t stream);
struct ScoreParams {
  double med[DEWI_NUM_SIGNALS];
  double scale[DEWI_NUM_SIGNALS];  // 1.4826 * mad, the reference's denominator
  double w[5];
  double delta;
  int mode;
};
// d_med / d_mad (both or neither): fp32 device statistics that replace sp.med / sp.scale inside the kernel.
hipError_t launch_score(const void* d_S, int is_f64, int64_t n, int64_t ld, const ScoreParams& sp, const float* d_med,
                        const float* d_mad, double* d_out, float* d_out32, hipStream_t stream);

}  // namespace dewi
