// extern "C" boundary of the DEWI hot path (declared in include/dewi_hip.h).
//
// Nothing here touches torch: callers hand over device pointers, sizes and a hipStream_t.  The
// functions validate arguments, carve the caller's workspace, choose launch shapes and enqueue
// kernels; no allocation, no device synchronisation (except dewi_timing_read).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "launch.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  return fail(DEWI_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

struct DeviceInfo {
  bool ready = false;
  int cus = 0;
  int wave = 0;
  size_t mem = 0;
};
// Facts of the CALLING THREAD's current device, cached per device ordinal (a process may drive
// several GPUs from several threads).
constexpr int kMaxDevices = 64;
DeviceInfo g_dev[kMaxDevices];
std::mutex g_dev_mu;

int ensure_device(DeviceInfo& out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  if (dev < 0 || dev >= kMaxDevices) return fail(DEWI_ERR_UNSUPPORTED, "device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_dev_mu);
  DeviceInfo& d = g_dev[dev];
  if (!d.ready) {
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
    d.cus = p.multiProcessorCount;
    d.wave = p.warpSize;
    d.mem = p.totalGlobalMem;
    if (d.wave != 64) return fail(DEWI_ERR_UNSUPPORTED, "wavefront size %d: this library is written for gfx950 (wave64)", d.wave);
    d.ready = true;
  }
  out = d;
  return DEWI_OK;
}

// Launch-shape overrides belong to the calling thread (dewi_tuning_set): a sweep or a test in one
// thread cannot change the plan — and with it the workspace layout — under another thread's calls.
thread_local dewi::Tuning g_tuning{0, 0, -1, 1};

// ---- timing ring -----------------------------------------------------------------------------
// Like the tuning, the measurement state belongs to the CALLING THREAD: dewi_timing_enable / _read and the brackets of
// the dewi_knn_* calls made from the same host thread share one ring, so two benchmarking threads never mix samples.
struct Timing {
  int every = 0;          // 0 = off; n = bracket every n-th scan with events
  unsigned long calls = 0;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
thread_local Timing g_timing;

// One bracket = two events around the dominant corpus-pass kernel of a call.  begin() decides whether this
// call is sampled; end() records the stop event of the bracket begin() opened on this thread.
thread_local hipEvent_t t_open_stop = nullptr;

}  // namespace

namespace dewi {
void timing_begin(hipStream_t stream) {
  t_open_stop = nullptr;
  if (g_timing.every <= 0) return;
  if ((g_timing.calls++ % static_cast<unsigned long>(g_timing.every)) != 0) return;
  if (g_timing.used == g_timing.start.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_timing.start.push_back(a);
    g_timing.stop.push_back(b);
  }
  (void)hipEventRecord(g_timing.start[g_timing.used], stream);
  t_open_stop = g_timing.stop[g_timing.used];
  ++g_timing.used;
}
void timing_end(hipStream_t stream) {
  if (t_open_stop) (void)hipEventRecord(t_open_stop, stream);
  t_open_stop = nullptr;
}
}  // namespace dewi

namespace {

struct ScanTimer {
  hipStream_t stream;
  explicit ScanTimer(hipStream_t s) : stream(s) { dewi::timing_begin(s); }
  ~ScanTimer() { dewi::timing_end(stream); }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct KnnLayout {
  dewi::ScanPlan plan;
  size_t keys_off, keys_bytes;
  size_t qn_off, qn_bytes;
  size_t big_off, big_bytes;  // 2 x [n_queries][p2] u64 scratch when n_candidates > kMaxSortCandidates
  int p2;
  size_t total;
};

KnnLayout layout_knn(int64_t n_rows, int dim, int elem_bytes, int n_queries, int n_candidates, int cus) {
  KnnLayout L;
  L.plan = dewi::plan_scan(n_rows, dim, elem_bytes, n_candidates, cus, g_tuning);
  L.keys_off = 0;
  L.keys_bytes = align_up(static_cast<size_t>(n_queries) * static_cast<size_t>(L.plan.keys_per_query) * 8, 256);
  L.qn_off = L.keys_off + L.keys_bytes;
  L.qn_bytes = align_up(static_cast<size_t>(n_queries) * dim * 4, 256);
  L.big_off = L.qn_off + L.qn_bytes;
  L.p2 = 0;
  L.big_bytes = 0;
  if (n_candidates > dewi::kMaxSortCandidates) {
    int p2 = 2;
    while (p2 < n_candidates) p2 <<= 1;
    L.p2 = p2;
    L.big_bytes = align_up(static_cast<size_t>(2) * n_queries * p2 * 8, 256);
  }
  L.total = L.big_off + L.big_bytes;
  return L;
}

dewi::RerankParams make_rerank(double eta, double pref, int transform = DEWI_SIM_RAW, int space = DEWI_SPACE_COSINE) {
  dewi::RerankParams rp;
  rp.transform = transform;
  rp.space = space;
  // NumPy treats the Python floats (1 - eta), eta, entropy_pref as weak scalars: each is rounded to
  // fp32 once and the array arithmetic stays fp32 (reference backends.py:461-465).
  rp.w_sim = static_cast<float>(1.0 - eta);
  rp.w_dewi = static_cast<float>(eta);
  rp.w_ent = static_cast<float>(pref);
  rp.use_ent = pref != 0.0 ? 1 : 0;
  return rp;
}

// Steps 1-3 for every query: fills the keys region of the workspace.
int run_scan(const KnnLayout& L, const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q,
             int n_queries, int n_candidates, int space, char* ws, hipStream_t stream) {
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws + L.keys_off);
  float* qn = reinterpret_cast<float*>(ws + L.qn_off);
  hipError_t e;
  if (!L.plan.raw_queries) {
    e = dewi::launch_prepare_queries(d_Q, qn, n_queries, dim, space, elem_type ? 1 : 0, stream);
    if (e != hipSuccess) return hip_fail(e, "prepare_queries");
  }
  ScanTimer timer(stream);
  int q = 0;
  while (q < n_queries) {
    // queries per corpus pass: 8 (fp32 row-per-wave kernel with one sorted list per workgroup), else 4, else 1
    static const bool nq8_enabled = [] { const char* e = getenv("DEWI_SCAN_NQ8"); return e == nullptr || atoi(e) != 0; }();
    const bool can8 = !elem_type && L.plan.fast && L.plan.slots == 1 && nq8_enabled;
    const int nq = (can8 && n_queries - q >= 8) ? 8 : ((L.plan.nq_max > 1 && n_queries - q >= L.plan.nq_max) ? L.plan.nq_max : 1);
    if (elem_type)
      e = dewi::launch_scan_bf16(L.plan, static_cast<const uint16_t*>(d_E), n_rows, dim, d_Q, L.plan.raw_queries ? nullptr : qn,
                                 q, nq, n_candidates, space, keys, stream);
    else
      e = dewi::launch_scan_f32(L.plan, static_cast<const float*>(d_E), n_rows, dim, d_Q, L.plan.raw_queries ? nullptr : qn, q,
                                nq, n_candidates, space, keys, stream);
    if (e != hipSuccess) return hip_fail(e, "scan launch");
    q += nq;
  }
  return DEWI_OK;
}

int check_common(const void* d_E, int64_t n_rows, int dim, const float* d_Q, int n_queries, int space) {
  if (!d_E || !d_Q) return fail(DEWI_ERR_INVALID_ARG, "null embedding or query pointer");
  if (n_rows <= 0) return fail(DEWI_ERR_INVALID_ARG, "n_rows must be positive (got %lld)", static_cast<long long>(n_rows));
  if (n_rows > 0xFFFFFFFFll) return fail(DEWI_ERR_UNSUPPORTED, "n_rows %lld exceeds 2^32-1 rows per device", static_cast<long long>(n_rows));
  if (dim <= 0) return fail(DEWI_ERR_INVALID_ARG, "dim must be positive (got %d)", dim);
  if (n_queries <= 0) return fail(DEWI_ERR_INVALID_ARG, "n_queries must be positive (got %d)", n_queries);
  if (space != DEWI_SPACE_COSINE && space != DEWI_SPACE_L2) return fail(DEWI_ERR_INVALID_ARG, "unknown space %d", space);
  return DEWI_OK;
}

// ---- one query batch = a SCAN step (steps 1-3 of ExactIndex.search for every query, into the workspace) and a
// SELECT step (the exact top-c cut, then either the re-rank or the shard's candidate records).  Three scan paths,
// chosen from the shapes alone so that dewi_knn_scan and dewi_knn_finish (two calls, two streams) agree:
//   Rows  : row-per-wave kernels, 1 / 4 / 8 queries per corpus pass (any shape, any space)
//   Depth : depth-split matrix-core pass, 32 queries per corpus pass (fp32 corpus from 5 queries; bf16 corpus for
//           2..32 queries, and for larger batches where the 256-query kernel cannot hold the dimension or the space
//           is l2); cosine and l2
//   Big   : 256-query matrix-core kernel (bf16 corpus, cosine, dim <= 768, more than 32 queries)
enum class BatchPath { Rows, Depth, Big };
struct BatchPlan {
  BatchPath path;
  int c_local;                 // candidates a shard of n_rows can contribute: min(n_candidates, n_rows)
  KnnLayout rows;              // Rows: the batch itself; Depth / Big: the REPAIR of refused queries (same row kernels, keys from offset 0)
  dewi::MfmaF32Layout depth;
  dewi::MfmaLayout big;
  size_t flags_off;            // Depth / Big: one u32 per query behind both layouts — raised by the select for a refused query
  size_t total;                // workspace bytes of the chosen path (+ its repair)
};

// A matrix-core pass may refuse a query (survivor segment overflowed, error band wider than the sort: adversarial corpora,
// ~4 % of the calls of scripts/fuzz_shadow_single.py).  The reference always answers (backends.py:414-481), so the batch does
// too, behind the boundary: the select raises the query's flag instead of writing it off, and two fixed-shape launches on the
// same stream — a row scan and a select that look at the flags and return at once when none is set — answer the flagged
// queries exactly.  `corpus_elem_bytes` is the matrix the repair scans (the fp32 rows for a pre-selection over a bf16 shadow).
void plan_repair(BatchPlan& P, size_t path_total, int64_t n_rows, int dim, int corpus_elem_bytes, int n_queries, int n_candidates,
                 int cus) {
  P.rows = layout_knn(n_rows, dim, corpus_elem_bytes, n_queries, n_candidates, cus);
  P.flags_off = align_up(path_total > P.rows.total ? path_total : P.rows.total, 256);
  P.total = P.flags_off + align_up(static_cast<size_t>(n_queries) * 4, 256);
}

BatchPlan plan_batch(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int space, int cus) {
  BatchPlan P{};
  P.c_local = n_candidates < n_rows ? n_candidates : static_cast<int>(n_rows);
  // the matrix-core paths select exactly n_candidates rows: a shard with fewer rows stays on the row kernels (padding)
  // space l2 on the matrix cores is scored 2<e,q> - ||e||^2 - ||q||^2: absolute error ~ulp(||e||^2 + ||q||^2), where the
  // reference's -sum((e - q)^2) (backends.py:434-436) has a small RELATIVE error of the distance — a near-duplicate of the
  // query would come back as +-1e-4 noise instead of ~0 and search_batch(Q)[j] would differ from search(Q[j]).  Parity
  // first: over an fp32 corpus the pass runs in EXACT-REFINE mode (error-widened cut, candidates re-scored with the row
  // kernels' arithmetic: batch_select); over a bf16 corpus l2 batches take the exact row kernels unless the calling thread
  // opted in to the approximate form (dewi_tuning_set batched_mfma = 2).
  const bool mfma = g_tuning.mfma != 0 && P.c_local == n_candidates &&
                    (space == DEWI_SPACE_COSINE || elem_type == 0 || g_tuning.mfma == 2) &&
                    dewi::scan_flagged_supported(dewi::plan_scan(n_rows, dim, elem_type ? 2 : 4, n_candidates, cus, g_tuning),
                                                 elem_type ? 2 : 4);
  const bool depth_ok = mfma && dewi::mfma_f32_path_supported(elem_type, n_rows, dim, n_queries, n_candidates, space);
  const bool big_ok = mfma && elem_type == 1 && dewi::mfma_path_supported(n_rows, dim, n_queries, n_candidates, space);
  if (depth_ok && (elem_type == 0 || n_queries <= 32 || !big_ok)) {
    P.path = BatchPath::Depth;
    P.depth = dewi::plan_mfma_f32(elem_type, n_rows, dim, n_queries, n_candidates, cus);
    plan_repair(P, P.depth.total, n_rows, dim, elem_type ? 2 : 4, n_queries, n_candidates, cus);
  } else if (big_ok) {
    P.path = BatchPath::Big;
    P.big = dewi::plan_mfma(n_rows, dim, n_queries, n_candidates, cus);
    plan_repair(P, P.big.total, n_rows, dim, elem_type ? 2 : 4, n_queries, n_candidates, cus);
  } else {
    P.path = BatchPath::Rows;
    P.rows = layout_knn(n_rows, dim, elem_type ? 2 : 4, n_queries, P.c_local, cus);
    P.total = P.rows.total;
  }
  return P;
}

int batch_scan(const BatchPlan& P, const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
               int n_candidates, int space, void* d_ws, size_t ws_bytes, int cus, hipStream_t stream) {
  if (!d_ws || ws_bytes < P.total) return fail(DEWI_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, P.total);
  char* ws = static_cast<char*>(d_ws);
  hipError_t e;
  switch (P.path) {
    case BatchPath::Depth:
      e = dewi::launch_mfma_f32(P.depth, elem_type, d_E, n_rows, dim, d_Q, n_queries, n_candidates, space, ws, stream);
      return e == hipSuccess ? DEWI_OK : hip_fail(e, "depth-split mfma scan launch");
    case BatchPath::Big:   // the launcher brackets its filter pass for dewi_timing_read itself
      e = dewi::launch_mfma_bf16(P.big, static_cast<const uint16_t*>(d_E), n_rows, dim, d_Q, n_queries, n_candidates, space, ws,
                                 cus, stream);
      return e == hipSuccess ? DEWI_OK : hip_fail(e, "mfma scan launch");
    default:
      return run_scan(P.rows, d_E, elem_type, n_rows, dim, d_Q, n_queries, P.c_local, space, ws, stream);
  }
}

// k > 0: ids / scores of the re-ranked top k.  k == 0: n_candidates records per query into d_out_cand (an overflowed
// query of a matrix-core path carries id -2 there, -1 in the id output).
// d_E / elem_type / dim / space: the corpus the scan ran over — the exact-refine mode of l2 over an fp32 corpus re-scores
// its candidates from the rows themselves.
// The repair of a matrix-core batch (plan_repair): flagged queries once more on the exact row kernels over `d_E`
// (`elem_type`: its element type), keys from offset 0 of the workspace — the pass's own regions are dead by now.
int batch_repair(const BatchPlan& P, char* ws, const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q,
                 int n_queries, int n_candidates, int space, int k, const dewi::RerankParams& rp, const float* d_dewi32,
                 const float* d_ent32, int64_t id_offset, int64_t* d_out_ids, float* d_out_scores, dewi_candidate* d_out_cand,
                 hipStream_t stream) {
  const KnnLayout& L = P.rows;
  uint32_t* flags = reinterpret_cast<uint32_t*>(ws + P.flags_off);
  uint64_t* keys = reinterpret_cast<uint64_t*>(ws + L.keys_off);
  hipError_t e = elem_type ? dewi::launch_scan_flagged_bf16(L.plan, static_cast<const uint16_t*>(d_E), n_rows, dim, d_Q, n_queries,
                                                            n_candidates, space, keys, flags, stream)
                           : dewi::launch_scan_flagged_f32(L.plan, static_cast<const float*>(d_E), n_rows, dim, d_Q, n_queries,
                                                           n_candidates, space, keys, flags, stream);
  if (e != hipSuccess) return hip_fail(e, "repair scan launch");
  if (n_candidates > dewi::kMaxSortCandidates) return fail(DEWI_ERR_UNSUPPORTED, "repair beyond %d candidates", dewi::kMaxSortCandidates);
  const int sorted = L.plan.slots == 1 ? L.plan.n_lists : 0;
  e = dewi::launch_select_rerank(keys, L.plan.keys_per_query, sorted, n_queries, n_candidates, k, rp, d_dewi32, d_ent32, id_offset,
                                 d_out_ids, d_out_scores, d_out_cand, nullptr, dewi::SegmentLayout{}, stream,
                                 dewi::RefineParams{nullptr, nullptr, nullptr, 0, 0.f, 0}, dewi::QueryFlags{flags, 2});
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "repair select launch");
}

// d_Q: the caller's RAW queries (the repair of a refused query scans for it again; exact-refine over a bf16 shadow
// re-scores with them: `shadow`).
int batch_select(const BatchPlan& P, void* d_ws, size_t ws_bytes, int n_queries, int n_candidates, int k,
                 const dewi::RerankParams& rp, const float* d_dewi32, const float* d_ent32, int64_t id_offset,
                 int64_t* d_out_ids, float* d_out_scores, dewi_candidate* d_out_cand, hipStream_t stream,
                 const void* d_E, int elem_type, int64_t n_rows, int dim, int space, const float* d_Q, bool shadow = false) {
  const float* d_Q_shadow = shadow ? d_Q : nullptr;
  if (!d_ws || ws_bytes < P.total) return fail(DEWI_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, P.total);
  char* ws = static_cast<char*>(d_ws);
  hipError_t e = hipSuccess;
  if (P.path == BatchPath::Rows && P.c_local > dewi::kMaxSortCandidates) {
    // k > 1024 (dense keys from the row kernels): the candidate arrays live in the workspace, not in LDS
    const KnnLayout& L = P.rows;
    uint64_t* g1 = reinterpret_cast<uint64_t*>(ws + L.big_off);
    e = dewi::launch_select_rerank_large(reinterpret_cast<const uint64_t*>(ws + L.keys_off), L.plan.keys_per_query, n_queries,
                                         P.c_local, L.p2, k, rp, d_dewi32, d_ent32, id_offset, g1,
                                         g1 + static_cast<size_t>(n_queries) * L.p2, d_out_ids, d_out_scores, d_out_cand,
                                         n_candidates, stream);
    return e == hipSuccess ? DEWI_OK : hip_fail(e, "select_rerank_large launch");
  }
  if (P.path == BatchPath::Rows) {
    const KnnLayout& L = P.rows;
    const int sorted = (L.plan.slots == 1 && P.c_local == n_candidates) ? L.plan.n_lists : 0;
    e = dewi::launch_select_rerank(reinterpret_cast<const uint64_t*>(ws + L.keys_off), L.plan.keys_per_query, sorted, n_queries,
                                   n_candidates, k, rp, d_dewi32, d_ent32, id_offset, d_out_ids, d_out_scores, d_out_cand,
                                   nullptr, dewi::SegmentLayout{}, stream);
    return e == hipSuccess ? DEWI_OK : hip_fail(e, "select launch");
  }
  // matrix-core paths: one select launch per query group (each group has its own survivor segments); up to 8192
  // survivors of a query (64 KiB) are staged in LDS
  const bool big = P.path == BatchPath::Big;
  const int per = big ? 256 : 32;
  const int groups = big ? P.big.groups : P.depth.groups;
  const int n_seg = big ? P.big.n_seg : P.depth.n_seg;
  const int seg_cap = big ? P.big.seg_cap : P.depth.seg_cap;
  const size_t cand_off = big ? P.big.cand_off : P.depth.cand_off, cnt_off = big ? P.big.cnt_off : P.depth.cnt_off;
  // counts: query-major for the 256-query pass (coalesced in the select kernel), segment-major for the depth-split pass
  // (LDS staging of a query's survivors: 8192 records; 12288 when the scores only pre-select for the exact re-scoring of
  // an fp32 corpus — its thresholds sit two error bounds lower, ~10 K survivors per query at k = 100)
  // DEWI_STAGE_KEYS (tests only): shrink the staging so that the over-capacity routes of the select kernel run on small inputs
  static const int stage_override = [] { const char* e = getenv("DEWI_STAGE_KEYS"); return e ? atoi(e) : 0; }();
  const int stage_keys = stage_override > 0 ? stage_override : ((big && d_Q_shadow != nullptr) ? 12288 : 8192);
  const dewi::SegmentLayout seg{n_seg, seg_cap, 1, static_cast<int64_t>(per) * seg_cap, big ? 1 : per, stage_keys, big ? n_seg : 1};
  // exact-refine modes: l2 on the depth-split pass over an fp32 corpus (queries and norms as the scan left them in the
  // workspace), or the 256-query pass over the bf16 SHADOW of an fp32 corpus (d_Q_shadow = the caller's raw queries)
  const bool refine_l2 = !big && space == DEWI_SPACE_L2 && elem_type == 0;
  const bool refine_shadow = d_Q_shadow != nullptr;
  if (!d_E || !d_Q) return fail(DEWI_ERR_INVALID_ARG, "a matrix-core batch needs the corpus and the raw queries in its finish step (repair of refused queries)");
  uint32_t* flags = reinterpret_cast<uint32_t*>(ws + P.flags_off);
  for (int g = 0; g < groups && e == hipSuccess; ++g) {
    const int q0 = g * per;
    const int nq = n_queries - q0 < per ? n_queries - q0 : per;
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(ws + cand_off) + static_cast<int64_t>(g) * n_seg * per * seg_cap;
    const uint32_t* counts = reinterpret_cast<const uint32_t*>(ws + cnt_off) + static_cast<int64_t>(g) * n_seg * per;
    dewi::RefineParams rf{nullptr, nullptr, nullptr, 0, 0.f, 0};
    if (refine_l2)   // the raw queries and their squared norms as the scan left them in the workspace
      rf = dewi::RefineParams{static_cast<const float*>(d_E),
                              reinterpret_cast<const float*>(ws + P.depth.qn_off) + static_cast<int64_t>(q0) * dim,
                              reinterpret_cast<const float*>(ws + P.depth.qn2_off) + q0, dim, dewi::depth_l2_margin(dim),
                              DEWI_SPACE_L2};
    else if (refine_shadow)
      rf = dewi::RefineParams{static_cast<const float*>(d_E), d_Q_shadow + static_cast<int64_t>(q0) * dim, nullptr, dim,
                              dewi::shadow_margin(dim), DEWI_SPACE_COSINE};
    e = dewi::launch_select_rerank(keys, 0, 0, nq, n_candidates, k, rp, d_dewi32, d_ent32, id_offset,
                                   d_out_ids ? d_out_ids + static_cast<int64_t>(q0) * k : nullptr,
                                   d_out_scores ? d_out_scores + static_cast<int64_t>(q0) * k : nullptr,
                                   d_out_cand ? d_out_cand + static_cast<int64_t>(q0) * n_candidates : nullptr, counts, seg, stream,
                                   rf, dewi::QueryFlags{flags + q0, 1});
  }
  if (e != hipSuccess) return hip_fail(e, "select launch");
  return batch_repair(P, ws, d_E, elem_type, n_rows, dim, d_Q, n_queries, n_candidates, space, k, rp, d_dewi32, d_ent32, id_offset,
                      d_out_ids, d_out_scores, d_out_cand, stream);
}

// n_candidates_override <= 0: the reference's cut, min(2k, n_rows) (backends.py:439).
int knn_rerank_impl(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                    const float* d_dewi32, const float* d_ent32, int k, double eta, double pref, int space,
                    int64_t* d_out_ids, float* d_out_scores, void* d_ws, size_t ws_bytes, void* stream_,
                    int n_candidates_override = 0, int transform = DEWI_SIM_RAW) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_common(d_E, n_rows, dim, d_Q, n_queries, space);
  if (rc) return rc;
  if (k <= 0) return DEWI_OK;  // candidate_count <= 0 -> [] (reference backends.py:439-441)
  if (k > n_rows)
    return fail(DEWI_ERR_K_OUT_OF_BOUNDS, "kth(=%lld) out of bounds (%lld)", static_cast<long long>(n_rows - k),
                static_cast<long long>(n_rows));
  if (!d_dewi32 || !d_ent32 || !d_out_ids || !d_out_scores) return fail(DEWI_ERR_INVALID_ARG, "null payload or output pointer");
  int64_t c64 = (2ll * k < n_rows) ? 2ll * k : n_rows;
  if (n_candidates_override > 0) {
    if (n_candidates_override < k)
      return fail(DEWI_ERR_INVALID_ARG, "n_candidates %d must be at least k = %d", n_candidates_override, k);
    c64 = n_candidates_override < n_rows ? n_candidates_override : n_rows;
  }
  if (c64 > (1ll << 30)) return fail(DEWI_ERR_UNSUPPORTED, "candidate count %lld exceeds 2^30", static_cast<long long>(c64));
  const int c = static_cast<int>(c64);
  DeviceInfo dev;
  rc = ensure_device(dev);
  if (rc) return rc;
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, c, space, dev.cus);
  rc = batch_scan(P, d_E, elem_type, n_rows, dim, d_Q, n_queries, c, space, d_ws, ws_bytes, dev.cus, stream);
  if (rc) return rc;
  const dewi::RerankParams rp = make_rerank(eta, pref, transform, space);
  return batch_select(P, d_ws, ws_bytes, n_queries, c, k, rp, d_dewi32, d_ent32, 0, d_out_ids, d_out_scores, nullptr, stream, d_E,
                      elem_type, n_rows, dim, space, d_Q);
}

// Which route a search over an fp32 corpus WITH a bf16 shadow takes (dewi_knn_rerank_f32_shadow), from the shapes alone.
enum class ShadowMode { Plain, Lists, Big, Depth };
struct ShadowPlan {
  ShadowMode mode;
  int c;              // the reference's cut min(2k, n_rows)
  int list_len;       // Lists: length of the per-workgroup lists the bf16 row kernel is asked for
  KnnLayout lists;    // Lists: that scan's layout
  BatchPlan P;        // the pass's layout (Big / Depth) and, for every mode but Plain, the repair + flags (plan_repair)
};

ShadowPlan plan_shadow(bool have_shadow, int64_t n_rows, int dim, int n_queries, int k, int space, int cus) {
  ShadowPlan S{};
  S.mode = ShadowMode::Plain;
  const int64_t c64 = (2ll * k < n_rows) ? 2ll * k : n_rows;
  S.c = static_cast<int>(c64 < (1ll << 30) ? c64 : (1ll << 30));
  // the shadow pre-selects only where a matrix-core pass runs over it and the one-query search of the same corpus takes the
  // row-per-wave kernel whose arithmetic the refinement repeats (dim 256 / 512 / 768 / 1024 / 1536: scan_rows_f32<U = dim / 256>, up to six
  // 16-byte units per lane in the re-scoring); everything else is the plain search.  dim 1024 / 1536 have no 256-query pass
  // (and 1536 no tuned bf16 row kernel: one query takes the depth-split pass too):
  // its batches run the depth-split pass over the shadow in groups of 32 (2 GB instead of 4 GB per group at 1 M rows)
  // (round 4: every dim % 8 == 0 — whole 16-byte units of the bf16 copy — from 136 to 1536 columns: the re-scoring repeats
  // scan_rows_any's arithmetic at the widths outside the dim = 256 U set, and the depth-split pass takes a partial last chunk;
  // up to 128 columns the one-query kernel is scan_short_rows_any, whose lane layout the re-scoring does not repeat)
  const bool usable = have_shadow && g_tuning.mfma != 0 && space == DEWI_SPACE_COSINE && k > 0 && k <= n_rows &&
                      dim % 8 == 0 && dim >= 136 && dim <= 1536;
  if (!usable) return S;
  const bool use_big = n_queries > 32 && c64 <= 512 && dewi::mfma_path_supported(n_rows, dim, n_queries, S.c, space);
  // (a SINGLE query takes the depth-split pass too: one pass over half the bytes + the exact re-scoring, 0.26 ms instead of
  // the fp32 row scan's 0.43 at 1 M x 768; the pass itself has no lower limit on the batch, kMfmaMinQueries is a choice
  // between it and the bf16 row kernels for a bf16 CORPUS)
  const bool use_depth = !use_big && c64 <= 256 &&
                         dewi::mfma_f32_path_supported(1, n_rows, dim, n_queries < dewi::kMfmaMinQueries ? dewi::kMfmaMinQueries : n_queries,
                                                       S.c, space);
  // ONE query with a small cut: the bf16 ROW kernel over the shadow (two launches instead of the pass's five: 0.222 ms scan
  // at 1 M x 768) with per-workgroup lists long enough for the rows inside the error band, then the same exact re-scoring
  // (select_rerank.hip refine_from_sorted_lists)
  const int list_len = (n_queries == 1 && n_rows >= 64 * 1024) ? dewi::shadow_list_len(static_cast<int>(c64 < 64 ? c64 : 64)) : 0;
  if (list_len > 0) {
    S.lists = layout_knn(n_rows, dim, 2, 1, list_len, cus);
    if (S.lists.plan.fast && S.lists.plan.slots == 1 && S.lists.plan.n_lists <= 4 * 64) {
      S.mode = ShadowMode::Lists;
      S.list_len = list_len;
      plan_repair(S.P, S.lists.total, n_rows, dim, 4, 1, S.c, cus);   // the repair of a refused query: the plain fp32 scan
      return S;
    }
  }
  S.P.c_local = S.c;
  if (use_big) {
    S.mode = ShadowMode::Big;
    S.P.path = BatchPath::Big;
    S.P.big = dewi::plan_mfma(n_rows, dim, n_queries, S.c, cus, true);
    plan_repair(S.P, S.P.big.total, n_rows, dim, 4, n_queries, S.c, cus);      // the repair scans the fp32 rows
  } else if (use_depth) {
    S.mode = ShadowMode::Depth;    // 1..32 queries (and larger batches the 256-query pass does not take): passes of 32 over the shadow
    S.P.path = BatchPath::Depth;
    S.P.depth = dewi::plan_mfma_f32(1, n_rows, dim, n_queries, S.c, cus, true);
    plan_repair(S.P, S.P.depth.total, n_rows, dim, 4, n_queries, S.c, cus);
  }
  return S;
}

}  // namespace

extern "C" {

int dewi_abi_version(void) { return DEWI_ABI_VERSION; }
const char* dewi_last_error(void) { return g_err; }

int dewi_device_info(int* out_compute_units, int* out_wavefront, size_t* out_total_mem) {
  DeviceInfo d;
  int rc = ensure_device(d);
  if (rc) return rc;
  if (out_compute_units) *out_compute_units = d.cus;
  if (out_wavefront) *out_wavefront = d.wave;
  if (out_total_mem) *out_total_mem = d.mem;
  return DEWI_OK;
}

int dewi_normalize_rows_f32(const float* d_src, float* d_dst, int64_t n_rows, int dim, void* stream) {
  if (n_rows < 0 || dim <= 0) return fail(DEWI_ERR_INVALID_ARG, "bad shape %lld x %d", static_cast<long long>(n_rows), dim);
  if (n_rows == 0) return DEWI_OK;
  if (!d_src || !d_dst) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  hipError_t e = dewi::launch_normalize_rows(d_src, d_dst, n_rows, dim, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "normalize_rows launch");
}

int dewi_row_cosine_f32(const float* d_a, const float* d_b, float* d_out, int64_t n_rows, int dim, void* stream) {
  if (n_rows < 0 || dim <= 0) return fail(DEWI_ERR_INVALID_ARG, "bad shape %lld x %d", static_cast<long long>(n_rows), dim);
  if (n_rows == 0) return DEWI_OK;
  if (!d_a || !d_b || !d_out) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  hipError_t e = dewi::launch_row_cosine(d_a, d_b, d_out, n_rows, dim, 1e-8f, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "row_cosine launch");
}

int dewi_convert_f32_to_bf16(const float* d_src, uint16_t* d_dst, int64_t n_elems, void* stream) {
  if (n_elems < 0) return fail(DEWI_ERR_INVALID_ARG, "negative element count");
  if (n_elems == 0) return DEWI_OK;
  if (!d_src || !d_dst) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  hipError_t e = dewi::launch_f32_to_bf16(d_src, d_dst, n_elems, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "f32_to_bf16 launch");
}

int dewi_payload_soa_f64(const double* d_dewi, const double* d_ht_mean, const double* d_hi_mean, float* d_dewi32,
                         float* d_ent32, int64_t n_rows, void* stream) {
  if (n_rows < 0) return fail(DEWI_ERR_INVALID_ARG, "negative row count");
  if (n_rows == 0) return DEWI_OK;
  if (!d_dewi || !d_ht_mean || !d_hi_mean || !d_dewi32 || !d_ent32) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  hipError_t e = dewi::launch_payload_soa(d_dewi, d_ht_mean, d_hi_mean, d_dewi32, d_ent32, n_rows,
                                          static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "payload_soa launch");
}

size_t dewi_knn_workspace_bytes(int64_t n_rows, int dim, int n_queries, int n_candidates) {
  DeviceInfo dev;
  if (ensure_device(dev)) return 0;
  if (n_rows <= 0 || dim <= 0 || n_queries <= 0 || n_candidates <= 0) return 0;
  size_t a = layout_knn(n_rows, dim, 4, n_queries, n_candidates, dev.cus).total;
  const size_t b = layout_knn(n_rows, dim, 2, n_queries, n_candidates, dev.cus).total;
  if (b > a) a = b;
  if (const int ll = dewi::shadow_list_len(n_candidates)) {   // one query through the bf16 shadow: longer per-workgroup lists
    const size_t s = layout_knn(n_rows, dim, 2, n_queries, ll, dev.cus).total;
    if (s > a) a = s;
  }
  if (dewi::mfma_path_supported(n_rows, dim, n_queries, n_candidates, DEWI_SPACE_COSINE)) {
    for (int pre = 0; pre < 2; ++pre) {       // (pre = 1: pre-selection over a bf16 shadow — finer sample, other segment sizes)
      const size_t m = dewi::plan_mfma(n_rows, dim, n_queries, n_candidates, dev.cus, pre != 0).total;
      if (m > a) a = m;
    }
  }
  for (int et = 0; et < 2; ++et) {
    // (a single query reaches the bf16 depth pass through the shadow entry: size for it as for two)
    const int nq = et == 1 && n_queries < dewi::kMfmaMinQueries ? dewi::kMfmaMinQueries : n_queries;
    if (dewi::mfma_f32_path_supported(et, n_rows, dim, nq, n_candidates, DEWI_SPACE_COSINE)) {
      for (int pre = 0; pre < 2; ++pre) {     // (pre = 1: the pass pre-selects over a bf16 shadow, finer sample)
        const size_t m = dewi::plan_mfma_f32(et, n_rows, dim, n_queries, n_candidates, dev.cus, pre != 0).total;
        if (m > a) a = m;
      }
    }
  }
  // valid for either element type and every path (small-batch scans, bf16 / fp32 matrix-core) + the per-query refusal
  // flags of a matrix-core batch behind the larger of its own layout and its repair's (plan_repair)
  return align_up(a, 256) + align_up(static_cast<size_t>(n_queries) * 4, 256);
}

int dewi_knn_scan_kernel(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int space, char* out,
                         size_t out_bytes) {
  if (!out || out_bytes < 16) return fail(DEWI_ERR_INVALID_ARG, "name buffer too small");
  if (n_rows <= 0 || dim <= 0 || n_queries <= 0 || n_candidates <= 0) return fail(DEWI_ERR_INVALID_ARG, "non-positive size");
  DeviceInfo dev;
  int rc = ensure_device(dev);
  if (rc) return rc;
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, n_candidates < n_rows ? n_candidates : static_cast<int>(n_rows),
                                 space, dev.cus);
  const char* e = elem_type ? "bf16" : "f32";
  if (P.path == BatchPath::Depth) {
    snprintf(out, out_bytes, "mfma_scan_f32<%s", elem_type ? "true" : "false");
  } else if (P.path == BatchPath::Big) {
    snprintf(out, out_bytes, "mfma_scan_bf16_s16");
  } else {
    const dewi::ScanPlan& p = P.rows.plan;
    const int nq = n_queries >= p.nq_max ? p.nq_max : 1;
    switch (p.kind) {
      case dewi::kScanFast: snprintf(out, out_bytes, "scan_rows_%s", e); break;
      case dewi::kScanAnyLong:
        if (p.odd_contig && nq == 1) {
          snprintf(out, out_bytes, "scan_rows_odd_contig<%d, %d, %d, %d, %d>", elem_type ? 1 : 0, p.u_pad, p.odd_contig_rows, space, p.slots);
          break;
        }
        snprintf(out, out_bytes, "scan_rows_any<%d, %d, %d, %d, %d, %d, %s>", elem_type ? 1 : 0, p.u_pad,
                                        nq > 1 ? p.rows_per_iter_batch : p.rows_per_iter, nq, space, p.slots,
                                        p.odd_rows ? "true" : "false"); break;
      case dewi::kScanAnyShort: snprintf(out, out_bytes, "scan_short_rows_any<%d, %d, %d, %d, %d, %s>", elem_type ? 1 : 0,
                                         p.rows_per_iter / (64 >> p.log2p), nq, space, p.slots, p.odd_rows ? "true" : "false"); break;
      default: snprintf(out, out_bytes, "scan_generic_%s", e); break;
    }
  }
  return DEWI_OK;
}

int dewi_knn_rerank_f32(const float* d_E, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                        const float* d_dewi32, const float* d_ent32, int k, double eta, double entropy_pref, int space,
                        int64_t* d_out_ids, float* d_out_scores, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  return knn_rerank_impl(d_E, 0, n_rows, dim, d_Q, n_queries, d_dewi32, d_ent32, k, eta, entropy_pref, space,
                         d_out_ids, d_out_scores, d_workspace, workspace_bytes, stream);
}

int dewi_knn_rerank_f32_shadow(const float* d_E, const uint16_t* d_E_bf16, int64_t n_rows, int dim, const float* d_Q,
                               int n_queries, const float* d_dewi32, const float* d_ent32, int k, double eta,
                               double entropy_pref, int space, int64_t* d_out_ids, float* d_out_scores, void* d_workspace,
                               size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_common(d_E, n_rows, dim, d_Q, n_queries, space);
  if (rc) return rc;
  DeviceInfo dev;
  rc = ensure_device(dev);
  if (rc) return rc;
  const ShadowPlan S = plan_shadow(d_E_bf16 != nullptr, n_rows, dim, n_queries, k, space, dev.cus);
  if (S.mode == ShadowMode::Plain)
    return knn_rerank_impl(d_E, 0, n_rows, dim, d_Q, n_queries, d_dewi32, d_ent32, k, eta, entropy_pref, space, d_out_ids,
                           d_out_scores, d_workspace, workspace_bytes, stream_);
  if (!d_dewi32 || !d_ent32 || !d_out_ids || !d_out_scores) return fail(DEWI_ERR_INVALID_ARG, "null payload or output pointer");
  const BatchPlan& P = S.P;
  if (!d_workspace || workspace_bytes < P.total) return fail(DEWI_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, P.total);
  char* ws = static_cast<char*>(d_workspace);
  const dewi::RerankParams rp = make_rerank(eta, entropy_pref, DEWI_SIM_RAW, space);
  if (S.mode == ShadowMode::Lists) {
    const KnnLayout& L = S.lists;
    rc = run_scan(L, d_E_bf16, 1, n_rows, dim, d_Q, 1, S.list_len, space, ws, stream);
    if (rc) return rc;
    const dewi::RefineParams rf{d_E, d_Q, nullptr, dim, dewi::shadow_margin(dim), DEWI_SPACE_COSINE, S.list_len};
    const hipError_t e = dewi::launch_select_rerank(reinterpret_cast<const uint64_t*>(ws + L.keys_off), L.plan.keys_per_query,
                                                    L.plan.n_lists, 1, S.c, k, rp, d_dewi32, d_ent32, 0, d_out_ids, d_out_scores,
                                                    nullptr, nullptr, dewi::SegmentLayout{}, stream, rf,
                                                    dewi::QueryFlags{reinterpret_cast<uint32_t*>(ws + P.flags_off), 1});
    if (e != hipSuccess) return hip_fail(e, "select launch (one query, bf16 shadow)");
    return batch_repair(P, ws, d_E, 0, n_rows, dim, d_Q, 1, S.c, space, k, rp, d_dewi32, d_ent32, 0, d_out_ids, d_out_scores,
                        nullptr, stream);
  }
  // scores from bf16(e), bf16(q) are within shadow_margin of the fp32 row kernels': the sample's c-th best minus the bound is
  // a lower bound of the exact c-th best, and a row may score that much lower here than exactly -> thresholds - 2 bounds
  const float bias = 2.f * dewi::shadow_margin(dim);
  hipError_t e = S.mode == ShadowMode::Big
                     ? dewi::launch_mfma_bf16(P.big, d_E_bf16, n_rows, dim, d_Q, n_queries, S.c, space, ws, dev.cus, stream, bias)
                     : dewi::launch_mfma_f32(P.depth, 1, d_E_bf16, n_rows, dim, d_Q, n_queries, S.c, space, ws, stream, bias);
  if (e != hipSuccess) return hip_fail(e, "mfma scan launch (bf16 shadow)");
  return batch_select(P, d_workspace, workspace_bytes, n_queries, S.c, k, rp, d_dewi32, d_ent32, 0, d_out_ids, d_out_scores, nullptr,
                      stream, d_E, 0, n_rows, dim, space, d_Q, true);
}

int dewi_knn_refusal_flags(int elem_type, int through_shadow, int64_t n_rows, int dim, int n_queries, int k, int n_candidates,
                           int space, size_t* out_offset_bytes) {
  if (!out_offset_bytes) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (n_rows <= 0 || dim <= 0 || n_queries <= 0) return fail(DEWI_ERR_INVALID_ARG, "non-positive size");
  DeviceInfo dev;
  int rc = ensure_device(dev);
  if (rc) return rc;
  *out_offset_bytes = static_cast<size_t>(-1);   // row kernels: nothing can be refused, no flags
  if (through_shadow && elem_type == 0) {
    const ShadowPlan S = plan_shadow(true, n_rows, dim, n_queries, k, space, dev.cus);
    if (S.mode != ShadowMode::Plain) {
      *out_offset_bytes = S.P.flags_off;
      return DEWI_OK;
    }
  }
  int64_t c64 = n_candidates > 0 ? n_candidates : 2ll * k;
  if (c64 > n_rows) c64 = n_rows;
  if (c64 <= 0 || c64 > (1ll << 30)) return DEWI_OK;
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, static_cast<int>(c64), space, dev.cus);
  if (P.path != BatchPath::Rows) *out_offset_bytes = P.flags_off;
  return DEWI_OK;
}

int dewi_knn_rerank_candidates(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                               const float* d_dewi32, const float* d_ent32, int k, int n_candidates, double eta,
                               double entropy_pref, int space, int sim_transform, int64_t* d_out_ids,
                               float* d_out_scores, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (n_candidates <= 0) return fail(DEWI_ERR_INVALID_ARG, "n_candidates must be positive (got %d)", n_candidates);
  if (sim_transform != DEWI_SIM_RAW && sim_transform != DEWI_SIM_ONE_MINUS_DIST && sim_transform != DEWI_SIM_INV_ONE_PLUS_DIST)
    return fail(DEWI_ERR_INVALID_ARG, "unknown sim_transform %d", sim_transform);
  return knn_rerank_impl(d_E, elem_type, n_rows, dim, d_Q, n_queries, d_dewi32, d_ent32, k, eta, entropy_pref, space,
                         d_out_ids, d_out_scores, d_workspace, workspace_bytes, stream, n_candidates, sim_transform);
}

int dewi_prepare_queries_bf16(const float* d_Q, int n_queries, int dim, int space, uint16_t* d_out, void* stream) {
  if (!d_Q || !d_out) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (n_queries <= 0 || dim <= 0) return fail(DEWI_ERR_INVALID_ARG, "non-positive size");
  if (space != DEWI_SPACE_COSINE && space != DEWI_SPACE_L2) return fail(DEWI_ERR_INVALID_ARG, "unknown space %d", space);
  hipError_t e = dewi::launch_prepare_queries_bf16(d_Q, d_out, n_queries, n_queries, dim, space, nullptr, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "prepare_queries_bf16 launch");
}

int dewi_knn_scan(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                  int n_candidates, int space, void* d_workspace, size_t workspace_bytes, void* stream_) {
  int rc = check_common(d_E, n_rows, dim, d_Q, n_queries, space);
  if (rc) return rc;
  if (n_candidates <= 0) return DEWI_OK;
  if (n_candidates > (1 << 30)) return fail(DEWI_ERR_UNSUPPORTED, "n_candidates %d exceeds 2^30", n_candidates);
  DeviceInfo dev;
  rc = ensure_device(dev);
  if (rc) return rc;
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, n_candidates, space, dev.cus);
  return batch_scan(P, d_E, elem_type, n_rows, dim, d_Q, n_queries, n_candidates, space, d_workspace, workspace_bytes, dev.cus,
                    static_cast<hipStream_t>(stream_));
}

int dewi_knn_finish(void* d_workspace, size_t workspace_bytes, const void* d_E, int elem_type, int64_t n_rows, int dim,
                    const float* d_Q, int n_queries, int n_candidates, int space, int k, double eta, double entropy_pref,
                    const float* d_dewi32, const float* d_ent32, int64_t id_offset, int64_t* d_out_ids,
                    float* d_out_scores, dewi_candidate* d_out_cand, void* stream_) {
  if (n_rows <= 0 || dim <= 0 || n_queries <= 0) return fail(DEWI_ERR_INVALID_ARG, "non-positive size");
  if (space != DEWI_SPACE_COSINE && space != DEWI_SPACE_L2) return fail(DEWI_ERR_INVALID_ARG, "unknown space %d", space);
  if (n_candidates <= 0) return DEWI_OK;
  if (n_candidates > (1 << 30)) return fail(DEWI_ERR_UNSUPPORTED, "n_candidates %d exceeds 2^30", n_candidates);
  if (!d_dewi32 || !d_ent32) return fail(DEWI_ERR_INVALID_ARG, "null payload pointer");
  const bool records = d_out_cand != nullptr;
  if (!records) {
    if (k <= 0) return DEWI_OK;
    if (k > n_candidates) return fail(DEWI_ERR_K_OUT_OF_BOUNDS, "kth(=%d) out of bounds (%d)", n_candidates - k, n_candidates);
    if (!d_out_ids || !d_out_scores) return fail(DEWI_ERR_INVALID_ARG, "null output pointer");
  } else if (id_offset < 0 || id_offset + n_rows > 0x7FFFFFFFll) {
    return fail(DEWI_ERR_UNSUPPORTED, "global row ids must fit int32");
  }
  DeviceInfo dev;
  int rc = ensure_device(dev);
  if (rc) return rc;
  // the same plan as dewi_knn_scan made (same shapes, same thread's tuning)
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, n_candidates, space, dev.cus);
  return batch_select(P, d_workspace, workspace_bytes, n_queries, n_candidates, records ? 0 : k,
                      make_rerank(records ? 0.0 : eta, records ? 0.0 : entropy_pref, DEWI_SIM_RAW, space), d_dewi32, d_ent32,
                      id_offset, records ? nullptr : d_out_ids, records ? nullptr : d_out_scores, d_out_cand,
                      static_cast<hipStream_t>(stream_), d_E, elem_type, n_rows, dim, space, d_Q);
}

int dewi_knn_rerank_bf16(const uint16_t* d_E, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                         const float* d_dewi32, const float* d_ent32, int k, double eta, double entropy_pref, int space,
                         int64_t* d_out_ids, float* d_out_scores, void* d_workspace, size_t workspace_bytes,
                         void* stream) {
  return knn_rerank_impl(d_E, 1, n_rows, dim, d_Q, n_queries, d_dewi32, d_ent32, k, eta, entropy_pref, space,
                         d_out_ids, d_out_scores, d_workspace, workspace_bytes, stream);
}

int dewi_knn_candidates(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                        const float* d_dewi32, const float* d_ent32, int n_candidates, int space, int64_t id_offset,
                        dewi_candidate* d_out, void* d_workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int rc = check_common(d_E, n_rows, dim, d_Q, n_queries, space);
  if (rc) return rc;
  if (n_candidates <= 0) return DEWI_OK;
  if (n_candidates > (1 << 30)) return fail(DEWI_ERR_UNSUPPORTED, "n_candidates %d exceeds 2^30", n_candidates);
  if (!d_dewi32 || !d_ent32 || !d_out) return fail(DEWI_ERR_INVALID_ARG, "null payload or output pointer");
  if (id_offset < 0 || id_offset + n_rows > 0x7FFFFFFFll)
    return fail(DEWI_ERR_UNSUPPORTED, "global row ids must fit int32 (offset %lld + %lld rows)",
                static_cast<long long>(id_offset), static_cast<long long>(n_rows));
  DeviceInfo dev;
  rc = ensure_device(dev);
  if (rc) return rc;
  // The select step writes n_candidates records per query; a shard with fewer rows than that selects every row and
  // pads the tail (id = -1, sim = -inf).
  const BatchPlan P = plan_batch(elem_type, n_rows, dim, n_queries, n_candidates, space, dev.cus);
  rc = batch_scan(P, d_E, elem_type, n_rows, dim, d_Q, n_queries, n_candidates, space, d_workspace, workspace_bytes, dev.cus, stream);
  if (rc) return rc;
  return batch_select(P, d_workspace, workspace_bytes, n_queries, n_candidates, 0, make_rerank(0.0, 0.0), d_dewi32, d_ent32,
                      id_offset, nullptr, nullptr, d_out, stream, d_E, elem_type, n_rows, dim, space, d_Q);
}

size_t dewi_merge_workspace_bytes(int n_lists, int n_queries, int list_len, int n_candidates) {
  if (n_lists <= 0 || n_queries <= 0 || list_len <= 0 || n_candidates <= 0) return 0;
  if (static_cast<int64_t>(n_lists) * list_len <= dewi::kMaxSortCandidates) return 0;   // sorted in LDS
  return dewi::merge_large_workspace_bytes(n_queries, n_candidates);
}

int dewi_merge_rerank(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len, int n_candidates, int k,
                      double eta, double entropy_pref, int64_t* d_out_ids, float* d_out_scores, void* d_workspace,
                      size_t workspace_bytes, void* stream) {
  if (!d_lists || !d_out_ids || !d_out_scores) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (n_lists <= 0 || n_queries <= 0 || list_len <= 0 || n_candidates <= 0)
    return fail(DEWI_ERR_INVALID_ARG, "non-positive size");
  if (k <= 0) return DEWI_OK;
  if (k > n_candidates) return fail(DEWI_ERR_K_OUT_OF_BOUNDS, "k %d exceeds candidate count %d", k, n_candidates);
  if (static_cast<int64_t>(n_lists) * list_len > 0x7FFFFFFFll || n_candidates > (1 << 30))
    return fail(DEWI_ERR_UNSUPPORTED, "n_lists*list_len = %lld records per query", static_cast<long long>(n_lists) * list_len);
  const size_t need = dewi_merge_workspace_bytes(n_lists, n_queries, list_len, n_candidates);
  hipError_t e;
  if (need == 0) {
    e = dewi::launch_merge_rerank(d_lists, n_lists, n_queries, list_len, n_candidates, k, make_rerank(eta, entropy_pref),
                                  d_out_ids, d_out_scores, static_cast<hipStream_t>(stream));
  } else {
    if (!d_workspace || workspace_bytes < need)
      return fail(DEWI_ERR_WORKSPACE, "merge workspace %zu B < required %zu B", workspace_bytes, need);
    e = dewi::launch_merge_rerank_large(d_lists, n_lists, n_queries, list_len, n_candidates, k, make_rerank(eta, entropy_pref),
                                        d_workspace, d_out_ids, d_out_scores, static_cast<hipStream_t>(stream));
  }
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "merge_rerank launch");
}

size_t dewi_robust_fit_workspace_bytes(int n_signals) {
  return n_signals > 0 ? dewi::robust_fit_workspace_bytes(n_signals) : 0;
}

int dewi_robust_fit_f32(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                        void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_S || !d_med || !d_mad) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (n <= 0 || n_signals <= 0 || ld < n) return fail(DEWI_ERR_INVALID_ARG, "bad shape n=%lld ld=%lld n_signals=%d",
                                                      static_cast<long long>(n), static_cast<long long>(ld), n_signals);
  if (n > 0xFFFFFFFFll) return fail(DEWI_ERR_UNSUPPORTED, "n exceeds 2^32-1");
  const size_t need = dewi::robust_fit_workspace_bytes(n_signals);
  if (!d_workspace || workspace_bytes < need) return fail(DEWI_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  hipError_t e = dewi::launch_robust_fit(d_S, n, ld, n_signals, d_med, d_mad, d_workspace, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "robust_fit launch");
}

// ---- sharded robust fit: the select of dewi_robust_fit_f32 split at its histogram boundaries ----
static int check_fit_step(int n_signals, int phase, int pass, void* d_workspace, size_t workspace_bytes) {
  if (n_signals <= 0) return fail(DEWI_ERR_INVALID_ARG, "n_signals %d", n_signals);
  if (phase < 0 || phase > 1 || pass < 0 || pass > 2) return fail(DEWI_ERR_INVALID_ARG, "phase %d / pass %d out of range", phase, pass);
  const size_t need = dewi::robust_fit_workspace_bytes(n_signals);
  if (!d_workspace || workspace_bytes < need) return fail(DEWI_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  return DEWI_OK;
}

int dewi_robust_fit_begin(int n_signals, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_fit_step(n_signals, 0, 0, d_workspace, workspace_bytes)) return rc;
  hipError_t e = dewi::launch_fit_begin(d_workspace, n_signals, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "robust_fit_begin");
}

int dewi_robust_fit_hist_f32(const float* d_S, int64_t n_local, int64_t ld, int n_signals, int phase, int pass,
                             const float* d_med, void* d_workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_fit_step(n_signals, phase, pass, d_workspace, workspace_bytes)) return rc;
  if (n_local < 0 || ld < n_local) return fail(DEWI_ERR_INVALID_ARG, "bad shape n_local=%lld ld=%lld", static_cast<long long>(n_local), static_cast<long long>(ld));
  if (n_local > 0 && !d_S) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (phase == 1 && !d_med) return fail(DEWI_ERR_INVALID_ARG, "the MAD phase needs the medians");
  if (n_local > 0xFFFFFFFFll) return fail(DEWI_ERR_UNSUPPORTED, "n exceeds 2^32-1");
  hipError_t e = dewi::launch_fit_hist(d_S, n_local, ld, n_signals, phase, pass, d_med, d_workspace, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "robust_fit_hist launch");
}

int dewi_robust_fit_region(int n_signals, int phase, int pass, int which, size_t* offset_bytes, size_t* count_u32) {
  if (n_signals <= 0 || phase < 0 || phase > 1 || pass < 0 || pass > 2 || which < 0 || which > 1 || !offset_bytes || !count_u32)
    return fail(DEWI_ERR_INVALID_ARG, "bad region request");
  dewi::robust_fit_region(n_signals, phase, pass, which, offset_bytes, count_u32);
  return DEWI_OK;
}

int dewi_robust_fit_pick(int64_t n_total, int n_signals, int phase, int pass, void* d_workspace, size_t workspace_bytes,
                         void* stream) {
  if (int rc = check_fit_step(n_signals, phase, pass, d_workspace, workspace_bytes)) return rc;
  if (n_total <= 0 || n_total > 0xFFFFFFFFll) return fail(DEWI_ERR_INVALID_ARG, "n_total %lld", static_cast<long long>(n_total));
  hipError_t e = dewi::launch_fit_pick(n_total, n_signals, phase, pass, d_workspace, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "robust_fit_pick launch");
}

int dewi_robust_fit_finish(int64_t n_total, int n_signals, int phase, void* d_workspace, size_t workspace_bytes,
                           float* d_out, void* stream) {
  if (int rc = check_fit_step(n_signals, phase, 0, d_workspace, workspace_bytes)) return rc;
  if (n_total <= 0 || !d_out) return fail(DEWI_ERR_INVALID_ARG, "bad arguments");
  hipError_t e = dewi::launch_fit_finish(n_total, n_signals, phase, d_workspace, d_out, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "robust_fit_finish launch");
}

static int score_impl(const void* d_S, int signals_are_f64, int64_t n, int64_t ld, const double* med, const double* mad,
                      const float* d_med, const float* d_mad, const double* weights, double delta, int mode, double* d_out,
                      float* d_out32, void* stream) {
  if (!d_S || !weights || (!d_out && !d_out32)) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  if (n < 0 || ld < n) return fail(DEWI_ERR_INVALID_ARG, "bad shape n=%lld ld=%lld", static_cast<long long>(n), static_cast<long long>(ld));
  if (mode != DEWI_MODE_STANDARD && mode != DEWI_MODE_CONDITIONAL) return fail(DEWI_ERR_INVALID_ARG, "unknown mode %d", mode);
  if (n == 0) return DEWI_OK;
  dewi::ScoreParams sp;
  for (int s = 0; s < DEWI_NUM_SIGNALS; ++s) {
    sp.med[s] = med ? med[s] : 0.0;
    sp.scale[s] = mad ? 1.4826 * mad[s] : 1.0;  // reference scorer.py:31 — the product is rounded before the division
  }
  for (int i = 0; i < 5; ++i) sp.w[i] = weights[i];
  sp.delta = delta;
  sp.mode = mode;
  hipError_t e = dewi::launch_score(d_S, signals_are_f64, n, ld, sp, d_med, d_mad, d_out, d_out32, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? DEWI_OK : hip_fail(e, "score launch");
}

int dewi_score_f64(const void* d_S, int signals_are_f64, int64_t n, int64_t ld, const double* med, const double* mad,
                   const double* weights, double delta, int mode, double* d_out, float* d_out32, void* stream) {
  if (!med || !mad) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  return score_impl(d_S, signals_are_f64, n, ld, med, mad, nullptr, nullptr, weights, delta, mode, d_out, d_out32, stream);
}

int dewi_score_f64_dev(const void* d_S, int signals_are_f64, int64_t n, int64_t ld, const float* d_med, const float* d_mad,
                       const double* weights, double delta, int mode, double* d_out, float* d_out32, void* stream) {
  if (!d_med || !d_mad) return fail(DEWI_ERR_INVALID_ARG, "null pointer");
  return score_impl(d_S, signals_are_f64, n, ld, nullptr, nullptr, d_med, d_mad, weights, delta, mode, d_out, d_out32, stream);
}

int dewi_timing_enable(int every) {
  g_timing.every = every > 0 ? every : 0;
  g_timing.calls = 0;
  g_timing.used = 0;
  return DEWI_OK;
}

int dewi_timing_read(double* out_mean_scan_ms, int* out_launches) {
  double total = 0.0;
  int n = 0;
  for (size_t i = 0; i < g_timing.used; ++i) {
    hipError_t e = hipEventSynchronize(g_timing.stop[i]);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, g_timing.start[i], g_timing.stop[i]);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
    total += ms;
    ++n;
  }
  g_timing.used = 0;
  if (out_mean_scan_ms) *out_mean_scan_ms = n ? total / n : 0.0;
  if (out_launches) *out_launches = n;
  return DEWI_OK;
}

int dewi_tuning_set(int scan_blocks, int rows_per_iter, int nontemporal, int batched_mfma) {
  g_tuning.scan_blocks = scan_blocks;
  g_tuning.rows_per_iter = rows_per_iter;
  g_tuning.nontemporal = nontemporal;
  g_tuning.mfma = batched_mfma;
  return DEWI_OK;
}

}  // extern "C"
