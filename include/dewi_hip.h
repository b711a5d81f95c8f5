/*
 * dewi_hip.h — C ABI of the MI355X-native DEWI scoring-and-retrieval hot path.
 *
 * The reference (lexsightllc/DEWI, pure Python) has no FFI: its hot path is the
 * NumPy code in src/dewi/backends.py (ExactIndex) and src/dewi/scorer.py.  Each
 * entry point below replaces one block of that code; the citation after "replaces"
 * is the reference file:line whose result it reproduces.  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add to call these from dewi.backends.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No torch / C++ types.
 *   - Every pointer named d_* is a DEVICE pointer (hipMalloc'd by the caller, e.g.
 *     torch.Tensor.data_ptr()).  The caller owns every buffer; nothing is retained
 *     after the call returns and nothing is allocated per call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *     All work is enqueued on it; no entry point synchronises the device unless its
 *     comment says so.
 *   - Return value: 0 on success, a negative DEWI_ERR_* code on failure;
 *     dewi_last_error() returns a thread-local, human-readable message.
 *   - Row indices ("ids") are positions in the embedding matrix; the host layer maps
 *     them to the reference's string doc ids.
 */
#ifndef DEWI_HIP_H
#define DEWI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEWI_ABI_VERSION 5

/* status codes */
#define DEWI_OK 0
#define DEWI_ERR_INVALID_ARG (-1)   /* NULL pointer, non-positive dim, unknown enum ... */
#define DEWI_ERR_K_OUT_OF_BOUNDS (-2) /* k > number of rows: the reference raises ValueError here (backends.py:468) */
#define DEWI_ERR_WORKSPACE (-3)     /* workspace smaller than dewi_*_workspace_bytes() */
#define DEWI_ERR_HIP (-4)           /* a HIP runtime call failed; see dewi_last_error() */
#define DEWI_ERR_UNSUPPORTED (-5)   /* shape outside what this build handles */

/* `space` argument (reference: ExactIndex(space=...), backends.py:389-392) */
#define DEWI_SPACE_COSINE 0 /* query is L2-normalised unless its norm is 0; score = <e, q>            */
#define DEWI_SPACE_L2 1     /* nothing is normalised; score = -sum((e - q)^2)                       */

/* `sim_transform` argument of dewi_knn_rerank_candidates: how a neighbour's raw score becomes the
 * similarity the blend uses in the reference's ANN backends.  `dist` is the distance the ANN library
 * would report for that neighbour: 1 - <e,q> in cosine space (hnswlib, fp32), the squared L2 distance
 * (= -score) in l2 space. */
#define DEWI_SIM_RAW 0               /* sim = score             faiss inner product  (backends.py:335-336)  */
#define DEWI_SIM_ONE_MINUS_DIST 1    /* sim = 1 - dist          hnswlib              (backends.py:229-231)  */
#define DEWI_SIM_INV_ONE_PLUS_DIST 2 /* sim = 1 / (1 + dist)    faiss L2             (backends.py:337-338)  */

/* `mode` argument of dewi_score_f64 (reference: DewiScorer.score / score_conditional) */
#define DEWI_MODE_STANDARD 0
#define DEWI_MODE_CONDITIONAL 1

/* number of per-document signals the scorer consumes, in this fixed order
 * (reference: DewiScorer._components, scorer.py:49-58):
 *   0 ht_mean  1 ht_q90  2 hi_mean  3 hi_q90  4 I_hat  5 redundancy  6 noise        */
#define DEWI_NUM_SIGNALS 7

/* One similarity candidate as exchanged between doc-id shards (16 bytes).
 * `sim` is the raw similarity, `dewi`/`ent` the two fp32 payload values the
 * re-rank reads (backends.py:450-458), `id` the GLOBAL row index (shard offset
 * already added) or -1 for padding. */
typedef struct dewi_candidate {
  float sim;
  float dewi;
  float ent;
  int32_t id;
} dewi_candidate;

int dewi_abi_version(void);
const char* dewi_last_error(void);

/* Facts of the calling thread's CURRENT device that the planner uses (compute units, wavefront size);
 * hipGetDeviceProperties is called once per device ordinal and cached under a lock. */
int dewi_device_info(int* out_compute_units, int* out_wavefront, size_t* out_total_mem);

/* ------------------------------------------------------------------------------------------
 * A1/A2  bulk ingest — replaces ExactIndex.add's `emb / np.linalg.norm(emb)` applied row by
 * row and ExactIndex.build's np.stack (backends.py:394-412).  Rows with zero norm become NaN,
 * exactly like the reference (no guard).  src and dst may alias.  fp32 in, fp32 out.
 * ------------------------------------------------------------------------------------------ */
int dewi_normalize_rows_f32(const float* d_src, float* d_dst, int64_t n_rows, int dim, void* stream);

/* F3  I_hat signal — replaces the post-embedding arithmetic of CrossModalDependency
 * (signals/cross_modal.py:69, 124-139): out[i] = F.cosine_similarity(A[i], B[i]) with torch's
 * semantics (each vector divided by max(||x||, 1e-8)).  A, B [n_rows][dim] fp32 row-major. */
int dewi_row_cosine_f32(const float* d_a, const float* d_b, float* d_out, int64_t n_rows, int dim, void* stream);

/* fp32 -> bf16 (round to nearest even, NaN preserved) for the bf16 corpus of config C3. */
int dewi_convert_f32_to_bf16(const float* d_src, uint16_t* d_dst, int64_t n_elems, void* stream);

/* payload SoA — replaces the per-candidate Python loop of backends.py:450-458:
 * dewi32 = fp32(dewi), ent32 = fp32((ht_mean + hi_mean) * 0.5) with the sum/scale in float64. */
int dewi_payload_soa_f64(const double* d_dewi, const double* d_ht_mean, const double* d_hi_mean,
                         float* d_dewi32, float* d_ent32, int64_t n_rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * A3+A4  search — replaces ExactIndex.search (backends.py:414-481) for a batch of queries:
 *   1. cosine: q <- q / ||q|| unless ||q|| == 0           (:420-424)
 *   2. sim[i] = <E[i], q>   or   -sum((E[i]-q)^2)          (:431-436)
 *   3. c = min(2k, n_rows) best rows by sim                (:439-447)   ties: lower row first
 *   4. adj = fp32(1-eta)*sim + fp32(eta)*dewi32[i]         (:461)       two rounded products, one add
 *      adj += fp32(pref)*ent32[i]   if pref != 0           (:464-465)
 *   5. k best by adj, descending                           (:468-481)   ties: higher sim, then lower row
 * d_E  [n_rows][dim] row-major; d_Q [n_queries][dim] raw (un-normalised) fp32 queries.
 *      ANY dim, as the reference's one BLAS call (:431-433).  Rows that are whole 16-byte units (fp32 dim % 4 == 0, bf16
 *      dim % 8 == 0): d_E 16-byte aligned (hipMalloc gives 256).  Other widths: d_E may be any element-aligned address — a
 *      shard that starts in the middle of a larger buffer; the kernels read whole aligned 16-byte units, so up to 15 bytes
 *      before the first and behind the last row are touched (never across a 16-byte boundary, hence never across a page).
 * d_out_ids [n_queries][k] int64 row indices, d_out_scores [n_queries][k] fp32.
 * k <= 0 writes nothing and returns DEWI_OK (reference returns []); k > n_rows returns
 * DEWI_ERR_K_OUT_OF_BOUNDS (reference: ValueError from np.argpartition).
 *
 * ALWAYS ANSWERED (ABI 5), as ExactIndex.search is (backends.py:414-481): batches run on matrix-core passes that can
 * refuse a query on adversarial corpora (a survivor segment overflowed; more rows inside an error band than the sort
 * holds).  Every entry point that may take such a pass — dewi_knn_rerank_f32 / _bf16 / _f32_shadow / _candidates,
 * dewi_knn_finish, dewi_knn_candidates — enqueues, behind the pass's select and on the same stream, two fixed-shape
 * REPAIR launches that read the per-query refusal flags from the workspace and answer the flagged queries on the
 * exact row kernels (they return at once when no flag is set).  No id -1 / -2 marker ever reaches the caller.
 * ------------------------------------------------------------------------------------------ */
size_t dewi_knn_workspace_bytes(int64_t n_rows, int dim, int n_queries, int n_candidates);

/* Name of the kernel that streams the corpus for this shape on the calling thread's current device and tuning, as
 * rocprofv3 prints it up to its template arguments ("scan_rows_f32", "scan_rows_any<0, 2, 3, 1, 0, 1, false>",
 * "scan_short_rows_any<0, 8, 1, 0, 1, false>", "scan_generic_f32", "mfma_scan_f32<false", "mfma_scan_bf16_s16"): what a
 * measurement harness labels its roofline line with.  (ABI 5.) */
int dewi_knn_scan_kernel(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int space, char* out,
                         size_t out_bytes);

int dewi_knn_rerank_f32(const float* d_E, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                        const float* d_dewi32, const float* d_ent32, int k, double eta, double entropy_pref,
                        int space, int64_t* d_out_ids, float* d_out_scores, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* The same search over an fp32 corpus that also has a bf16 SHADOW copy (d_E_bf16: the stored rows rounded with
 * dewi_convert_f32_to_bf16; +50 % memory): a matrix-core pass runs over the shadow — more than 32 queries: the 256-query
 * pass (half the bytes, 256 queries per corpus pass instead of 32); 1-32 queries: the depth-split pass in its bf16
 * geometry (half the bytes; c = min(2k, n_rows) <= 256); ONE query with c <= 32: the bf16 row kernel with per-workgroup
 * lists long enough for the error band (two launches) — as a PRE-SELECTION: its scores are within a
 * proven bound of the fp32 ones (bf16 rounding of unit vectors: 2^-8 plus accumulation), the candidate cut is widened by
 * that bound — and the candidates are re-scored from the fp32 rows with the row kernels' arithmetic, so ids and scores
 * equal dewi_knn_rerank_f32's one-query results bit for bit.  The bound assumes STORED rows of norm <= 1.0001 (what
 * dewi_normalize_rows_f32 leaves; the host layer checks it once).  Cosine, every dim % 32 == 0 from 160 to 1536 columns (ABI 5; ABI 4: 256 / 512 / 768 / 1024 / 1536), corpus >= 64 K rows;
 * any other call (and d_E_bf16 == NULL) behaves exactly as dewi_knn_rerank_f32.  A query with more candidates inside
 * the error band than the sort holds is answered by the repair launches on the plain fp32 scan (ABI 5; ABI 4 returned
 * it refused, id -1).  (ABI 4.) */
int dewi_knn_rerank_f32_shadow(const float* d_E, const uint16_t* d_E_bf16, int64_t n_rows, int dim, const float* d_Q,
                               int n_queries, const float* d_dewi32, const float* d_ent32, int k, double eta,
                               double entropy_pref, int space, int64_t* d_out_ids, float* d_out_scores,
                               void* d_workspace, size_t workspace_bytes, void* stream);

/* Monitoring / tests: where in the workspace the per-query refusal flags of such a call live (one uint32 per query: 1 =
 * the matrix-core pass refused the query and the repair launches answered it; valid once the call's work on the stream
 * has finished).  *out_offset_bytes = (size_t)-1 when the shape takes the row kernels, which refuse nothing.
 * through_shadow: the call is dewi_knn_rerank_f32_shadow with a shadow; n_candidates <= 0: the default cut min(2k, n_rows).
 * (ABI 5.) */
int dewi_knn_refusal_flags(int elem_type, int through_shadow, int64_t n_rows, int dim, int n_queries, int k, int n_candidates,
                           int space, size_t* out_offset_bytes);

/* A10 / F4  the same search with an explicit candidate count instead of min(2k, n_rows): n_candidates =
 * k reproduces the re-rank rule of the reference's HNSWIndex / FAISSIndex.search (backends.py:204-241,
 * 309-356: the library returns exactly k neighbours, which are then blended and sorted) on top of an
 * exact neighbour search.  k <= n_candidates; elem_type 0 fp32, 1 bf16.  sim_transform (DEWI_SIM_*) selects the
 * similarity those backends blend: the raw inner product (faiss), `1 - dist` (hnswlib) or `1/(1+dist)` (faiss
 * L2); fp32 arithmetic, one rounding per operation.  hnswlib / faiss themselves are not part of this build
 * and were not available to check against: parity of this entry point is UNPINNED (restated by reading). */
int dewi_knn_rerank_candidates(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                               const float* d_dewi32, const float* d_ent32, int k, int n_candidates, double eta,
                               double entropy_pref, int space, int sim_transform, int64_t* d_out_ids,
                               float* d_out_scores, void* d_workspace, size_t workspace_bytes, void* stream);

/* Step 1 of the bf16 search alone (backends.py:420-424 followed by the bf16 rounding of config C3): q / ||q||
 * in fp32 unless the norm is 0 (cosine), then round-to-nearest-even to bf16.  This is the kernel the batched
 * matrix-core path runs on its queries; exposed so that parity tests can check the normalisation on its own
 * and feed the oracle the very same prepared queries.  d_out [n_queries][dim] bf16. */
int dewi_prepare_queries_bf16(const float* d_Q, int n_queries, int dim, int space, uint16_t* d_out, void* stream);

/* The same search split at the kernel boundary, for callers that keep several queries in flight:
 * dewi_knn_scan enqueues steps 1-3a (corpus scan, per-workgroup candidate lists or survivor segments ->
 * workspace) and dewi_knn_finish steps 3b-5 (select, blend, top-k) from that workspace.  The two may be
 * enqueued on DIFFERENT streams (order them with events) so that the finish of batch i overlaps the scan of
 * batch i+1 on a second workspace; dewi_knn_rerank_* == scan + finish on one stream, on the same kernels: a
 * batch takes the same path (row kernels / matrix-core passes) either way; the repair of a refused query
 * (see "always answered" above) is part of dewi_knn_finish, which therefore takes the RAW queries d_Q that
 * dewi_knn_scan was given (ABI 5).
 * dewi_knn_finish writes final results (d_out_cand == NULL) or, for doc-id shards, the shard's
 * `n_candidates` best rows as dewi_candidate records (d_out_cand != NULL; d_out_ids/scores unused,
 * k ignored), exactly as dewi_knn_candidates.  elem_type/n_rows/dim/n_queries/n_candidates/space must equal
 * the values given to dewi_knn_scan, and both calls must come from the same host thread's tuning (together
 * they determine the path and the workspace layout).  elem_type: 0 fp32, 1 bf16.  (ABI 3: `space` added.  ABI 4:
 * any n_candidates up to 2^30 — above 2048 the finish step sorts in the workspace, which is therefore no longer const;
 * d_E, the corpus dewi_knn_scan ran over: an l2 batch over an fp32 corpus re-scores its candidates from the rows.) */
int dewi_knn_scan(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                  int n_candidates, int space, void* d_workspace, size_t workspace_bytes, void* stream);

int dewi_knn_finish(void* d_workspace, size_t workspace_bytes, const void* d_E, int elem_type, int64_t n_rows, int dim,
                    const float* d_Q, int n_queries, int n_candidates, int space, int k, double eta, double entropy_pref,
                    const float* d_dewi32, const float* d_ent32, int64_t id_offset, int64_t* d_out_ids,
                    float* d_out_scores, dewi_candidate* d_out_cand, void* stream);

/* Same contract with a bf16 corpus (rows already normalised in fp32, then rounded to bf16); queries
 * arrive as fp32, are normalised in fp32 and rounded to bf16; products are exact, accumulation fp32. */
int dewi_knn_rerank_bf16(const uint16_t* d_E, int64_t n_rows, int dim, const float* d_Q, int n_queries,
                         const float* d_dewi32, const float* d_ent32, int k, double eta, double entropy_pref,
                         int space, int64_t* d_out_ids, float* d_out_scores, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Doc-id sharding (new; the reference is single-process).  Steps 1-3 on one shard, emitting the
 * shard's best `n_candidates` rows per query as dewi_candidate records sorted by (sim desc, id
 * asc); records past min(n_candidates, n_rows) are padding (id = -1, sim = -inf).
 * `id_offset` is added to the local row index.  d_out [n_queries][n_candidates].
 * elem_type: 0 = fp32 corpus, 1 = bf16 corpus.  A bf16 shard with >= 2 queries takes the batched
 * matrix-core path (same conditions as dewi_knn_rerank_bf16); a query whose survivor buffer
 * overflowed there is answered by the repair launches (ABI 5), so every record is real or padding.
 * (dewi_merge_rerank still maps an id -2 record — an ABI-4 shard — to id -1 for its query.)
 * ------------------------------------------------------------------------------------------ */
int dewi_knn_candidates(const void* d_E, int elem_type, int64_t n_rows, int dim, const float* d_Q,
                        int n_queries, const float* d_dewi32, const float* d_ent32, int n_candidates,
                        int space, int64_t id_offset, dewi_candidate* d_out, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* Steps 3-5 over the concatenation of `n_lists` candidate lists per query (the all-gather result,
 * laid out [n_lists][n_queries][list_len]): global top-`n_candidates` by (sim desc, id asc), then
 * the blend and the top-k exactly as dewi_knn_rerank_f32.  Up to 2048 records per query (n_lists * list_len)
 * are sorted in LDS and need no workspace (dewi_merge_workspace_bytes returns 0, d_workspace may be NULL);
 * beyond that (k > 128 at eight shards) the sorted shard lists are rank-merged through a caller-owned workspace
 * of dewi_merge_workspace_bytes(...) bytes, so that a sharded search answers every k the single device
 * answers (the reference has no limit: backends.py:439-471).  (ABI 4: workspace arguments added.) */
size_t dewi_merge_workspace_bytes(int n_lists, int n_queries, int list_len, int n_candidates);

int dewi_merge_rerank(const dewi_candidate* d_lists, int n_lists, int n_queries, int list_len,
                      int n_candidates, int k, double eta, double entropy_pref, int64_t* d_out_ids,
                      float* d_out_scores, void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A6  robust statistics — replaces scorer.RobustStats.fit (scorer.py:18-26) for n_signals
 * columns of n fp32 values each, stored SoA: column s starts at d_S + s*ld.
 *   med[s] = np.median(col)            exact fp32 order statistic; even n: fp32 (a+b)/2
 *   mad[s] = np.median(|col - med[s]|) subtraction in fp32
 * Any NaN in a column makes its median NaN (NumPy semantics).  The `mad or 1e-8` substitution is
 * host-side float64 logic and stays in the caller.  Outputs are device fp32 arrays [n_signals].
 * ------------------------------------------------------------------------------------------ */
size_t dewi_robust_fit_workspace_bytes(int n_signals);

int dewi_robust_fit_f32(const float* d_S, int64_t n, int64_t ld, int n_signals, float* d_med, float* d_mad,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * A6 over doc-id shards ("fit_stats shards too").  The same exact select, split at
 * its histogram boundaries so that the rows of a column may live on several GPUs:
 *   begin                                   zero the workspace
 *   for phase in (0 median, 1 MAD):         (phase 1 needs d_med = the medians phase 0 produced)
 *     for pass in (0, 1, 2):                11 + 11 + 10 key bits
 *       hist(local rows)  ->  caller SUMS the u32 regions `which` = 0 (histograms, every pass) and
 *       `which` = 1 (NaN counts, pass 0 only) over ranks (RCCL all-reduce)  ->  pick(n_total)
 *     finish(n_total) -> d_out[n_signals]   identical on every rank
 * With one rank and no reduction the sequence IS dewi_robust_fit_f32.  n_local may be 0.
 * dewi_robust_fit_region reports where a region lies inside the workspace (byte offset, u32 count).
 * ------------------------------------------------------------------------------------------ */
int dewi_robust_fit_begin(int n_signals, void* d_workspace, size_t workspace_bytes, void* stream);
int dewi_robust_fit_hist_f32(const float* d_S, int64_t n_local, int64_t ld, int n_signals, int phase, int pass,
                             const float* d_med, void* d_workspace, size_t workspace_bytes, void* stream);
int dewi_robust_fit_region(int n_signals, int phase, int pass, int which, size_t* offset_bytes, size_t* count_u32);
int dewi_robust_fit_pick(int64_t n_total, int n_signals, int phase, int pass, void* d_workspace,
                         size_t workspace_bytes, void* stream);
int dewi_robust_fit_finish(int64_t n_total, int n_signals, int phase, void* d_workspace, size_t workspace_bytes,
                           float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * A7+A8  DEWI score — replaces RobustStats.z, DewiScorer._components, score, score_conditional
 * (scorer.py:28-31, 49-89), float64 arithmetic in the reference's operation order:
 *   z = (x - med) / (1.4826 * mad);  Ht = .5(z0+z1)  Hi = .5(z2+z3)  I = z4  R = z5  N = z6
 *   standard:    U = at*Ht + ai*Hi - am*I - ar*R - an*N
 *   conditional: U = at*(Ht-I) + ai*(Hi-I) - ar*R - an*N
 *   out = 1 / (1 + exp(-clip(U, -delta, delta)))
 * d_S: DEWI_NUM_SIGNALS columns (fixed order above), column s at element offset s*ld, of fp32
 * (signals_are_f64 = 0) or float64 (= 1) values.  med/mad/weights are HOST arrays
 * (7, 7 and 5 doubles: alpha_t, alpha_i, alpha_m, alpha_r, alpha_n).  d_out: n float64 values;
 * d_out32: the same rounded to fp32, ready to be the index's dewi32 column (either may be NULL, not both).
 * ------------------------------------------------------------------------------------------ */
int dewi_score_f64(const void* d_S, int signals_are_f64, int64_t n, int64_t ld, const double* med,
                   const double* mad, const double* weights, double delta, int mode, double* d_out,
                   float* d_out32, void* stream);

/* The same score with the statistics read from DEVICE memory: d_med / d_mad are the fp32 [DEWI_NUM_SIGNALS]
 * outputs of dewi_robust_fit_f32 (or of the sharded fit) as they are.  What the host does between fit and
 * score in the reference — widening to float64 and the `mad or 1e-8` substitution of scorer.py:24 — happens
 * inside the kernel, bit for bit, so a fit -> score -> index-build chain (reference pipelines.py:180-223) runs
 * on one stream without a host round trip.  (ABI 4.) */
int dewi_score_f64_dev(const void* d_S, int signals_are_f64, int64_t n, int64_t ld, const float* d_med,
                       const float* d_mad, const double* weights, double delta, int mode, double* d_out,
                       float* d_out32, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): dewi_timing_enable(n) makes every n-th dewi_knn_* call bracket its
 * corpus-scan kernel with hipEvents on `stream` (n = 1: every call; 0: off — the two event records
 * cost ~5 us of stream time, so throughput runs sample); dewi_timing_read synchronises those
 * events and returns the mean scan-kernel duration in milliseconds and the number of launches
 * averaged, then resets.  The state belongs to the CALLING THREAD (thread-local, like dewi_tuning_set): a
 * thread's brackets, its enable and its read go together, two measuring threads never mix samples.
 * ------------------------------------------------------------------------------------------ */
int dewi_timing_enable(int every);
int dewi_timing_read(double* out_mean_scan_ms, int* out_launches);

/* Launch-shape overrides for tuning sweeps (0 / -1 = planner default).  batched_mfma = 0 disables the
 * matrix-core paths (every batch then takes the small-batch scan kernels); 1 (default) = cosine batches on the
 * matrix cores; l2 batches over an fp32 corpus too, in exact-refine mode (the pass scores 2<e,q> - ||e||^2 - ||q||^2,
 * whose ABSOLUTE error is ~ulp(||e||^2 + ||q||^2) where the reference's -sum((e-q)^2), backends.py:434-436, has a
 * relative error of the distance; the candidate cut is widened by that bound and the candidates are re-scored with the
 * row kernels' arithmetic, so results equal the one-query search bit for bit); l2 batches over a bf16 corpus on the
 * exact row kernels; 2 = those on the matrix cores as well, WITHOUT refinement (near-duplicates of a query then score
 * +-1e-4 instead of ~0 at ||e||^2 ~ 500: an opt-in for throughput, not the parity path).  The setting belongs to the
 * CALLING THREAD (thread-local): it changes the plan, and with it dewi_knn_workspace_bytes, only for calls
 * made from the same thread, so one thread's sweep cannot invalidate another thread's workspace. */
int dewi_tuning_set(int scan_blocks, int rows_per_iter, int nontemporal, int batched_mfma);

#ifdef __cplusplus
}
#endif
#endif /* DEWI_HIP_H */
