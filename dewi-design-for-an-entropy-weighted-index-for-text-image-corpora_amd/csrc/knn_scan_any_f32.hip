// fp32 instantiation of the any-width row scan (scan_any.hpp): dim % 4 == 0 outside the dim = 256 U set of knn_scan.hip.
#include "scan_any.hpp"

namespace dewi {

hipError_t launch_scan_any_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                               int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream) {
  return launch_scan_any_impl<0>(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
}

hipError_t launch_scan_any_flagged_f32(const ScanPlan& plan, const float* d_E, int64_t n_rows, int dim, const float* d_q_raw,
                                       int n_queries, int n_candidates, int space, uint64_t* d_keys, const uint32_t* d_flags,
                                       hipStream_t stream) {
  (void)dim;
  return launch_scan_any_flagged_impl<0>(plan, d_E, n_rows, d_q_raw, n_queries, n_candidates, space, d_keys, d_flags, stream);
}

}  // namespace dewi
