// bf16 instantiation of the any-width row scan for rows that are NOT whole 16-byte units (scan_any.hpp, PH = true): dim % 8 != 0.
#include "scan_any.hpp"

namespace dewi {

hipError_t launch_scan_odd_bf16(const ScanPlan& plan, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                                int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream) {
  return launch_scan_any_impl<1, true>(plan, d_E, n_rows, dim, d_q_raw, q0, nq, n_candidates, space, d_keys, stream);
}

}  // namespace dewi
