// selective-scan forward, f32 I/O instantiations (see ssm_scan_fwd_kernel.hpp)
#include "ssm_scan_fwd_kernel.hpp"

namespace dimsum {
template int ssm_scan_fwd_dispatch<float>(const dimsum_ssm_params_t &, hipStream_t);
}  // namespace dimsum

// diagnostics (not part of the public header): resident workgroups per CU the runtime reports for the headline variant
extern "C" int dimsum_debug_scan_fwd_occupancy(void) {
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dimsum::ssm_scan_fwd_kernel<float, 16, true, true, true>, dimsum::kWave, 0) != hipSuccess) return -1;
    return n;
}
