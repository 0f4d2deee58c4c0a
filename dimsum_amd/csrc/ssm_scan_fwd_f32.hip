// selective-scan forward, f32 I/O instantiations (see ssm_scan_fwd_kernel.hpp)
#include "ssm_scan_fwd_kernel.hpp"

namespace dimsum {
template int ssm_scan_fwd_dispatch<float>(const dimsum_ssm_params_t &, hipStream_t);
}  // namespace dimsum

