// selective-scan forward, f16 I/O instantiations (see ssm_scan_fwd_kernel.hpp)
#include "ssm_scan_fwd_kernel.hpp"

namespace dimsum {
template int ssm_scan_fwd_dispatch<__half>(const dimsum_ssm_params_t &, hipStream_t);
}  // namespace dimsum
