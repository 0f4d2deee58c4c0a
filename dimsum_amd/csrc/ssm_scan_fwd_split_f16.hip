// selective-scan forward, f16 I/O: the state-split kernels, 2 and 4 lanes per channel (ssm_scan_fwd_split.hpp)
#include "ssm_scan_fwd_split.hpp"

namespace dimsum {
DIMSUM_INSTANTIATE_FWD_SPLIT(__half)
}  // namespace dimsum
