// selective-scan forward, bf16 I/O: the state-split kernels -- 2 and 4 lanes per channel (ssm_scan_fwd_split.hpp), one lane
// per state (ssm_scan_fwd_lanes.hpp)
#include "ssm_scan_fwd_lanes.hpp"

namespace dimsum {
DIMSUM_INSTANTIATE_FWD_SPLIT(__hip_bfloat16)
DIMSUM_INSTANTIATE_FWD_LANES(__hip_bfloat16)
}  // namespace dimsum
