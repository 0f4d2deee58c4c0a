// selective-scan forward, bf16 I/O: the 64-channels-per-wave kernel (ssm_scan_fwd_kernel.hpp). One translation unit per dtype
// and kernel family: the scheduled inner blocks make every instantiation slow to compile.
#include "ssm_scan_fwd_kernel.hpp"

namespace dimsum {
DIMSUM_INSTANTIATE_FWD_V0(__hip_bfloat16)
}  // namespace dimsum
