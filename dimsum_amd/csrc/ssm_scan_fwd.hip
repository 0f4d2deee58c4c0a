// ssm_scan_fwd.hip -- C entry point of the selective-scan forward (kernel: ssm_scan_fwd_kernel.hpp).
#include <cstdlib>

#include "common.hpp"

namespace dimsum {

template <typename T> int ssm_scan_fwd_dispatch(const dimsum_ssm_params_t &p, hipStream_t stream);   // ssm_scan_fwd_{f32,f16,bf16}.hip

int ssm_check(const dimsum_ssm_params_t *p, bool forward) {
    if (!p || !p->A_ptr || !p->B_ptr || !p->C_ptr || !p->u_ptr || !p->delta_ptr) return DIMSUM_ERR_NULL;
    if (forward && p->z_ptr && !p->out_z_ptr) return DIMSUM_ERR_NULL;   // in the backward out_z is the optional recompute
    if (p->batch <= 0 || p->dim <= 0 || p->seqlen <= 0 || p->n_groups <= 0 || p->dim % p->n_groups != 0) return DIMSUM_ERR_SHAPE;
    if (p->dstate > 256) return DIMSUM_ERR_SHAPE;  // selective_scan.cpp:262
    if (p->n_chunks != (p->seqlen + 2047) / 2048) return DIMSUM_ERR_SHAPE;
    return DIMSUM_OK;
}

// Which forward kernel serves a shape. DIMSUM_SCAN_SPLIT=0/1 forces one (experiments).
static int g_force_split = -1;      // -1: automatic; 0 / 1: forced (tests, experiments)

bool ssm_scan_fwd_use_split(const dimsum_ssm_params_t &p) {
    static const char *env = getenv("DIMSUM_SCAN_SPLIT");
    if (g_force_split == 0 || g_force_split == 1) return g_force_split == 1 && p.dstate % 4 == 0;
    if (env && (env[0] == '0' || env[0] == '1')) return env[0] == '1';
    // The 64-channel kernel keeps 8 waves per CU resident (2048 on the chip). A launch that does not even fill those slots
    // once is latency-bound per wave: the split kernel gives it twice the waves, each with half the sequential work
    // (measured: (64, 1152, 1024) 534 -> 475 us, (16, 1152, 4096) 1225 -> 785 us; (256, 1024, 256) equal).
    const int64_t dpg = p.dim / p.n_groups;
    const int64_t waves = (int64_t)p.batch * p.n_groups * ((dpg + kWave - 1) / kWave);
    return p.dstate % 4 == 0 && waves < 2048;
}

}  // namespace dimsum

// diagnostics (not part of the public header): force one of the two forward kernels (-1 = automatic choice)
extern "C" void dimsum_debug_scan_fwd_force_split(int mode) { dimsum::g_force_split = mode; }

extern "C" int dimsum_ssm_scan_fwd(const dimsum_ssm_params_t *p, void *stream) {
    using namespace dimsum;
    const int rc = ssm_check(p, true);
    if (rc != DIMSUM_OK) return rc;
    if (p->batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (p->dtype) {
        case DIMSUM_F32: return ssm_scan_fwd_dispatch<float>(*p, s);
        case DIMSUM_F16: return ssm_scan_fwd_dispatch<__half>(*p, s);
        case DIMSUM_BF16: return ssm_scan_fwd_dispatch<__hip_bfloat16>(*p, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}
