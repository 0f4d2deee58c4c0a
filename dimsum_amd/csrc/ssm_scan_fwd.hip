// ssm_scan_fwd.hip -- C entry point of the selective-scan forward (kernel: ssm_scan_fwd_kernel.hpp).
#include <type_traits>
#include <cstdlib>

#include "common.hpp"

namespace dimsum {

// kernel launchers, instantiated in ssm_scan_fwd_{f32,f16,bf16}.hip / ssm_scan_fwd_split_{f32,f16,bf16}.hip
template <typename T, int kN> void ssm_scan_fwd_launch_v0(const ssm_args_t &p, hipStream_t stream, int tiles, bool vec, bool full);
template <typename T, int kN, int kSP> void ssm_scan_fwd_launch_split(const ssm_args_t &p, hipStream_t stream, int tiles, bool vec, bool full);
template <typename T> void ssm_scan_fwd_launch_lanes(const ssm_args_t &p, hipStream_t stream, int tiles, bool vec, bool full);

int ssm_check(const ssm_args_t *p, bool forward) {
    if (!p || !p->A_ptr || !p->B_ptr || !p->C_ptr || !p->u_ptr || (!p->delta_ptr && !(forward && p->dt_w_ptr))) return DIMSUM_ERR_NULL;
    if (p->dt_w_ptr) {          // fused dt_proj (forward only)
        if (!forward) return DIMSUM_ERR_UNSUPPORTED;
        if (!p->dt_x_ptr) return DIMSUM_ERR_NULL;
        if (p->dt_rank <= 0 || p->dt_rank > 32 || p->dt_rank % 4 != 0) return DIMSUM_ERR_SHAPE;
        if (p->dt_w_row_stride % 4 != 0 || p->dt_w_row_stride < p->dt_rank || p->dt_x_row_stride < (int64_t)p->batch * p->seqlen ||
            reinterpret_cast<uintptr_t>(p->dt_w_ptr) % 16 != 0 || reinterpret_cast<uintptr_t>(p->dt_x_ptr) % 4 != 0 ||
            (int64_t)8 * p->dt_x_row_stride * 4 >= ((int64_t)1 << 31))           // (32-bit byte offsets of the 8 r rows a lane reads)
            return DIMSUM_ERR_STRIDE;
    }
    if (forward && p->z_ptr && !p->out_z_ptr) return DIMSUM_ERR_NULL;   // in the backward out_z is the optional recompute
    if (p->batch <= 0 || p->dim <= 0 || p->seqlen <= 0 || p->n_groups <= 0 || p->dim % p->n_groups != 0) return DIMSUM_ERR_SHAPE;
    if (p->dstate > 256) return DIMSUM_ERR_SHAPE;  // selective_scan.cpp:262
    if (p->n_chunks != (p->seqlen + 2047) / 2048) return DIMSUM_ERR_SHAPE;
    return DIMSUM_OK;
}

// Which forward kernel serves a call, as lanes per channel: 1 = lane = channel (64 channels per wave), 2 / 4 = the state-split
// kernel (32 / 16 channels per wave), 16 = one lane per state (dstate 16: 4 channels per wave). p.kernel_variant != 0 asks
// for one of them (tests, tuning); a pure function of the parameters.
constexpr int64_t kLanesBelowWaves = 2048;

static bool variant_ok(const ssm_args_t &p, int v) {
    return v == 1 || (v == 2 && p.dstate % 4 == 0) || (v == 4 && p.dstate % 8 == 0) || (v == 16 && p.dstate == 16);
}

int ssm_scan_fwd_variant(const ssm_args_t &p) {
    if (p.kernel_variant != 0) return variant_ok(p, p.kernel_variant) ? p.kernel_variant : 1;
    // The 64-channel kernel keeps 8 waves per CU resident (2048 on the chip) and is HBM-bound when they are all there.
    // A launch that does not fill those slots is latency-bound per wave: splitting the states over 2 or 4 lanes gives it
    // 2x / 4x the waves, each with 1/2 / 1/4 of the sequential work per step. Measured (fp32, dstate 16):
    //   (256, 1024, 256): 64-channel 0.33 ms, kSP = 2 equal;   (64, 1152, 1024): 0.534 / 0.475 / see DESIGN.md section 3.1
    const int64_t dpg = p.dim / p.n_groups;
    const int64_t waves = (int64_t)p.batch * p.n_groups * ((dpg + kWave - 1) / kWave);
    if (waves >= 2048) return 1;
    // fewer than 2 waves per SIMD even at 16 channels per wave: one lane per state (4 channels per wave). Measured (fp32):
    //   (16, 1152, 4096): 0.71 -> 0.57 ms, (8, 1152, 4096): 0.54 -> 0.34 ms; (32, 1152, 1024), 2304 waves: equal
    const int64_t waves4 = (int64_t)p.batch * p.n_groups * ((dpg + 15) / 16);
    if (waves4 < kLanesBelowWaves && variant_ok(p, 16)) return 16;
    if (variant_ok(p, 4)) return 4;
    return variant_ok(p, 2) ? 2 : 1;
}

template <typename T, int kN>
static int launch_fwd(const ssm_args_t &p, hipStream_t stream) {
    const int dpg = p.dim / p.n_groups;
    const int sp = ssm_scan_fwd_variant(p);                // lanes per channel: 1, 2, 4 or 16
    const int cpw = kWave / sp;                            // channels per wave
    const int tiles = p.batch * p.n_groups * ((dpg + cpw - 1) / cpw);
    const size_t va = 4 * sizeof(T);  // vector path: every row base 4-element aligned
    bool vec = (p.seqlen % 4 == 0) && aligned_to<T>(p.u_ptr, va) && aligned_to<T>(p.delta_ptr, va) &&
               aligned_to<T>(p.B_ptr, va) && aligned_to<T>(p.C_ptr, va) && (p.u_batch_stride % 4 == 0) &&
               (p.u_d_stride % 4 == 0) && (p.delta_batch_stride % 4 == 0) && (p.delta_d_stride % 4 == 0) &&
               (p.B_batch_stride % 4 == 0) && (p.B_group_stride % 4 == 0) && (p.B_dstate_stride % 4 == 0) &&
               (p.C_batch_stride % 4 == 0) && (p.C_group_stride % 4 == 0) && (p.C_dstate_stride % 4 == 0);
    if (p.out_ptr) vec = vec && aligned_to<T>(p.out_ptr, va) && (p.out_batch_stride % 4 == 0) && (p.out_d_stride % 4 == 0);
    if (p.z_ptr)
        vec = vec && aligned_to<T>(p.z_ptr, va) && aligned_to<T>(p.out_z_ptr, va) && (p.z_batch_stride % 4 == 0) &&
              (p.z_d_stride % 4 == 0) && (p.out_z_batch_stride % 4 == 0) && (p.out_z_d_stride % 4 == 0);
    if (p.out_z_lo_offset != 0) {      // out_z as its split-bf16 pair of planes: float32 I/O, vector path, 8-byte aligned 4-element stores
        if (!std::is_same<T, float>::value || !p.z_ptr) return DIMSUM_ERR_UNSUPPORTED;
        if (p.seqlen % 8 != 0) return DIMSUM_ERR_SHAPE;
        if (!vec || !aligned_to<char>(p.out_z_ptr, 16) || p.out_z_lo_offset % 8 != 0 || p.out_z_batch_stride % 8 != 0 || p.out_z_d_stride % 8 != 0) return DIMSUM_ERR_STRIDE;
    }
    if (p.x_ptr && !aligned_to<float>(p.x_ptr, 16)) return DIMSUM_ERR_STRIDE;
    // In-tile offsets are 32-bit BYTE offsets (saddr + voffset addressing): the farthest element of a tile is
    // (channels_per_wave - 1) * d_stride + seqlen elements from the tile base.
    if (!offsets_fit_32bit<T>(p.seqlen, kWave, {p.u_d_stride, p.delta_d_stride, p.out_ptr ? p.out_d_stride : 0, p.z_ptr ? p.z_d_stride : 0,
                                                p.z_ptr ? p.out_z_d_stride : 0}) ||
        !offsets_fit_32bit<T>(p.seqlen, p.dstate, {p.B_dstate_stride, p.C_dstate_stride}))
        return DIMSUM_ERR_STRIDE;
    const bool full = vec && (dpg % cpw == 0);
    if ((p.dt_w_ptr || p.out_z_f16) && !(sp == 1 && full && p.z_ptr && !p.ckpt_ptr && std::is_same<T, float>::value && kN == 16 && p.seqlen % 4 == 0))
        return DIMSUM_ERR_UNSUPPORTED;        // the fused dt_proj / fp16 out_z ride on the 64-channel kernel's full fp32 inference path only
    if (p.out_z_f16) {
        if (!p.out_z_scale_ptr) return DIMSUM_ERR_NULL;
        if (p.seqlen % 32 != 0 || p.out_z_lo_offset != 0) return DIMSUM_ERR_UNSUPPORTED;
        if (!aligned_to<char>(p.out_z_ptr, 16) || p.out_z_batch_stride % 8 != 0 || p.out_z_d_stride % 8 != 0 || p.out_z_scale_ld < p.dim / 64) return DIMSUM_ERR_STRIDE;
    }
    if (sp == 16) {
        if constexpr (kN == 16) ssm_scan_fwd_launch_lanes<T>(p, stream, tiles, vec, full);
    } else if (sp == 4) {
        if constexpr (kN % 8 == 0) ssm_scan_fwd_launch_split<T, kN, 4>(p, stream, tiles, vec, full);
    } else if (sp == 2) {
        ssm_scan_fwd_launch_split<T, kN, 2>(p, stream, tiles, vec, full);
    } else {
        ssm_scan_fwd_launch_v0<T, kN>(p, stream, tiles, vec, full);
    }
    return launch_status();
}

template <typename T>
static int ssm_scan_fwd_dispatch(const ssm_args_t &p, hipStream_t stream) {
    switch (p.dstate) {
        case 4: return launch_fwd<T, 4>(p, stream);
        case 8: return launch_fwd<T, 8>(p, stream);
        case 32: return launch_fwd<T, 32>(p, stream);
        case 16: return launch_fwd<T, 16>(p, stream);
        default: return DIMSUM_ERR_SHAPE;
    }
}

}  // namespace dimsum

// flat block -> kernels (also the entry of the backward's state-rebuild sweep, ssm_scan_bwd.hip)
namespace dimsum {
int ssm_scan_fwd_run(const ssm_args_t &a, hipStream_t s) {
    const int rc = ssm_check(&a, true);
    if (rc != DIMSUM_OK) return rc;
    if (a.batch == 0) return DIMSUM_OK;
    switch (a.dtype) {
        case DIMSUM_F32: return ssm_scan_fwd_dispatch<float>(a, s);
        case DIMSUM_F16: return ssm_scan_fwd_dispatch<__half>(a, s);
        case DIMSUM_BF16: return ssm_scan_fwd_dispatch<__hip_bfloat16>(a, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}
}  // namespace dimsum

extern "C" int dimsum_ssm_scan_fwd_variant(const dimsum_ssm_params_t *p) {
    dimsum::ssm_args_t a;
    if (dimsum::ssm_args_from(p, a, true) != DIMSUM_OK || a.n_groups <= 0) return -1;
    return dimsum::ssm_scan_fwd_variant(a);
}

extern "C" int dimsum_ssm_scan_fwd(const dimsum_ssm_params_t *p, void *stream) {
    dimsum::ssm_args_t a;
    const int rc = dimsum::ssm_args_from(p, a, true);
    if (rc != DIMSUM_OK) return rc;
    return dimsum::ssm_scan_fwd_run(a, reinterpret_cast<hipStream_t>(stream));
}
