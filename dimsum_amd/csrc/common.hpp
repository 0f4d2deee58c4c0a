// common.hpp -- shared device/host helpers for libdimsum_hip (gfx950 only; wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <string.h>

#include <initializer_list>

#include "../../include/dimsum_hip.h"

namespace dimsum {

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct alignas(16) f32x4 { float v[4]; };

// ---- 4-element vector load/store of the I/O dtype, widened to f32 ------------------------------------------------
template <typename T> struct Raw4;                                   // raw register image of 4 elements
template <> struct Raw4<float> { float4 r; };
template <> struct Raw4<__half> { uint2 r; };
template <> struct Raw4<__hip_bfloat16> { uint2 r; };

template <typename T> __device__ __forceinline__ Raw4<T> ld4(const T *p);
template <> __device__ __forceinline__ Raw4<float> ld4<float>(const float *p) { return {*reinterpret_cast<const float4 *>(p)}; }
template <> __device__ __forceinline__ Raw4<__half> ld4<__half>(const __half *p) { return {*reinterpret_cast<const uint2 *>(p)}; }
template <> __device__ __forceinline__ Raw4<__hip_bfloat16> ld4<__hip_bfloat16>(const __hip_bfloat16 *p) { return {*reinterpret_cast<const uint2 *>(p)}; }

__device__ __forceinline__ f32x4 widen(const Raw4<float> &a) { return {{a.r.x, a.r.y, a.r.z, a.r.w}}; }
__device__ __forceinline__ f32x4 widen(const Raw4<__half> &a) {
    const __half2 lo = *reinterpret_cast<const __half2 *>(&a.r.x), hi = *reinterpret_cast<const __half2 *>(&a.r.y);
    const float2 l = __half22float2(lo), h = __half22float2(hi);
    return {{l.x, l.y, h.x, h.y}};
}
__device__ __forceinline__ f32x4 widen(const Raw4<__hip_bfloat16> &a) {
    return {{__uint_as_float(a.r.x << 16), __uint_as_float(a.r.x & 0xffff0000u), __uint_as_float(a.r.y << 16),
             __uint_as_float(a.r.y & 0xffff0000u)}};
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T> __device__ __forceinline__ void st4(T *p, const f32x4 &a);
template <> __device__ __forceinline__ void st4<float>(float *p, const f32x4 &a) {
    *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
template <> __device__ __forceinline__ void st4<__half>(__half *p, const f32x4 &a) {
    const __half2 lo = __floats2half2_rn(a.v[0], a.v[1]), hi = __floats2half2_rn(a.v[2], a.v[3]);
    uint2 r;
    r.x = *reinterpret_cast<const uint32_t *>(&lo);
    r.y = *reinterpret_cast<const uint32_t *>(&hi);
    *reinterpret_cast<uint2 *>(p) = r;
}
template <> __device__ __forceinline__ void st4<__hip_bfloat16>(__hip_bfloat16 *p, const f32x4 &a) {
    __hip_bfloat16 t[4] = {__float2bfloat16(a.v[0]), __float2bfloat16(a.v[1]), __float2bfloat16(a.v[2]), __float2bfloat16(a.v[3])};
    *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(t);
}

// ---- math --------------------------------------------------------------------------------------------------------
// ---- split-bf16 operand images --------------------------------------------------------------------------------------
// fp32 x = hi + lo (+ 2^-17 relative): hi = bf16(x), lo = bf16(x - hi), both round to nearest even (v_cvt_pk_bf16_f32).
// Three bf16 MFMA products hi*hi + hi*lo + lo*hi reproduce the fp32 product to ~4e-6 -- what hipBLASLt does for fp32 operands
// under the reference's allow_tf32 policy. A producer kernel can hand the library the operand already split ("split3" rows:
// [hi | hi | lo] for the left operand, [hi | lo | hi] for the weights, 3 K bf16 per row), which turns the product into ONE
// plain bf16 GEMM over 3 K -- the library's fastest kernels.
__device__ __forceinline__ void split2(float x0, float x1, unsigned &hi, unsigned &lo) {   // packed [x0 | x1 << 16]
#pragma clang fp contract(off)      // lo = bf16(x - hi) of the ROUNDED x: a producer's last multiply must not fuse into the subtraction
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = __builtin_convertvector(f2{x0, x1}, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = __builtin_convertvector(f2{r0, r1}, bf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}
// 4 consecutive columns c .. c + 3 of a split3 row of logical width N (row: 3 N bf16, 8-byte aligned pieces): kLeft = [hi | hi | lo]
template <bool kLeft> __device__ __forceinline__ void st_split3(unsigned short *row, int64_t c, int64_t N, const f32x4 &v) {
    unsigned h0, l0, h1, l1;
    split2(v.v[0], v.v[1], h0, l0);
    split2(v.v[2], v.v[3], h1, l1);
    *reinterpret_cast<uint2 *>(row + c) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(row + N + c) = kLeft ? make_uint2(h0, h1) : make_uint2(l0, l1);
    *reinterpret_cast<uint2 *>(row + 2 * N + c) = kLeft ? make_uint2(l0, l1) : make_uint2(h0, h1);
}

// the left image of 4 consecutive columns, either as the three pieces [hi | hi | lo] (row: 3 N) or as the pair [hi | lo] (row: 2 N) that the
// hand-written GEMM reads as [hi | hi | lo] (dimsum_gemm_params_t.a_alias_rows): the producers' y_split3 / out_split3 == 3
__device__ __forceinline__ void st_split_left(unsigned short *row, int64_t c, int64_t N, const f32x4 &v, bool pair) {
    unsigned h0, l0, h1, l1;
    split2(v.v[0], v.v[1], h0, l0);
    split2(v.v[2], v.v[3], h1, l1);
    *reinterpret_cast<uint2 *>(row + c) = make_uint2(h0, h1);
    if (pair) {
        *reinterpret_cast<uint2 *>(row + N + c) = make_uint2(l0, l1);
    } else {
        *reinterpret_cast<uint2 *>(row + N + c) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(row + 2 * N + c) = make_uint2(l0, l1);
    }
}

// ---- scaled-fp16 operand images ("f16s") ------------------------------------------------------------------------------
// The reference multiplies under TF32 (train.py:20-21): operands rounded to 10 mantissa bits, fp32 accumulation. fp16 has exactly that
// mantissa; what it lacks is range, so a row travels as fp16(row * 2^s) plus the exact power of two 2^-s that the consuming GEMM's
// epilogue multiplies back (dimsum_gemm_params_t.a_inv_scale / b_inv_scale). s comes from an upper bound m >= max|row|: m * 2^s lies in
// [2^14, 2^15) < 65504, so nothing overflows, and every element down to 2^-28 m keeps its full 11-bit significand (below that it is
// a subnormal / zero with absolute error <= 2^-39 m, invisible next to the 2^-12 m rounding of the row's leading elements, which
// TF32 has too). The exponent of m is clamped to [-112, 127] so that both powers of two are normal fp32 numbers: rows below 2^-112
// (incl. all-zero rows) take s = 126.
__device__ __forceinline__ void f16s_scales(float m, float &scale, float &inv) {
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    e = min(max(e, -112), 127);
    scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
    inv = __uint_as_float((unsigned)(127 - 14 + e) << 23);
}
__device__ __forceinline__ uint2 f16s_pack4(const f32x4 &v, float scale) {     // 4 consecutive columns, round to nearest even
    const __half2 a = __floats2half2_rn(v.v[0] * scale, v.v[1] * scale), b = __floats2half2_rn(v.v[2] * scale, v.v[3] * scale);
    return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
// maximum over the 64 lanes, in every lane, on the VALU only: 4 DPP steps inside each row of 16 lanes (quad_perm, row_half_mirror,
// row_mirror), then the two row exchanges v_permlane16_swap / v_permlane32_swap. (__shfl_xor is a ds_bpermute: an LDS round trip per
// step -- 96 of them per thread in the token pass that reduces 16 row maxima.)
template <int kCtrl> __device__ __forceinline__ float dpp_mov(float x) {
    const int v = __builtin_bit_cast(int, x);                  // (old = the source: see gemm_nt_kernel.hpp dpp_row_ror8)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_allmax(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));        // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, dpp_mov<0x4E>(v));        // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, dpp_mov<0x141>(v));       // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));       // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }        // v_exp_f32
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }   // v_log_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_fast(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// softplus with the reference's threshold (selective_scan_fwd_kernel.cuh:153-155): x <= 20 ? log1p(exp(x)) : x.
// log1p(e) ~= log(w) + (e - (w - 1)) / w with w = 1 + e recovers the bits 1 + e rounds away (e can be ~1e-3 here:
// dt_min = 0.001), keeping the relative error of a small step size at fp32 roundoff.
// The correction (e - (w-1)) is at most one ulp of w, so 1/w is replaced by max(2 - w, 0): exact at w = 1 where the
// correction matters, within 2e-8 absolute elsewhere -- and no v_rcp_f32 (8 issue cycles) on the per-step path.
__device__ __forceinline__ float softplus_ref(float x) {
    const float e = fast_exp(x);
    const float w = 1.0f + e;
    const float r = fmaf(e - (w - 1.0f), fmaxf(2.0f - w, 0.0f), fast_log(w));
    return x <= 20.0f ? r : x;
}

// `flag ? softplus_ref(x) : x` for a wave-uniform runtime flag WITHOUT a branch: the empty asm pins the evaluation into
// the straight-line code, so that the exp/log chains of neighbouring elements interleave (a uniform branch around each
// element ends the basic block and serialises four ~80-cycle dependent chains per 4 time steps).
__device__ __forceinline__ float softplus_if(float x, bool flag) {
    float sp = softplus_ref(x);
    asm volatile("" : "+v"(sp));
    return flag ? sp : x;
}

// ---- internal parameter blocks -----------------------------------------------------------------------------------------
// The kernels take ONE flat parameter block by value. The public structs (include/dimsum_hip.h) are versioned: a reference-shaped base
// struct + an optional extension behind `ext`. The entry points validate the sizes and flatten base + extension into these blocks
// (args_from below); fields of an extension the caller's struct_size does not cover read as 0 / NULL.
struct ssm_args_t {
    int32_t batch, dim, seqlen, dstate, n_groups, n_chunks;
    int32_t delta_softplus, dtype;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t z_batch_stride, z_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t out_z_batch_stride, out_z_d_stride;
    const void *A_ptr, *B_ptr, *C_ptr, *D_ptr, *u_ptr, *delta_ptr, *delta_bias_ptr, *z_ptr;
    void *out_ptr, *x_ptr, *out_z_ptr, *ckpt_ptr;
    int32_t kernel_variant;
    void *timing_start_event, *timing_stop_event;
    int64_t out_z_lo_offset;
    const void *dt_w_ptr, *dt_x_ptr;
    int64_t dt_w_row_stride, dt_x_row_stride;
    int32_t dt_rank, out_z_f16;
    void *out_z_scale_ptr;
    int64_t out_z_scale_ld;
};

struct ssm_bwd_args_t {
    ssm_args_t fwd;
    int64_t dout_batch_stride, dout_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_dstate_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_dstate_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t dz_batch_stride, dz_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    const void *dout_ptr;
    void *dA_ptr, *dB_ptr, *dC_ptr, *dD_ptr, *du_ptr, *dz_ptr, *ddelta_ptr, *ddelta_bias_ptr, *workspace_ptr;
    int64_t workspace_bytes;
};

// a caller's extension struct copied into a zero-filled one of the library's size: DIMSUM_ERR_ABI when it is larger than the library knows
// (or too small to hold its own struct_size); NULL = all zeros
template <typename Ext> inline int ext_from(const Ext *e, Ext &out) {
    memset(&out, 0, sizeof(Ext));
    if (!e) return DIMSUM_OK;
    const uint32_t n = e->struct_size;
    if (n < sizeof(uint32_t) || n > sizeof(Ext)) return DIMSUM_ERR_ABI;
    memcpy(&out, e, n);
    return DIMSUM_OK;
}

// base (+ ext) -> flat block. check_size: the entry points of the forward check p->struct_size; inside a backward struct only the outer
// struct's size is checked (include/dimsum_hip.h, "Versioning")
inline int ssm_args_from(const dimsum_ssm_params_t *p, ssm_args_t &a, bool check_size) {
    if (!p) return DIMSUM_ERR_NULL;
    if (check_size && p->struct_size != sizeof(dimsum_ssm_params_t)) return DIMSUM_ERR_ABI;
    dimsum_ssm_ext_t e;
    const int rc = ext_from(p->ext, e);
    if (rc != DIMSUM_OK) return rc;
    a.batch = p->batch; a.dim = p->dim; a.seqlen = p->seqlen; a.dstate = p->dstate; a.n_groups = p->n_groups; a.n_chunks = p->n_chunks;
    a.delta_softplus = p->delta_softplus; a.dtype = p->dtype;
    a.A_d_stride = p->A_d_stride; a.A_dstate_stride = p->A_dstate_stride;
    a.B_batch_stride = p->B_batch_stride; a.B_group_stride = p->B_group_stride; a.B_dstate_stride = p->B_dstate_stride;
    a.C_batch_stride = p->C_batch_stride; a.C_group_stride = p->C_group_stride; a.C_dstate_stride = p->C_dstate_stride;
    a.u_batch_stride = p->u_batch_stride; a.u_d_stride = p->u_d_stride;
    a.delta_batch_stride = p->delta_batch_stride; a.delta_d_stride = p->delta_d_stride;
    a.z_batch_stride = p->z_batch_stride; a.z_d_stride = p->z_d_stride;
    a.out_batch_stride = p->out_batch_stride; a.out_d_stride = p->out_d_stride;
    a.out_z_batch_stride = p->out_z_batch_stride; a.out_z_d_stride = p->out_z_d_stride;
    a.A_ptr = p->A_ptr; a.B_ptr = p->B_ptr; a.C_ptr = p->C_ptr; a.D_ptr = p->D_ptr; a.u_ptr = p->u_ptr; a.delta_ptr = p->delta_ptr;
    a.delta_bias_ptr = p->delta_bias_ptr; a.z_ptr = p->z_ptr;
    a.out_ptr = p->out_ptr; a.x_ptr = p->x_ptr; a.out_z_ptr = p->out_z_ptr;
    a.ckpt_ptr = e.ckpt_ptr; a.kernel_variant = e.kernel_variant;
    a.timing_start_event = e.timing_start_event; a.timing_stop_event = e.timing_stop_event;
    a.out_z_lo_offset = e.out_z_lo_offset;
    a.dt_w_ptr = e.dt_w_ptr; a.dt_x_ptr = e.dt_x_ptr; a.dt_w_row_stride = e.dt_w_row_stride; a.dt_x_row_stride = e.dt_x_row_stride;
    a.dt_rank = e.dt_rank; a.out_z_f16 = e.out_z_f16; a.out_z_scale_ptr = e.out_z_scale_ptr; a.out_z_scale_ld = e.out_z_scale_ld;
    return DIMSUM_OK;
}

// ---- host ----------------------------------------------------------------------------------------------------------
inline int launch_status() { return hipGetLastError() == hipSuccess ? DIMSUM_OK : DIMSUM_ERR_LAUNCH; }

// Kernel-boundary timing (dimsum_ssm_ext_t.timing_start_event / timing_stop_event): hipExtLaunchKernelGGL records the events
// at the begin / end of the kernel's own dispatch packet -- what rocprofv3 reports -- instead of as separate commands around it.
#define DIMSUM_LAUNCH_EV(KERNEL, GRID, BLOCK, STREAM, EV0, EV1, ...)                                              \
    do {                                                                                                          \
        if ((EV0) || (EV1)) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, EV0, EV1, 0, __VA_ARGS__);      \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, STREAM, __VA_ARGS__);                                     \
    } while (0)

template <typename T> inline bool aligned_to(const void *p, size_t bytes) { return (reinterpret_cast<uintptr_t>(p) % bytes) == 0; }

// The kernels address inside a tile with ONE 32-bit byte offset per lane (base in SGPRs + voffset). A tile spans `rows`
// rows of `row_stride` elements plus `seqlen` along the row: every such offset must stay below 2^32 bytes (and below
// 2^31 elements: the row * stride products are formed in int). Negative strides are not supported.
template <typename T> inline bool offsets_fit_32bit(int64_t seqlen, int64_t rows, std::initializer_list<int64_t> row_strides) {
    for (int64_t rs : row_strides)
        if (rs < 0 || (rows * rs + seqlen) * (int64_t)sizeof(T) >= ((int64_t)1 << 32) || rows * rs + seqlen >= ((int64_t)1 << 31)) return false;
    return true;
}

}  // namespace dimsum
