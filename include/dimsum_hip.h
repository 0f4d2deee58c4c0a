/*
 * dimsum_hip.h -- C ABI of libdimsum_hip.so: the MI355X (gfx950) native kernels of the DiMSUM denoiser hot path.
 *
 * This is the drop-in boundary. Each entry point replaces one function of the reference's two pybind extension
 * modules (or the Triton kernels it has no ROCm path for) and is what a reference-side FFI would bind:
 *
 *   dimsum_ssm_scan_fwd        <- selective_scan_cuda.fwd      mamba/csrc/selective_scan/selective_scan.cpp:226-336
 *   dimsum_ssm_scan_bwd        <- selective_scan_cuda.bwd      mamba/csrc/selective_scan/selective_scan.cpp:338-492
 *   dimsum_causal_conv1d_fwd   <- causal_conv1d_cuda.causal_conv1d_fwd[_cond]   causal-conv1d/csrc/causal_conv1d.cpp:221-336
 *   dimsum_causal_conv1d_bwd   <- causal_conv1d_cuda.causal_conv1d_bwd[_cond]   causal-conv1d/csrc/causal_conv1d.cpp:338-509
 *   dimsum_norm_fwd / _bwd     <- _layer_norm_fwd / _layer_norm_bwd (Triton)     mamba/mamba_ssm/ops/triton/layernorm.py:120-364
 *   dimsum_token_transform     <- einops/flip/local_scan/DWT/DCT chains          dimsum/models_dim.py:572-604,656-705,876-928,1496-1524
 *   dimsum_xattn_fusion_fwd/_bwd <- F.scaled_dot_product_attention x2 (+ autograd)  dimsum/attention_fusion.py:44-75
 *   dimsum_gated_gelu_fwd/_bwd <- gelu_tanh(x1) * x2                             dimsum/mlp.py:66-70
 *   dimsum_gemm_nt             <- nn.Linear / F.linear of the bias-free projections (cuBLAS TF32 GEMMs under train.py:20-21), and
 *                                 w12 + bias + gelu_tanh(x1) * x2 of the GatedMLP as ONE kernel    dimsum/mlp.py:49-70
 *
 * Conventions (same as the reference's host wrappers, minus ATen):
 *   - plain pointers, sizes and ELEMENT strides; no torch types. The caller allocates every output, including the
 *     zero-filled fp32 accumulators of the backward passes (selective_scan.cpp:458-466, causal_conv1d.cpp:405-407).
 *   - asynchronous on the hipStream_t passed in (`stream`, a `void*` so that plain C / ctypes can include this);
 *     no allocation, no synchronisation, no global state (the library has no mutable statics at all). Re-entrant.
 *   - return value: DIMSUM_OK or an error code; dimsum_status_string() gives the message the host layer raises
 *     (the reference raises RuntimeError from TORCH_CHECK at the same places).
 *   - innermost (sequence / feature) stride must be 1 for every activation tensor, like selective_scan.cpp:252-253.
 *
 * Versioning (ABI 17). Every parameter struct starts with `struct_size`: sizeof() of the struct AS THE CALLER COMPILED IT. An entry point
 * whose struct_size differs from the library's own sizeof returns DIMSUM_ERR_ABI before it reads anything else -- a caller built against an
 * older or newer header can never make a kernel read past its struct. (Inside the *_bwd_params_t structs only the outer struct_size is
 * checked; `fwd.struct_size` is ignored, `fwd.ext` is honoured.)
 * The two structs that mirror reference structs (dimsum_ssm_params_t <- SSMParamsBase, dimsum_gemm_params_t <- the arguments of F.linear)
 * hold ONLY the reference interface; everything this library offers beyond it (inference fusions, saved states, timing events, tuning)
 * lives behind `ext`, a pointer to a dimsum_*_ext_t that may be NULL (= the reference interface, nothing else). An ext struct carries its
 * own struct_size and only ever grows at its end: the library accepts any struct_size up to its own sizeof and reads the fields past the
 * caller's struct_size as 0 / NULL, so a new kernel option does not move the ABI version; an ext LARGER than the library's is
 * DIMSUM_ERR_ABI (the caller asks for something this library does not know).
 */
#ifndef DIMSUM_HIP_H
#define DIMSUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIMSUM_ABI_VERSION 17

typedef enum {
    DIMSUM_OK = 0,
    DIMSUM_ERR_NULL = 1,          /* a required pointer is NULL */
    DIMSUM_ERR_DTYPE = 2,         /* unsupported dtype code */
    DIMSUM_ERR_SHAPE = 3,         /* bad size (dstate, width, n_groups, ...) */
    DIMSUM_ERR_STRIDE = 4,        /* stride not supported (innermost != 1, overflow) */
    DIMSUM_ERR_UNSUPPORTED = 5,   /* valid in the reference but out of scope here (complex A, constant B/C) */
    DIMSUM_ERR_LAUNCH = 6,        /* hipGetLastError() after the launch */
    DIMSUM_ERR_ABI = 7            /* struct_size of a parameter struct does not match this library (stale / foreign header) */
} dimsum_status_t;

typedef enum { DIMSUM_F32 = 0, DIMSUM_F16 = 1, DIMSUM_BF16 = 2 } dimsum_dtype_t;

const char *dimsum_status_string(int status);
int dimsum_abi_version(void);

/* Measurement helpers (benchmarks; no reference counterpart): HIP events (hipEvent_t behind void*) for the per-call
 * `timing_start_event` / `timing_stop_event` fields of dimsum_ssm_ext_t. Stateless wrappers of hipEventCreate / Destroy / ElapsedTime. */
void *dimsum_event_create(void);
void dimsum_event_destroy(void *event);
float dimsum_event_elapsed_ms(void *start_event, void *stop_event);   /* after the stream has been synchronised; < 0 on error */
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char *dimsum_target_arch(void);

/* ---------------------------------------------------------------------------------------------------------------
 * Selective scan (Mamba S6), real A, input-dependent B and C.   Mirrors SSMParamsBase / SSMParamsBwd
 * (mamba/csrc/selective_scan/selective_scan.h:26-101). Shapes:
 *   u, delta, z, out, out_z, dout, du, ddelta, dz : (batch, dim, seqlen)      dtype `dtype`, innermost stride 1
 *   A : (dim, dstate) f32      D, delta_bias : (dim) f32 or NULL
 *   B, C : (batch, n_groups, dstate, seqlen)  dtype `dtype`, innermost stride 1
 *   x : (batch, dim, n_chunks, 2*dstate) f32 contiguous, n_chunks = ceil(seqlen/2048)   (selective_scan.cpp:307-313)
 *       x[...,2n] = running product of exp(delta*A_n), x[...,2n+1] = state h_n at the end of each 2048-chunk
 *   out   = C.h + D*u          out_z = out * silu(z)  (written iff z_ptr != NULL)
 * ------------------------------------------------------------------------------------------------------------- */
/* Everything beyond the reference interface (SSMParamsBase has none of it). All 0 / NULL by default; nothing here is process state. */
typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_ssm_ext_t) as the caller compiled it (see "Versioning" at the top) */
    int32_t kernel_variant;   /* forward kernel: 0 = automatic (dimsum_ssm_scan_fwd_variant tells which), else lanes per channel:
                                 1 (64 channels per wave), 2, 4, 16 (one lane per state; dstate 16). A variant the shape does
                                 not support (dstate % 4, % 8, != 16) falls back to 1. For tests and tuning. */
    void *ckpt_ptr;   /* optional (batch, ceil(seqlen/8), dstate, dim) f32: the state h BEFORE every 8th step.
                         forward: written when non-NULL (training callers keep it for the backward);
                         backward: read when non-NULL, otherwise rebuilt into workspace_ptr by one extra sweep.
                         Not part of the reference interface (its backward re-scans whole rows instead). */
    void *timing_start_event, *timing_stop_event;   /* optional hipEvent_t pair: recorded at the begin of the call's first kernel
                                 and the end of its last one (hipExtLaunchKernel: the kernels' own dispatch timestamps, the
                                 durations rocprofv3 reports). In dimsum_ssm_bwd_params_t.fwd they bracket the whole backward
                                 call incl. the state-rebuild sweep of a caller without ckpt_ptr. Not capturable into a hipGraph. */
    int64_t out_z_lo_offset;  /* forward, float32 I/O only; 0 = out_z is a `dtype` tensor (the reference's interface). != 0: out_z is written as
                                 its split-bf16 pair -- out_z_ptr = the bfloat16 `hi` plane (out_z_*_stride in bfloat16 elements), the `lo`
                                 plane lies out_z_lo_offset elements behind it; hi = bf16(x), lo = bf16(x - hi): the same 4 bytes per
                                 element (seqlen % 8 == 0), already the operand image of out_proj's GEMM (dimsum_gemm_tn with a_alias_rows), which then
                                 needs no conversion pass (mamba_simple.py:352-354 out_proj under allow_tf32). */
    /* Fused dt_proj (forward, inference, float32 I/O; no reference counterpart as a kernel: selective_scan_interface.py:840-841 computes
     * delta = dt_proj.weight @ x_dbl[:, :R]^T as a TF32 GEMM of its own and hands the (batch, dim, seqlen) result to the scan). With
     * dt_w_ptr != NULL the scan forms delta[b, d, t] = sum_r dt_w[d, r] * dt_x[r, b * seqlen + t] itself, tile by tile on the matrix
     * cores (three bf16 products per fp32 product: the arithmetic of the library GEMM it replaces), delta_ptr is NOT read (it may be
     * NULL) and the (batch, dim, seqlen) delta tensor never exists: one launch and 2 B D L 4 bytes of traffic less per mixer.
     * Served by the 64-channels-per-wave kernel on its full vector path (dim / n_groups % 64 == 0, z given, no ckpt_ptr); any other
     * call with dt_w_ptr set is DIMSUM_ERR_UNSUPPORTED (dimsum_ssm_scan_fwd_variant() == 1 tells the host beforehand). */
    const void *dt_w_ptr;     /* (dim, dt_rank) f32, 16-byte aligned rows, dt_rank % 4 == 0, dt_rank <= 32 */
    const void *dt_x_ptr;     /* (dt_rank, batch * seqlen) f32: the first dt_rank rows of x_proj's output written r-major (x_proj.weight @ conv_out
                                 as a (R + 2N, batch seqlen) matrix, the layout whose B / C rows the scan reads without a transposing copy) */
    int64_t dt_w_row_stride, dt_x_row_stride;     /* in elements; dt_w_row_stride % 4 == 0 */
    int32_t dt_rank;
    /* Block-scaled fp16 out_z (forward, inference, float32 I/O; same kernel path as dt_w_ptr): out_z_f16 != 0 -> out_z_ptr is a float16 buffer
     * (out_z_*_stride in float16 elements, rows 16-byte aligned, seqlen % 32 == 0): the 64 channels x 32 steps a wave finishes at a time are
     * stored as fp16(out_z * 2^s) with 2^-s (f32) at out_z_scale_ptr[((b * seqlen + t) / 32) * out_z_scale_ld + d / 64] -- the A operand of
     * out_proj as ONE fp16 product per element (dimsum_gemm_tn with a_block_inv_ptr): the reference multiplies out_z under TF32
     * (selective_scan_interface.py:954-981 under train.py:20-21), i.e. with 10-bit mantissas. */
    int32_t out_z_f16;
    void *out_z_scale_ptr;
    int64_t out_z_scale_ld;
} dimsum_ssm_ext_t;

/* The reference interface: field for field what set_ssm_params_fwd fills into SSMParamsBase (selective_scan.cpp:64-143;
 * selective_scan.h:26-68), plus the dtype code ATen carries in the tensors. `ext` = NULL is exactly that interface. */
typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_ssm_params_t) as the caller compiled it; anything else -> DIMSUM_ERR_ABI */
    int32_t batch, dim, seqlen, dstate, n_groups, n_chunks;
    int32_t delta_softplus;   /* bool */
    int32_t dtype;            /* dimsum_dtype_t of u/delta/z/B/C/out/out_z */
    int32_t reserved;         /* 0 */

    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t z_batch_stride, z_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t out_z_batch_stride, out_z_d_stride;

    const void *A_ptr, *B_ptr, *C_ptr, *D_ptr, *u_ptr, *delta_ptr, *delta_bias_ptr, *z_ptr;
    void *out_ptr;    /* may be NULL: inference-only callers that need just out_z skip the store */
    void *x_ptr;      /* may be NULL: skip the chunk-state store */
    void *out_z_ptr;  /* required iff z_ptr != NULL */
    const dimsum_ssm_ext_t *ext;   /* NULL = the reference interface */
} dimsum_ssm_params_t;

typedef struct {
    uint32_t struct_size;      /* sizeof(dimsum_ssm_bwd_params_t) */
    uint32_t reserved;         /* 0 */
    dimsum_ssm_params_t fwd;   /* forward operands (out_ptr = forward `out`, needed when z_ptr != NULL;
                                  out_z_ptr != NULL => recompute_out_z, selective_scan.cpp:443-449); fwd.ext->ckpt_ptr = the saved states */
    int64_t dout_batch_stride, dout_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_dstate_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_dstate_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t dz_batch_stride, dz_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    const void *dout_ptr;
    void *dA_ptr;          /* (dim, dstate) f32, zero-filled by the caller, accumulated with atomics */
    void *dB_ptr, *dC_ptr; /* (batch, n_groups, dstate, seqlen) F32 always; fully overwritten (no zero-fill needed): the
                              per-workgroup (64-channel) partial sums go through the workspace and are added in a fixed order */
    void *dD_ptr;          /* (dim) f32 zero-filled, or NULL */
    void *du_ptr, *dz_ptr, *ddelta_ptr;
    void *ddelta_bias_ptr; /* (dim) f32 zero-filled, or NULL */
    void *workspace_ptr;   /* scratch, 16-byte aligned, contents undefined on entry and exit:
                              per-workgroup partial dB / dC (2 * ceil(dim/64) * batch * dstate * seqlen f32) and, iff
                              fwd.ckpt_ptr == NULL, the rebuilt states. dimsum_ssm_scan_bwd_workspace_bytes(...) bytes
                              always suffice. */
    int64_t workspace_bytes;
} dimsum_ssm_bwd_params_t;

int dimsum_ssm_scan_fwd(const dimsum_ssm_params_t *p, void *stream);
int dimsum_ssm_scan_bwd(const dimsum_ssm_bwd_params_t *p, void *stream);
/* upper bound of the backward's workspace (partial dB / dC + rebuilt states) */
int64_t dimsum_ssm_scan_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t seqlen, int32_t dstate, int32_t n_groups);
/* Which forward kernel dimsum_ssm_scan_fwd launches for these parameters, p->kernel_variant included (a pure function of *p;
 * measurement / diagnostics; no reference counterpart: the reference has one kernel, selective_scan_fwd_kernel.cuh:67-303).
 * Lanes per channel:
 *   1 = ssm_scan_fwd_kernel        lane = channel, 64 channels per wave (launches that fill the chip)
 *   2 = ssm_scan_fwd_split_kernel  lane = (channel, state half), 32 channels per wave
 *   4 = ssm_scan_fwd_split_kernel  lane = (channel, state quarter), 16 channels per wave (few channels, long sequences)
 *  16 = ssm_scan_fwd_lanes_kernel  lane = (channel, state), 4 channels per wave, dstate 16 (fewer channels still)
 * -1 on invalid parameters. */
int dimsum_ssm_scan_fwd_variant(const dimsum_ssm_params_t *p);

/* ---------------------------------------------------------------------------------------------------------------
 * Causal depthwise conv1d, width 2..4, optional bias, optional SiLU.  Mirrors ConvParamsBase / ConvParamsBwd
 * (causal-conv1d/csrc/causal_conv1d.h:9-52).  x, out, dout, dx : (batch, dim, seqlen) innermost stride 1.
 * weight (dim, width), bias (dim): f32.  dweight/dbias: f32 zero-filled by the caller (atomics across batch).
 * The reference's `_cond` entry points alias `out` to the caller's init_x buffer and are otherwise identical
 * (SURVEY finding 1): pass that buffer as out_ptr.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_conv_params_t) */
    int32_t batch, dim, seqlen, width;
    int32_t silu_activation;  /* bool */
    int32_t dtype;            /* of x/out */
    int32_t reserved;         /* 0 */
    int64_t x_batch_stride, x_c_stride;
    int64_t weight_c_stride, weight_width_stride;
    int64_t out_batch_stride, out_c_stride;
    const void *x_ptr, *weight_ptr, *bias_ptr;
    void *out_ptr;
} dimsum_conv_params_t;

typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_conv_bwd_params_t) */
    uint32_t reserved;        /* 0 */
    dimsum_conv_params_t fwd; /* out_ptr unused */
    int64_t dout_batch_stride, dout_c_stride;
    int64_t dx_batch_stride, dx_c_stride;
    int64_t dweight_c_stride, dweight_width_stride;
    const void *dout_ptr;
    void *dx_ptr, *dweight_ptr, *dbias_ptr;
} dimsum_conv_bwd_params_t;

int dimsum_causal_conv1d_fwd(const dimsum_conv_params_t *p, void *stream);
int dimsum_causal_conv1d_bwd(const dimsum_conv_bwd_params_t *p, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused residual-add + RMSNorm / LayerNorm over rows (M, N), f32 statistics.
 *   r = x (+ x_bias) (+ residual) ; residual_out = r (if residual_out_ptr) ; y = norm(r) * weight (+ bias)
 *   optionally followed by the adaLN modulation of the row's batch element (rows_per_batch consecutive rows each):
 *       y = y * (1 + mod_scale[row / rows_per_batch, :]) + mod_shift[row / rows_per_batch, :]
 *   (x_bias = the bias of the Linear that produced x, so that the GEMM runs without a bias epilogue; with it and the
 *   modulation, "h + proj(x) -> RMSNorm -> modulate" of a DiM block is ONE pass instead of three.)
 *   rstd (M) f32 always written; mean (M) f32 written for LayerNorm.
 *   y_split3: y is written as the split-bf16 operand image the following Linear consumes (below, dimsum_split3).
 * bwd: dx = d(norm)/dr . dy (+ dresidual_out) ; dweight/dbias (N) f32 accumulated with atomics into zero-filled
 * buffers (the Triton reference reduces per-SM partials on the host, layernorm.py:324-359).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_norm_params_t) */
    int32_t rows, cols;
    int32_t is_rms_norm;
    int32_t x_dtype, residual_dtype, out_dtype; /* dimsum_dtype_t; residual_dtype covers residual and residual_out */
    float eps;
    int64_t x_row_stride, residual_row_stride, y_row_stride, residual_out_row_stride;
    const void *x_ptr, *residual_ptr, *weight_ptr, *bias_ptr;
    void *y_ptr, *residual_out_ptr, *mean_ptr, *rstd_ptr;
    const void *xbias_ptr;                      /* (N) f32 or NULL */
    const void *mod_scale_ptr, *mod_shift_ptr;  /* (M / rows_per_batch, N) f32, both or none */
    int64_t mod_row_stride;
    int32_t rows_per_batch;
    int32_t y_split3;   /* 1: y is a split-bf16 left operand image, rows of 3 N bf16 [hi | hi | lo] (out_dtype BF16, N % 4 == 0,
                         * y_row_stride >= 3 N): see dimsum_split3.  2: y is a scaled-fp16 operand image (out_dtype F16, N % 4 == 0):
                         * row r = fp16(y_r * 2^s_r) with 2^-s_r written to y_inv_scale_ptr[r]: see dimsum_rows_f16s.
                         * 3: like 1 but as the PAIR [hi | lo], rows of 2 N bf16 (y_row_stride >= 2 N), for a consumer that reads it as
                         * [hi | hi | lo] (dimsum_gemm_ext_t.a_alias_rows / b_alias_rows): a third less image traffic */
    void *y_inv_scale_ptr;                      /* (M) f32, y_split3 == 2 only */
} dimsum_norm_params_t;

typedef struct {
    uint32_t struct_size;     /* sizeof(dimsum_norm_bwd_params_t) */
    int32_t rows, cols;
    int32_t is_rms_norm;
    float eps;
    int32_t reserved;         /* 0 */
    int64_t r_row_stride, dy_row_stride, dres_row_stride, dx_row_stride;
    const void *r_ptr;       /* saved residual_out (f32) = the normalised input */
    const void *weight_ptr, *mean_ptr, *rstd_ptr;
    const void *dy_ptr;      /* f32 */
    const void *dres_ptr;    /* gradient flowing into residual_out, or NULL */
    void *dx_ptr;            /* f32; equals dresidual_in when a residual was added */
    void *dweight_ptr, *dbias_ptr; /* (N) f32 zero-filled; dbias may be NULL */
} dimsum_norm_bwd_params_t;

int dimsum_norm_fwd(const dimsum_norm_params_t *p, void *stream);
int dimsum_norm_bwd(const dimsum_norm_bwd_params_t *p, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Split-bf16 operand images for the library GEMMs (csrc/operand_split.hip). No reference counterpart: the reference's
 * Linears are TF32 GEMMs (allow_tf32, dimsum/train.py:20-21); gfx950 has no TF32 MFMA and hipBLASLt splits fp32 operands
 * into hi + lo bf16 inside the GEMM. x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a (rows, cols) f32 matrix becomes
 * (rows, 3 cols) bf16:  left != 0: [hi | hi | lo]  (activations),  left == 0: [hi | lo | hi]  (weights), so that
 *     left_image (M, 3K) . weight_image (N, 3K)^T  =  hi.hi + hi.lo + lo.hi   accumulated in f32 by ONE bf16 GEMM.
 * Producer kernels write the left image directly (dimsum_norm_params_t.y_split3, dimsum_gated_gelu_fwd_split3).
 * left == 2: the PAIR [hi | lo] ((rows, 2 cols) bf16) for dimsum_gemm_ext_t.a_alias_rows / tn_pair_*_cols.
 * cols % 4 == 0, src rows 16-byte aligned (src_row_stride % 4 == 0), dst contiguous and 8-byte aligned.
 * ------------------------------------------------------------------------------------------------------------- */
int dimsum_split3(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int32_t left, void *stream);
/* weight (rows = N, cols = K) float32 -> (3 K, N) bfloat16: the ROW stack [hi; lo; hi] of its transpose -- the b operand of dimsum_gemm_tn when
   the a operand is a d-major activation pair of planes (out_proj behind the scan's out_z_lo_offset output). rows % 2 == 0. */
int dimsum_split3_t(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, void *stream);

/* Scaled-fp16 operand image ("f16s") -- the single-product carrier of the same policy: TF32 keeps 10 mantissa bits of each operand and
 * accumulates in fp32; fp16 has exactly that mantissa, and its narrow exponent is taken out of the picture by an exact power-of-two
 * scale per row: dst[r, :] = fp16(src[r, :] * 2^s_r), 2^s_r * max|src[r, :]| in [2^14, 2^15), inv_scale[r] = 2^-s_r. The GEMM multiplies
 * the images (v_mfma_f32_16x16x32_f16, fp32 accumulation) and its epilogue applies a_inv_scale[m] * b_inv_scale[n] (dimsum_gemm_nt).
 *   src (rows, cols) f32, cols % 4 == 0; dst (rows, cols) f16; inv_scale (rows) f32;
 *   l1max: NULL, or one f32 (zeroed by the caller) that receives max_r sum_c |src[r, c]| (weights: the bound the gated epilogue uses).
 * Producer kernels write the image directly (dimsum_norm_params_t.y_split3 == 2, dimsum_tt_params_t.y_split3 == 2, GATED_GELU_F16). */
int dimsum_rows_f16s(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride,
                     void *inv_scale, void *l1max, void *stream);

/* Block-scaled fp16 image of a d-major f32 matrix (rows = channels, cols = batch x tokens): every 64 rows x 32 cols block as fp16(x 2^s), 2^-s (f32) at
 * table[(col / 32) * table_ld + row / 64] -- the layout dimsum_ssm_ext_t.out_z_f16 makes the 64-channel scan kernel write, produced here from the f32
 * out_z of a launch the state-split scan kernels serve, for dimsum_gemm_tn's a_block_inv_ptr (out_proj as one fp16 product). rows % 64 == 0,
 * cols % 256 == 0, src rows 16-byte aligned. */
int dimsum_rows_block_f16s(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride, void *table,
                           int64_t table_ld, void *stream);

/* Many scaled-fp16 conversions in ONE launch (per 24 jobs): the weight images, largest row L1 norms and bias maxima a whole denoiser forward
 * needs under the scaled-fp16 policy (host: dimsum_amd/gemm.py forward_scope). No reference counterpart: the reference's cuBLAS TF32 GEMMs
 * read the fp32 weights directly (dimsum/train.py:20-21). Same image as dimsum_rows_f16s, bit for bit. */
typedef struct {
    const void *src;            /* (rows, cols) f32 rows, 16-byte aligned, cols % 4 == 0, src_row_stride % 4 == 0 */
    void *dst;                  /* (rows, cols) f16 image, or NULL: reduce only (a bias vector: rows = 1) */
    void *inv_scale_ptr;        /* (rows) f32, required with dst */
    void *l1max_ptr;            /* 1 f32, ZERO-FILLED by the caller, or NULL: l1_factor * max_r sum_k |x_rk| */
    void *absmax_ptr;           /* 1 f32, ZERO-FILLED by the caller, or NULL: max |x| over the matrix */
    int64_t rows, cols, src_row_stride, dst_row_stride;
    float l1_factor;            /* 0 = 1 */
    int32_t reserved;
} dimsum_f16s_job_t;
int dimsum_rows_f16s_multi(const dimsum_f16s_job_t *jobs, int32_t n_jobs, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Token-space transforms on (batch, L = grid*grid tokens, channels) f32 tensors: ONE pass that fuses
 *   - a per-channel gate at load:             v[s, c] = x[b, in_index[s], c] * gate[b, c]
 *   - a transform T on every 4x4 token block of the grid (position s = row-major grid index):
 *       HAAR_FWD / HAAR_INV : 2-level Haar DWT / inverse incl. the reference's subband->channel regrouping
 *                             (WaveDiMBlock._dwt_fast / _idwt_fast, dimsum/models_dim.py:572-604)
 *       DCT_FWD / DCT_INV   : 4x4 DCT-II / inverse (dimsum/dct_layer.py:6-84, models_dim.py:876-882, 919-928)
 *   - a token permutation at load (in_index) and/or at store (out_index): int32 tables composed on the host from
 *     transpose / continuity / flip / window-scan / zigzag orders (dimsum/scanning_orders.py, models_dim.py:1496-1524)
 *   - the adaLN affine and a residual at store:
 *       y[b, out_index[s], c] = T(v)[s, c] * (1 + scale[b, c]) + shift[b, c] + residual[b, out_index[s], c]
 *   - for the backward passes (the adjoint of an orthogonal T is its inverse up to a constant, so the adjoints of
 *     both block-side fusions are again this kernel), two per-(batch, channel) reductions against a weight tensor w
 *     indexed like y:   wdot[b, c] += sum_s T(v)[s, c] * w[b, out_index[s], c]      wsum[b, c] += sum_s w[b, out_index[s], c]
 *     and the plain token sum   tsum[b, c] += sum_s T(v)[s, c]
 *     (f32 atomics into caller-zeroed (batch, channels) arrays: d scale / d gate and d shift of the adaLN modulation).
 *     With y_ptr == NULL only the reductions are produced.
 * Every optional pointer may be NULL (identity / 0). Streaming op: 2*B*L*C*4 bytes (+ residual / + w).
 * ------------------------------------------------------------------------------------------------------------- */
typedef enum {
    DIMSUM_TT_NONE = 0, DIMSUM_TT_HAAR_FWD = 1, DIMSUM_TT_HAAR_INV = 2, DIMSUM_TT_DCT_FWD = 3, DIMSUM_TT_DCT_INV = 4
} dimsum_tt_kind_t;

typedef struct {
    uint32_t struct_size;                  /* sizeof(dimsum_tt_params_t) */
    int32_t batch, tokens, channels, grid; /* tokens = grid*grid, grid % 4 == 0 unless kind == NONE */
    int32_t kind;                          /* dimsum_tt_kind_t */
    int32_t y_split3;                      /* 1: y is the split-bf16 left operand image of the Linear that consumes it: rows of
                                              3 C bf16 [hi | hi | lo] (dimsum_split3); y strides in bf16 elements, channels % 4 == 0.
                                              2: y is the scaled-fp16 image (dimsum_rows_f16s): fp16 rows of C (strides in fp16 elements,
                                              channels % 4 == 0, channels <= 1024), y_inv_scale_ptr[b * tokens + token] = 2^-s.
                                              3: like 1 as the pair [hi | lo], rows of 2 C bf16 (see dimsum_norm_params_t.y_split3) */
    int64_t x_batch_stride, x_token_stride;           /* channel stride 1 everywhere */
    int64_t res_batch_stride, res_token_stride;
    int64_t y_batch_stride, y_token_stride;
    int64_t mod_batch_stride;                         /* row stride of gate / scale / shift, each (batch, channels) */
    int64_t w_batch_stride, w_token_stride;
    int64_t red_batch_stride;                         /* row stride of wdot / wsum, each (batch, channels) */
    const void *x_ptr;
    const int32_t *in_index_ptr, *out_index_ptr;      /* (tokens) or NULL */
    const void *gate_ptr, *scale_ptr, *shift_ptr, *residual_ptr;
    void *y_ptr;                                      /* may be NULL when only the reductions are wanted */
    const void *w_ptr;                                /* (batch, tokens, channels) f32 or NULL */
    void *wdot_ptr, *wsum_ptr;                        /* (batch, channels) f32 accumulators (atomicAdd) or NULL */
    void *tsum_ptr;                                   /* (batch, channels) f32: tsum[b, c] += sum_s T(v)[s, c], or NULL */
    void *y_inv_scale_ptr;                            /* (batch, tokens) f32, y_split3 == 2 only */
    int32_t y_f16s_lds_offset;                        /* internal (set by the library) */
    int32_t reserved;                                 /* 0 */
} dimsum_tt_params_t;

int dimsum_token_transform(const dimsum_tt_params_t *p, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Cross-attention fusion core (attention_fusion.py:72-74, swap_k = False), MFMA QK^T / PV, f32 in/out:
 *   out[b, i, h*hd + e]        = softmax_j(q1[b,h,i,:].k2[b,h,j,:] * scale) v2[b,h,j,e]        (x12)
 *   out[b, i, C/2 + h*hd + e]  = softmax_j(q2.k1 * scale) v1                                    (x21)
 * qkv1, qkv2 : (batch, L, 3*heads*hd) as produced by the qkv Linear (q|k|v, head-major inside each).
 * lse (batch, 2, heads, L) f32 saved for the backward, or NULL.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t struct_size;   /* sizeof(dimsum_xattn_params_t) */
    int32_t batch, seqlen, heads, head_dim;
    float scale;
    int32_t n_dirs;   /* 0 or 2: the two swapped-KV directions above. 1: plain self-attention out = softmax(q1 k1^T) v1
                         (DiTBlock's attention, models_dim.py:1540): qkv2 / bias2 unused, out (batch, L, heads*hd),
                         lse (batch, 1, heads, L); in the backward dqkv2_ptr is unused and dqkv1 receives dq, dk, dv */
    int64_t qkv_batch_stride, qkv_token_stride;
    int64_t out_batch_stride, out_token_stride;
    const void *qkv1_ptr, *qkv2_ptr;
    void *out_ptr, *lse_ptr;
    const void *bias1_ptr, *bias2_ptr;   /* optional (3*heads*hd) f32: the qkv Linear biases, added while q / k / v are
                                            fetched, so that the qkv GEMMs can run without a bias epilogue */
    int32_t precision;  /* 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32). 1: split-bf16 -- every fp32 operand
                           x = hi + lo (two bf16), products hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 with fp32
                           accumulation, ~1e-5 relative: the same arithmetic hipBLASLt uses for the reference's
                           torch.backends.cuda.matmul.allow_tf32 = True policy (train.py:20-21) on gfx950 */
    int32_t out_split3; /* forward, precision 1 only. != 0: out rows are the split-bf16 operand image of the proj Linear:
                           3 x (n_dirs' x heads x hd) bf16 [hi | hi | lo] (dimsum_split3); out strides in bf16 elements. 3 = the pair
                           [hi | lo] (2 x ...), see dimsum_norm_params_t.y_split3.
                           precision 2: 0 (fp32 out) or 2 = the scaled-fp16 image (dimsum_rows_f16s): fp16 rows of n_dirs' x heads x hd
                           (strides in fp16 elements) + out_inv_ptr[b * L + token] */
    /* precision 2 (forward only): ONE fp16 MFMA product per element, the TF32-equivalent arithmetic (10-bit mantissas, fp32
     * accumulation). q is scaled per query row (exact), k / v per batch element from the bound
     *   |k|, |v| <= max_t (2^15 x_inv[b, t]) * wl1 + bmax,   kv_bound = {wl1 of qkv1's weight, max|bias1|, wl1 of qkv2's, max|bias2|}
     * where x*_inv are the inverse row scales (batch, L) of the scaled-fp16 images the qkv GEMMs consumed (wl1 = max_n sum_c |W_nc|). */
    const void *x1_inv_ptr, *x2_inv_ptr, *kv_bound_ptr;
    void *out_inv_ptr;  /* (batch, L) f32, out_split3 == 2 */
    int32_t qkv_f16;    /* precision 2 only. 1: qkv1 / qkv2 are the fp16 outputs of dimsum_gemm_nt's F16_QKV epilogue (biases included, scaled as
                           described there; qkv strides in fp16 elements, bias pointers unused): half the bytes of the fp32 qkv tensors */
    int32_t reserved2;
} dimsum_xattn_params_t;

int dimsum_xattn_fusion_fwd(const dimsum_xattn_params_t *p, void *stream);

/* Backward of the core (replaces the autograd of the two F.scaled_dot_product_attention calls).
 *   fwd      : the forward's operands; out_ptr = the forward's `out`, lse_ptr = its saved log-sum-exp (both required)
 *   dout     : (batch, L, 2*heads*hd) f32, strides like `out`
 *   dqkv1/2  : (batch, L, 3*heads*hd) f32, fully overwritten (direction 0 -> dq1, dk2, dv2; direction 1 -> dq2, dk1, dv1)
 *   delta    : (batch, 2, heads, L) f32 scratch (row sums of dout o out); TWICE that under fwd.precision == 2
 *   fwd.precision: 0 exact fp32 MFMA, 1 split-bf16 (three products), 2 = ONE fp16 product per element (qkv_f16 must be 0: the fp32 qkv
 *                  tensors are read). Under 2 q (times scale), k, v go to fp16 as they are (|.| <= 65504); every dout row (one query of
 *                  one head) is scaled by an exact power of two that brings its maximum to [2^-5, 2^-4), D with it, P travels as 2^8 P, and
 *                  dq is unscaled on store; in the sums over queries (dk, dv) P also carries g_min / g_q <= 1, so that all rows enter
 *                  at the scale of the (batch, head, direction)'s largest gradient row (the second half of `delta` carries the row
 *                  scales from the dq kernel to the dk / dv kernel). Range: max|v| <= 32 at head_dim 64 in the worst case. */
typedef struct {
    uint32_t struct_size;   /* sizeof(dimsum_xattn_bwd_params_t) */
    uint32_t reserved;      /* 0 */
    dimsum_xattn_params_t fwd;
    int64_t dqkv_batch_stride, dqkv_token_stride;
    const void *dout_ptr;
    void *dqkv1_ptr, *dqkv2_ptr, *delta_ptr;
} dimsum_xattn_bwd_params_t;

int dimsum_xattn_fusion_bwd(const dimsum_xattn_bwd_params_t *p, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GatedMLP epilogue (mlp.py:66-70, GELU tanh), x12 : (M, 2H) f32 = output of the w12 GEMM WITHOUT its bias:
 *   h[m, j] = gelu_tanh(x12[m, j] + bias[j]) * (x12[m, H + j] + bias[H + j])          bias (2H) f32 or NULL
 * bwd: dx12 (M, 2H); dbias (2H) f32 zero-filled by the caller (column sums of dx12, f32 atomics) or NULL.
 * ------------------------------------------------------------------------------------------------------------- */
int dimsum_gated_gelu_fwd(const void *x12, const void *bias, void *h, int64_t rows, int64_t hidden, void *stream);
/* same, h3 = the split-bf16 left operand image of the w3 GEMM: (rows, 3 hidden) bf16, [hi | hi | lo] (dimsum_split3) */
int dimsum_gated_gelu_fwd_split3(const void *x12, const void *bias, void *h3, int64_t rows, int64_t hidden, void *stream);
int dimsum_gated_gelu_bwd(const void *x12, const void *bias, const void *dh, void *dx12, void *dbias, int64_t rows,
                          int64_t hidden, void *stream);
/* same, dx12_image = dx12 as a split-bf16 operand image in WEIGHT order: (rows, 3 x 2 hidden) bf16 [hi | lo | hi] (dimsum_split3),
 * paired with left-order images of W12^T (input gradient) and, through the (3 rows, .) view of both, of the MLP input (weight gradient) */
int dimsum_gated_gelu_bwd_split3(const void *x12, const void *bias, const void *dh, void *dx12_image, void *dbias, int64_t rows,
                                 int64_t hidden, void *stream);
/* the same with dx12 as the PAIR [hi | lo] (rows of 2 x 2 hidden bf16): see dimsum_gemm_ext_t.a_alias_weight_order / tn_pair_a_cols */
int dimsum_gated_gelu_bwd_pair(const void *x12, const void *bias, const void *dh, void *dx12_pair, void *dbias, int64_t rows, int64_t hidden, void *stream);
/* the same with dx12 as a scaled-fp16 operand image (dimsum_rows_f16s): dx12_image (rows, 2 hidden) float16, inv_scale (rows) f32 (exact row maxima):
 * the operand of both backward GEMMs of w12 under the scaled-fp16 policy (dimsum_gemm_nt; dimsum_gemm_tn with k_scale_ptr). hidden <= 5120. */
int dimsum_gated_gelu_bwd_f16s(const void *x12, const void *bias, const void *dh, void *dx12_image, void *inv_scale, void *dbias, int64_t rows,
                               int64_t hidden, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * NT GEMM with a fused Linear epilogue: C (m, n) = A (m, k) . B (n, k)^T, 16-bit operands (bf16 or fp16 rows, k contiguous), fp32
 * accumulation on v_mfma_f32_16x16x32_*. Replaces the library GEMM behind F.linear(x, weight) for the large bias-free projections
 * of the denoiser (dimsum/mlp.py:66-70 w12 / w3, mamba_simple.py in_proj / out_proj, attention_fusion.py qkv / proj). The operands are
 * what the producer kernels of this library write: split-bf16 images over 3 K ([hi | hi | lo] rows against [hi | lo | hi] weights:
 * the reference's TF32 policy, dimsum_split3) or scaled fp16 rows.
 *   epilogue F32            : C fp32 (m, n), row stride ldc
 *            F32_BIAS       : C = A B^T + bias[n]
 *            GATED_GELU_SPLIT3 : B = the w12 weight (n = 2 F rows: x1 rows [0, F), x2 rows [F, 2 F)), bias (2 F) fp32 or NULL;
 *                             C = the LEFT split-bf16 image (m, 3 F) of h = gelu_tanh(x1 + b1) * (x2 + b2), ldc in bf16 elements:
 *                             the fp32 x12 tensor of mlp.py:68 never exists
 *            GATED_GELU_F16 : same, C = fp16 (m, F) of h * out_scale
 *            F32_GATE_RESIDUAL : C = residual + gate[row / rows_per_batch] * (A B^T + bias): the residual tail of a block
 *                             ("x = x + gate * mlp(...)", models_dim.py:1107-1113) in the epilogue of its last Linear; residual (m, n) f32,
 *                             gate (m / rows_per_batch, n) f32 or NULL (= 1), rows_per_batch % 256 == 0, bias NULL or (n)
 * Shapes: m % 256 == 0, k % 64 == 0, k >= 128, n % 4 == 0 (gated: n % 16 == 0, ldc % 8 == 0); lda, ldb % 8 == 0; a_ptr, b_ptr, c_ptr
 * 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------------- */
typedef enum {
    DIMSUM_GEMM_EPI_F32 = 0, DIMSUM_GEMM_EPI_GATED_GELU_SPLIT3 = 1, DIMSUM_GEMM_EPI_GATED_GELU_F16 = 2, DIMSUM_GEMM_EPI_F32_BIAS = 3,
    DIMSUM_GEMM_EPI_F32_GATE_RESIDUAL = 4,
    DIMSUM_GEMM_EPI_F32_CONV = 6,   /* in_proj of a Mamba mixer with the causal conv1d + bias + SiLU of its x half in the epilogue (mamba_simple.py in_proj followed
                                       by causal_conv1d_fn, selective_scan_interface.py:616): C (m, n) f32 d-major (rows = channels, columns = tokens); rows
                                       [0, conv_rows) hold silu(conv(A B^T) + conv_bias) along the columns, in sequences of conv_seq columns (256 % conv_seq == 0,
                                       n % conv_seq == 0: zero history at every sequence start), the other rows the plain product. conv_weight (conv_rows,
                                       conv_width 2..4) f32 row stride conv_weight_ld, conv_bias (conv_rows) f32 or NULL. conv_rows % 128 == 0. */
    DIMSUM_GEMM_EPI_F16_QKV = 5     /* the qkv Linear of the attention fusion under the scaled-fp16 policy (attention_fusion.py:52-60, models_dim.py:1470):
                                       C (m, n = 3 C') fp16 = fp16((A B^T + bias) 2^s), scaled-fp16 operands (a / b_inv_scale_ptr required). Columns
                                       [0, qkv_q_cols) (q) take one power-of-two scale per ROW, the rest (k, v) one per BATCH ELEMENT (rows_per_batch rows,
                                       a multiple of 256), both from the bound |x W^T + b| <= 2^15 a_inv * wl1 + bmax with gate_bound_ptr = {wl1, bmax}:
                                           s_row = scale(2 (2^15 a_inv[row] wl1 + bmax)),  s_batch = scale(2 (2^15 max_t a_inv[b, t] wl1 + bmax))
                                       -- exactly what dimsum_xattn_fusion_fwd (precision 2, qkv_f16 = 1) assumes of its fp16 inputs: it recomputes the
                                       same scales from the same a_inv and bound. Halves the bytes between the two kernels (no fp32 qkv tensor). */
} dimsum_gemm_epilogue_t;

/* Everything beyond C = A B^T (+ bias): fused epilogue operands, operand-image read modes, timing, tuning. All 0 / NULL by default. */
typedef struct {
    uint32_t struct_size;         /* sizeof(dimsum_gemm_ext_t) as the caller compiled it (see "Versioning" at the top) */
    int32_t rows_per_batch;       /* F32_GATE_RESIDUAL with a gate, F16_QKV */
    void *timing_start_event, *timing_stop_event;   /* optional hipEvent_t pair recorded at the kernel's own dispatch boundaries */
    /* tune_variant: 0 = the library's choice of tile shape / schedule (dimsum_gemm_nt_kernel_for reports it). 512 = 128 x 256 tiles by 4-wave
     * workgroups (scaled-fp16 operands only), 513 = 256 x 256 tiles, one workgroup per tile (never the persistent stream), 514 = the persistent
     * K-tile stream (one workgroup per CU walking the tile list) where the shape allows it (k / 64 even and >= 4, more tiles than CUs, no
     * aliased operand), else as 513. Other values: tuning builds only (-DDIMSUM_GEMM_TUNE), DIMSUM_ERR_UNSUPPORTED otherwise.
     * tune_group_m: tile rows per L2 patch of the tile walk (0 = automatic). tune_reserved: start delay of the odd CUs (tuning builds). */
    int32_t tune_variant, tune_group_m, tune_reserved;
    int32_t c_image_pieces;       /* GATED_GELU_SPLIT3: 0 / 3 = the image [hi | hi | lo] (ldc >= 3 F); 2 = the pair [hi | lo] (ldc >= 2 F), for a consumer that
                                     reads it with a_alias_rows: a third less image traffic, the same three products */
    /* GATED_GELU_F16 over scaled operands: the h image gets one power-of-two scale per row, derived WITHOUT a row reduction from the bound
     * |x1|, |x2| <= max|a_m| * gate_bound[0] + gate_bound[1]  (gate_bound = {max_n sum_k |w_nk|, max |bias|}, 2 f32 on the device);
     * its inverse goes to h_inv_scale[m] -- the a_inv_scale of the w3 GEMM. NULL: out_scale for every row. */
    const void *gate_bound_ptr;
    void *h_inv_scale_ptr;        /* (m) f32 */
    /* F32_GATE_RESIDUAL */
    const void *residual_ptr, *gate_ptr;
    int64_t residual_ld, gate_ld;
    /* GATED_GELU_SPLIT3, training forward (mlp.py:66-70 under autograd): when non-NULL the bias-free accumulators [x1 | x2] are ALSO stored
       as float32 (m, n) rows with stride x12_ld -- what the gated-GeLU adjoint of the backward reads; bf16 images only. */
    void *x12_ptr;
    int64_t x12_ld;
    /* != 0 = the A operand's reduction indices r >= a_alias_rows are the indices r - a_alias_rows of a_ptr (k = 3 a_alias_rows, % 64 == 0): a
       [hi | lo] pair (dimsum_gemm_nt: 2 C columns per row; dimsum_gemm_tn: a [hi; lo] pair of planes) serves as the left operand image
       [hi | hi | lo] without storing hi twice. */
    int64_t a_alias_rows;
    int64_t b_alias_rows;         /* dimsum_gemm_nt, F32 epilogue: the same for the B rows (in_proj: the activation image is the right operand) */
    /* training (dimsum/mlp.py:66-70 under autograd): the gradient images as pairs too.
       a_alias_weight_order (dimsum_gemm_nt, with a_alias_rows = C): the pair is read in WEIGHT order [hi | lo | hi] (dx = dy_w . (W^T)image).
       tn_pair_a_cols / tn_pair_b_cols (dimsum_gemm_tn): both operands are pairs -- a_ptr rows [hi | lo] with lo at column tn_pair_a_cols (read
       in weight order), b_ptr rows [hi | lo] with lo at column tn_pair_b_cols (read in left order); k = the rows of one piece, splits = 3 x the
       number of row ranges: partial result (piece, range) pairs A's piece with B's piece over that range. */
    int32_t a_alias_weight_order, qkv_q_cols;      /* qkv_q_cols: F16_QKV only (% 16 == 0) */
    const void *conv_weight_ptr, *conv_bias_ptr;   /* F32_CONV only */
    int32_t conv_rows, conv_width, conv_seq, conv_weight_ld;
    /* dimsum_gemm_tn, float16 operands, splits == 1, k <= 4096: block-scaled A (the scan's fp16 out_z, dimsum_ssm_ext_t.out_z_f16, with its
     * table as it is): a_block_inv_ptr is a (m / 32, a_block_inv_ld >= k / 64) float32 table of inverse scales (powers of two): the A values of
     * tokens [32 g, 32 g + 32) in reduction rows [64 t, 64 t + 64) stand for value * a_block_inv[g][t]. The kernel puts a token group on ONE
     * scale -- its row's largest inverse -- by multiplying the blocks by exact powers of two <= 1 as it reads them, and multiplies the result by
     * that scale and b_inv_scale_ptr[n]; a_inv_scale_ptr must be NULL. */
    const void *a_block_inv_ptr;
    int64_t a_block_inv_ld;
    int64_t tn_pair_a_cols, tn_pair_b_cols;
    /* dimsum_gemm_tn, float16 operands: the weight gradient dW = dY^T X of a Linear under the scaled-fp16 policy (torch.mm(dy.t(), x) under
     * train.py:20-21's TF32). Both operands are scaled-fp16 images with one power-of-two scale per ROW -- the images the forward and the input
     * gradient use -- but their rows are the reduction index here: k_scale_ptr[r] (k float16: a_inv[r] b_inv[r] / max_r(a_inv b_inv), powers of
     * two <= 1) multiplies A's row r as it is read and *c_scale_ptr (one f32 on the device: that maximum) multiplies the result;
     * a_inv_scale_ptr / b_inv_scale_ptr NULL. k / splits <= 16384 reduction rows per range. */
    const void *k_scale_ptr;
    const void *c_scale_ptr;
    /* the same factors formed INSIDE the kernel (no dimsum_row_factors launch in front of the product): k_inv_a_ptr (k f32) and k_inv_b_ptr (k f32
     * or NULL = 1) are the two images' inverse row scales; every workgroup multiplies the pair over its OWN reduction range, normalises by the
     * range's maximum (powers of two: exact), keeps the factors in LDS as float16 and multiplies its partial result by that maximum -- ranges
     * are summed in fp32 afterwards, so no global maximum is needed. k_scale_ptr / c_scale_ptr must be NULL then. */
    const void *k_inv_a_ptr;
    const void *k_inv_b_ptr;
} dimsum_gemm_ext_t;

/* The Linear itself: what F.linear(x, weight, bias) is given (plus the operand dtype and the power-of-two scales the scaled-fp16 operand
 * images carry). `ext` = NULL: C = A B^T (+ bias), fp32 (or the gated epilogues with a constant out_scale). */
typedef struct {
    uint32_t struct_size;         /* sizeof(dimsum_gemm_params_t) as the caller compiled it; anything else -> DIMSUM_ERR_ABI */
    int32_t m, n, k;
    int32_t operand_dtype;        /* DIMSUM_BF16 or DIMSUM_F16 (both operands) */
    int32_t epilogue;             /* dimsum_gemm_epilogue_t */
    float out_scale;              /* GATED_GELU_F16 only */
    int32_t reserved;             /* 0 */
    int64_t lda, ldb, ldc;        /* row strides in elements of the respective dtype */
    const void *a_ptr, *b_ptr, *bias_ptr;
    void *c_ptr;
    /* scaled-fp16 operand images (dimsum_rows_f16s): C[m, n] = acc * a_inv_scale[m] * b_inv_scale[n] (exact powers of two), both or none */
    const void *a_inv_scale_ptr;  /* (m) f32 */
    const void *b_inv_scale_ptr;  /* (n) f32 */
    const dimsum_gemm_ext_t *ext; /* NULL = the plain Linear */
} dimsum_gemm_params_t;

int dimsum_gemm_nt(const dimsum_gemm_params_t *p, void *stream);
/* Which kernel dimsum_gemm_nt launches for these parameters (a pure function of *p and of the current device's CU count; nothing is launched):
 *   0 = gemm_nt_kernel         256 x 256 tiles, one 8-wave workgroup per tile
 *   1 = gemm_nt_m128_kernel    128 x 256 tiles, 4-wave workgroups, two per CU (short-K launches with fp32-family epilogues)
 *   2 = gemm_nt_persist_kernel one workgroup per CU walking the tile list as one K-tile stream (the gated w12 GEMM under the scaled-fp16 policy)
 * or -(error status) for parameters dimsum_gemm_nt would refuse. Measurement / tests; no reference counterpart. */
int dimsum_gemm_nt_kernel_for(const dimsum_gemm_params_t *p);

/* The weight-gradient shape of the same Linears under autograd (torch.mm(dy.t(), x) in the reference's autograd graph; dimsum/mlp.py:66-70,
   attention_fusion.py:44-79): C[s] (m, n) float32 = sum over the rows r of reduction range s of A[r, 0..m)^T B[r, 0..n). a_ptr: (k, m) rows
   with stride lda, b_ptr: (k, n) rows with stride ldb (16-bit, the split-bf16 images viewed as (3 rows, features) stacks), k = all
   reduction rows, cut into `splits` equal ranges (k % (64 splits) == 0) whose partial results lie c_split_stride floats apart in c_ptr --
   the caller adds them (a fixed order: bitwise reproducible). m % 256 == 0; n % 256 == 0, or n % 4 == 0 with B's rows ZERO-PADDED to the next multiple
   of 256 columns (ldb >= that multiple: the last column tile reads the padding; only columns < n are stored); epilogue F32 only; the other fields as above. */
int dimsum_gemm_tn(const dimsum_gemm_params_t *p, int32_t splits, int64_t c_split_stride, void *stream);

/* The weight-gradient shape of the Mamba projections under autograd (selective_scan_interface.py:954-981: "eB,dB->ed" d out_proj.weight,
   d in_proj.weight = dxz x): one operand is a d-major activation -- A (m, k) float16 rows CONTIGUOUS along the reduction index (channels x tokens,
   lda) -- the other token-major -- B (k, n) float16 rows OVER the reduction index (ldb): C[s] (m, n) float32 = sum over the rows r of range s of
   A[0..m, r] B[r, 0..n). Scaled-fp16 images only: a_inv_scale_ptr (m) = A's row scales (its rows are output rows; NULL = 1), ext->k_scale_ptr (k)
   float16 = B's row scales as per-reduction-row factors (/ their maximum), ext->c_scale_ptr = that maximum -- or ext->k_inv_a_ptr (k f32) = B's row
   scales themselves, normalised inside the kernel; b_inv_scale_ptr NULL. `splits`
   ranges of k / splits <= 16384 rows, k % (64 splits) == 0; m % 256 == 0, n % 256 == 0; partial results c_split_stride floats apart. */
int dimsum_gemm_nn(const dimsum_gemm_params_t *p, int32_t splits, int64_t c_split_stride, void *stream);
/* the k_scale_ptr / c_scale_ptr operands of the two entry points above from the row scales of the two images: k_scale[r] = float16(a_inv[r] b_inv[r] /
   c_scale), c_scale = max_r a_inv[r] b_inv[r]; a_inv, b_inv (n) f32 inverse row scales (b_inv NULL = 1), k_scale (n) float16, c_scale 1 f32. One launch. */
int dimsum_row_factors(const void *a_inv, const void *b_inv, int64_t n, void *k_scale, void *c_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIMSUM_HIP_H */
