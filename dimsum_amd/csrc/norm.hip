// norm.hip -- fused residual-add + RMSNorm / LayerNorm, forward and backward, for gfx950.
//
// Replaces the Triton kernels the reference uses unconditionally for RMSNorm (mamba/mamba_ssm/ops/triton/layernorm.py:
// _layer_norm_fwd_1pass_kernel :61-117, _layer_norm_bwd_kernel :190-285; semantics = rms_norm_ref/layer_norm_ref :19-45):
//     r = x (+ residual)           residual_out = r   (fp32 when residual_in_fp32)
//     RMS: y = r * rsqrt(mean(r^2) + eps) * w (+ b)      LN: y = (r - mean) * rsqrt(var + eps) * w + b
// Streaming op: forward prenorm moves 4*M*N*4 bytes (x, residual in; y, residual_out out).
//
// MI355X design: one wave64 per row, the row lives in registers (16 B per lane per piece, up to 8 pieces = 2048 columns;
// wider rows take a re-reading path), statistics by cross-lane butterflies -- no LDS, no block barrier. The backward
// keeps per-lane dweight/dbias partial sums in registers over all the rows a wave owns and flushes them with one
// atomic per column per wave (the Triton reference writes per-SM partials and reduces them on the host).
#include "common.hpp"

namespace dimsum {

constexpr int kMaxPieces = 8;   // 8 * 256 = 2048 columns held in registers

__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

template <typename T> __device__ __forceinline__ f32x4 ld_cols(const T *row, int c, int N, bool vec) {
    if (vec) return c < N ? widen(ld4<T>(row + c)) : f32x4{{0.f, 0.f, 0.f, 0.f}};
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = (c + e < N) ? to_f32<T>(row[c + e]) : 0.f;
    return r;
}
template <typename T> __device__ __forceinline__ void st_cols(T *row, int c, int N, bool vec, const f32x4 &v) {
    if (vec) { if (c < N) st4<T>(row + c, v); return; }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < N) row[c + e] = from_f32<T>(v.v[e]);
}

// TX: dtype of x, TR: dtype of residual / residual_out, TY: dtype of y. Weights fp32.
template <typename TX, typename TR, typename TY, int kPieces>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const dimsum_norm_params_t p, const bool vec) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = p.cols;
    const float inv_n = 1.0f / (float)N;
    const float *w = reinterpret_cast<const float *>(p.weight_ptr);
    const float *bb = reinterpret_cast<const float *>(p.bias_ptr);
    const float *xb = reinterpret_cast<const float *>(p.xbias_ptr);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)gridDim.x * 4) {
        const TX *x = reinterpret_cast<const TX *>(p.x_ptr) + row * p.x_row_stride;
        const float *msc = nullptr, *msh = nullptr;
        if (p.mod_scale_ptr) {
            const int64_t bi = row / p.rows_per_batch;
            msc = reinterpret_cast<const float *>(p.mod_scale_ptr) + bi * p.mod_row_stride;
            msh = reinterpret_cast<const float *>(p.mod_shift_ptr) + bi * p.mod_row_stride;
        }
        const TR *res = p.residual_ptr ? reinterpret_cast<const TR *>(p.residual_ptr) + row * p.residual_row_stride : nullptr;
        TR *ro = p.residual_out_ptr ? reinterpret_cast<TR *>(p.residual_out_ptr) + row * p.residual_out_row_stride : nullptr;
        TY *y = reinterpret_cast<TY *>(p.y_ptr) + row * p.y_row_stride;
        f32x4 r[kPieces];
        float s = 0.f;
        // one branch per OPERAND, not per piece: no load moves across a branch, so `if (res)` inside the piece loop made every piece's loads wait
        // for the previous piece's (the same finding as in the blocked token passes)
#pragma unroll
        for (int i = 0; i < kPieces; ++i) r[i] = ld_cols<TX>(x, (i * kWave + lane) * 4, N, vec);
        if (xb) {
            f32x4 q[kPieces];
#pragma unroll
            for (int i = 0; i < kPieces; ++i) q[i] = ld_cols<float>(xb, (i * kWave + lane) * 4, N, vec);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i].v[e] += q[i].v[e];
        }
        if (res) {
            f32x4 q[kPieces];
#pragma unroll
            for (int i = 0; i < kPieces; ++i) q[i] = ld_cols<TR>(res, (i * kWave + lane) * 4, N, vec);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i].v[e] += q[i].v[e];
        }
        if (ro) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) st_cols<TR>(ro, (i * kWave + lane) * 4, N, vec, r[i]);
        }
#pragma unroll
        for (int i = 0; i < kPieces; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += p.is_rms_norm ? r[i].v[e] * r[i].v[e] : r[i].v[e];
        s = wave_allsum(s);
        float mean = 0.f, var;
        if (p.is_rms_norm) {
            var = s * inv_n;
        } else {   // two-pass variance on the register copy, like the Triton kernel (layernorm.py:93-96)
            mean = s * inv_n;
            float v2 = 0.f;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * kWave + lane) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = (c + e < N) ? r[i].v[e] - mean : 0.f; v2 += dlt * dlt; }
            }
            var = wave_allsum(v2) * inv_n;
        }
        const float rstd = 1.0f / sqrtf(var + p.eps);
        if (lane == 0) {
            if (p.rstd_ptr) reinterpret_cast<float *>(p.rstd_ptr)[row] = rstd;
            if (p.mean_ptr && !p.is_rms_norm) reinterpret_cast<float *>(p.mean_ptr)[row] = mean;
        }
        float ymax = 0.f;                           // y_split3 == 2: the row waits in r[] for its exact maximum (scaled-fp16 image)
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int c = (i * kWave + lane) * 4;
            if (c < N) {
                const f32x4 wv = ld_cols<float>(w, c, N, vec);
                const f32x4 bv = bb ? ld_cols<float>(bb, c, N, vec) : f32x4{{0.f, 0.f, 0.f, 0.f}};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = (r[i].v[e] - mean) * rstd * wv.v[e] + bv.v[e];
                if (msc) {
                    const f32x4 sc = ld_cols<float>(msc, c, N, vec), sh = ld_cols<float>(msh, c, N, vec);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = fmaf(o.v[e], 1.0f + sc.v[e], sh.v[e]);
                }
                if (p.y_split3 == 2) {
                    r[i] = o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ymax = fmaxf(ymax, fabsf(o.v[e]));
                } else if (p.y_split3) st_split_left(reinterpret_cast<unsigned short *>(p.y_ptr) + row * p.y_row_stride, c, N, o, p.y_split3 == 3);
                else st_cols<TY>(y, c, N, vec, o);
            }
        }
        if (p.y_split3 == 2) {
            float sc, inv;
            f16s_scales(wave_allmax(ymax), sc, inv);
            __half *yh = reinterpret_cast<__half *>(p.y_ptr) + row * p.y_row_stride;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * kWave + lane) * 4;
                if (c < N) *reinterpret_cast<uint2 *>(yh + c) = f16s_pack4(r[i], sc);
            }
            if (lane == 0) reinterpret_cast<float *>(p.y_inv_scale_ptr)[row] = inv;
        }
    }
}

// backward (fp32 buffers). dx = (w*dy - xhat*c1 [- c2]) * rstd (+ dres);  c1 = mean(xhat*w*dy), c2 = mean(w*dy)
template <int kPieces>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const dimsum_norm_bwd_params_t p, const bool vec) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = p.cols;
    const float inv_n = 1.0f / (float)N;
    const float *w = reinterpret_cast<const float *>(p.weight_ptr);
    f32x4 wreg[kPieces], dw[kPieces], db[kPieces];
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int c = (i * kWave + lane) * 4;
        wreg[i] = ld_cols<float>(w, c, N, vec);
        dw[i] = {{0.f, 0.f, 0.f, 0.f}};
        db[i] = {{0.f, 0.f, 0.f, 0.f}};
    }
    // TWO rows per trip, all six row streams requested before the first reduction: a wave that walks its rows one at a time exposes a full
    // memory round trip per row (the grid is capped at ~2 waves per SIMD because every workgroup ends in N atomics), 3.4 TB/s at 16384 rows.
    // dres is only needed after the row reduction but is requested up front with the rest.
    const int64_t stride = (int64_t)gridDim.x * 4;
    constexpr int kRows = kPieces <= 5 ? 2 : 1;          // (8 pieces: two rows in flight would need 336 VGPRs)
    for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < p.rows; row0 += kRows * stride) {
        f32x4 rv[kRows][kPieces], g[kRows][kPieces], dr[kRows][kPieces];
        float rstd[kRows], mean[kRows];
        bool live[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            const int64_t row = row0 + u * stride;
            live[u] = row < p.rows;
            const int64_t lr = live[u] ? row : row0;          // (a missing second row re-reads the first: no branch around the loads)
            const float *r = reinterpret_cast<const float *>(p.r_ptr) + lr * p.r_row_stride;
            const float *dy = reinterpret_cast<const float *>(p.dy_ptr) + lr * p.dy_row_stride;
            const float *dres = p.dres_ptr ? reinterpret_cast<const float *>(p.dres_ptr) + lr * p.dres_row_stride : nullptr;
            rstd[u] = reinterpret_cast<const float *>(p.rstd_ptr)[lr];
            mean[u] = (p.is_rms_norm || !p.mean_ptr) ? 0.f : reinterpret_cast<const float *>(p.mean_ptr)[lr];
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * kWave + lane) * 4;
                rv[u][i] = ld_cols<float>(r, c, N, vec);
                g[u][i] = ld_cols<float>(dy, c, N, vec);
                dr[u][i] = (dres && c < N) ? ld_cols<float>(dres, c, N, vec) : f32x4{{0.f, 0.f, 0.f, 0.f}};
            }
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
            if (!live[u]) continue;                            // (wave-uniform)
            float *dx = reinterpret_cast<float *>(p.dx_ptr) + (row0 + u * stride) * p.dx_row_stride;
            f32x4 xh[kPieces], wdy[kPieces];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * kWave + lane) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = c + e < N;
                    xh[i].v[e] = in ? (rv[u][i].v[e] - mean[u]) * rstd[u] : 0.f;
                    wdy[i].v[e] = wreg[i].v[e] * g[u][i].v[e];
                    c1 += xh[i].v[e] * wdy[i].v[e];
                    c2 += wdy[i].v[e];
                    dw[i].v[e] = fmaf(g[u][i].v[e], xh[i].v[e], dw[i].v[e]);
                    db[i].v[e] += g[u][i].v[e];
                }
            }
            c1 = wave_allsum(c1) * inv_n;
            c2 = p.is_rms_norm ? 0.f : wave_allsum(c2) * inv_n;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * kWave + lane) * 4;
                if (c < N) {
                    f32x4 o = dr[u][i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] += (wdy[i].v[e] - (xh[i].v[e] * c1 + c2)) * rstd[u];
                    st_cols<float>(dx, c, N, vec, o);
                }
            }
        }
    }
    // The 4 waves of the workgroup add their column sums in LDS first: one atomic per column per WORKGROUP (the atomics were
    // ~40 % of the kernel's time with one per wave), which also leaves room for twice the waves in flight.
    __shared__ __attribute__((aligned(16))) float sred[4 * kPieces * kWave * 4];
    float *dwp = reinterpret_cast<float *>(p.dweight_ptr), *dbp = reinterpret_cast<float *>(p.dbias_ptr);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {          // (unrolled: `pass ? db : dw` on a runtime index put both arrays into scratch)
        if (pass && !dbp) break;
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < kPieces; ++i)
            *reinterpret_cast<f32x4 *>(&sred[(wave * kPieces * kWave + i * kWave + lane) * 4]) = pass ? db[i] : dw[i];
        __syncthreads();
        float *dst = pass ? dbp : dwp;
        for (int i = threadIdx.x; i < kPieces * kWave; i += 256) {          // 16-byte column group i
            f32x4 acc = *reinterpret_cast<const f32x4 *>(&sred[i * 4]);
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(&sred[(wv * kPieces * kWave + i) * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc.v[e] += t.v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i * 4 + e < N) atomicAdd(dst + i * 4 + e, acc.v[e]);
        }
    }
}

template <typename TX, typename TR, typename TY>
static int launch_norm_fwd(const dimsum_norm_params_t &p, hipStream_t s) {
    const int N = p.cols;
    bool vec = N % 4 == 0 && aligned_to<TX>(p.x_ptr, 4 * sizeof(TX)) && aligned_to<TY>(p.y_ptr, 4 * sizeof(TY)) &&
               p.x_row_stride % 4 == 0 && p.y_row_stride % 4 == 0 && aligned_to<float>(p.weight_ptr, 16) &&
               (!p.bias_ptr || aligned_to<float>(p.bias_ptr, 16));
    if (p.residual_ptr) vec = vec && aligned_to<TR>(p.residual_ptr, 4 * sizeof(TR)) && p.residual_row_stride % 4 == 0;
    if (p.residual_out_ptr) vec = vec && aligned_to<TR>(p.residual_out_ptr, 4 * sizeof(TR)) && p.residual_out_row_stride % 4 == 0;
    if (p.xbias_ptr) vec = vec && aligned_to<float>(p.xbias_ptr, 16);
    if (p.mod_scale_ptr) vec = vec && aligned_to<float>(p.mod_scale_ptr, 16) && aligned_to<float>(p.mod_shift_ptr, 16) && p.mod_row_stride % 4 == 0;
    const int pieces = (N + 255) / 256;
    const int64_t blocks = (p.rows + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), block(256);
#define DIMSUM_NF(K) hipLaunchKernelGGL((norm_fwd_kernel<TX, TR, TY, K>), grid, block, 0, s, p, vec)
    if (pieces <= 1) DIMSUM_NF(1);
    else if (pieces <= 2) DIMSUM_NF(2);
    else if (pieces <= 4) DIMSUM_NF(4);
    else if (pieces <= 5) DIMSUM_NF(5);
    else if (pieces <= kMaxPieces) DIMSUM_NF(8);
    else return DIMSUM_ERR_SHAPE;
#undef DIMSUM_NF
    return launch_status();
}

template <typename TX, typename TR>
static int dispatch_out(const dimsum_norm_params_t &p, hipStream_t s) {
    switch (p.out_dtype) {
        case DIMSUM_F32: return launch_norm_fwd<TX, TR, float>(p, s);
        case DIMSUM_F16: return launch_norm_fwd<TX, TR, __half>(p, s);
        case DIMSUM_BF16: return launch_norm_fwd<TX, TR, __hip_bfloat16>(p, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}
template <typename TX>
static int dispatch_res(const dimsum_norm_params_t &p, hipStream_t s) {
    switch (p.residual_dtype) {   // the reference keeps the residual stream in fp32 (residual_in_fp32) or in x's dtype
        case DIMSUM_F32: return dispatch_out<TX, float>(p, s);
        case DIMSUM_F16: return dispatch_out<TX, __half>(p, s);
        case DIMSUM_BF16: return dispatch_out<TX, __hip_bfloat16>(p, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}

}  // namespace dimsum

extern "C" int dimsum_norm_fwd(const dimsum_norm_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_norm_params_t)) return DIMSUM_ERR_ABI;
    if (!p->x_ptr || !p->weight_ptr || !p->y_ptr) return DIMSUM_ERR_NULL;
    if (p->rows < 0 || p->cols <= 0) return DIMSUM_ERR_SHAPE;
    if ((p->mod_scale_ptr == nullptr) != (p->mod_shift_ptr == nullptr)) return DIMSUM_ERR_NULL;
    if (p->mod_scale_ptr && p->rows_per_batch <= 0) return DIMSUM_ERR_SHAPE;
    // scaled-fp16 image: fp16 rows of N + one inverse scale per row
    if (p->y_split3 == 2) {
        if (!p->y_inv_scale_ptr) return DIMSUM_ERR_NULL;
        if (p->out_dtype != DIMSUM_F16 || p->cols % 4 != 0 || p->y_row_stride % 4 != 0 || p->y_row_stride < p->cols || !dimsum::aligned_to<char>(p->y_ptr, 8))
            return DIMSUM_ERR_STRIDE;
    } else
    // split3 output: bf16 rows of 3 N, written 8 bytes at a time
    if (p->y_split3 && (p->out_dtype != DIMSUM_BF16 || p->cols % 4 != 0 || p->y_row_stride % 4 != 0 || p->y_row_stride < (p->y_split3 == 3 ? 2 : 3) * (int64_t)p->cols ||
                        !dimsum::aligned_to<char>(p->y_ptr, 8)))
        return DIMSUM_ERR_STRIDE;
    if (p->rows == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (p->x_dtype) {
        case DIMSUM_F32: return dispatch_res<float>(*p, s);
        case DIMSUM_F16: return dispatch_res<__half>(*p, s);
        case DIMSUM_BF16: return dispatch_res<__hip_bfloat16>(*p, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}

extern "C" int dimsum_norm_bwd(const dimsum_norm_bwd_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_norm_bwd_params_t)) return DIMSUM_ERR_ABI;
    if (!p->r_ptr || !p->weight_ptr || !p->rstd_ptr || !p->dy_ptr || !p->dx_ptr || !p->dweight_ptr) return DIMSUM_ERR_NULL;
    if (p->rows < 0 || p->cols <= 0) return DIMSUM_ERR_SHAPE;
    if (p->rows == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int N = p->cols;
    bool vec = N % 4 == 0 && aligned_to<float>(p->r_ptr, 16) && aligned_to<float>(p->dy_ptr, 16) && aligned_to<float>(p->dx_ptr, 16) &&
               aligned_to<float>(p->weight_ptr, 16) && p->r_row_stride % 4 == 0 && p->dy_row_stride % 4 == 0 && p->dx_row_stride % 4 == 0;
    if (p->dres_ptr) vec = vec && aligned_to<float>(p->dres_ptr, 16) && p->dres_row_stride % 4 == 0;
    const int pieces = (N + 255) / 256;
    // few, fat workgroups: every workgroup ends with N atomics, so cap the grid at ~2 waves per SIMD
    const int64_t blocks = (p->rows + 3) / 4;
    // measured with two rows per trip (tools/scratch/norm_bwd_time.py), 65536 x 1024: 256 workgroups 382 us, 384: 303, 512: 266 (one round at 2 per CU),
    // 768: 336, 1024: 301; 16384 x 1024 (a training step at 64 latents; the N atomics per workgroup weigh more): 256: 87, 384: 83, 512: 93, 768: 112
    const int64_t cap = p->rows >= 32768 ? 512 : 384;
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap)), block(256);
#define DIMSUM_NB(K) hipLaunchKernelGGL((norm_bwd_kernel<K>), grid, block, 0, s, *p, vec)
    if (pieces <= 1) DIMSUM_NB(1);
    else if (pieces <= 2) DIMSUM_NB(2);
    else if (pieces <= 4) DIMSUM_NB(4);
    else if (pieces <= 5) DIMSUM_NB(5);
    else if (pieces <= kMaxPieces) DIMSUM_NB(8);
    else return DIMSUM_ERR_SHAPE;
#undef DIMSUM_NB
    return launch_status();
}
