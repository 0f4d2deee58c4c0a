// operand_split.hip -- split-bf16 operand images for the library GEMMs (gfx950).
//
// No reference counterpart: the reference's Linears run as TF32 GEMMs (torch.backends.cuda.matmul.allow_tf32 = True,
// dimsum/train.py:20-21, sample_ddp.py:56). gfx950 has no TF32 MFMA; hipBLASLt serves fp32 operands under that flag by splitting
// them into hi + lo bf16 inside the GEMM (3 products, ~370 TFLOP/s-equivalent). Given the SAME three products as one plain bf16
// GEMM over a 3 K reduction -- rows [hi | hi | lo] on the left, [hi | lo | hi] on the weights, fp32 accumulate and output -- the
// library's bf16 kernels reach ~410-430 TFLOP/s-equivalent (tools/scratch/ksplit_probe.py). The left images are written by the
// kernels that produce the activations (norm.hip y_split3, gated GeLU split3, ...); this file converts what has no producer
// of its own: the weights (per call: a cached image could not see parameter updates made through .data) and plain fp32 rows.
#include "common.hpp"

namespace dimsum {

template <bool kLeft>
__global__ __launch_bounds__(256) void split3_kernel(const float *src, int64_t rows, int64_t cols, int64_t src_row_stride, unsigned short *dst) {
    const int64_t q = cols / 4, total = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q, c = (i - r * q) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(src + r * src_row_stride + c);
        st_split3<kLeft>(dst + r * 3 * cols, c, cols, f32x4{{v.x, v.y, v.z, v.w}});
    }
}

// the pair [hi | lo] (rows of 2 cols bf16): what the hand-written GEMM reads as [hi | hi | lo] or [hi | lo | hi] by aliasing K tiles
__global__ __launch_bounds__(256) void split_pair_kernel(const float *src, int64_t rows, int64_t cols, int64_t src_row_stride, unsigned short *dst) {
    const int64_t q = cols / 4, total = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / q, c = (i - r * q) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(src + r * src_row_stride + c);
        st_split_left(dst + r * 2 * cols, c, cols, f32x4{{v.x, v.y, v.z, v.w}}, true);
    }
}

// weight (N, K) fp32 -> the ROW stack [hi; lo; hi] (3 K, N) bf16 of its transpose: the right operand of dimsum_gemm_tn when the left one is
// a d-major activation (out_proj: y = out_z^T W^T with out_z (d_inner, tokens)). One thread per (k, 2 n): the matrix is small and
// L2-resident, the transposing reads cost nothing next to a launch.
__global__ __launch_bounds__(256) void split3_t_kernel(const float *src, int64_t N, int64_t K, int64_t src_row_stride, unsigned short *dst) {
    const int64_t half = N / 2, total = K * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / half, n = (i - k * half) * 2;
        unsigned hi, lo;
        split2(src[n * src_row_stride + k], src[(n + 1) * src_row_stride + k], hi, lo);
        *reinterpret_cast<unsigned *>(dst + k * N + n) = hi;
        *reinterpret_cast<unsigned *>(dst + (K + k) * N + n) = lo;
        *reinterpret_cast<unsigned *>(dst + (2 * K + k) * N + n) = hi;
    }
}

// scaled-fp16 image of fp32 rows (common.hpp, f16s): one wave per row, exact row maximum. Converts what has no producer kernel of its
// own: the weights (whose largest row L1 norm -- the bound sum_k |w_nk| on |x W^T| / max|x| -- the gated epilogue of the GEMM needs).
template <int kPieces>      // kPieces * 256 >= cols: the row lives in registers (one pass); 0: two passes over a row of any length
__global__ __launch_bounds__(256) void rows_f16s_kernel(const float *src, int64_t rows, int64_t cols, int64_t src_stride, __half *dst, int64_t dst_stride,
                                                       float *inv_scale, float *l1max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float l1top = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float *x = src + row * src_stride;
        float m = 0.f, l1 = 0.f;
        float4 r[kPieces > 0 ? kPieces : 1];
        if constexpr (kPieces > 0) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int64_t c = (int64_t)(i * 64 + lane) * 4;
                r[i] = c < cols ? *reinterpret_cast<const float4 *>(x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(r[i].x), fabsf(r[i].y))), fmaxf(fabsf(r[i].z), fabsf(r[i].w)));
                l1 += (fabsf(r[i].x) + fabsf(r[i].y)) + (fabsf(r[i].z) + fabsf(r[i].w));
            }
        } else {
            for (int64_t c = lane * 4; c < cols; c += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                l1 += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
            }
        }
        m = wave_allmax(m);
        float sc, inv;
        f16s_scales(m, sc, inv);
        if constexpr (kPieces > 0) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int64_t c = (int64_t)(i * 64 + lane) * 4;
                if (c < cols) *reinterpret_cast<uint2 *>(dst + row * dst_stride + c) = f16s_pack4(f32x4{{r[i].x, r[i].y, r[i].z, r[i].w}}, sc);
            }
        } else {
            for (int64_t c = lane * 4; c < cols; c += 256) {         // (the row is re-read from the cache)
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                *reinterpret_cast<uint2 *>(dst + row * dst_stride + c) = f16s_pack4(f32x4{{v.x, v.y, v.z, v.w}}, sc);
            }
        }
        if (l1max) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o, kWave);
            l1top = fmaxf(l1top, l1);
        }
        if (lane == 0) inv_scale[row] = inv;
    }
    if (l1max) {
        // one atomic per workgroup, and only when it can still raise the maximum: thousands of waves hitting ONE L2 line serialise (the
        // 8192 x 1024 w12 weight took 46 us of which ~35 were this queue). Non-negative floats order like their bit patterns.
        __shared__ float top[4];
        if (lane == 0) top[wave] = l1top;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = fmaxf(fmaxf(top[0], top[1]), fmaxf(top[2], top[3]));
            if (t > __builtin_nontemporal_load(l1max)) atomicMax(reinterpret_cast<int *>(l1max), __float_as_int(t));
        }
    }
}

// The same image for LONG rows (a d-major activation: channels x (batch x tokens), 65536 columns at DiM-L/2 batch 256 -- the operand of the Mamba
// projections' training GEMMs whose reduction runs over the tokens): one workgroup per row, 4 independent 16-byte loads per thread in flight
// (a wave per row as above keeps 1 KB in flight per row: a quarter of the chip's rate on 2048 rows), two passes over the row.
__global__ __launch_bounds__(256) void rows_f16s_long_kernel(const float *src, int64_t rows, int64_t cols, int64_t src_stride, __half *dst, int64_t dst_stride,
                                                            float *inv_scale) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *x = src + row * src_stride;
        float m = 0.f;
        int64_t c = (int64_t)threadIdx.x * 4;
        for (; c + 3 * 1024 < cols; c += 4 * 1024) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(x + c + k * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
        }
        for (; c < cols; c += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(x + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        m = wave_allmax(m);
        __syncthreads();                      // (the previous row's maxima have been read)
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sc, inv;
        f16s_scales(m, sc, inv);
        if (threadIdx.x == 0) inv_scale[row] = inv;
        __half *d = dst + row * dst_stride;
        c = (int64_t)threadIdx.x * 4;
        for (; c + 3 * 1024 < cols; c += 4 * 1024) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(x + c + k * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2 *>(d + c + k * 1024) = f16s_pack4(f32x4{{v[k].x, v[k].y, v[k].z, v[k].w}}, sc);
        }
        for (; c < cols; c += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(x + c);
            *reinterpret_cast<uint2 *>(d + c) = f16s_pack4(f32x4{{v.x, v.y, v.z, v.w}}, sc);
        }
    }
}

// Many conversions in ONE launch (dimsum_rows_f16s_multi): what a denoiser forward under the scaled-fp16 policy needs of its weights --
// the images of every large Linear (7 per DiMBlockCombined), their largest row L1 norms and the bias maxima of the bound-derived scales
// -- used to be ~150 launches of 5-30 us per DiM-L/2 forward (2.2 ms of small grids + ~1 ms of torch reductions between the big kernels);
// the job table travels in the kernel arguments, a workgroup finds its job by its block index. Rows of any length: two passes over a row
// (the second read hits the cache); a job without `dst` only reduces (bias vectors: rows = 1).
constexpr int kMultiJobs = 24;
struct F16sJob {
    const float *src;
    __half *dst;
    float *inv_scale, *l1max, *absmax;
    int64_t rows, cols, src_stride, dst_stride;
    float l1_factor;
    int first_block;            // blocks [first_block, next job's first_block) work on this job
};
struct F16sJobs {
    F16sJob job[kMultiJobs];
    int n;
};
__global__ __launch_bounds__(256) void rows_f16s_multi_kernel(const F16sJobs t) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int j = 0;
#pragma unroll 1
    for (int i = 1; i < t.n; ++i) j = ((int)blockIdx.x >= t.job[i].first_block) ? i : j;
    const F16sJob &q = t.job[j];
    const int nblocks = (j + 1 < t.n ? t.job[j + 1].first_block : (int)gridDim.x) - q.first_block;
    float l1top = 0.f, mtop = 0.f;
    for (int64_t row = (int64_t)((int)blockIdx.x - q.first_block) * 4 + wave; row < q.rows; row += (int64_t)nblocks * 4) {
        const float *x = q.src + row * q.src_stride;
        float m = 0.f, l1 = 0.f;
        for (int64_t c = lane * 4; c < q.cols; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(x + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            l1 += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
        }
        m = wave_allmax(m);
        mtop = fmaxf(mtop, m);
        if (q.dst) {
            float sc, inv;
            f16s_scales(m, sc, inv);
            for (int64_t c = lane * 4; c < q.cols; c += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(x + c);
                *reinterpret_cast<uint2 *>(q.dst + row * q.dst_stride + c) = f16s_pack4(f32x4{{v.x, v.y, v.z, v.w}}, sc);
            }
            if (lane == 0) q.inv_scale[row] = inv;
        }
        if (q.l1max) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o, kWave);
            l1top = fmaxf(l1top, l1);
        }
    }
    if (q.l1max || q.absmax) {          // (wave-uniform per workgroup: every wave of a workgroup works on the same job)
        __shared__ float top[2][4];
        if (lane == 0) { top[0][wave] = l1top; top[1][wave] = mtop; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float a = fmaxf(fmaxf(top[0][0], top[0][1]), fmaxf(top[0][2], top[0][3])) * q.l1_factor;
            const float b = fmaxf(fmaxf(top[1][0], top[1][1]), fmaxf(top[1][2], top[1][3]));
            if (q.l1max && a > __builtin_nontemporal_load(q.l1max)) atomicMax(reinterpret_cast<int *>(q.l1max), __float_as_int(a));
            if (q.absmax && b > __builtin_nontemporal_load(q.absmax)) atomicMax(reinterpret_cast<int *>(q.absmax), __float_as_int(b));
        }
    }
}

}  // namespace dimsum

extern "C" int dimsum_rows_f16s_multi(const dimsum_f16s_job_t *jobs, int32_t n_jobs, void *stream) {
    using namespace dimsum;
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return DIMSUM_ERR_NULL;
    for (int32_t i = 0; i < n_jobs; ++i) {
        const dimsum_f16s_job_t &q = jobs[i];
        if (!q.src || (q.dst && !q.inv_scale_ptr) || (!q.dst && !q.l1max_ptr && !q.absmax_ptr)) return DIMSUM_ERR_NULL;
        if (q.rows <= 0 || q.cols <= 0 || q.cols % 4 != 0) return DIMSUM_ERR_SHAPE;
        if (q.src_row_stride % 4 != 0 || q.src_row_stride < q.cols || !aligned_to<float>(q.src, 16) ||
            (q.dst && (q.dst_row_stride % 4 != 0 || q.dst_row_stride < q.cols || !aligned_to<char>(q.dst, 8))))
            return DIMSUM_ERR_STRIDE;
    }
    for (int32_t base = 0; base < n_jobs; base += kMultiJobs) {
        F16sJobs t{};
        t.n = n_jobs - base < kMultiJobs ? n_jobs - base : kMultiJobs;
        int blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            const dimsum_f16s_job_t &q = jobs[base + i];
            F16sJob &o = t.job[i];
            o.src = reinterpret_cast<const float *>(q.src);
            o.dst = reinterpret_cast<__half *>(q.dst);
            o.inv_scale = reinterpret_cast<float *>(q.inv_scale_ptr);
            o.l1max = reinterpret_cast<float *>(q.l1max_ptr);
            o.absmax = reinterpret_cast<float *>(q.absmax_ptr);
            o.rows = q.rows; o.cols = q.cols; o.src_stride = q.src_row_stride; o.dst_stride = q.dst_row_stride;
            o.l1_factor = q.l1_factor != 0.f ? q.l1_factor : 1.0f;
            o.first_block = blocks;
            // enough workgroups to stream the job at the chip's rate, few enough that the per-workgroup atomics on ONE address stay short
            const int64_t by_rows = (q.rows + 3) / 4, by_bytes = (q.rows * q.cols * 4 + 65535) / 65536;
            int64_t nb = by_rows < by_bytes ? by_rows : by_bytes;
            nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
            blocks += (int)nb;
        }
        hipLaunchKernelGGL(rows_f16s_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t);
        if (launch_status() != DIMSUM_OK) return DIMSUM_ERR_LAUNCH;
    }
    return DIMSUM_OK;
}

extern "C" int dimsum_rows_f16s(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride,
                                void *inv_scale, void *l1max, void *stream) {
    using namespace dimsum;
    if (!src || !dst || !inv_scale) return DIMSUM_ERR_NULL;
    if (rows < 0 || cols <= 0 || cols % 4 != 0) return DIMSUM_ERR_SHAPE;
    if (src_row_stride % 4 != 0 || src_row_stride < cols || dst_row_stride % 4 != 0 || dst_row_stride < cols || !aligned_to<float>(src, 16) ||
        !aligned_to<char>(dst, 8))
        return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
    if (cols > 8192 && !l1max) {           // long rows (d-major activations): one workgroup per row
        const int64_t nb = rows < 8192 ? rows : 8192;
        hipLaunchKernelGGL(rows_f16s_long_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float *>(src),
                           rows, cols, src_row_stride, reinterpret_cast<__half *>(dst), dst_row_stride, reinterpret_cast<float *>(inv_scale));
        return launch_status();
    }
    int64_t blocks = (rows + 3) / 4;
    const int64_t cap = l1max ? 512 : 256 * 8;      // (with the L1 maximum every workgroup ends in an atomic on ONE address: ~35 ns each, serialised)
    if (blocks > cap) blocks = cap;
#define DIMSUM_F16S(P)                                                                                                                             \
    hipLaunchKernelGGL(rows_f16s_kernel<P>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float *>(src), \
                       rows, cols, src_row_stride, reinterpret_cast<__half *>(dst), dst_row_stride, reinterpret_cast<float *>(inv_scale),                  \
                       reinterpret_cast<float *>(l1max))
    if (cols <= 512) DIMSUM_F16S(2);
    else if (cols <= 1024) DIMSUM_F16S(4);
    else if (cols <= 2048) DIMSUM_F16S(8);
    else if (cols <= 4096) DIMSUM_F16S(16);
    else DIMSUM_F16S(0);
#undef DIMSUM_F16S
    return launch_status();
}

extern "C" int dimsum_split3(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int32_t left, void *stream) {
    using namespace dimsum;
    if (!src || !dst) return DIMSUM_ERR_NULL;
    if (rows < 0 || cols <= 0 || cols % 4 != 0) return DIMSUM_ERR_SHAPE;
    if (src_row_stride % 4 != 0 || src_row_stride < cols || !aligned_to<float>(src, 16) || !aligned_to<char>(dst, 8)) return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
    const int64_t total = rows * (cols / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (left == 2) hipLaunchKernelGGL(split_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float *>(src), rows, cols, src_row_stride, reinterpret_cast<unsigned short *>(dst));
    else if (left) hipLaunchKernelGGL(split3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float *>(src), rows, cols, src_row_stride, reinterpret_cast<unsigned short *>(dst));
    else hipLaunchKernelGGL(split3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float *>(src), rows, cols, src_row_stride, reinterpret_cast<unsigned short *>(dst));
    return launch_status();
}

extern "C" int dimsum_split3_t(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, void *stream) {
    using namespace dimsum;
    if (!src || !dst) return DIMSUM_ERR_NULL;
    if (rows <= 0 || cols <= 0 || rows % 2 != 0) return DIMSUM_ERR_SHAPE;
    if (src_row_stride < cols || !aligned_to<char>(dst, 4)) return DIMSUM_ERR_STRIDE;
    const int64_t total = cols * (rows / 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split3_t_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float *>(src), rows, cols,
                       src_row_stride, reinterpret_cast<unsigned short *>(dst));
    return launch_status();
}

// ---- per-reduction-row factors of a weight-gradient product over two row-scaled images (dimsum_gemm_ext_t.k_scale_ptr) --------------------------------
// fac[r] = fp16(a_inv[r] b_inv[r] / top), top = max_r a_inv[r] b_inv[r] (b_inv NULL = 1): ONE small launch (a single workgroup: n <= a few 10^5)
// instead of five torch launches (product, max, divide, cast) in front of every such GEMM -- 160 of them per DiM-L/2 training step.
namespace dimsum {
__global__ __launch_bounds__(1024) void row_factors_kernel(const float *a_inv, const float *b_inv, int64_t n, __half *fac, float *top) {
    __shared__ float red[16];
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, a_inv[i] * (b_inv ? b_inv[i] : 1.0f));
    m = wave_allmax(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    if (threadIdx.x == 0) *top = m;
    const float r = m > 0.f ? 1.0f / m : 0.f;            // (powers of two: exact; an all-zero pair of operands has factors 0)
    for (int64_t i = threadIdx.x; i < n; i += 1024) fac[i] = __float2half_rn(a_inv[i] * (b_inv ? b_inv[i] : 1.0f) * r);
}
}  // namespace dimsum

extern "C" int dimsum_row_factors(const void *a_inv, const void *b_inv, int64_t n, void *k_scale, void *c_scale, void *stream) {
    using namespace dimsum;
    if (!a_inv || !k_scale || !c_scale) return DIMSUM_ERR_NULL;
    if (n <= 0) return DIMSUM_ERR_SHAPE;
    hipLaunchKernelGGL(row_factors_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float *>(a_inv),
                       reinterpret_cast<const float *>(b_inv), n, reinterpret_cast<__half *>(k_scale), reinterpret_cast<float *>(c_scale));
    return launch_status();
}

// ---- block-scaled fp16 image of a d-major fp32 matrix (channels x tokens) ---------------------------------------------------------------------------
// The layout the 64-channel scan kernel writes out_z in (dimsum_ssm_ext_t.out_z_f16): every 64 channels x 32 tokens block as fp16(x 2^s) with the
// block's own power-of-two scale, 2^-s in table[token / 32][channel / 64] -- the A operand of out_proj as ONE fp16 product (dimsum_gemm_tn with
// a_block_inv_ptr). For launches the state-split scan kernels serve (16 channels per wave: no wave sees a whole block), this pass converts their fp32
// out_z: a workgroup = 64 channels x 256 tokens, a wave 16 channel rows (16 x 16 B in flight per lane, 1-KB row segments), block maxima over the 8
// lanes of a token group (3 DPP steps) and the 4 waves (LDS). 6 bytes per element: 0.45 GB in ~90 us at DiM-XL/2 512 px, batch 64.
namespace dimsum {
__global__ __launch_bounds__(256) void rows_block_f16s_kernel(const float *src, int64_t src_stride, __half *dst, int64_t dst_stride, float *table, int64_t table_ld) {
    __shared__ float red[4][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t d0 = (int64_t)blockIdx.y * 64 + wave * 16, m0 = (int64_t)blockIdx.x * 256 + lane * 4;
    float4 v[16];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4 *>(src + (d0 + i) * src_stride + m0);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    mx = fmaxf(mx, dpp_mov<0xB1>(mx));         // the 8 lanes of a 32-token group: quad_perm, quad_perm, row_half_mirror
    mx = fmaxf(mx, dpp_mov<0x4E>(mx));
    mx = fmaxf(mx, dpp_mov<0x141>(mx));
    if ((lane & 7) == 0) red[wave][lane >> 3] = mx;
    __syncthreads();
    const int g = lane >> 3;
    mx = fmaxf(fmaxf(red[0][g], red[1][g]), fmaxf(red[2][g], red[3][g]));
    float sc, inv;
    f16s_scales(mx, sc, inv);
    if (wave == 0 && (lane & 7) == 0) table[((int64_t)blockIdx.x * 8 + g) * table_ld + blockIdx.y] = inv;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<uint2 *>(dst + (d0 + i) * dst_stride + m0) = f16s_pack4(f32x4{{v[i].x, v[i].y, v[i].z, v[i].w}}, sc);
}
}  // namespace dimsum

extern "C" int dimsum_rows_block_f16s(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride, void *table,
                                      int64_t table_ld, void *stream) {
    using namespace dimsum;
    if (!src || !dst || !table) return DIMSUM_ERR_NULL;
    if (rows <= 0 || cols <= 0 || rows % 64 != 0 || cols % 256 != 0 || cols / 256 > 0x7fffffff || rows / 64 > 65535) return DIMSUM_ERR_SHAPE;
    if (src_row_stride % 4 != 0 || src_row_stride < cols || dst_row_stride % 4 != 0 || dst_row_stride < cols || table_ld < rows / 64 || !aligned_to<char>(src, 16) ||
        !aligned_to<char>(dst, 8))
        return DIMSUM_ERR_STRIDE;
    hipLaunchKernelGGL(rows_block_f16s_kernel, dim3((unsigned)(cols / 256), (unsigned)(rows / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float *>(src), src_row_stride, reinterpret_cast<__half *>(dst), dst_row_stride, reinterpret_cast<float *>(table), table_ld);
    return launch_status();
}
