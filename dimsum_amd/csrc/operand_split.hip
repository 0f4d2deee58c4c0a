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

}  // namespace dimsum

extern "C" int dimsum_rows_f16s(const void *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride,
                                void *inv_scale, void *l1max, void *stream) {
    using namespace dimsum;
    if (!src || !dst || !inv_scale) return DIMSUM_ERR_NULL;
    if (rows < 0 || cols <= 0 || cols % 4 != 0) return DIMSUM_ERR_SHAPE;
    if (src_row_stride % 4 != 0 || src_row_stride < cols || dst_row_stride % 4 != 0 || dst_row_stride < cols || !aligned_to<float>(src, 16) ||
        !aligned_to<char>(dst, 8))
        return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
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
