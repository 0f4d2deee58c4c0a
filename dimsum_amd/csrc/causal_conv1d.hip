// causal_conv1d.hip -- depthwise causal conv1d (width 2..4) + bias + SiLU, forward and backward, for gfx950.
//
// Replaces causal_conv1d_cuda.causal_conv1d_fwd[_cond] / _bwd[_cond] (causal-conv1d/csrc/causal_conv1d.cpp:221-509;
// kernels causal_conv1d_fwd.cu:39-130, causal_conv1d_bwd.cu:46-240):
//     out[b,d,t] = act(bias[d] + sum_k W[d,k] * x[b,d,t-(width-1-k)]),   x[.., t<0] = 0,   act = SiLU or identity
// Pure streaming: 2*B*D*L*s bytes forward, 3*B*D*L*s backward.
//
// MI355X design: one wave64 walks (b,d) rows, four at a time with their loads issued back to back, 256 elements per step
// and row (16 B per lane, fully coalesced 1-KiB requests);
// the 3-element halo comes from the neighbouring lane (one cross-lane move per value) and, across 256-element steps,
// from registers -- no LDS, no block barrier (the reference stages a 128-thread block through shared memory and idles
// half of it at L = 256). Weights and bias are wave-uniform (scalar loads). The backward walks the row from the end
// so the gradient halo is carried the same way, keeps dweight/dbias partial sums in registers across all the batch
// rows a wave owns and issues ONE atomic per wave per tap (the reference: one per block per row).
#include "common.hpp"

namespace dimsum {

__device__ __forceinline__ float lane_up(float v, float carry_for_lane0) {   // value of lane-1, lane 0 gets the carry
    const float t = __shfl_up(v, 1, kWave);
    return (threadIdx.x & (kWave - 1)) == 0 ? carry_for_lane0 : t;
}
__device__ __forceinline__ float lane_down(float v, float carry_for_lane63) {
    const float t = __shfl_down(v, 1, kWave);
    return (threadIdx.x & (kWave - 1)) == kWave - 1 ? carry_for_lane63 : t;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float silu_grad(float pre) {   // d/dpre [pre * sigmoid(pre)]   (causal_conv1d_bwd.cu:153-164)
    const float sg = sigmoidf_fast(pre);
    return sg * (1.0f + pre * (1.0f - sg));
}

template <typename T, bool kVec>
__device__ __forceinline__ f32x4 load_row4(const T *row, int t, int L) {
    if constexpr (kVec) {
        if (t < L) return widen(ld4<T>(row + t));
        return {{0.f, 0.f, 0.f, 0.f}};
    } else {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r.v[e] = (t + e < L) ? to_f32<T>(row[t + e]) : 0.f;
        return r;
    }
}
template <typename T, bool kVec>
__device__ __forceinline__ void store_row4(T *row, int t, int L, const f32x4 &v) {
    if constexpr (kVec) {
        if (t < L) st4<T>(row + t, v);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t + e < L) row[t + e] = from_f32<T>(v.v[e]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool kVec>
__global__ __launch_bounds__(256) void causal_conv1d_fwd_kernel(const dimsum_conv_params_t p) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t rows = (int64_t)p.batch * p.dim;
    const int L = p.seqlen, W = p.width;
    // kR rows per wave and iteration, a grid stride apart: their loads are issued back to back (kR x 1 KiB in flight per wave)
    constexpr int kR = 4;
    const int64_t gstride = (int64_t)gridDim.x * 4;
    for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < rows; row0 += kR * gstride) {
        const T *x[kR];
        T *o[kR];
        float w4[kR][4], bias[kR], c1[kR], c2[kR], c3[kR];   // taps right-aligned into 4 slots: w4[3] multiplies x[t], w4[2] x[t-1], ...
        bool ok[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int64_t row = min(row0 + r * gstride, rows - 1);
            ok[r] = row0 + r * gstride < rows;
            const int b = (int)(row / p.dim), d = (int)(row - (int64_t)b * p.dim);
            x[r] = reinterpret_cast<const T *>(p.x_ptr) + (int64_t)b * p.x_batch_stride + (int64_t)d * p.x_c_stride;
            o[r] = reinterpret_cast<T *>(p.out_ptr) + (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_c_stride;
            const float *wp = reinterpret_cast<const float *>(p.weight_ptr) + (int64_t)d * p.weight_c_stride;
#pragma unroll
            for (int k = 0; k < 4; ++k) w4[r][k] = (k >= 4 - W) ? wp[(k - (4 - W)) * p.weight_width_stride] : 0.f;
            bias[r] = p.bias_ptr ? reinterpret_cast<const float *>(p.bias_ptr)[d] : 0.f;
            c1[r] = c2[r] = c3[r] = 0.f;                     // x[t0-1], x[t0-2], x[t0-3] carried across 256-element steps
        }
        for (int t0 = 0; t0 < L; t0 += 4 * kWave) {
            const int t = t0 + lane * 4;
            f32x4 v[kR];
#pragma unroll
            for (int r = 0; r < kR; ++r) v[r] = load_row4<T, kVec>(x[r], t, L);
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const float m1 = lane_up(v[r].v[3], c1[r]), m2 = lane_up(v[r].v[2], c2[r]), m3 = lane_up(v[r].v[1], c3[r]);
                f32x4 q;
                q.v[0] = bias[r] + w4[r][0] * m3 + w4[r][1] * m2 + w4[r][2] * m1 + w4[r][3] * v[r].v[0];
                q.v[1] = bias[r] + w4[r][0] * m2 + w4[r][1] * m1 + w4[r][2] * v[r].v[0] + w4[r][3] * v[r].v[1];
                q.v[2] = bias[r] + w4[r][0] * m1 + w4[r][1] * v[r].v[0] + w4[r][2] * v[r].v[1] + w4[r][3] * v[r].v[2];
                q.v[3] = bias[r] + w4[r][0] * v[r].v[0] + w4[r][1] * v[r].v[1] + w4[r][2] * v[r].v[2] + w4[r][3] * v[r].v[3];
                if (p.silu_activation) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) q.v[e] *= sigmoidf_fast(q.v[e]);   // out / (1 + exp(-out)), causal_conv1d_fwd.cu:113-118
                }
                if (ok[r]) store_row4<T, kVec>(o[r], t, L, q);
                c1[r] = __shfl(v[r].v[3], kWave - 1, kWave);
                c2[r] = __shfl(v[r].v[2], kWave - 1, kWave);
                c3[r] = __shfl(v[r].v[1], kWave - 1, kWave);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward. grid = (dim, n_split): wave (d, s) owns batch rows b = s, s + n_split*4, ... of channel d.
template <typename T, bool kVec>
__global__ __launch_bounds__(256) void causal_conv1d_bwd_kernel(const dimsum_conv_bwd_params_t q) {
    const dimsum_conv_params_t &p = q.fwd;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = blockIdx.x;
    const int L = p.seqlen, W = p.width;
    const float *wp = reinterpret_cast<const float *>(p.weight_ptr) + (int64_t)d * p.weight_c_stride;
    float w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = (k >= 4 - W) ? wp[(k - (4 - W)) * p.weight_width_stride] : 0.f;
    const float bias = p.bias_ptr ? reinterpret_cast<const float *>(p.bias_ptr)[d] : 0.f;
    float dw[4] = {0.f, 0.f, 0.f, 0.f}, db = 0.f;
    const int n_steps = (L + 4 * kWave - 1) / (4 * kWave);

    for (int b = blockIdx.y * 4 + wave; b < p.batch; b += gridDim.y * 4) {
        const T *x = reinterpret_cast<const T *>(p.x_ptr) + (int64_t)b * p.x_batch_stride + (int64_t)d * p.x_c_stride;
        const T *go = reinterpret_cast<const T *>(q.dout_ptr) + (int64_t)b * q.dout_batch_stride + (int64_t)d * q.dout_c_stride;
        T *dx = reinterpret_cast<T *>(q.dx_ptr) + (int64_t)b * q.dx_batch_stride + (int64_t)d * q.dx_c_stride;
        float g1 = 0.f, g2 = 0.f, g3 = 0.f;   // g[t0+256], g[t0+257], g[t0+258] of the step processed before (later in time)
        for (int step = n_steps - 1; step >= 0; --step) {
            const int t0 = step * 4 * kWave, t = t0 + lane * 4;
            const f32x4 v = load_row4<T, kVec>(x, t, L);
            f32x4 g = load_row4<T, kVec>(go, t, L);
            // x halo from the left: neighbour lane, or straight from memory for lane 0 (3 scalar loads per 256 elements)
            float c1 = 0.f, c2 = 0.f, c3 = 0.f;
            if (t0 > 0) { c1 = to_f32<T>(x[t0 - 1]); c2 = to_f32<T>(x[t0 - 2]); c3 = to_f32<T>(x[t0 - 3]); }
            const float m1 = lane_up(v.v[3], c1), m2 = lane_up(v.v[2], c2), m3 = lane_up(v.v[1], c3);
            const float xs[7] = {m3, m2, m1, v.v[0], v.v[1], v.v[2], v.v[3]};   // x[t-3] .. x[t+3]
            if (p.silu_activation) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pre = bias + w4[0] * xs[e] + w4[1] * xs[e + 1] + w4[2] * xs[e + 2] + w4[3] * xs[e + 3];
                    g.v[e] *= silu_grad(pre);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                db += g.v[e];
#pragma unroll
                for (int k = 0; k < 4; ++k) dw[k] = fmaf(g.v[e], xs[e + k], dw[k]);
            }
            // dx[s] = sum_k w4[k] * g[s + 3 - k]  -> needs g[t+4], g[t+5], g[t+6] from the right neighbour
            const float p1 = lane_down(g.v[0], g1), p2 = lane_down(g.v[1], g2), p3 = lane_down(g.v[2], g3);
            const float gs[7] = {g.v[0], g.v[1], g.v[2], g.v[3], p1, p2, p3};   // g[t] .. g[t+6]
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r.v[e] = w4[3] * gs[e] + w4[2] * gs[e + 1] + w4[1] * gs[e + 2] + w4[0] * gs[e + 3];
            store_row4<T, kVec>(dx, t, L, r);
            g1 = __shfl(g.v[0], 0, kWave);
            g2 = __shfl(g.v[1], 0, kWave);
            g3 = __shfl(g.v[2], 0, kWave);
        }
    }
    // one atomic per wave per tap (fp32 accumulators zero-filled by the caller, causal_conv1d.cpp:405-407)
    float *dwp = reinterpret_cast<float *>(q.dweight_ptr) + (int64_t)d * q.dweight_c_stride;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float s = wave_sum(dw[k]);
        if (lane == 0 && k >= 4 - W) atomicAdd(dwp + (k - (4 - W)) * q.dweight_width_stride, s);
    }
    if (q.dbias_ptr) {
        const float s = wave_sum(db);
        if (lane == 0) atomicAdd(reinterpret_cast<float *>(q.dbias_ptr) + d, s);
    }
}

static int conv_check(const dimsum_conv_params_t &p) {
    if (!p.x_ptr || !p.weight_ptr) return DIMSUM_ERR_NULL;
    if (p.width < 2 || p.width > 4) return DIMSUM_ERR_SHAPE;   // causal_conv1d.cpp:248
    if (p.batch < 0 || p.dim <= 0 || p.seqlen <= 0) return DIMSUM_ERR_SHAPE;
    if (p.dtype < DIMSUM_F32 || p.dtype > DIMSUM_BF16) return DIMSUM_ERR_DTYPE;
    return DIMSUM_OK;
}
template <typename T> static bool vec_ok(const void *ptr, int64_t s0, int64_t s1, int L) {
    return L % 4 == 0 && aligned_to<T>(ptr, 4 * sizeof(T)) && s0 % 4 == 0 && s1 % 4 == 0;
}

template <typename T> static int launch_conv_fwd(const dimsum_conv_params_t &p, hipStream_t s) {
    const bool vec = vec_ok<T>(p.x_ptr, p.x_batch_stride, p.x_c_stride, p.seqlen) &&
                     vec_ok<T>(p.out_ptr, p.out_batch_stride, p.out_c_stride, p.seqlen);
    const int64_t rows = (int64_t)p.batch * p.dim;
    const int grid = (int)((rows + 15) / 16 < 256 * 32 ? (rows + 15) / 16 : 256 * 32);   // 4 waves x 4 rows per workgroup and pass (a 2x larger grid measured slower)
    if (vec) hipLaunchKernelGGL((causal_conv1d_fwd_kernel<T, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((causal_conv1d_fwd_kernel<T, false>), dim3(grid), dim3(256), 0, s, p);
    return launch_status();
}
template <typename T> static int launch_conv_bwd(const dimsum_conv_bwd_params_t &q, hipStream_t s) {
    const dimsum_conv_params_t &p = q.fwd;
    const bool vec = vec_ok<T>(p.x_ptr, p.x_batch_stride, p.x_c_stride, p.seqlen) &&
                     vec_ok<T>(q.dout_ptr, q.dout_batch_stride, q.dout_c_stride, p.seqlen) &&
                     vec_ok<T>(q.dx_ptr, q.dx_batch_stride, q.dx_c_stride, p.seqlen);
    // enough waves to fill the chip while keeping >= 4 batch rows per wave when the batch allows it
    int split = 1;
    while ((int64_t)p.dim * split < 2048 && split * 4 < p.batch) split *= 2;
    if (vec) hipLaunchKernelGGL((causal_conv1d_bwd_kernel<T, true>), dim3(p.dim, split), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((causal_conv1d_bwd_kernel<T, false>), dim3(p.dim, split), dim3(256), 0, s, q);
    return launch_status();
}

}  // namespace dimsum

extern "C" int dimsum_causal_conv1d_fwd(const dimsum_conv_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_conv_params_t)) return DIMSUM_ERR_ABI;
    if (!p->out_ptr) return DIMSUM_ERR_NULL;
    const int rc = conv_check(*p);
    if (rc != DIMSUM_OK) return rc;
    if (p->batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (p->dtype) {
        case DIMSUM_F32: return launch_conv_fwd<float>(*p, s);
        case DIMSUM_F16: return launch_conv_fwd<__half>(*p, s);
        default: return launch_conv_fwd<__hip_bfloat16>(*p, s);
    }
}

extern "C" int dimsum_causal_conv1d_bwd(const dimsum_conv_bwd_params_t *q, void *stream) {
    using namespace dimsum;
    if (!q) return DIMSUM_ERR_NULL;
    if (q->struct_size != sizeof(dimsum_conv_bwd_params_t)) return DIMSUM_ERR_ABI;
    if (!q->dout_ptr || !q->dx_ptr || !q->dweight_ptr) return DIMSUM_ERR_NULL;
    const int rc = conv_check(q->fwd);
    if (rc != DIMSUM_OK) return rc;
    if (q->fwd.batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (q->fwd.dtype) {
        case DIMSUM_F32: return launch_conv_bwd<float>(*q, s);
        case DIMSUM_F16: return launch_conv_bwd<__half>(*q, s);
        default: return launch_conv_bwd<__hip_bfloat16>(*q, s);
    }
}
