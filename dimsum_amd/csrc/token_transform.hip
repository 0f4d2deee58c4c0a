// token_transform.hip -- fused token-space transforms of the DiM blocks for gfx950 (see include/dimsum_hip.h).
//
// One pass over (batch, L = grid*grid tokens, C channels) fp32:
//     v[s, c]  = x[b, in_index[s], c] * gate[b, c]
//     t        = T(v) on every 4x4 token block  (identity | 2-level Haar | its inverse | 4x4 DCT-II | its inverse)
//     y[b, out_index[s], c] = t[s, c] * (1 + scale[b, c]) + shift[b, c] + residual[b, out_index[s], c]
// It replaces, per mixer call, the reference's chain of ~10 full-tensor passes: einops rearranges, flips, local_scan,
// 8 grouped stride-2 convs + cats of the DWT, modulate, and the residual add (dimsum/models_dim.py:572-604, 656-705,
// 876-928, 1496-1524; dimsum/wavelet_layer.py; dimsum/dct_layer.py). Streaming: 2 (3 with residual) tensors of traffic.
//
// Layout: a workgroup owns one 4x4 token block of one batch element; a thread owns 4 adjacent channels (16 B) of all
// 16 tokens -- every global access is a coalesced 16-B-per-lane row segment; the transform is pure register
// butterflies (Haar: 64 add/sub per channel; DCT: separable 2 x 16 x 4 FMA).
//
// Haar detail (WaveDiMBlock._dwt_fast): after the two DWT levels the reference concatenates the 16 subbands
// band-major, shuffles them (i -> i%4*4 + i/4) and then regroups the flat (band, channel) axis as (channel', p1, p2):
// output token p of the block, channel c' holds band q of input channel c with  q*C + c = c'*16 + p.  That regrouping
// mixes channels across threads, so the coefficients go through an LDS image indexed by the flat j = q*C + c, stored
// as [c'/4][p][c'%4] with 68-dword rows: 16-B reads by the storing thread (4 channels of one token) are conflict free.
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"

namespace dimsum {

constexpr int kTTThreads = 256;
__device__ __forceinline__ int haar_lds(int j) {      // flat j = c'*16 + p  ->  LDS dword index
    const int cp = j >> 4, p = j & 15;
    return (cp >> 2) * 68 + p * 4 + (cp & 3);
}

// 2x2 Haar butterfly (wavelet_layer.py:8-22): (a b / c d) -> ll, lh (rows high-pass), hl, hh; each * 1/2
__device__ __forceinline__ void haar2(float a, float b, float c, float d, float &ll, float &lh, float &hl, float &hh) {
    const float p = a + b, q = a - b, r = c + d, s = c - d;
    ll = (p + r) * 0.5f; lh = (p - r) * 0.5f; hl = (q + s) * 0.5f; hh = (q - s) * 0.5f;
}
// inverse (wavelet_layer.py:41-54)
__device__ __forceinline__ void ihaar2(float ll, float lh, float hl, float hh, float &a, float &b, float &c, float &d) {
    const float p = ll + lh, q = ll - lh, r = hl + hh, s = hl - hh;
    a = (p + r) * 0.5f; b = (p - r) * 0.5f; c = (q + s) * 0.5f; d = (q - s) * 0.5f;
}

// X[y*4+x] (16 tokens of one channel) -> Y[q] with q = k1*4 + k2 (the shuffled list position), incl. the 1/4 scale
__device__ __forceinline__ void haar_fwd16(const float *X, float *Y) {
    float b1[4][4];   // [k1][i*2+j] first-level bands on the 2x2 grid of sub-blocks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            haar2(X[(2 * i) * 4 + 2 * j], X[(2 * i) * 4 + 2 * j + 1], X[(2 * i + 1) * 4 + 2 * j], X[(2 * i + 1) * 4 + 2 * j + 1],
                  b1[0][i * 2 + j], b1[1][i * 2 + j], b1[2][i * 2 + j], b1[3][i * 2 + j]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float o[4];
        haar2(b1[k1][0], b1[k1][1], b1[k1][2], b1[k1][3], o[0], o[1], o[2], o[3]);   // o[k2]
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) Y[k1 * 4 + k2] = o[k2] * 0.25f;
    }
}
__device__ __forceinline__ void haar_inv16(const float *Y, float *X) {   // Y[q = k1*4 + k2] (already * 4)
    float b1[4][4];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
        ihaar2(Y[k1 * 4 + 0], Y[k1 * 4 + 1], Y[k1 * 4 + 2], Y[k1 * 4 + 3], b1[k1][0], b1[k1][1], b1[k1][2], b1[k1][3]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            ihaar2(b1[0][i * 2 + j], b1[1][i * 2 + j], b1[2][i * 2 + j], b1[3][i * 2 + j], X[(2 * i) * 4 + 2 * j],
                   X[(2 * i) * 4 + 2 * j + 1], X[(2 * i + 1) * 4 + 2 * j], X[(2 * i + 1) * 4 + 2 * j + 1]);
}

// 4-point DCT-II basis m[v][y] = c_v * cos((2y+1) v pi / 8) * sqrt(2/4)   (dct_layer.py:21-29: (2 C_v C_u / 4) cos cos)
__device__ __forceinline__ void dct4(const float *in, int stride, float *out, int ostride, bool inverse) {
    constexpr float kA = 0.5f;                    // sqrt(2/4) * (1/sqrt 2)
    constexpr float kC1 = 0.6532814824381883f;    // sqrt(2/4) * cos(pi/8)
    constexpr float kC2 = 0.5f;                   // sqrt(2/4) * cos(2 pi/8)
    constexpr float kC3 = 0.2705980500730985f;    // sqrt(2/4) * cos(3 pi/8)
    const float x0 = in[0], x1 = in[stride], x2 = in[2 * stride], x3 = in[3 * stride];
    if (!inverse) {
        const float s03 = x0 + x3, d03 = x0 - x3, s12 = x1 + x2, d12 = x1 - x2;
        out[0] = kA * (s03 + s12);
        out[ostride] = kC1 * d03 + kC3 * d12;
        out[2 * ostride] = kC2 * (s03 - s12);
        out[3 * ostride] = kC3 * d03 - kC1 * d12;
    } else {      // transpose (orthonormal basis)
        const float e = kA * x0 + kC2 * x2, f = kA * x0 - kC2 * x2, g = kC1 * x1 + kC3 * x3, h = kC3 * x1 - kC1 * x3;
        out[0] = e + g;
        out[ostride] = f + h;
        out[2 * ostride] = f - h;
        out[3 * ostride] = e - g;
    }
}
__device__ __forceinline__ void dct16(const float *X, float *Y, bool inverse) {   // X[y*4+x] <-> Y[v*4+u]
    float t[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) dct4(X + r * 4, 1, t + r * 4, 1, inverse);       // along x (u)
#pragma unroll
    for (int c = 0; c < 4; ++c) dct4(t + c, 4, Y + c, 4, inverse);               // along y (v)
}

// F16S: y is the scaled-fp16 operand image of the Linear that consumes it (common.hpp, f16s): a token's row of C channels is spread over
// the workgroup's threads, so the 16 tokens' outputs wait in registers for their exact row maxima (wave butterflies + 4 x 16 floats of
// LDS); C <= 4 * kTTThreads (one channel group per thread).
// kFix (blocked kinds with the image output: the two passes around the frequency branch's mixer): which optional operands exist is a COMPILE-TIME
// fact -- 1 = scale + shift only (the pre-mixer pass), 2 = gate + residual only (the post-mixer pass), 0 = decided at run time. A run-time
// `if (ptr)` around a vector load is a uniform branch per token and per operand: ~2800 scalar instructions in the unrolled 16-token body, and
// no load moves across a branch, so every token's loads waited for the previous token's (Haar forward + image 94 us = 2.1 TB/s at 256 latents).
// With the presence known and the 16 token indices fetched up front the body is straight-line code.
template <int KIND, int VEC, bool F16S = false, int kFix = 0>
__global__ __launch_bounds__(kTTThreads) void token_transform_kernel(const dimsum_tt_params_t p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int nthreads = (F16S || KIND != DIMSUM_TT_NONE) ? (int)blockDim.x : kTTThreads;   // blocked kinds / images: the block is sized to the channel count
    const int b = blockIdx.y;
    const int C = p.channels, G = p.grid;
    // 16 positions s of this workgroup: a 4x4 block of the grid (or 16 consecutive tokens when KIND == NONE)
    int s_of[16];
    if (KIND == DIMSUM_TT_NONE) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s_of[k] = blockIdx.x * 16 + k;
    } else {
        const int gb = G / 4, bh = blockIdx.x / gb, bw = blockIdx.x - bh * gb;
#pragma unroll
        for (int k = 0; k < 16; ++k) s_of[k] = (bh * 4 + (k >> 2)) * G + bw * 4 + (k & 3);
    }
    // KIND NONE with reductions: a workgroup walks several 16-token groups (grid-stride over the groups) and flushes its per-(batch, channel)
    // partial sums ONCE -- the atomics were most of such a pass (64 x 256 x 512 with three sums: 56 us against 18 for the plain pass)
    int grp = blockIdx.x;
    const int ngrp = KIND == DIMSUM_TT_NONE ? (p.tokens + 15) / 16 : 1;
    auto pos_of = [&](int k) { return KIND == DIMSUM_TT_NONE ? grp * 16 + k : s_of[k]; };   // NONE: k may be dynamic
    // blocked kinds: the source / destination token of the 16 positions, fetched once (ONE branch per table instead of one per token and use)
    int tin[KIND == DIMSUM_TT_NONE ? 1 : 16], tout[KIND == DIMSUM_TT_NONE ? 1 : 16];
    if constexpr (KIND != DIMSUM_TT_NONE) {
        if (p.in_index_ptr) {
#pragma unroll
            for (int k = 0; k < 16; ++k) tin[k] = p.in_index_ptr[s_of[k]];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) tin[k] = s_of[k];
        }
        if (p.out_index_ptr) {
#pragma unroll
            for (int k = 0; k < 16; ++k) tout[k] = p.out_index_ptr[s_of[k]];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) tout[k] = s_of[k];
        }
    }
    float stash[F16S ? 16 : 1][VEC], tmax[F16S ? 16 : 1];      // F16S: the outputs of this thread's channel group, the running row maxima
    if constexpr (F16S) {
#pragma unroll
        for (int k = 0; k < 16; ++k) tmax[k] = 0.f;
    }
    const float *xb = reinterpret_cast<const float *>(p.x_ptr) + (int64_t)b * p.x_batch_stride;
    float *yb = p.y_ptr ? reinterpret_cast<float *>(p.y_ptr) + (int64_t)b * p.y_batch_stride : nullptr;
    const float *rb = p.residual_ptr ? reinterpret_cast<const float *>(p.residual_ptr) + (int64_t)b * p.res_batch_stride : nullptr;
    const float *gate = p.gate_ptr ? reinterpret_cast<const float *>(p.gate_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;
    const float *scale = p.scale_ptr ? reinterpret_cast<const float *>(p.scale_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;
    const float *shift = p.shift_ptr ? reinterpret_cast<const float *>(p.shift_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;

    const float *wb = p.w_ptr ? reinterpret_cast<const float *>(p.w_ptr) + (int64_t)b * p.w_batch_stride : nullptr;
    // kFix:            0 run time | 1 scale + shift | 2 gate + residual | 3 scale + w | 4 w, no y | 5 scale + w + token sums
    //                  (1, 2: the forward passes around a mixer and the pre-mixer's adjoint; 3, 4, 5: the training backward's passes with reductions)
    const bool has_gate = kFix == 0 ? gate != nullptr : kFix == 2;
    const bool has_mod = kFix == 0 ? scale != nullptr : (kFix == 1 || kFix == 3 || kFix == 5);
    const bool has_shift = kFix == 0 ? shift != nullptr : kFix == 1;
    const bool has_res = kFix == 0 ? rb != nullptr : kFix == 2;
    const bool has_w = kFix == 0 ? wb != nullptr : kFix >= 3;
    const bool has_tsum = kFix == 0 ? p.tsum_ptr != nullptr : kFix == 5;
    const bool has_y = kFix == 0 ? yb != nullptr : kFix != 4;
    float wdot[VEC], wsum[VEC], tsum[VEC];       // this thread's partial reductions for the channel group it is storing
#pragma unroll
    for (int e = 0; e < VEC; ++e) { wdot[e] = 0.f; wsum[e] = 0.f; tsum[e] = 0.f; }
    auto flush_red = [&](int c) {     // one atomic per channel per workgroup; resets the partials
        if (wb || p.tsum_ptr) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (p.wdot_ptr) atomicAdd(reinterpret_cast<float *>(p.wdot_ptr) + (int64_t)b * p.red_batch_stride + c + e, wdot[e]);
                if (p.wsum_ptr) atomicAdd(reinterpret_cast<float *>(p.wsum_ptr) + (int64_t)b * p.red_batch_stride + c + e, wsum[e]);
                if (p.tsum_ptr) atomicAdd(reinterpret_cast<float *>(p.tsum_ptr) + (int64_t)b * p.red_batch_stride + c + e, tsum[e]);
                wdot[e] = 0.f; wsum[e] = 0.f; tsum[e] = 0.f;
            }
        }
    };

    auto load_vec = [&](const float *ptr, float *dst) {
        if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4 *>(ptr); dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w; }
        else dst[0] = *ptr;
    };
    auto store_out = [&](int k, int c, const float *val) {       // position k of the block, channels c..c+VEC-1
#pragma clang fp contract(off)      // (the image of y must be the image of the fp32 pass's y: with the operands' presence known at compile time the
                                    //  multiply by (1 + scale), the shift and the residual add sit in one basic block and would fuse into FMAs)
        int tok;
        if constexpr (KIND == DIMSUM_TT_NONE) {
            if (pos_of(k) >= p.tokens) return;
            tok = p.out_index_ptr ? p.out_index_ptr[pos_of(k)] : pos_of(k);
        } else {
            tok = tout[k];                                        // (blocked kinds: the grid covers whole 4 x 4 blocks)
        }
        float o[VEC], t[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = val[e];
        if (has_w) { load_vec(wb + (int64_t)tok * p.w_token_stride + c, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { wdot[e] = fmaf(o[e], t[e], wdot[e]); wsum[e] += t[e]; } }
        if (has_tsum) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) tsum[e] += o[e]; }
        if (!has_y) return;
        if (has_mod) { load_vec(scale + c, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] *= 1.0f + t[e]; }
        if (has_shift) { load_vec(shift + c, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] += t[e]; }
        if (has_res) { load_vec(rb + (int64_t)tok * p.res_token_stride + c, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] += t[e]; }
        if constexpr (F16S) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { stash[k][e] = o[e]; tmax[k] = fmaxf(tmax[k], fabsf(o[e])); }
            return;
        }
        if constexpr (VEC == 4) {
            if (p.y_split3) {       // split-bf16 operand image of the Linear that consumes y: rows of 3 C bf16, strides in bf16 elements
                st_split_left(reinterpret_cast<unsigned short *>(p.y_ptr) + (int64_t)b * p.y_batch_stride + (int64_t)tok * p.y_token_stride, c, C,
                              f32x4{{o[0], o[1], o[2], o[3]}}, p.y_split3 == 3);
                return;
            }
        }
        float *dst = yb + (int64_t)tok * p.y_token_stride + c;
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else dst[0] = o[0];
    };
    auto load_in = [&](int k, int c, float *dst) {               // gated input of position k
#pragma clang fp contract(off)
        int tok;
        if constexpr (KIND == DIMSUM_TT_NONE) {
            if (pos_of(k) >= p.tokens) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) dst[e] = 0.f;
                return;
            }
            tok = p.in_index_ptr ? p.in_index_ptr[pos_of(k)] : pos_of(k);
        } else {
            tok = tin[k];
        }
        load_vec(xb + (int64_t)tok * p.x_token_stride + c, dst);
        if (has_gate) { float t[VEC]; load_vec(gate + c, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[e] *= t[e]; }
    };

    if constexpr (KIND == DIMSUM_TT_NONE) {
        // No transform ties the 16 positions together: when the channel groups do not fill the workgroup (the half-width
        // x1 / x2 slices: 128 groups for 256 threads), split the 16 tokens over the idle threads as well -- every thread
        // active, 8 (or 4) tokens with all their loads in flight per thread.
        const int cg = (C + VEC - 1) / VEC;
        const int nsub = cg <= kTTThreads / 4 ? 4 : (cg <= kTTThreads / 2 ? 2 : 1);
        if (!F16S && nsub > 1 && C % VEC == 0) {
            const int sub = tid / cg, c = (tid - sub * cg) * VEC;
            if (sub < nsub) {
                auto pass = [&](auto perc) {
                    constexpr int PER = decltype(perc)::value;
                    for (grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
                        float v[PER][VEC];
#pragma unroll
                        for (int k = 0; k < PER; ++k) load_in(sub * PER + k, c, v[k]);
#pragma unroll
                        for (int k = 0; k < PER; ++k) store_out(sub * PER + k, c, v[k]);
                    }
                    flush_red(c);
                };
                if (nsub == 2) pass(std::integral_constant<int, 8>{});
                else pass(std::integral_constant<int, 4>{});
            }
            return;
        }
    }

    if constexpr (KIND == DIMSUM_TT_HAAR_INV) {
        // phase 1: the (token p, channel c') image goes to LDS at flat j = c'*16 + p
        for (int c = tid * VEC; c < C; c += nthreads * VEC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float v[VEC];
                load_in(k, c, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) lds[haar_lds((c + e) * 16 + k)] = v[e] * 4.0f;
            }
        }
        __syncthreads();
    }

    for (int c = tid * VEC; c < C; c += nthreads * VEC) {
        float X[VEC][16], Y[VEC][16];
        if constexpr (KIND == DIMSUM_TT_HAAR_INV) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int e = 0; e < VEC; ++e) Y[e][q] = lds[haar_lds(q * C + c + e)];
#pragma unroll
            for (int e = 0; e < VEC; ++e) haar_inv16(Y[e], X[e]);
#pragma unroll
            for (int k = 0; k < 16; ++k) { float o[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = X[e][k];
                store_out(k, c, o); }
            flush_red(c);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) { float v[VEC]; load_in(k, c, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) X[e][k] = v[e]; }
            if constexpr (KIND == DIMSUM_TT_HAAR_FWD) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) haar_fwd16(X[e], Y[e]);
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) lds[haar_lds(q * C + c + e)] = Y[e][q];
            } else {
                if constexpr (KIND == DIMSUM_TT_DCT_FWD || KIND == DIMSUM_TT_DCT_INV) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dct16(X[e], Y[e], KIND == DIMSUM_TT_DCT_INV);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) { float o[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] = (KIND == DIMSUM_TT_NONE) ? X[e][k] : Y[e][k];
                    store_out(k, c, o); }
                if constexpr (KIND == DIMSUM_TT_NONE && !F16S) {        // the further token groups of this workgroup (X holds group blockIdx.x)
                    for (grp = blockIdx.x + gridDim.x; grp < ngrp; grp += gridDim.x) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) { float v[VEC]; load_in(k, c, v);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) X[e][k] = v[e]; }
#pragma unroll
                        for (int k = 0; k < 16; ++k) { float o[VEC];
#pragma unroll
                            for (int e = 0; e < VEC; ++e) o[e] = X[e][k];
                            store_out(k, c, o); }
                    }
                    grp = blockIdx.x;
                }
                flush_red(c);
            }
        }
    }

    if constexpr (KIND == DIMSUM_TT_HAAR_FWD) {
        __syncthreads();
        // phase 2: output token p, channels c'..c'+VEC-1 <- flat j = c'*16 + p  (16-byte LDS reads when VEC == 4)
        for (int c = tid * VEC; c < C; c += nthreads * VEC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float o[VEC];
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(&lds[(c >> 2) * 68 + k * 4]);
                    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
                } else {
                    o[0] = lds[haar_lds(c * 16 + k)];
                }
                store_out(k, c, o);
            }
            flush_red(c);
        }
    }

    if constexpr (F16S && VEC == 4) {
        // exact row maxima of the 16 tokens: butterflies inside each wave, then the 4 waves through LDS (behind the Haar image)
        float *red = lds + p.y_f16s_lds_offset;
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int k = 0; k < 16; ++k) tmax[k] = wave_allmax(tmax[k]);
        if (lane < 16) {
            float m = tmax[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) m = lane == k ? tmax[k] : m;
            red[wave * 16 + lane] = m;
        }
        __syncthreads();
        const int c = tid * VEC;
        __half *yh = reinterpret_cast<__half *>(p.y_ptr) + (int64_t)b * p.y_batch_stride;
        float *inv_out = reinterpret_cast<float *>(p.y_inv_scale_ptr) + (int64_t)b * p.tokens;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == DIMSUM_TT_NONE && pos_of(k) >= p.tokens) continue;
            float m = red[k];
            for (int w = 1; w < (nthreads >> 6); ++w) m = fmaxf(m, red[w * 16 + k]);
            float sc, inv;
            f16s_scales(m, sc, inv);
            int tok;
            if constexpr (KIND == DIMSUM_TT_NONE) tok = p.out_index_ptr ? p.out_index_ptr[pos_of(k)] : pos_of(k);
            else tok = tout[k];
            if (c < C) *reinterpret_cast<uint2 *>(yh + (int64_t)tok * p.y_token_stride + c) = f16s_pack4(f32x4{{stash[k][0], stash[k][1], stash[k][2], stash[k][3]}}, sc);
            if (tid == k) inv_out[tok] = inv;
        }
    }
}

// kind NONE at inference (no reductions): no transform ties tokens together, so ONE WAVE owns a token row (like the norm kernel): whole-row
// 16-byte loads, and for the scaled-fp16 image the row waits in registers for its exact maximum (6 VALU steps, no LDS, no barrier).
// kOut: 0 = fp32 y, 1 = split-bf16 image [hi | hi | lo], 2 = scaled-fp16 image + inverse scale. 5.5 TB/s where the 16-tokens-per-workgroup
// form reaches 4.2-4.8 (tools/bench_token.py).
template <int kPieces, int kOut>
__global__ __launch_bounds__(256) void token_rows_kernel(const dimsum_tt_params_t p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.channels, T = p.tokens;
    const int64_t rows = (int64_t)p.batch * T;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / T), s = (int)(row - (int64_t)b * T);
        const int src = p.in_index_ptr ? p.in_index_ptr[s] : s, dst = p.out_index_ptr ? p.out_index_ptr[s] : s;
        const float *x = reinterpret_cast<const float *>(p.x_ptr) + (int64_t)b * p.x_batch_stride + (int64_t)src * p.x_token_stride;
        const float *res = p.residual_ptr ? reinterpret_cast<const float *>(p.residual_ptr) + (int64_t)b * p.res_batch_stride + (int64_t)dst * p.res_token_stride : nullptr;
        const float *gate = p.gate_ptr ? reinterpret_cast<const float *>(p.gate_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;
        const float *scale = p.scale_ptr ? reinterpret_cast<const float *>(p.scale_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;
        const float *shift = p.shift_ptr ? reinterpret_cast<const float *>(p.shift_ptr) + (int64_t)b * p.mod_batch_stride : nullptr;
        f32x4 v[kPieces];
        float m = 0.f;
        // one branch per optional OPERAND, not per piece (no load moves across a branch: DESIGN 3.3); the order of the operations per element is
        // the old one -- gate, 1 + scale, shift, residual -- and they stay separate instructions (separate basic blocks: no FMA contraction)
        auto ld = [&](const float *ptr, f32x4 (&q)[kPieces]) {
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * 64 + lane) * 4;
                const float4 t = c < C ? *reinterpret_cast<const float4 *>(ptr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                q[i] = f32x4{{t.x, t.y, t.z, t.w}};
            }
        };
        f32x4 q[kPieces];
        ld(x, v);
        if (gate) { ld(gate, q);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i].v[e] *= q[i].v[e]; }
        if (scale) { ld(scale, q);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i].v[e] *= 1.0f + q[i].v[e]; }
        if (shift) { ld(shift, q);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i].v[e] += q[i].v[e]; }
        if (res) { ld(res, q);
#pragma unroll
            for (int i = 0; i < kPieces; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i].v[e] += q[i].v[e]; }
#pragma unroll
        for (int i = 0; i < kPieces; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[i].v[e]));
        if constexpr (kOut == 2) {
            float sc, inv;
            f16s_scales(wave_allmax(m), sc, inv);
            __half *y = reinterpret_cast<__half *>(p.y_ptr) + (int64_t)b * p.y_batch_stride + (int64_t)dst * p.y_token_stride;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < C) *reinterpret_cast<uint2 *>(y + c) = f16s_pack4(v[i], sc);
            }
            if (lane == 0) reinterpret_cast<float *>(p.y_inv_scale_ptr)[(int64_t)b * T + dst] = inv;
        } else if constexpr (kOut == 1) {
            unsigned short *y = reinterpret_cast<unsigned short *>(p.y_ptr) + (int64_t)b * p.y_batch_stride + (int64_t)dst * p.y_token_stride;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < C) st_split_left(y, c, C, v[i], p.y_split3 == 3);
            }
        } else {
            float *y = reinterpret_cast<float *>(p.y_ptr) + (int64_t)b * p.y_batch_stride + (int64_t)dst * p.y_token_stride;
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < C) *reinterpret_cast<float4 *>(y + c) = make_float4(v[i].v[0], v[i].v[1], v[i].v[2], v[i].v[3]);
            }
        }
    }
}

template <int VEC>
static int launch_tt(const dimsum_tt_params_t &p, hipStream_t s) {
    const bool blocked = p.kind != DIMSUM_TT_NONE;
    const int nblk = blocked ? (p.grid / 4) * (p.grid / 4) : (p.tokens + 15) / 16;
    // a thread owns 4 channels of all 16 tokens of a 4 x 4 block: with 512 channels (a branch of DiM-L/2) a 256-thread block would idle half
    // its threads at 2 blocks per CU (230-240 VGPRs): the blocked kinds get a block of the channel-group count instead
    const int cgs = (p.channels + VEC - 1) / VEC;
    // KIND NONE with reductions: fewer, longer workgroups (see the kernel) as long as ~2 per CU remain
    int gx = nblk;
    if (!blocked && (p.w_ptr || p.tsum_ptr) && p.y_split3 != 2) {
        // measured (tools/scratch/tt_red_time.py; us at 64 / 256 latents x 256 tokens x 512 channels): sums only 38 / 123 with one group per
        // workgroup, 28 / 63 down to 512 workgroups; y + wdot 33 / 122 -> 33 / 108 down to 1024 but 43 / 157 at 512 (the stores want the waves)
        const int min_wgs = p.y_ptr ? 1024 : 512;
        while (gx % 2 == 0 && (int64_t)(gx / 2) * p.batch >= min_wgs) gx /= 2;
    }
    const dim3 grid(gx, p.batch), block(blocked ? (cgs <= 64 ? 64 : (cgs <= 128 ? 128 : kTTThreads)) : kTTThreads);
    size_t lds = (p.kind == DIMSUM_TT_HAAR_FWD || p.kind == DIMSUM_TT_HAAR_INV) ? (size_t)((p.channels + 3) / 4) * 68 * 4 : 0;
    if (lds > 160 * 1024) return DIMSUM_ERR_SHAPE;
    // kind NONE without reductions (every inference token pass outside the frequency branch): one wave per token row
    if (p.kind == DIMSUM_TT_NONE && p.y_ptr && !p.w_ptr && !p.tsum_ptr && p.channels <= 2048) {
        if constexpr (VEC == 4) {
            const int64_t rows = (int64_t)p.batch * p.tokens;
            int64_t blocks = (rows + 3) / 4;
            if (blocks > 256 * 16) blocks = 256 * 16;
            const dim3 g1((unsigned)blocks), b1(256);
#define DIMSUM_TTR(O)                                                                                    \
            if (p.channels <= 512) hipLaunchKernelGGL((token_rows_kernel<2, O>), g1, b1, 0, s, p);          \
            else if (p.channels <= 1024) hipLaunchKernelGGL((token_rows_kernel<4, O>), g1, b1, 0, s, p);    \
            else hipLaunchKernelGGL((token_rows_kernel<8, O>), g1, b1, 0, s, p)
            if (p.y_split3 == 2) { DIMSUM_TTR(2); }
            else if (p.y_split3) { DIMSUM_TTR(1); }
            else { DIMSUM_TTR(0); }
#undef DIMSUM_TTR
            return launch_status();
        }
        if (p.y_split3 == 2) return DIMSUM_ERR_STRIDE;
    }
    if (p.y_split3 == 2) {
        if constexpr (VEC == 4) {
            dimsum_tt_params_t q = p;
            q.y_f16s_lds_offset = (int32_t)(lds / 4);
            lds += 64 * 4;
            const int cg = (p.channels + 3) / 4;
            const dim3 blockh(cg <= 64 ? 64 : (cg <= 128 ? 128 : kTTThreads));
            // the two passes around the frequency branch's mixer as the model issues them: operand presence fixed at compile time (kFix, see the kernel)
            static const bool fix = !(getenv("DIMSUM_TT_FIX") && atoi(getenv("DIMSUM_TT_FIX")) == 0);          // (A / B: 0 = the run-time checks)
            const bool pre = fix && p.scale_ptr && p.shift_ptr && !p.gate_ptr && !p.residual_ptr, post = fix && p.gate_ptr && p.residual_ptr && !p.scale_ptr && !p.shift_ptr;
#define DIMSUM_TTH(K) do { if (blocked && pre) hipLaunchKernelGGL((token_transform_kernel<K, 4, true, (K) == DIMSUM_TT_NONE ? 0 : 1>), grid, blockh, lds, s, q);          \
                           else if (blocked && post) hipLaunchKernelGGL((token_transform_kernel<K, 4, true, (K) == DIMSUM_TT_NONE ? 0 : 2>), grid, blockh, lds, s, q);    \
                           else hipLaunchKernelGGL((token_transform_kernel<K, 4, true, 0>), grid, blockh, lds, s, q); } while (0)
            switch (p.kind) {
                case DIMSUM_TT_NONE: DIMSUM_TTH(DIMSUM_TT_NONE); break;
                case DIMSUM_TT_HAAR_FWD: DIMSUM_TTH(DIMSUM_TT_HAAR_FWD); break;
                case DIMSUM_TT_HAAR_INV: DIMSUM_TTH(DIMSUM_TT_HAAR_INV); break;
                case DIMSUM_TT_DCT_FWD: DIMSUM_TTH(DIMSUM_TT_DCT_FWD); break;
                case DIMSUM_TT_DCT_INV: DIMSUM_TTH(DIMSUM_TT_DCT_INV); break;
                default: return DIMSUM_ERR_SHAPE;
            }
#undef DIMSUM_TTH
            return launch_status();
        }
        return DIMSUM_ERR_STRIDE;
    }
    // (the same compile-time operand presence for the fp32 / split-bf16 outputs of the blocked kinds: the training forward's two passes and the
    // pre-mixer's adjoint with the tail gradient as its residual)
    if constexpr (VEC == 4) {
        static const bool fix = !(getenv("DIMSUM_TT_FIX") && atoi(getenv("DIMSUM_TT_FIX")) == 0);
        const bool sc = p.scale_ptr != nullptr, sh = p.shift_ptr != nullptr, ga = p.gate_ptr != nullptr, re = p.residual_ptr != nullptr;
        const bool w = p.w_ptr != nullptr, ts = p.tsum_ptr != nullptr, y = p.y_ptr != nullptr;
        int kf = 0;
        if (fix && !w && !ts && y) kf = (blocked && sc && sh && !ga && !re) ? 1 : (blocked && ga && re && !sc && !sh) ? 2 : 0;
        else if (fix && w && !sh && !ga && !re) kf = (sc && !ts && y) ? 3 : (!sc && !ts && !y) ? 4 : (sc && ts && y) ? 5 : 0;
        if (kf) {
#define DIMSUM_TTF(K) do { if (kf == 1) hipLaunchKernelGGL((token_transform_kernel<K, 4, false, 1>), grid, block, lds, s, p);      \
                           else if (kf == 2) hipLaunchKernelGGL((token_transform_kernel<K, 4, false, 2>), grid, block, lds, s, p); \
                           else if (kf == 3) hipLaunchKernelGGL((token_transform_kernel<K, 4, false, 3>), grid, block, lds, s, p); \
                           else if (kf == 4) hipLaunchKernelGGL((token_transform_kernel<K, 4, false, 4>), grid, block, lds, s, p); \
                           else hipLaunchKernelGGL((token_transform_kernel<K, 4, false, 5>), grid, block, lds, s, p); } while (0)
            switch (p.kind) {
                case DIMSUM_TT_NONE: DIMSUM_TTF(DIMSUM_TT_NONE); break;
                case DIMSUM_TT_HAAR_FWD: DIMSUM_TTF(DIMSUM_TT_HAAR_FWD); break;
                case DIMSUM_TT_HAAR_INV: DIMSUM_TTF(DIMSUM_TT_HAAR_INV); break;
                case DIMSUM_TT_DCT_FWD: DIMSUM_TTF(DIMSUM_TT_DCT_FWD); break;
                case DIMSUM_TT_DCT_INV: DIMSUM_TTF(DIMSUM_TT_DCT_INV); break;
                default: return DIMSUM_ERR_SHAPE;
            }
#undef DIMSUM_TTF
            return launch_status();
        }
    }
#define DIMSUM_TT(K) hipLaunchKernelGGL((token_transform_kernel<K, VEC>), grid, block, lds, s, p)
    switch (p.kind) {
        case DIMSUM_TT_NONE: DIMSUM_TT(DIMSUM_TT_NONE); break;
        case DIMSUM_TT_HAAR_FWD: DIMSUM_TT(DIMSUM_TT_HAAR_FWD); break;
        case DIMSUM_TT_HAAR_INV: DIMSUM_TT(DIMSUM_TT_HAAR_INV); break;
        case DIMSUM_TT_DCT_FWD: DIMSUM_TT(DIMSUM_TT_DCT_FWD); break;
        case DIMSUM_TT_DCT_INV: DIMSUM_TT(DIMSUM_TT_DCT_INV); break;
        default: return DIMSUM_ERR_SHAPE;
    }
#undef DIMSUM_TT
    return launch_status();
}

// ---- GatedMLP epilogue (mlp.py:66-70): h = gelu_tanh(x1) * x2 --------------------------------------------------------
__device__ __forceinline__ float tanh_fast(float x) {           // tanh(x) = 1 - 2 / (exp(2x) + 1)
    return 1.0f - 2.0f * fast_rcp(fast_exp(2.0f * x) + 1.0f);
}
__device__ __forceinline__ float gelu_tanh(float a) {
    const float u = 0.7978845608028654f * (a + 0.044715f * a * a * a);
    return 0.5f * a * (1.0f + tanh_fast(u));
}
__device__ __forceinline__ float gelu_tanh_grad(float a) {
    const float u = 0.7978845608028654f * (a + 0.044715f * a * a * a);
    const float t = tanh_fast(u);
    return 0.5f * (1.0f + t) + 0.5f * a * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * a * a);
}

// backward: one workgroup = a strip of 1024 columns (4 per thread, 16 B) x a chunk of kGGRows rows: bias in registers,
// fully coalesced rows, the column sums of dx12 (= d bias) accumulate in registers: one atomic per column per workgroup.
constexpr int kGGRows = 64;
// forward: one thread = 4 columns of one row, consecutive threads = consecutive 16-byte pieces of h (and of each half of
// x12); 4 independent pieces in flight per thread, a grid stride apart
// kSplit: h is the split-bf16 left operand image of the w3 GEMM (rows of 3 H bf16 [hi | hi | lo], common.hpp) instead of fp32
template <bool kSplit>
__global__ __launch_bounds__(256) void gated_gelu_fwd_kernel(const float *x12, const float *bias, void *hv, int64_t rows, int64_t H) {
    float *h = reinterpret_cast<float *>(hv);
    const int64_t q = H / 4, total = rows * q;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        float4 a[4], g[4];
        int64_t rr[4], cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = min(i0 + k * stride, total - 1);
            rr[k] = i / q; cc[k] = (i - rr[k] * q) * 4;
            a[k] = *reinterpret_cast<const float4 *>(x12 + rr[k] * 2 * H + cc[k]);
            g[k] = *reinterpret_cast<const float4 *>(x12 + rr[k] * 2 * H + H + cc[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k * stride >= total) break;
            float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bg = ba;
            if (bias) { ba = *reinterpret_cast<const float4 *>(bias + cc[k]); bg = *reinterpret_cast<const float4 *>(bias + H + cc[k]); }
            const float4 av = make_float4(a[k].x + ba.x, a[k].y + ba.y, a[k].z + ba.z, a[k].w + ba.w);
            const float4 gv = make_float4(g[k].x + bg.x, g[k].y + bg.y, g[k].z + bg.z, g[k].w + bg.w);
            const f32x4 o = {{gelu_tanh(av.x) * gv.x, gelu_tanh(av.y) * gv.y, gelu_tanh(av.z) * gv.z, gelu_tanh(av.w) * gv.w}};
            if constexpr (kSplit) st_split3<true>(reinterpret_cast<unsigned short *>(hv) + rr[k] * 3 * H, cc[k], H, o);
            else *reinterpret_cast<float4 *>(h + rr[k] * H + cc[k]) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
        }
    }
}
// kSplit: dx12 is written as the split-bf16 operand image of the two GEMMs that consume it (d input = dx12 W12, d weight = dx12^T h):
// rows of 3 x 2H bf16 in WEIGHT order [hi | lo | hi] (common.hpp), to be paired with left-order images of W12^T and of h
template <int kSplit>        // 0: fp32 dx12, 1: the weight-order image [hi | lo | hi], 2: the pair [hi | lo]
__global__ __launch_bounds__(256) void gated_gelu_bwd_kernel(const float *x12, const float *bias, const float *dh, void *dx12v, float *dbias,
                                                             int64_t rows, int64_t H) {
    float *dx12 = reinterpret_cast<float *>(dx12v);
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= H) return;
    float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bg = ba, sa = ba, sg = ba;
    if (bias) { ba = *reinterpret_cast<const float4 *>(bias + c); bg = *reinterpret_cast<const float4 *>(bias + H + c); }
    const int64_t r0 = (int64_t)blockIdx.y * kGGRows, r1 = min(rows, r0 + kGGRows);
    for (int64_t r = r0; r < r1; ++r) {
        float4 a = *reinterpret_cast<const float4 *>(x12 + r * 2 * H + c);
        float4 g = *reinterpret_cast<const float4 *>(x12 + r * 2 * H + H + c);
        const float4 d = *reinterpret_cast<const float4 *>(dh + r * H + c);
        a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w;
        g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
        const float4 da = make_float4(d.x * g.x * gelu_tanh_grad(a.x), d.y * g.y * gelu_tanh_grad(a.y), d.z * g.z * gelu_tanh_grad(a.z), d.w * g.w * gelu_tanh_grad(a.w));
        const float4 dg = make_float4(d.x * gelu_tanh(a.x), d.y * gelu_tanh(a.y), d.z * gelu_tanh(a.z), d.w * gelu_tanh(a.w));
        if constexpr (kSplit == 2) {
            unsigned short *row = reinterpret_cast<unsigned short *>(dx12v) + r * 4 * H;
            st_split_left(row, c, 2 * H, f32x4{{da.x, da.y, da.z, da.w}}, true);
            st_split_left(row, H + c, 2 * H, f32x4{{dg.x, dg.y, dg.z, dg.w}}, true);
        } else if constexpr (kSplit == 1) {
            unsigned short *row = reinterpret_cast<unsigned short *>(dx12v) + r * 6 * H;
            st_split3<false>(row, c, 2 * H, f32x4{{da.x, da.y, da.z, da.w}});
            st_split3<false>(row, H + c, 2 * H, f32x4{{dg.x, dg.y, dg.z, dg.w}});
        } else {
            *reinterpret_cast<float4 *>(dx12 + r * 2 * H + c) = da;
            *reinterpret_cast<float4 *>(dx12 + r * 2 * H + H + c) = dg;
        }
        sa.x += da.x; sa.y += da.y; sa.z += da.z; sa.w += da.w;
        sg.x += dg.x; sg.y += dg.y; sg.z += dg.z; sg.w += dg.w;
    }
    if (dbias) {
        atomicAdd(dbias + c, sa.x); atomicAdd(dbias + c + 1, sa.y); atomicAdd(dbias + c + 2, sa.z); atomicAdd(dbias + c + 3, sa.w);
        atomicAdd(dbias + H + c, sg.x); atomicAdd(dbias + H + c + 1, sg.y); atomicAdd(dbias + H + c + 2, sg.z); atomicAdd(dbias + H + c + 3, sg.w);
    }
}

// The same adjoint with dx12 written as a scaled-fp16 operand image (common.hpp, f16s): rows of 2H fp16 = fp16(dx12_r 2^s_r) with the exact row
// maximum's power of two, inv[r] = 2^-s_r -- the operand of BOTH backward GEMMs of w12 under the scaled-fp16 policy (d input = dx12 W12 as an NT
// product, d weight = dx12^T h as a TN product with per-reduction-row factors, dimsum_gemm_ext_t.k_scale_ptr). A row's maximum needs the whole
// row: one workgroup walks rows_per_wg rows (rows / 512: one round of 512 workgroups), a thread holding its 4-column pieces of both halves (H <= 1024 kStrips) in registers between the
// maximum and the store; the column sums (d bias) accumulate in registers across the rows.
template <int kStrips>
__global__ __launch_bounds__(256) void gated_gelu_bwd_f16s_kernel(const float *x12, const float *bias, const float *dh, __half *img, float *inv, float *dbias,
                                                                  int64_t rows, int64_t H, int rows_per_wg) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 ba[kStrips], bg[kStrips], sa[kStrips], sg[kStrips];
#pragma unroll
    for (int s = 0; s < kStrips; ++s) {
        const int64_t c = ((int64_t)s * 256 + threadIdx.x) * 4;
        ba[s] = bg[s] = sa[s] = sg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && c < H) { ba[s] = *reinterpret_cast<const float4 *>(bias + c); bg[s] = *reinterpret_cast<const float4 *>(bias + H + c); }
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    for (int64_t r = r0; r < r1; ++r) {
        float4 da[kStrips], dg[kStrips];
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < kStrips; ++s) {
            const int64_t c = ((int64_t)s * 256 + threadIdx.x) * 4;
            da[s] = dg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < H) {
                float4 a = *reinterpret_cast<const float4 *>(x12 + r * 2 * H + c);
                float4 g = *reinterpret_cast<const float4 *>(x12 + r * 2 * H + H + c);
                const float4 d = *reinterpret_cast<const float4 *>(dh + r * H + c);
                a.x += ba[s].x; a.y += ba[s].y; a.z += ba[s].z; a.w += ba[s].w;
                g.x += bg[s].x; g.y += bg[s].y; g.z += bg[s].z; g.w += bg[s].w;
                da[s] = make_float4(d.x * g.x * gelu_tanh_grad(a.x), d.y * g.y * gelu_tanh_grad(a.y), d.z * g.z * gelu_tanh_grad(a.z), d.w * g.w * gelu_tanh_grad(a.w));
                dg[s] = make_float4(d.x * gelu_tanh(a.x), d.y * gelu_tanh(a.y), d.z * gelu_tanh(a.z), d.w * gelu_tanh(a.w));
                m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(da[s].x), fabsf(da[s].y)), fmaxf(fabsf(da[s].z), fabsf(da[s].w))),
                                   fmaxf(fmaxf(fabsf(dg[s].x), fabsf(dg[s].y)), fmaxf(fabsf(dg[s].z), fabsf(dg[s].w)))));
                sa[s].x += da[s].x; sa[s].y += da[s].y; sa[s].z += da[s].z; sa[s].w += da[s].w;
                sg[s].x += dg[s].x; sg[s].y += dg[s].y; sg[s].z += dg[s].z; sg[s].w += dg[s].w;
            }
        }
        m = wave_allmax(m);
        const int par = (int)(r & 1);               // two slots: the next row's maxima are written while slow waves still read this row's
        if (lane == 0) red[par][w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[par][0], red[par][1]), fmaxf(red[par][2], red[par][3]));
        float scale, iv;
        f16s_scales(m, scale, iv);
        if (threadIdx.x == 0) inv[r] = iv;
        __half *row = img + r * 2 * H;
#pragma unroll
        for (int s = 0; s < kStrips; ++s) {
            const int64_t c = ((int64_t)s * 256 + threadIdx.x) * 4;
            if (c < H) {
                *reinterpret_cast<uint2 *>(row + c) = f16s_pack4(f32x4{{da[s].x, da[s].y, da[s].z, da[s].w}}, scale);
                *reinterpret_cast<uint2 *>(row + H + c) = f16s_pack4(f32x4{{dg[s].x, dg[s].y, dg[s].z, dg[s].w}}, scale);
            }
        }
    }
    if (dbias) {
#pragma unroll
        for (int s = 0; s < kStrips; ++s) {
            const int64_t c = ((int64_t)s * 256 + threadIdx.x) * 4;
            if (c < H) {
                atomicAdd(dbias + c, sa[s].x); atomicAdd(dbias + c + 1, sa[s].y); atomicAdd(dbias + c + 2, sa[s].z); atomicAdd(dbias + c + 3, sa[s].w);
                atomicAdd(dbias + H + c, sg[s].x); atomicAdd(dbias + H + c + 1, sg[s].y); atomicAdd(dbias + H + c + 2, sg[s].z); atomicAdd(dbias + H + c + 3, sg[s].w);
            }
        }
    }
}

}  // namespace dimsum

extern "C" int dimsum_token_transform(const dimsum_tt_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_tt_params_t)) return DIMSUM_ERR_ABI;
    if (!p->x_ptr) return DIMSUM_ERR_NULL;
    if (!p->y_ptr && !p->tsum_ptr && !(p->w_ptr && (p->wdot_ptr || p->wsum_ptr))) return DIMSUM_ERR_NULL;   // nothing to produce
    if ((p->wdot_ptr || p->wsum_ptr) && !p->w_ptr) return DIMSUM_ERR_NULL;
    if (p->batch < 0 || p->tokens <= 0 || p->channels <= 0) return DIMSUM_ERR_SHAPE;
    if (p->kind != DIMSUM_TT_NONE && (p->grid % 4 != 0 || p->grid * p->grid != p->tokens)) return DIMSUM_ERR_SHAPE;
    if (p->batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bool vec = p->channels % 4 == 0 && aligned_to<float>(p->x_ptr, 16) && (!p->y_ptr || aligned_to<float>(p->y_ptr, 16)) &&
               p->x_batch_stride % 4 == 0 && p->x_token_stride % 4 == 0 && p->y_batch_stride % 4 == 0 && p->y_token_stride % 4 == 0 &&
               p->mod_batch_stride % 4 == 0 && (!p->gate_ptr || aligned_to<float>(p->gate_ptr, 16)) &&
               (!p->scale_ptr || aligned_to<float>(p->scale_ptr, 16)) && (!p->shift_ptr || aligned_to<float>(p->shift_ptr, 16));
    if (p->residual_ptr) vec = vec && aligned_to<float>(p->residual_ptr, 16) && p->res_batch_stride % 4 == 0 && p->res_token_stride % 4 == 0;
    if (p->w_ptr) vec = vec && aligned_to<float>(p->w_ptr, 16) && p->w_batch_stride % 4 == 0 && p->w_token_stride % 4 == 0;
    if (p->y_split3 == 2) {       // scaled-fp16 image rows + (batch, tokens) inverse scales; no reductions alongside
        if (!p->y_ptr || !p->y_inv_scale_ptr) return DIMSUM_ERR_NULL;
        if (!vec || p->y_token_stride < p->channels || p->channels > (p->kind == DIMSUM_TT_NONE ? 2048 : 4 * kTTThreads) || p->w_ptr || p->tsum_ptr)
            return DIMSUM_ERR_STRIDE;
    } else
    if (p->y_split3 && (!vec || !p->y_ptr || p->y_token_stride < (p->y_split3 == 3 ? 2 : 3) * (int64_t)p->channels)) return DIMSUM_ERR_STRIDE;   // image rows: 8-byte pieces
    return vec ? launch_tt<4>(*p, s) : launch_tt<1>(*p, s);
}

template <bool kSplit>
static int launch_gated_gelu_fwd(const void *x12, const void *bias, void *h, int64_t rows, int64_t hidden, void *stream) {
    using namespace dimsum;
    if (!x12 || !h) return DIMSUM_ERR_NULL;
    if (rows < 0 || hidden <= 0 || hidden % 4 != 0) return DIMSUM_ERR_SHAPE;
    if (!aligned_to<float>(x12, 16) || !aligned_to<float>(h, 16) || (bias && !aligned_to<float>(bias, 16))) return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
    // flat mapping: 4 pieces per thread a quarter of the tensor apart, no row loop: 0.60 ms at (65536, 2 x 4096) against 0.67 ms
    // for the strip-per-workgroup form the backward keeps (it needs the row loop for the d bias column sums)
    const int64_t total = rows * (hidden / 4);
    const int64_t blocks = (total + 256 * 4 - 1) / (256 * 4);
    if (blocks > 0x7fffffff) return DIMSUM_ERR_SHAPE;
    hipLaunchKernelGGL(gated_gelu_fwd_kernel<kSplit>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float *>(x12), reinterpret_cast<const float *>(bias), h, rows, hidden);
    return launch_status();
}

extern "C" int dimsum_gated_gelu_fwd(const void *x12, const void *bias, void *h, int64_t rows, int64_t hidden, void *stream) {
    return launch_gated_gelu_fwd<false>(x12, bias, h, rows, hidden, stream);
}

extern "C" int dimsum_gated_gelu_fwd_split3(const void *x12, const void *bias, void *h3, int64_t rows, int64_t hidden, void *stream) {
    return launch_gated_gelu_fwd<true>(x12, bias, h3, rows, hidden, stream);
}

template <int kSplit>
static int launch_gated_gelu_bwd(const void *x12, const void *bias, const void *dh, void *dx12, void *dbias, int64_t rows, int64_t hidden, void *stream) {
    using namespace dimsum;
    if (!x12 || !dh || !dx12) return DIMSUM_ERR_NULL;
    if (rows < 0 || hidden <= 0 || hidden % 4 != 0) return DIMSUM_ERR_SHAPE;
    if (!aligned_to<float>(x12, 16) || !aligned_to<float>(dh, 16) || !aligned_to<float>(dx12, 16) || (bias && !aligned_to<float>(bias, 16)))
        return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
    const dim3 grid((unsigned)((hidden / 4 + 255) / 256), (unsigned)((rows + kGGRows - 1) / kGGRows));
    hipLaunchKernelGGL(gated_gelu_bwd_kernel<kSplit>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float *>(x12), reinterpret_cast<const float *>(bias), reinterpret_cast<const float *>(dh),
                       dx12, reinterpret_cast<float *>(dbias), rows, hidden);
    return launch_status();
}

extern "C" int dimsum_gated_gelu_bwd(const void *x12, const void *bias, const void *dh, void *dx12, void *dbias, int64_t rows,
                                     int64_t hidden, void *stream) {
    return launch_gated_gelu_bwd<0>(x12, bias, dh, dx12, dbias, rows, hidden, stream);
}

extern "C" int dimsum_gated_gelu_bwd_split3(const void *x12, const void *bias, const void *dh, void *dx12_image, void *dbias, int64_t rows,
                                            int64_t hidden, void *stream) {
    return launch_gated_gelu_bwd<1>(x12, bias, dh, dx12_image, dbias, rows, hidden, stream);
}

extern "C" int dimsum_gated_gelu_bwd_pair(const void *x12, const void *bias, const void *dh, void *dx12_pair, void *dbias, int64_t rows,
                                          int64_t hidden, void *stream) {
    return launch_gated_gelu_bwd<2>(x12, bias, dh, dx12_pair, dbias, rows, hidden, stream);
}

/* dx12 as the scaled-fp16 image (rows, 2 hidden) float16 + inv_scale (rows) f32: see gated_gelu_bwd_f16s_kernel */
extern "C" int dimsum_gated_gelu_bwd_f16s(const void *x12, const void *bias, const void *dh, void *dx12_image, void *inv_scale, void *dbias, int64_t rows,
                                          int64_t hidden, void *stream) {
    using namespace dimsum;
    if (!x12 || !dh || !dx12_image || !inv_scale) return DIMSUM_ERR_NULL;
    if (rows < 0 || hidden <= 0 || hidden % 4 != 0 || hidden > 5 * 1024) return DIMSUM_ERR_SHAPE;
    if (!aligned_to<char>(x12, 16) || !aligned_to<char>(dh, 16) || !aligned_to<char>(dx12_image, 8) || (bias && !aligned_to<char>(bias, 16))) return DIMSUM_ERR_STRIDE;
    if (rows == 0) return DIMSUM_OK;
    // rows per workgroup: ONE round of 512 workgroups (two per CU; three fit), whatever the batch size. Measured (tools/scratch/gg_bwd_time.py,
    // rows per workgroup = rows / div): 16384 rows: 128 workgroups 538 us, 256: 320, 390-512: 262, 780: 348 (a dozen workgroups left over for a
    // second round run alone for a whole workgroup's duration), 1024: 309, 2048+: 350; 65536 rows: 512 workgroups 862 us, 1024: 937, 2048: 882,
    // 8192: 1117 (the column sums cost one atomic per column and workgroup: 8192 each).
    int rpw = (int)((rows + 511) / 512);
    rpw = rpw < 8 ? 8 : rpw;
    const dim3 grid((unsigned)((rows + rpw - 1) / rpw));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int strips = (int)((hidden + 1023) / 1024);
#define DIMSUM_GGF(K) hipLaunchKernelGGL(gated_gelu_bwd_f16s_kernel<K>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(x12), reinterpret_cast<const float *>(bias), \
                                         reinterpret_cast<const float *>(dh), reinterpret_cast<__half *>(dx12_image), reinterpret_cast<float *>(inv_scale),                \
                                         reinterpret_cast<float *>(dbias), rows, hidden, rpw)
    switch (strips) {
        case 1: DIMSUM_GGF(1); break;
        case 2: DIMSUM_GGF(2); break;
        case 3: DIMSUM_GGF(3); break;
        case 4: DIMSUM_GGF(4); break;
        default: DIMSUM_GGF(5); break;
    }
#undef DIMSUM_GGF
    return launch_status();
}
