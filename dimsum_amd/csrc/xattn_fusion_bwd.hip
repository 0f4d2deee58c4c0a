// xattn_fusion_bwd.hip -- backward of the cross-attention fusion core on the matrix cores (gfx950), fp32.
//
// Forward (xattn_fusion.hip; dimsum/attention_fusion.py:64-79): per (batch, head, direction), with (q, k, v) =
// (q1, k2, v2) for direction 0 and (q2, k1, v1) for direction 1:  S = scale q k^T,  P = softmax(S),  o = P v.
// Backward, with do = d(out) and the saved log-sum-exp rows:
//     D_i = sum_e do_ie o_ie        dP = do v^T        dS = P o (dP - D)        P = exp(S - lse)
//     dq = scale dS k               dk = scale dS^T q  dv = P^T do
// Two kernels, no atomics, every output element written exactly once (direction 0 owns dq1, dk2, dv2; direction 1 owns
// dq2, dk1, dv1 -- the six slices of dqkv1 / dqkv2):
//   xattn_bwd_dq_kernel   one wave = 16 queries, walks key tiles (the forward's structure): S^T = K Q^T and
//                         dP^T = V dO^T in the TRANSPOSED form so that the C-layout registers of dS^T are directly the B
//                         operand of dQ^T = K^T dS^T; also emits D (needed by the second kernel).
//   xattn_bwd_dkv_kernel  one wave = 16 keys, walks query tiles: S = Q K^T and dP = dO V^T NON-transposed, so that the C
//                         registers of P / dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS.
// S and dP are recomputed in both (7 GEMM-equivalents instead of 5): cheaper on fp32 MFMA than contended fp32 atomics
// on dq, and bitwise reproducible. All products are v_mfma_f32_16x16x4_f32 (exact fp32).
// Optional qkv Linear biases are added while q / k / v are fetched, like in the forward.
#include "xattn_common.hpp"

namespace dimsum {

constexpr int kBKT = 64;         // keys per tile (dq kernel)
constexpr int kBQT = 32;         // queries per tile (dkv kernel)

#define MFMA4(ACC, A4, B4)                                                       \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).x, (B4).x, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).y, (B4).y, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).z, (B4).z, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).w, (B4).w, ACC, 0, 0, 0)

struct XSrc {
    const float *q, *k, *v, *qb, *kb, *vb;     // rows of this (batch, head): token stride ts; biases or nullptr
    float *dq, *dk, *dv;
};
__device__ __forceinline__ XSrc xattn_src(const dimsum_xattn_bwd_params_t &p, int b, int h, int dir, int HD) {
    const int C = p.fwd.heads * HD;
    const bool kv1 = dir == 1 || p.fwd.n_dirs == 1;        // k, v (and dk, dv) live in tensor 1: direction 1, or self-attention
    const int64_t off = (int64_t)b * p.fwd.qkv_batch_stride + h * HD, doff = (int64_t)b * p.dqkv_batch_stride + h * HD;
    const float *qs = reinterpret_cast<const float *>(dir == 0 ? p.fwd.qkv1_ptr : p.fwd.qkv2_ptr) + off;
    const float *kvs = reinterpret_cast<const float *>(kv1 ? p.fwd.qkv1_ptr : p.fwd.qkv2_ptr) + off;
    const float *qbias = reinterpret_cast<const float *>(dir == 0 ? p.fwd.bias1_ptr : p.fwd.bias2_ptr);
    const float *kvbias = reinterpret_cast<const float *>(kv1 ? p.fwd.bias1_ptr : p.fwd.bias2_ptr);
    float *dqs = reinterpret_cast<float *>(dir == 0 ? p.dqkv1_ptr : p.dqkv2_ptr) + doff;
    float *dkvs = reinterpret_cast<float *>(kv1 ? p.dqkv1_ptr : p.dqkv2_ptr) + doff;
    XSrc s;
    s.q = qs; s.k = kvs + C; s.v = kvs + 2 * C;
    s.qb = qbias ? qbias + h * HD : nullptr;
    s.kb = kvbias ? kvbias + C + h * HD : nullptr;
    s.vb = kvbias ? kvbias + 2 * C + h * HD : nullptr;
    s.dq = dqs; s.dk = dkvs + C; s.dv = dkvs + 2 * C;
    return s;
}
__device__ __forceinline__ float4 add4(const float4 &a, const float4 &b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 ld_bias4(const float *row, const float *bias, int e) {
    float4 t = *reinterpret_cast<const float4 *>(row + e);
    if (bias) { const float4 bb = *reinterpret_cast<const float4 *>(bias + e); t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w; }
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// dq (+ D): workgroup = 4 waves = 64 queries of one (batch, head, direction)
// ---------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void xattn_bwd_dq_kernel(const dimsum_xattn_bwd_params_t p) {
    constexpr int KS = HD + 4;               // [key][e] tiles (K and V): 16-B aligned rows, conflict-free b128 reads
    constexpr int TS = kBKT + 4;             // [e][key] tile (K^T)
    constexpr int ET = (HD + 15) / 16;
    constexpr int EC = HD / 16;
    constexpr bool kTail8 = (HD % 16) == 8;
    constexpr int NC = EC + (kTail8 ? 1 : 0);
    __shared__ __attribute__((aligned(16))) float Ks[kBKT * KS];
    __shared__ __attribute__((aligned(16))) float Vs[kBKT * KS];
    __shared__ __attribute__((aligned(16))) float Kt[ET * 16 * TS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.fwd.seqlen, H = p.fwd.heads;
    const int qblocks = (L + 63) / 64;
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, qblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int qblk = idx % qblocks; idx /= qblocks;
    const int ndir = p.fwd.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const XSrc s = xattn_src(p, b, h, dir, HD);
    const int64_t ts = p.fwd.qkv_token_stride, dts = p.dqkv_token_stride;

    const int qi = lane & 15, kg = lane >> 4;
    const int q_tok = qblk * 64 + wave * 16 + qi;
    const int q_ld = min(q_tok, L - 1);
    const float qscale = p.fwd.scale * kLog2e;
    const float *dorow = reinterpret_cast<const float *>(p.dout_ptr) + (int64_t)b * p.fwd.out_batch_stride + (int64_t)q_ld * p.fwd.out_token_stride + dir * C + h * HD;
    const float *orow = reinterpret_cast<const float *>(p.fwd.out_ptr) + (int64_t)b * p.fwd.out_batch_stride + (int64_t)q_ld * p.fwd.out_token_stride + dir * C + h * HD;
    // Q^T (scaled into the log2 domain) and dO^T fragments: chunk c holds e = 16c + 4 kg .. +3 of this lane's query
    f4 qf[NC], dof[NC];
    float dpart = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool tail = kTail8 && c == EC;
        const int e = tail ? 16 * EC + 4 * (kg & 1) : 16 * c + 4 * kg;
        const float m = (tail && kg >= 2) ? 0.f : 1.f;        // 8-wide tail: k-groups 2, 3 idle
        const float4 t = ld_bias4(s.q + (int64_t)q_ld * ts, s.qb, e);
        const float4 g = *reinterpret_cast<const float4 *>(dorow + e), o = *reinterpret_cast<const float4 *>(orow + e);
        qf[c] = f4{t.x * qscale * m, t.y * qscale * m, t.z * qscale * m, t.w * qscale * m};
        dof[c] = f4{g.x * m, g.y * m, g.z * m, g.w * m};
        dpart += m * (g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w);
    }
    dpart = quad_sum(dpart);                                  // D of this lane's query
    const int64_t stat = (((int64_t)b * ndir + dir) * H + h) * L + q_ld;
    const float lse2 = reinterpret_cast<const float *>(p.fwd.lse_ptr)[stat] * kLog2e;
    if (q_tok < L && kg == 0) reinterpret_cast<float *>(p.delta_ptr)[stat] = dpart;

    f4 acc[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) acc[e] = f4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < L; k0 += kBKT) {
        __syncthreads();
        for (int i = tid; i < kBKT * (HD / 4); i += 256) {
            const int key = i / (HD / 4), e4 = i - key * (HD / 4);
            const int tok = min(k0 + key, L - 1);
            const float4 kv = ld_bias4(s.k + (int64_t)tok * ts, s.kb, e4 * 4);
            const float4 vv = ld_bias4(s.v + (int64_t)tok * ts, s.vb, e4 * 4);
            *reinterpret_cast<float4 *>(&Ks[key * KS + e4 * 4]) = kv;
            *reinterpret_cast<float4 *>(&Vs[key * KS + e4 * 4]) = vv;
            Kt[(e4 * 4 + 0) * TS + key] = kv.x; Kt[(e4 * 4 + 1) * TS + key] = kv.y;
            Kt[(e4 * 4 + 2) * TS + key] = kv.z; Kt[(e4 * 4 + 3) * TS + key] = kv.w;
        }
        if constexpr (ET * 16 > HD) {
            for (int i = tid; i < (ET * 16 - HD) * kBKT; i += 256) Kt[(HD + i / kBKT) * TS + (i % kBKT)] = 0.f;
        }
        __syncthreads();

        f4 ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f4 sacc = f4{0.f, 0.f, 0.f, 0.f}, pacc = f4{0.f, 0.f, 0.f, 0.f};
            const float *krow = &Ks[(kt * 16 + qi) * KS], *vrow = &Vs[(kt * 16 + qi) * KS];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int e = (kTail8 && c == EC) ? 16 * EC + 4 * (kg & 1) : 16 * c + 4 * kg;
                const float4 kf = *reinterpret_cast<const float4 *>(krow + e);
                const float4 vf = *reinterpret_cast<const float4 *>(vrow + e);
                MFMA4(sacc, kf, qf[c]);          // S^T  (log2 domain)
                MFMA4(pacc, vf, dof[c]);         // dP^T
            }
            // dS^T = P^T o (dP^T - D); keys beyond L contribute nothing
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = (k0 + kt * 16 + kg * 4 + r < L) ? fast_exp2(sacc[r] - lse2) : 0.f;
                ds[kt][r] = pr * (pacc[r] - dpart);
            }
        }
        // dQ^T += K^T dS^T: K-step (kt, r) covers keys kt*16 + {r, 4+r, 8+r, 12+r}; its B operand is ds[kt][r]
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const float *trow = &Kt[(e * 16 + qi) * TS];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float4 kf = *reinterpret_cast<const float4 *>(trow + kt * 16 + kg * 4);
                MFMA4(acc[e], kf, ds[kt]);
            }
        }
    }
    if (q_tok < L) {
        float *dst = s.dq + (int64_t)q_tok * dts;
        const float sc = p.fwd.scale;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int e0 = e * 16 + kg * 4;
            if (e0 < HD) *reinterpret_cast<float4 *>(dst + e0) = make_float4(acc[e][0] * sc, acc[e][1] * sc, acc[e][2] * sc, acc[e][3] * sc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dk, dv: workgroup = 4 waves = 64 keys of one (batch, head, direction), walks query tiles of 32
// ---------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void xattn_bwd_dkv_kernel(const dimsum_xattn_bwd_params_t p) {
    constexpr int RS = HD + 4;               // [query][e] tiles (Q and dO)
    constexpr int TS = kBQT + 4;             // [e][query] tiles (Q^T and dO^T)
    constexpr int ET = (HD + 15) / 16;
    constexpr int EC = HD / 16;
    constexpr bool kTail8 = (HD % 16) == 8;
    constexpr int NC = EC + (kTail8 ? 1 : 0);
    __shared__ __attribute__((aligned(16))) float Qs[kBQT * RS];
    __shared__ __attribute__((aligned(16))) float Gs[kBQT * RS];
    __shared__ __attribute__((aligned(16))) float Qt[ET * 16 * TS];
    __shared__ __attribute__((aligned(16))) float Gt[ET * 16 * TS];
    __shared__ __attribute__((aligned(16))) float sL[kBQT], sD[kBQT];      // lse (log2 domain) and D of the tile's queries

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.fwd.seqlen, H = p.fwd.heads;
    const int kblocks = (L + 63) / 64;
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, kblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int kblk = idx % kblocks; idx /= kblocks;
    const int ndir = p.fwd.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const XSrc s = xattn_src(p, b, h, dir, HD);
    const int64_t ts = p.fwd.qkv_token_stride, dts = p.dqkv_token_stride;
    const float *dobase = reinterpret_cast<const float *>(p.dout_ptr) + (int64_t)b * p.fwd.out_batch_stride + dir * C + h * HD;
    const int64_t stat0 = (((int64_t)b * ndir + dir) * H + h) * L;
    const float *lse = reinterpret_cast<const float *>(p.fwd.lse_ptr) + stat0;
    const float *dlt = reinterpret_cast<const float *>(p.delta_ptr) + stat0;

    const int ki = lane & 15, kg = lane >> 4;
    const int k_tok = kblk * 64 + wave * 16 + ki;
    const int k_ld = min(k_tok, L - 1);
    const float kscale = p.fwd.scale * kLog2e;
    // K^T (scaled) and V^T fragments of this lane's key: chunk c holds e = 16c + 4 kg .. +3   (B operands)
    f4 kf[NC], vf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool tail = kTail8 && c == EC;
        const int e = tail ? 16 * EC + 4 * (kg & 1) : 16 * c + 4 * kg;
        const float m = (tail && kg >= 2) ? 0.f : 1.f;
        const float4 kk = ld_bias4(s.k + (int64_t)k_ld * ts, s.kb, e);
        const float4 vv = ld_bias4(s.v + (int64_t)k_ld * ts, s.vb, e);
        kf[c] = f4{kk.x * kscale * m, kk.y * kscale * m, kk.z * kscale * m, kk.w * kscale * m};
        vf[c] = f4{vv.x * m, vv.y * m, vv.z * m, vv.w * m};
    }
    f4 dk[ET], dv[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) { dk[e] = f4{0.f, 0.f, 0.f, 0.f}; dv[e] = f4{0.f, 0.f, 0.f, 0.f}; }
    const bool key_live = k_tok < L;

    for (int q0 = 0; q0 < L; q0 += kBQT) {
        __syncthreads();
        for (int i = tid; i < kBQT * (HD / 4); i += 256) {
            const int q = i / (HD / 4), e4 = i - q * (HD / 4);
            const int tok = min(q0 + q, L - 1);
            const float4 qv = ld_bias4(s.q + (int64_t)tok * ts, s.qb, e4 * 4);
            const float4 gv = *reinterpret_cast<const float4 *>(dobase + (int64_t)tok * p.fwd.out_token_stride + e4 * 4);
            *reinterpret_cast<float4 *>(&Qs[q * RS + e4 * 4]) = qv;
            *reinterpret_cast<float4 *>(&Gs[q * RS + e4 * 4]) = gv;
            Qt[(e4 * 4 + 0) * TS + q] = qv.x; Qt[(e4 * 4 + 1) * TS + q] = qv.y; Qt[(e4 * 4 + 2) * TS + q] = qv.z; Qt[(e4 * 4 + 3) * TS + q] = qv.w;
            Gt[(e4 * 4 + 0) * TS + q] = gv.x; Gt[(e4 * 4 + 1) * TS + q] = gv.y; Gt[(e4 * 4 + 2) * TS + q] = gv.z; Gt[(e4 * 4 + 3) * TS + q] = gv.w;
        }
        if constexpr (ET * 16 > HD) {
            for (int i = tid; i < (ET * 16 - HD) * kBQT; i += 256) { Qt[(HD + i / kBQT) * TS + (i % kBQT)] = 0.f; Gt[(HD + i / kBQT) * TS + (i % kBQT)] = 0.f; }
        }
        if (tid < kBQT) {
            const int tok = q0 + tid;
            sL[tid] = tok < L ? lse[tok] * kLog2e : 1e30f;       // queries beyond L: P = exp2(S - inf) = 0
            sD[tid] = tok < L ? dlt[tok] : 0.f;
        }
        __syncthreads();

#pragma unroll
        for (int qt = 0; qt < kBQT / 16; ++qt) {
            // S (queries x keys) = Q K^T, dP = dO V^T: A rows = this tile's query (lane & 15)
            f4 sacc = f4{0.f, 0.f, 0.f, 0.f}, pacc = f4{0.f, 0.f, 0.f, 0.f};
            const float *qrow = &Qs[(qt * 16 + ki) * RS], *grow = &Gs[(qt * 16 + ki) * RS];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int e = (kTail8 && c == EC) ? 16 * EC + 4 * (kg & 1) : 16 * c + 4 * kg;
                const float4 qa = *reinterpret_cast<const float4 *>(qrow + e);
                const float4 ga = *reinterpret_cast<const float4 *>(grow + e);
                MFMA4(sacc, qa, kf[c]);
                MFMA4(pacc, ga, vf[c]);
            }
            // C layout: key = lane & 15 (column), queries qt*16 + kg*4 + r (rows)
            const float4 l4 = *reinterpret_cast<const float4 *>(&sL[qt * 16 + kg * 4]);
            const float4 d4 = *reinterpret_cast<const float4 *>(&sD[qt * 16 + kg * 4]);
            f4 pp, dsv;
            pp[0] = key_live ? fast_exp2(sacc[0] - l4.x) : 0.f; pp[1] = key_live ? fast_exp2(sacc[1] - l4.y) : 0.f;
            pp[2] = key_live ? fast_exp2(sacc[2] - l4.z) : 0.f; pp[3] = key_live ? fast_exp2(sacc[3] - l4.w) : 0.f;
            dsv[0] = pp[0] * (pacc[0] - d4.x); dsv[1] = pp[1] * (pacc[1] - d4.y);
            dsv[2] = pp[2] * (pacc[2] - d4.z); dsv[3] = pp[3] * (pacc[3] - d4.w);
            // dV^T += dO^T P, dK^T += Q^T dS: K-step r covers queries qt*16 + {r, 4+r, 8+r, 12+r} = C register r
#pragma unroll
            for (int e = 0; e < ET; ++e) {
                const float4 ga = *reinterpret_cast<const float4 *>(&Gt[(e * 16 + ki) * TS + qt * 16 + kg * 4]);
                const float4 qa = *reinterpret_cast<const float4 *>(&Qt[(e * 16 + ki) * TS + qt * 16 + kg * 4]);
                MFMA4(dv[e], ga, pp);
                MFMA4(dk[e], qa, dsv);
            }
        }
    }
    if (key_live) {
        float *dkd = s.dk + (int64_t)k_tok * dts, *dvd = s.dv + (int64_t)k_tok * dts;
        const float sc = p.fwd.scale;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int e0 = e * 16 + kg * 4;
            if (e0 < HD) {
                *reinterpret_cast<float4 *>(dkd + e0) = make_float4(dk[e][0] * sc, dk[e][1] * sc, dk[e][2] * sc, dk[e][3] * sc);
                *reinterpret_cast<float4 *>(dvd + e0) = make_float4(dv[e][0], dv[e][1], dv[e][2], dv[e][3]);
            }
        }
    }
}

// =====================================================================================================================
// Matrix-core variants on v_mfma_f32_16x16x32_{bf16,f16}: the same two kernels with every fp32 operand carried
//   fwd.precision = 1 (F16 = false): as hi + lo bf16, three products per element (xattn_common.hpp);
//   fwd.precision = 2 (F16 = true):  as ONE fp16 image, one product per element -- the TF32-equivalent arithmetic of the reference's
//                                    allow_tf32 policy (train.py:20-21 covers the backward matmuls too), the training counterpart of
//                                    xattn_fusion_f16.hip.
// The register-resident C results (dS^T; P and dS) become the B operand of the next product through the slot convention of
// xattn_fusion.hip: K-slot 8g + j of a 32-deep chunk is the row the lane already holds in the C registers of two stacked 16-row tiles,
// and the transposed LDS images (K^T; Q^T, dO^T) are staged with that permutation (cslot), so no operand changes layout. Softmax
// probabilities, D and all accumulators stay fp32.
// fp16 carrier, what keeps the exponent range: q (times scale log2 e), k, v are post-projection activations and go to fp16 as they are
// (|.| <= 65504; the scores only need ABSOLUTE accuracy, 2^-11 per product like TF32). Gradients have no a-priori magnitude: every dO
// row (one query of one head) is scaled by an exact power of two g_q that brings its maximum to [2^-5, 2^-4) (row_pow2_scale), D with it;
// dP, dS of that row live in the scaled domain and dq is unscaled on store (exact). P travels as 2^8 P (kPShift, folded into the
// log-sum-exp the accumulators start from): 11 bits down to P = 2^-22 where fp16 alone would fade below 2^-14. The dq kernel, which owns
// D, also leaves g_q behind the D rows of the scratch (delta: 2 x (batch, n_dirs', heads, L) floats under this carrier). dk / dv sum over
// queries: there P also carries g_min / g_q <= 1 (g_min = the scale of the (batch, head, direction)'s LARGEST gradient row; a power of
// two, so it too lives in the log-sum-exp), every row enters the sums at the scale of the largest one -- rows more than 2^10 below it
// fade, as they do in the fp32 sum -- and the result is multiplied by 2^-8 / g_min on store. Range: |dP - D| <= 2 hd 2^-4 max|v|, times
// 2^8: inside fp16 for max|v| <= 32 at head_dim 64 even if every product aligned.
// =====================================================================================================================
constexpr int kSQW = 128;        // queries per workgroup of the matrix-core dq kernel

// QT = 16-query tiles per wave, NW waves per workgroup (NW * QT * 16 = 128 queries either way). QT = 2 (4 waves): every K / V / K^T
// fragment read serves two query tiles and the two tiles' MFMA chains interleave -- used for head_dim <= 32, where two tiles take 163
// VGPRs (3 waves per SIMD): 0.778 -> 0.717 ms for the backward pair at head_dim 24. At head_dim 48 / 64 two tiles need ~250 VGPRs
// (2 waves per SIMD) and measured 1.33 against 1.24 ms for QT = 1 at 128 VGPRs (8 waves per workgroup, 4 per SIMD): latency hiding by
// occupancy wins there; head_dim 72 does not fit two tiles at all.
// (head_dim 72: 76 KB of LDS allow 2 workgroups per CU anyway -- asking for 4 waves per SIMD capped the kernel at 128 VGPRs and 292 B of
// scratch per lane: 5.2 ms per launch at 1024 tokens, profiles/r03_xattn_bwd_pmc.txt)
template <int HD, int QT, bool F16>
__global__ __launch_bounds__(kSQW / QT * 4, (QT == 1 && HD <= 64) ? 4 : 2) void xattn_bwd_dq_split_kernel(const dimsum_xattn_bwd_params_t p) {
    constexpr int NT = kSQW / QT * 4;        // threads per workgroup: 512 (QT = 1) or 256 (QT = 2)
    constexpr int EP = (HD + 31) / 32 * 32, EC = EP / 32, ET = (HD + 15) / 16;
    constexpr int KS = EP + 8;               // [key][e] rows of K and V (16-bit elements, 16 B of padding)
    constexpr int TS = kBKT + 8;             // [e][key slot] rows of K^T
    constexpr int LO = F16 ? 0 : 1;          // the lo images exist under the split-bf16 carrier only
    __shared__ __attribute__((aligned(16))) unsigned short Kh[kBKT * KS], Vh[kBKT * KS], Th[ET * 16 * TS];
    __shared__ __attribute__((aligned(16))) unsigned short Kl[LO * kBKT * KS + 8], Vl[LO * kBKT * KS + 8], Tl[LO * ET * 16 * TS + 8];
    static_assert((KS / 8) % 2 == 1 && (TS / 8) % 2 == 1, "odd number of 16-byte slots per row");
    // conflict-free fragment reads (xattn_fusion.hip): a ds_read_b128 is serviced in four fixed groups of 16 lanes, in which rows 4-11
    // of a 16-row fragment read k-slot a ^ 1 while rows 0-3 / 12-15 read slot a -- rows 4-11 keep their 16-byte slot PAIRS swapped
    // (8 elements), in the staging writes and in the lanes' base addresses alike
    auto flip = [](int row) { return (((row & 15) + 4) & 8); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.fwd.seqlen, H = p.fwd.heads;
    const int qblocks = (L + kSQW - 1) / kSQW;
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, qblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int qblk = idx % qblocks; idx /= qblocks;
    const int ndir = p.fwd.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const XSrc s = xattn_src(p, b, h, dir, HD);
    const int64_t ts = p.fwd.qkv_token_stride, dts = p.dqkv_token_stride;
    const int64_t nstat = (int64_t)p.fwd.batch * ndir * H * L;            // fp16 carrier: the row scales follow the D rows

    // The K / V rows of the NEXT key tile are requested right after the current tile has been staged (register-staged
    // prefetch: one thread = 2 keys x 4 e, kIt items per tile), so their HBM latency hides under the tile's MFMA work.
    constexpr int kItems = (kBKT / 2) * (HD / 4), kIt = (kItems + NT - 1) / NT;
    // (the loads stay RAW: adding the bias inside fetch would wait for the data there and expose the latency the prefetch is meant to
    // hide -- the thread's bias values are loop constants, added when the tile is staged)
    float4 pka[kIt], pkb[kIt], pva[kIt], pvb[kIt], kbias[kIt], vbias[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = min(tid + it * NT, kItems - 1), e4 = i % (HD / 4);
        kbias[it] = s.kb ? *reinterpret_cast<const float4 *>(s.kb + e4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        vbias[it] = s.vb ? *reinterpret_cast<const float4 *>(s.vb + e4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = min(tid + it * NT, kItems - 1);
            const int kp = i / (HD / 4), e4 = i - kp * (HD / 4), key = 2 * kp;
            const int tok0 = min(k0 + key, L - 1), tok1 = min(k0 + key + 1, L - 1);
            pka[it] = *reinterpret_cast<const float4 *>(s.k + (int64_t)tok0 * ts + e4 * 4); pkb[it] = *reinterpret_cast<const float4 *>(s.k + (int64_t)tok1 * ts + e4 * 4);
            pva[it] = *reinterpret_cast<const float4 *>(s.v + (int64_t)tok0 * ts + e4 * 4); pvb[it] = *reinterpret_cast<const float4 *>(s.v + (int64_t)tok1 * ts + e4 * 4);
        }
    };
    fetch(0);      // (first: its latency overlaps the Q / dO / O loads of the prologue)

    const int qi = lane & 15, kg = lane >> 4;
    const float qscale = p.fwd.scale * kLog2e;
    // Q^T (scaled into the log2 domain) and dO^T fragments (B operands) of the wave's QT query tiles: chunk c, slots j <-> e = 32c + 8 kg + j
    int q_tok[QT];
    float dpart[QT], lse2[QT], ginv[F16 ? QT : 1];
    Frag<F16> qf[QT][EC], gf[QT][EC];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        q_tok[t] = qblk * kSQW + (wave * QT + t) * 16 + qi;
        const int q_ld = min(q_tok[t], L - 1);
        const float *dorow = reinterpret_cast<const float *>(p.dout_ptr) + (int64_t)b * p.fwd.out_batch_stride + (int64_t)q_ld * p.fwd.out_token_stride + dir * C + h * HD;
        const float *orow = reinterpret_cast<const float *>(p.fwd.out_ptr) + (int64_t)b * p.fwd.out_batch_stride + (int64_t)q_ld * p.fwd.out_token_stride + dir * C + h * HD;
        float dp = 0.f, gmax = 0.f;
        float gkeep[F16 ? EC : 1][8];         // fp16 carrier: the dO row waits for its scale
#pragma unroll
        for (int c = 0; c < EC; ++c) {
            float qv[8], gv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e0 = 32 * c + 8 * kg + 4 * half;
                float4 tq = make_float4(0.f, 0.f, 0.f, 0.f), g = tq;
                if (e0 < HD) {
                    tq = ld_bias4(s.q + (int64_t)q_ld * ts, s.qb, e0);
                    g = *reinterpret_cast<const float4 *>(dorow + e0);
                    const float4 o = *reinterpret_cast<const float4 *>(orow + e0);
                    dp += g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
                    if constexpr (F16) gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
                }
                qv[4 * half + 0] = tq.x * qscale; qv[4 * half + 1] = tq.y * qscale; qv[4 * half + 2] = tq.z * qscale; qv[4 * half + 3] = tq.w * qscale;
                gv[4 * half + 0] = g.x; gv[4 * half + 1] = g.y; gv[4 * half + 2] = g.z; gv[4 * half + 3] = g.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                frag_put2<F16>(qf[t][c], i, qv[2 * i], qv[2 * i + 1]);
                if constexpr (!F16) frag_put2<F16>(gf[t][c], i, gv[2 * i], gv[2 * i + 1]);
            }
            if constexpr (F16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) gkeep[c][i] = gv[i];
            }
        }
        float gs = 1.f;
        if constexpr (F16) {
            gs = row_pow2_scale(quad_max(gmax));     // the query's 4 lanes hold its whole dO row between them
#pragma unroll
            for (int c = 0; c < EC; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) frag_put2<F16>(gf[t][c], i, gkeep[c][2 * i] * gs, gkeep[c][2 * i + 1] * gs);
        }
        dpart[t] = quad_sum(dp);                              // D of this lane's query
        const int64_t stat = (((int64_t)b * ndir + dir) * H + h) * L + q_ld;
        lse2[t] = reinterpret_cast<const float *>(p.fwd.lse_ptr)[stat] * kLog2e;
        if (q_tok[t] < L && kg == 0) {
            reinterpret_cast<float *>(p.delta_ptr)[stat] = dpart[t];
            if constexpr (F16) reinterpret_cast<float *>(p.delta_ptr)[nstat + stat] = gs;
        }
        if constexpr (F16) {      // D in the row's scaled domain; P as 2^kPShift P (folded into the log-sum-exp); both undone, exactly, on store
            dpart[t] *= gs;
            lse2[t] -= (float)kPShift;
            ginv[t] = (1.f / gs) * (1.f / (float)(1 << kPShift));
        }
    }

    f4 acc[QT][ET];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int e = 0; e < ET; ++e) acc[t][e] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EP > HD) {      // padding that is never rewritten: columns e in [hd, EP) of K and V
        for (int i = tid; i < kBKT * (EP - HD); i += NT) { const int key = i / (EP - HD), e = (HD + i % (EP - HD)) ^ flip(key); img_zero<F16>(Kh, Kl, key * KS + e); img_zero<F16>(Vh, Vl, key * KS + e); }
    }
    if constexpr (ET * 16 > HD) {  // rows e in [hd, ET*16) of K^T
        for (int i = tid; i < (ET * 16 - HD) * kBKT; i += NT) { const int e = HD + i / kBKT, k = (i % kBKT) ^ flip(e); img_zero<F16>(Th, Tl, e * TS + k); }
    }


    for (int k0 = 0; k0 < L; k0 += kBKT) {
        __syncthreads();
        // ---- stage K, V [key][e] and K^T [e][slot(key)] as operand images; one thread = 2 keys x 4 e ---------------------------
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = tid + it * NT;
            if (i >= kItems) break;
            const int kp = i / (HD / 4), e4 = i - kp * (HD / 4), key = 2 * kp;
            const float4 ka = add4(pka[it], kbias[it]), kb = add4(pkb[it], kbias[it]), va = add4(pva[it], vbias[it]), vb = add4(pvb[it], vbias[it]);
            const int ke = (e4 * 4) ^ flip(key);                    // key even: key and key + 1 are rows of the same kind
            img_st4<F16>(Kh, Kl, key * KS + ke, ka); img_st4<F16>(Kh, Kl, (key + 1) * KS + ke, kb);
            img_st4<F16>(Vh, Vl, key * KS + ke, va); img_st4<F16>(Vh, Vl, (key + 1) * KS + ke, vb);
            const int pos = ((key & ~31) + cslot(key & 31)) ^ flip(e4 * 4);      // rows e4*4 .. +3 of K^T: one kind
            const float a4[4] = {ka.x, ka.y, ka.z, ka.w}, b4[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) img_st2<F16>(Th, Tl, (e4 * 4 + e) * TS + pos, a4[e], b4[e]);
        }
        __syncthreads();
        if (k0 + kBKT < L) fetch(k0 + kBKT);

        f4 ds[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            // the accumulators start at -lse and -D: the MFMA chains then end in S^T - lse and dP^T - D
            f4 sacc[QT], pacc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) { sacc[t] = f4{-lse2[t], -lse2[t], -lse2[t], -lse2[t]}; pacc[t] = f4{-dpart[t], -dpart[t], -dpart[t], -dpart[t]}; }
            const int krow = (kt * 16 + qi) * KS + ((8 * kg) ^ flip(qi));
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const Frag<F16> kf = frag_ld<F16>(Kh, Kl, krow + 32 * c), vf = frag_ld<F16>(Vh, Vl, krow + 32 * c);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    sacc[t] = frag_mfma<F16>(kf, qf[t][c], sacc[t]);      // S^T - lse  (log2 domain)
                    pacc[t] = frag_mfma<F16>(vf, gf[t][c], pacc[t]);      // dP^T - D   (fp16 carrier: times the query's g)
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[t][kt][r] = fast_exp2(sacc[t][r]) * pacc[t][r];      // dS^T = P^T o (dP^T - D)
        }
        if (k0 + kBKT > L) {      // the last tile: keys beyond L (clamped loads) contribute nothing
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + kt * 16 + kg * 4 + r >= L) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) ds[t][kt][r] = 0.f;
                    }
        }
        // ---- dQ^T += K^T dS^T: chunk c = key tiles 2c, 2c + 1 -----------------------------------------------------------------
        Frag<F16> df[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            df[t][0] = frag_c2<F16>(ds[t][0], ds[t][1]);
            df[t][1] = frag_c2<F16>(ds[t][2], ds[t][3]);
        }
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int trow = (e * 16 + qi) * TS + ((8 * kg) ^ flip(qi));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const Frag<F16> tf = frag_ld<F16>(Th, Tl, trow + 32 * c);
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t][e] = frag_mfma<F16>(tf, df[t][c], acc[t][e]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (q_tok[t] >= L) continue;
        float *dst = s.dq + (int64_t)q_tok[t] * dts;
        float sc = p.fwd.scale;
        if constexpr (F16) sc *= ginv[t];
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int e0 = e * 16 + kg * 4;
            if (e0 < HD) *reinterpret_cast<float4 *>(dst + e0) = make_float4(acc[t][e][0] * sc, acc[t][e][1] * sc, acc[t][e][2] * sc, acc[t][e][3] * sc);
        }
    }
}

// KT = 16-key tiles per wave: a workgroup covers 64 * KT keys, so the staging of a query tile (loads, 16 hi / lo splits and 24 LDS
// writes per thread: more VALU work than the tile's 48 MFMAs per key tile take on the matrix cores) and every A-operand
// ds_read_b128 are shared by KT key tiles -- the forward's QT = 2 trick; 2 workgroups per CU instead of 3 (head_dim <= 64).
template <int HD, int KT, bool F16>
__global__ __launch_bounds__(256, (HD > 64 || KT == 2) ? 2 : 3) void xattn_bwd_dkv_split_kernel(const dimsum_xattn_bwd_params_t p) {
    constexpr int EP = (HD + 31) / 32 * 32, EC = EP / 32, ET = (HD + 15) / 16;
    constexpr int RS = EP + 8;               // [query][e] rows of Q and dO
    constexpr int TS = kBQT + 8;             // [e][query slot] rows of Q^T and dO^T
    constexpr int LO = F16 ? 0 : 1;
    __shared__ __attribute__((aligned(16))) unsigned short Qh[kBQT * RS], Gh[kBQT * RS], QTh[ET * 16 * TS], GTh[ET * 16 * TS];
    __shared__ __attribute__((aligned(16))) unsigned short Ql[LO * kBQT * RS + 8], Gl[LO * kBQT * RS + 8], QTl[LO * ET * 16 * TS + 8], GTl[LO * ET * 16 * TS + 8];
    __shared__ __attribute__((aligned(16))) float sL[kBQT], sD[kBQT];      // lse (log2 domain; fp16 carrier: minus the row's shifts) and D of the tile's queries
    __shared__ float sRed[4];
    static_assert((RS / 8) % 2 == 1 && (TS / 8) % 2 == 1, "odd number of 16-byte slots per row");
    auto flip = [](int row) { return (((row & 15) + 4) & 8); };      // slot-pair swap of rows 4-11 of a fragment (see the dq kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.fwd.seqlen, H = p.fwd.heads;
    const int kblocks = (L + 64 * KT - 1) / (64 * KT);
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, kblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int kblk = idx % kblocks; idx /= kblocks;
    const int ndir = p.fwd.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const XSrc s = xattn_src(p, b, h, dir, HD);
    const int64_t ts = p.fwd.qkv_token_stride, dts = p.dqkv_token_stride;
    const float *dobase = reinterpret_cast<const float *>(p.dout_ptr) + (int64_t)b * p.fwd.out_batch_stride + dir * C + h * HD;
    const int64_t stat0 = (((int64_t)b * ndir + dir) * H + h) * L;
    const float *lse = reinterpret_cast<const float *>(p.fwd.lse_ptr) + stat0;
    const float *dlt = reinterpret_cast<const float *>(p.delta_ptr) + stat0;
    const float *gsc = dlt + (int64_t)p.fwd.batch * ndir * H * L;        // fp16 carrier: the dO row scales the dq kernel left

    // The Q / dO rows of the NEXT query tile are requested right after the current tile has been staged, so their HBM
    // latency hides under the tile's MFMA work (register-staged prefetch: one thread = 2 queries x 4 e, kIt items per tile).
    constexpr int kItems = (kBQT / 2) * (HD / 4), kIt = (kItems + 255) / 256;
    // (raw loads, the bias is added when the tile is staged: see the dq kernel; the tile's per-query statistics -- lse, D, the dO row
    // scale -- ride along in the first kBQT threads)
    float4 pqa[kIt], pqb[kIt], pga[kIt], pgb[kIt], qbias[kIt];
    float2 pgs[kIt];
    float pl = 0.f, pd = 0.f, pg = 1.f;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int i = min(tid + it * 256, kItems - 1), e4 = i % (HD / 4);
        qbias[it] = s.qb ? *reinterpret_cast<const float4 *>(s.qb + e4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto fetch = [&](int q0) {
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = min(tid + it * 256, kItems - 1);
            const int qp = i / (HD / 4), e4 = i - qp * (HD / 4), q = 2 * qp;
            const int tok0 = min(q0 + q, L - 1), tok1 = min(q0 + q + 1, L - 1);
            pqa[it] = *reinterpret_cast<const float4 *>(s.q + (int64_t)tok0 * ts + e4 * 4); pqb[it] = *reinterpret_cast<const float4 *>(s.q + (int64_t)tok1 * ts + e4 * 4);
            pga[it] = *reinterpret_cast<const float4 *>(dobase + (int64_t)tok0 * p.fwd.out_token_stride + e4 * 4);
            pgb[it] = *reinterpret_cast<const float4 *>(dobase + (int64_t)tok1 * p.fwd.out_token_stride + e4 * 4);
            if constexpr (F16) pgs[it] = make_float2(gsc[tok0], gsc[tok1]);
        }
        if (tid < kBQT) {
            const int tok = min(q0 + tid, L - 1);
            pl = lse[tok]; pd = dlt[tok];
            if constexpr (F16) pg = gsc[tok];
        }
    };
    fetch(0);      // (first: its latency overlaps the g_min reduction and the K / V fragment loads)

    // fp16 carrier: g_min = the scale of this (batch, head, direction)'s largest gradient row
    float gsmin = 1.f;
    if constexpr (F16) {
        float m = 3.0e38f;
        for (int i = tid; i < L; i += 256) m = fminf(m, gsc[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fminf(m, __shfl_xor(m, o));
        if (lane == 0) sRed[wave] = m;
        __syncthreads();
        gsmin = fminf(fminf(sRed[0], sRed[1]), fminf(sRed[2], sRed[3]));
    }

    const int ki = lane & 15, kg = lane >> 4;
    const float kscale = p.fwd.scale * kLog2e;
    // K^T (scaled) and V^T fragments of this lane's keys (B operands): chunk c, slots j <-> e = 32c + 8 kg + j
    int k_tok[KT];
    bool key_live[KT];
    Frag<F16> kf[KT][EC], vf[KT][EC];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        k_tok[t] = kblk * (64 * KT) + (wave * KT + t) * 16 + ki;
        key_live[t] = k_tok[t] < L;
        const int k_ld = min(k_tok[t], L - 1);
#pragma unroll
        for (int c = 0; c < EC; ++c) {
            float kv[8], vv[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e0 = 32 * c + 8 * kg + 4 * half;
                float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), v4 = kk;
                if (e0 < HD) { kk = ld_bias4(s.k + (int64_t)k_ld * ts, s.kb, e0); v4 = ld_bias4(s.v + (int64_t)k_ld * ts, s.vb, e0); }
                kv[4 * half + 0] = kk.x * kscale; kv[4 * half + 1] = kk.y * kscale; kv[4 * half + 2] = kk.z * kscale; kv[4 * half + 3] = kk.w * kscale;
                vv[4 * half + 0] = v4.x; vv[4 * half + 1] = v4.y; vv[4 * half + 2] = v4.z; vv[4 * half + 3] = v4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { frag_put2<F16>(kf[t][c], i, kv[2 * i], kv[2 * i + 1]); frag_put2<F16>(vf[t][c], i, vv[2 * i], vv[2 * i + 1]); }
        }
    }
    f4 dk[KT][ET], dv[KT][ET];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int e = 0; e < ET; ++e) { dk[t][e] = f4{0.f, 0.f, 0.f, 0.f}; dv[t][e] = f4{0.f, 0.f, 0.f, 0.f}; }
    if constexpr (EP > HD) {
        for (int i = tid; i < kBQT * (EP - HD); i += 256) { const int q = i / (EP - HD), e = (HD + i % (EP - HD)) ^ flip(q); img_zero<F16>(Qh, Ql, q * RS + e); img_zero<F16>(Gh, Gl, q * RS + e); }
    }
    if constexpr (ET * 16 > HD) {
        for (int i = tid; i < (ET * 16 - HD) * kBQT; i += 256) { const int e = HD + i / kBQT, q = (i % kBQT) ^ flip(e); img_zero<F16>(QTh, QTl, e * TS + q); img_zero<F16>(GTh, GTl, e * TS + q); }
    }


    for (int q0 = 0; q0 < L; q0 += kBQT) {
        __syncthreads();
        // ---- stage Q, dO [query][e] and Q^T, dO^T [e][slot(query)] as operand images; one thread = 2 queries x 4 e ---------------
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int i = tid + it * 256;
            if (i >= kItems) break;
            const int qp = i / (HD / 4), e4 = i - qp * (HD / 4), q = 2 * qp;
            const float4 qa = add4(pqa[it], qbias[it]), qb = add4(pqb[it], qbias[it]);
            float4 ga = pga[it], gb = pgb[it];
            if constexpr (F16) {                                     // dO rows in their scaled domain (exact)
                const float s0 = pgs[it].x, s1 = pgs[it].y;
                ga.x *= s0; ga.y *= s0; ga.z *= s0; ga.w *= s0; gb.x *= s1; gb.y *= s1; gb.z *= s1; gb.w *= s1;
            }
            const int qe = (e4 * 4) ^ flip(q);                      // q even: q and q + 1 are rows of the same kind
            img_st4<F16>(Qh, Ql, q * RS + qe, qa); img_st4<F16>(Qh, Ql, (q + 1) * RS + qe, qb);
            img_st4<F16>(Gh, Gl, q * RS + qe, ga); img_st4<F16>(Gh, Gl, (q + 1) * RS + qe, gb);
            const int pos = cslot(q) ^ flip(e4 * 4);                // rows e4*4 .. +3 of Q^T / dO^T: one kind
            const float qa4[4] = {qa.x, qa.y, qa.z, qa.w}, qb4[4] = {qb.x, qb.y, qb.z, qb.w};
            const float ga4[4] = {ga.x, ga.y, ga.z, ga.w}, gb4[4] = {gb.x, gb.y, gb.z, gb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                img_st2<F16>(QTh, QTl, (e4 * 4 + e) * TS + pos, qa4[e], qb4[e]);
                img_st2<F16>(GTh, GTl, (e4 * 4 + e) * TS + pos, ga4[e], gb4[e]);
            }
        }
        if (tid < kBQT) {
            const bool live = q0 + tid < L;
            float l = live ? pl * kLog2e : 1e30f;                // queries beyond L: P = exp2(S - inf) = 0
            float d = live ? pd : 0.f;
            if constexpr (F16) {
                // P of this query travels as 2^kPShift (g_min / g) P: both factors are powers of two and live in the log-sum-exp
                const unsigned ef = (__float_as_uint(gsmin / pg) >> 23) & 0xffu;      // g_min / g is exact; more than 2^126 apart: the row adds nothing
                l = ef == 0u ? 1e30f : l - (float)((int)ef - 127 + kPShift);
                d *= pg;
            }
            sL[tid] = l;
            sD[tid] = d;
        }
        __syncthreads();
        if (q0 + kBQT < L) fetch(q0 + kBQT);

        // ---- S = Q K^T, dP = dO V^T for the two 16-query tiles: C layout key = lane & 15 (column), queries qt*16 + kg*4 + r;
        //      every Q / dO fragment read serves the wave's KT key tiles ---------------------------------------------------------
        f4 pp[KT][2], dsv[KT][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // the accumulators start at -lse and -D of their rows (queries): the MFMA chains end in S - lse and dP - D
            const float4 l4 = *reinterpret_cast<const float4 *>(&sL[qt * 16 + kg * 4]);
            const float4 d4 = *reinterpret_cast<const float4 *>(&sD[qt * 16 + kg * 4]);
            f4 sacc[KT], pacc[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) { sacc[t] = f4{-l4.x, -l4.y, -l4.z, -l4.w}; pacc[t] = f4{-d4.x, -d4.y, -d4.z, -d4.w}; }
            const int qrow = (qt * 16 + ki) * RS + ((8 * kg) ^ flip(ki));
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const Frag<F16> qf = frag_ld<F16>(Qh, Ql, qrow + 32 * c), gf = frag_ld<F16>(Gh, Gl, qrow + 32 * c);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    sacc[t] = frag_mfma<F16>(qf, kf[t][c], sacc[t]);
                    pacc[t] = frag_mfma<F16>(gf, vf[t][c], pacc[t]);
                }
            }
            // (keys beyond L -- clamped loads -- fill columns of dK^T / dV^T that are never stored: no mask)
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pp[t][qt][r] = fast_exp2(sacc[t][r]);
                    dsv[t][qt][r] = pp[t][qt][r] * pacc[t][r];
                }
        }
        // ---- dV^T += dO^T P, dK^T += Q^T dS: ONE 32-deep chunk whose slots are the tile's queries --------------------------------
        Frag<F16> pf[KT], sf[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) { pf[t] = frag_c2<F16>(pp[t][0], pp[t][1]); sf[t] = frag_c2<F16>(dsv[t][0], dsv[t][1]); }
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int row = (e * 16 + ki) * TS + ((8 * kg) ^ flip(ki));
            const Frag<F16> gt = frag_ld<F16>(GTh, GTl, row), qt_ = frag_ld<F16>(QTh, QTl, row);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                dv[t][e] = frag_mfma<F16>(gt, pf[t], dv[t][e]);
                dk[t][e] = frag_mfma<F16>(qt_, sf[t], dk[t][e]);
            }
        }
    }
    const float unscale = F16 ? (1.f / gsmin) * (1.f / (float)(1 << kPShift)) : 1.f;        // exact
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (!key_live[t]) continue;
        float *dkd = s.dk + (int64_t)k_tok[t] * dts, *dvd = s.dv + (int64_t)k_tok[t] * dts;
        const float sc = p.fwd.scale * unscale;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int e0 = e * 16 + kg * 4;
            if (e0 < HD) {
                *reinterpret_cast<float4 *>(dkd + e0) = make_float4(dk[t][e][0] * sc, dk[t][e][1] * sc, dk[t][e][2] * sc, dk[t][e][3] * sc);
                *reinterpret_cast<float4 *>(dvd + e0) = make_float4(dv[t][e][0] * unscale, dv[t][e][1] * unscale, dv[t][e][2] * unscale, dv[t][e][3] * unscale);
            }
        }
    }
}

template <int HD, bool F16>
static int launch_xbwd_mc(const dimsum_xattn_bwd_params_t &p, hipStream_t s) {
    const int64_t nblk = (int64_t)p.fwd.batch * p.fwd.heads * (p.fwd.n_dirs == 1 ? 1 : 2) * ((p.fwd.seqlen + 63) / 64);
    const int64_t nq = (int64_t)p.fwd.batch * p.fwd.heads * (p.fwd.n_dirs == 1 ? 1 : 2) * ((p.fwd.seqlen + kSQW - 1) / kSQW);
    // (fp16 carrier at head_dim 64: two query tiles per wave fit -- 214 VGPRs -- and measured 1.18 against 1.05 ms for the pair; one key tile
    // per wave in the dk / dv kernel at 4 waves per SIMD: 0.856 against 0.838 ms. The split carrier's choices stand.)
    if constexpr (HD <= 32) hipLaunchKernelGGL((xattn_bwd_dq_split_kernel<HD, 2, F16>), dim3((unsigned)nq), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((xattn_bwd_dq_split_kernel<HD, 1, F16>), dim3((unsigned)nq), dim3(512), 0, s, p);
    if (launch_status() != DIMSUM_OK) return DIMSUM_ERR_LAUNCH;
    // two key tiles per wave (128 keys per workgroup) halve the per-key staging work; short sequences and the wide head keep one
    bool two = false;
    if constexpr (HD <= 64) {
        if (p.fwd.seqlen >= 128) {
            const int64_t nk2 = (int64_t)p.fwd.batch * p.fwd.heads * (p.fwd.n_dirs == 1 ? 1 : 2) * ((p.fwd.seqlen + 127) / 128);
            hipLaunchKernelGGL((xattn_bwd_dkv_split_kernel<HD, 2, F16>), dim3((unsigned)nk2), dim3(256), 0, s, p);
            two = true;
        }
    }
    if (!two) hipLaunchKernelGGL((xattn_bwd_dkv_split_kernel<HD, 1, F16>), dim3((unsigned)nblk), dim3(256), 0, s, p);
    return launch_status();
}

template <int HD>
static int launch_xbwd(const dimsum_xattn_bwd_params_t &p, hipStream_t s) {
    const int64_t nblk = (int64_t)p.fwd.batch * p.fwd.heads * (p.fwd.n_dirs == 1 ? 1 : 2) * ((p.fwd.seqlen + 63) / 64);
    if (nblk > 0x7fffffff) return DIMSUM_ERR_SHAPE;
    if (p.fwd.precision == 2) return launch_xbwd_mc<HD, true>(p, s);
    if (p.fwd.precision == 1) return launch_xbwd_mc<HD, false>(p, s);
    hipLaunchKernelGGL(xattn_bwd_dq_kernel<HD>, dim3((unsigned)nblk), dim3(256), 0, s, p);
    if (launch_status() != DIMSUM_OK) return DIMSUM_ERR_LAUNCH;
    hipLaunchKernelGGL(xattn_bwd_dkv_kernel<HD>, dim3((unsigned)nblk), dim3(256), 0, s, p);
    return launch_status();
}

}  // namespace dimsum

extern "C" int dimsum_xattn_fusion_bwd(const dimsum_xattn_bwd_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_xattn_bwd_params_t)) return DIMSUM_ERR_ABI;
    const bool self_attn = p->fwd.n_dirs == 1;
    if (!p->fwd.qkv1_ptr || (!self_attn && !p->fwd.qkv2_ptr) || !p->fwd.out_ptr || !p->fwd.lse_ptr || !p->dout_ptr || !p->dqkv1_ptr ||
        (!self_attn && !p->dqkv2_ptr) || !p->delta_ptr)
        return DIMSUM_ERR_NULL;
    const dimsum_xattn_params_t &f = p->fwd;
    if (f.batch < 0 || f.seqlen <= 0 || f.heads <= 0 || (f.n_dirs != 0 && f.n_dirs != 1 && f.n_dirs != 2)) return DIMSUM_ERR_SHAPE;
    if (f.precision != 0 && f.precision != 1 && f.precision != 2) return DIMSUM_ERR_SHAPE;
    if (f.precision == 2 && f.qkv_f16 != 0) return DIMSUM_ERR_SHAPE;      // the backward reads the fp32 qkv tensors
    if (!self_attn && (f.bias1_ptr == nullptr) != (f.bias2_ptr == nullptr)) return DIMSUM_ERR_NULL;
    const void *ptrs[] = {f.qkv1_ptr, self_attn ? nullptr : f.qkv2_ptr, f.out_ptr, p->dout_ptr, p->dqkv1_ptr, self_attn ? nullptr : p->dqkv2_ptr,
                          f.bias1_ptr, self_attn ? nullptr : f.bias2_ptr};
    for (const void *q : ptrs)
        if (q && !aligned_to<float>(q, 16)) return DIMSUM_ERR_STRIDE;
    if (f.qkv_batch_stride % 4 || f.qkv_token_stride % 4 || f.out_batch_stride % 4 || f.out_token_stride % 4 || p->dqkv_batch_stride % 4 ||
        p->dqkv_token_stride % 4)
        return DIMSUM_ERR_STRIDE;
    if (f.batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (f.head_dim) {
        case 24: return launch_xbwd<24>(*p, s);
        case 32: return launch_xbwd<32>(*p, s);
        case 48: return launch_xbwd<48>(*p, s);
        case 64: return launch_xbwd<64>(*p, s);
        case 72: return launch_xbwd<72>(*p, s);
        default: return DIMSUM_ERR_SHAPE;
    }
}
