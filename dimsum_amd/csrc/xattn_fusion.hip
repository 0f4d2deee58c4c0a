// xattn_fusion.hip -- cross-attention fusion core on the matrix cores (gfx950), fp32 in / fp32 out.
//
// Replaces the two F.scaled_dot_product_attention calls + transposes + concat of CrossAttentionFusion.forward
// (dimsum/attention_fusion.py:64-79, swap_k = False), reading q/k/v straight from the qkv GEMM outputs
// (batch, L, 3*heads*hd laid out [q | k | v], head-major) and writing the concatenated proj input (batch, L, 2*heads*hd):
//     out[b, i, h*hd + e]       = sum_j softmax_j(q1_i . k2_j / sqrt(hd)) v2_j[e]      ("x12")
//     out[b, i, C + h*hd + e]   = sum_j softmax_j(q2_i . k1_j / sqrt(hd)) v1_j[e]      ("x21")
//
// MFMA mapping (v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation, 157 TFLOP/s peak = the fp32 vector rate,
// MI355X_MICROARCH.md) -- everything is computed TRANSPOSED so that no operand ever changes layout:
//     S^T (keys x queries) = K Q^T     A = K[key = lane&15][e],  B = Q^T[e][query = lane&15]
//         C layout: lane holds query (lane&15) and keys (lane>>4)*4 + r, r = 0..3 of each 16-key tile
//     O^T (hd x queries)   = V^T P^T   B operand of K-step r  ==  C register r of the S^T tile, untouched
//                                      (the 4 keys of a K-step are {r, 4+r, 8+r, 12+r}), A = V^T[e = lane&15][those keys]
// so a lane only ever works for ONE query: softmax statistics are an in-lane reduction + two cross-lane steps
// (lanes l, l^16, l^32, l^48), and the online-softmax rescale of O^T is a per-lane scalar.
// The reduction index e is consumed 16 at a time: one ds_read_b128 of K (4 consecutive e) feeds 4 MFMA K-steps, with the
// matching 4 consecutive e of Q in registers (any partition of e into groups of 4 is a valid K-step).
//
// Workgroup = 4 waves = 64 queries of one (batch, head, direction); K and V^T tiles of 64 keys are staged through LDS
// once per workgroup (34 KB -> 4 workgroups per CU). 2 * (QK^T + PV) = 4*L*L*hd FLOP per (b, h, direction).
#include "xattn_common.hpp"

namespace dimsum {

constexpr int kKT = 64;          // keys per tile
constexpr int kQW = 16;          // queries per wave

template <int HD>
__global__ __launch_bounds__(256) void xattn_fusion_fwd_kernel(const dimsum_xattn_params_t p) {
    constexpr int KS = HD + 4;               // K tile row stride (floats): 16-B aligned rows, conflict-free b128 reads
    constexpr int VS = kKT + 4;              // V^T tile row stride
    constexpr int ET = (HD + 15) / 16;       // 16-row output tiles along e
    constexpr int EC = HD / 16;              // full 16-wide chunks of the reduction index
    constexpr bool kTail8 = (HD % 16) == 8;  // plus one 8-wide chunk (hd = 24, 72)
    static_assert(HD % 8 == 0, "head_dim must be a multiple of 8");
    __shared__ __attribute__((aligned(16))) float Ks[kKT * KS];
    __shared__ __attribute__((aligned(16))) float Vt[ET * 16 * VS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.seqlen, H = p.heads;
    const int qblocks = (L + 63) / 64;
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, qblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int qb = idx % qblocks; idx /= qblocks;
    const int ndir = p.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    // direction 0: q1, k2, v2   direction 1: q2, k1, v1   (self-attention, n_dirs = 1: q1, k1, v1)
    const bool kv_from_1 = dir == 1 || ndir == 1;
    const float *qsrc = reinterpret_cast<const float *>(dir == 0 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const float *kvsrc = reinterpret_cast<const float *>(kv_from_1 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const float *ksrc = kvsrc + C, *vsrc = kvsrc + 2 * C;
    const int64_t ts = p.qkv_token_stride;
    // optional qkv Linear biases (the GEMMs then run without a bias epilogue): same [q | k | v], head-major layout
    const float *qbv = reinterpret_cast<const float *>(dir == 0 ? p.bias1_ptr : p.bias2_ptr);
    const float *kvb = reinterpret_cast<const float *>(kv_from_1 ? p.bias1_ptr : p.bias2_ptr);
    const float *qbias = qbv ? qbv + h * HD : nullptr;
    const float *kbias = kvb ? kvb + C + h * HD : nullptr, *vbias = kvb ? kvb + 2 * C + h * HD : nullptr;

    const int qi = lane & 15, kg = lane >> 4;                 // this lane's query (within the wave) and k-index group
    const int q_tok = qb * 64 + wave * kQW + qi;
    const int q_ld = min(q_tok, L - 1);
    const float qscale = p.scale * kLog2e;                    // scores live in the log2 domain
    // Q^T fragments: chunk c holds e = 16c + 4*kg .. +3
    f4 qf[EC + (kTail8 ? 1 : 0)];
#pragma unroll
    for (int c = 0; c < EC; ++c) {
        float4 t = *reinterpret_cast<const float4 *>(qsrc + (int64_t)q_ld * ts + 16 * c + 4 * kg);
        if (qbias) { const float4 bq = *reinterpret_cast<const float4 *>(qbias + 16 * c + 4 * kg); t.x += bq.x; t.y += bq.y; t.z += bq.z; t.w += bq.w; }
        qf[c] = f4{t.x * qscale, t.y * qscale, t.z * qscale, t.w * qscale};
    }
    if constexpr (kTail8) {                                   // 8-wide tail: k-groups 0,1 -> e = 16*EC + 4*(kg&1) ..; groups 2,3 idle
        float4 t = *reinterpret_cast<const float4 *>(qsrc + (int64_t)q_ld * ts + 16 * EC + 4 * (kg & 1));
        if (qbias) { const float4 bq = *reinterpret_cast<const float4 *>(qbias + 16 * EC + 4 * (kg & 1)); t.x += bq.x; t.y += bq.y; t.z += bq.z; t.w += bq.w; }
        const float m = (kg < 2) ? qscale : 0.f;
        qf[EC] = f4{t.x * m, t.y * m, t.z * m, t.w * m};
    }

    f4 o[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) o[e] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    for (int k0 = 0; k0 < L; k0 += kKT) {
        __syncthreads();
        // ---- stage K [key][e] and V^T [e][key] for keys k0 .. k0+63 ---------------------------------------------------
        for (int i = tid; i < kKT * (HD / 4); i += 256) {
            const int key = i / (HD / 4), e4 = i - key * (HD / 4);
            const int tok = min(k0 + key, L - 1);
            float4 kv = *reinterpret_cast<const float4 *>(ksrc + (int64_t)tok * ts + e4 * 4);
            float4 vv = *reinterpret_cast<const float4 *>(vsrc + (int64_t)tok * ts + e4 * 4);
            if (kbias) {
                const float4 bk = *reinterpret_cast<const float4 *>(kbias + e4 * 4), bv = *reinterpret_cast<const float4 *>(vbias + e4 * 4);
                kv.x += bk.x; kv.y += bk.y; kv.z += bk.z; kv.w += bk.w;
                vv.x += bv.x; vv.y += bv.y; vv.z += bv.z; vv.w += bv.w;
            }
            *reinterpret_cast<float4 *>(&Ks[key * KS + e4 * 4]) = kv;
            Vt[(e4 * 4 + 0) * VS + key] = vv.x; Vt[(e4 * 4 + 1) * VS + key] = vv.y;
            Vt[(e4 * 4 + 2) * VS + key] = vv.z; Vt[(e4 * 4 + 3) * VS + key] = vv.w;
        }
        if constexpr (ET * 16 > HD) {                         // zero the padding rows of V^T (e >= hd)
            for (int i = tid; i < (ET * 16 - HD) * kKT; i += 256) Vt[(HD + i / kKT) * VS + (i % kKT)] = 0.f;
        }
        __syncthreads();

        // ---- S^T = K Q^T for the 4 key tiles of 16 ------------------------------------------------------------------
        f4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            const float *krow = &Ks[(kt * 16 + qi) * KS];      // A operand row: key = kt*16 + (lane&15)
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const float4 kf = *reinterpret_cast<const float4 *>(krow + 16 * c + 4 * kg);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, acc, 0, 0, 0);
            }
            if constexpr (kTail8) {
                const float4 kf = *reinterpret_cast<const float4 *>(krow + 16 * EC + 4 * (kg & 1));
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[EC].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[EC].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[EC].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[EC].w, acc, 0, 0, 0);
            }
            s[kt] = acc;                                       // s[kt][r]: key k0 + kt*16 + kg*4 + r, query qi
        }
        // ---- online softmax for this lane's query ---------------------------------------------------------------------
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + kt * 16 + kg * 4 + r >= L) s[kt][r] = -1e30f;
                mx = fmaxf(mx, s[kt][r]);
            }
        mx = quad_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] = fast_exp2(s[kt][r] - m_new); rs += s[kt][r]; }
        rs = quad_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < ET; ++e) o[e] *= alpha;
        // ---- O^T += V^T P^T: K-step (kt, r) covers keys kt*16 + {r, 4+r, 8+r, 12+r}; its B operand is s[kt][r] -------------
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const float *vrow = &Vt[(e * 16 + qi) * VS];       // A operand row: e = e*16 + (lane&15)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float4 vf = *reinterpret_cast<const float4 *>(vrow + kt * 16 + kg * 4);   // keys kt*16 + kg*4 + r
                o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, s[kt][0], o[e], 0, 0, 0);
                o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, s[kt][1], o[e], 0, 0, 0);
                o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, s[kt][2], o[e], 0, 0, 0);
                o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, s[kt][3], o[e], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: O^T C layout = query (lane&15), e = et*16 + kg*4 + r  -> 16-byte stores --------------------------------
    if (q_tok < L) {
        const float inv = 1.0f / l_run;
        float *dst = reinterpret_cast<float *>(p.out_ptr) + (int64_t)b * p.out_batch_stride + (int64_t)q_tok * p.out_token_stride + dir * C + h * HD;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int e0 = e * 16 + kg * 4;
            if (e0 < HD) *reinterpret_cast<float4 *>(dst + e0) = make_float4(o[e][0] * inv, o[e][1] * inv, o[e][2] * inv, o[e][3] * inv);
        }
        if (p.lse_ptr && kg == 0)
            reinterpret_cast<float *>(p.lse_ptr)[(((int64_t)b * ndir + dir) * H + h) * L + q_tok] = (m_run + __builtin_amdgcn_logf(l_run)) * kLn2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 variant (precision = 1). Same transposed formulation, on v_mfma_f32_16x16x32_bf16 (2.5 PFLOP/s dense peak,
// 16x the fp32 MFMA rate): every fp32 operand x is split into hi = bf16(x), lo = bf16(x - hi) and a product a.b becomes
// hi.hi + hi.lo + lo.hi with fp32 accumulation (the dropped lo.lo term is ~2^-16 relative) -- the arithmetic hipBLASLt uses
// for the library GEMMs under the reference's allow_tf32 policy on gfx950, so attention and GEMMs share one precision
// policy. 3 bf16 MFMAs of K = 32 replace 8 fp32 MFMAs of K = 4: 5.3x fewer matrix-core cycles per tile.
//   * K-slot j of lane group g (= lane >> 4) of a 32-deep chunk is reduction index 8g + j. For QK^T that is e = 32c + 8g + j
//     (one ds_read_b128 of the bf16 K tile); for PV the slots of chunk c are the keys the lane ALREADY holds in its two S^T
//     C-register quadruples: j < 4 -> key 32c + 4g + j (tile 2c), j >= 4 -> key 32c + 16 + 4g + (j - 4) (tile 2c + 1).
//     V^T is staged with exactly that key permutation inside each 32-key chunk, so its A operand is one ds_read_b128 too
//     and P^T never changes layout.
//   * K, V^T tiles are split once per workgroup while they are staged (hi and lo images, bf16, rows padded by 16 B: an odd
//     number of 16-byte slots per row). A ds_read_b128 is serviced in four FIXED groups of 16 lanes ({0-3, 12-15, 20-27}, ...:
//     MI355X_MICROARCH.md, LDS): a group reads all 16 rows of a fragment, rows 0-3 / 12-15 at k-slot a and rows 4-11 at a ^ 1,
//     which the padding alone cannot separate (PMC: LDS 34 % busy, most of it bank-conflict cycles). Rows 4-11 of every 16-row
//     fragment therefore keep their slot PAIRS swapped (slot ^ 1): the lanes of rows 4-11 read k-slot kg ^ 1, every group then
//     reads ONE slot index of 16 rows with an odd slot stride -- conflict-free, at no cost in the loop (the flip is part of
//     the lane's base address).
// QT = 16-query tiles per wave: a workgroup covers 64 * QT queries, so the K / V^T staging (load, bias, hi / lo split, LDS
// writes: more VALU work than the softmax itself) and every A-operand ds_read_b128 are shared by QT query tiles.
template <int HD, int QT>
__global__ __launch_bounds__(256, (QT == 2 && HD <= 64) ? 3 : (QT == 2 ? 2 : 1)) void xattn_fusion_fwd_split_kernel(const dimsum_xattn_params_t p) {
    constexpr int EP = (HD + 31) / 32 * 32;  // reduction length of QK^T, padded with zeros to whole 32-deep chunks
    constexpr int EC = EP / 32;
    constexpr int ET = (HD + 15) / 16;       // 16-row output tiles along e
    constexpr int KS = EP + 8;               // K tile row stride in bf16 elements (16 B of padding)
    constexpr int VS = kKT + 8;              // V^T tile row stride
    static_assert(HD % 8 == 0, "head_dim must be a multiple of 8");
    static_assert((KS / 8) % 2 == 1 && (VS / 8) % 2 == 1, "odd number of 16-byte slots per row");
    auto flip = [](int row) { return (((row & 15) + 4) & 8); };     // 8 elements (one slot) for rows 4-11 of a 16-row fragment
    __shared__ __attribute__((aligned(16))) unsigned short Kh[kKT * KS], Kl[kKT * KS];
    __shared__ __attribute__((aligned(16))) unsigned short Vh[ET * 16 * VS], Vl[ET * 16 * VS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.seqlen, H = p.heads;
    const int qblocks = (L + 64 * QT - 1) / (64 * QT);
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, qblocks);      // (xattn_common.hpp: a group's blocks on ONE XCD)
    const int qb = idx % qblocks; idx /= qblocks;
    const int ndir = p.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const bool kv_from_1 = dir == 1 || ndir == 1;
    const float *qsrc = reinterpret_cast<const float *>(dir == 0 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const float *kvsrc = reinterpret_cast<const float *>(kv_from_1 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const float *ksrc = kvsrc + C, *vsrc = kvsrc + 2 * C;
    const int64_t ts = p.qkv_token_stride;
    const float *qbv = reinterpret_cast<const float *>(dir == 0 ? p.bias1_ptr : p.bias2_ptr);
    const float *kvb = reinterpret_cast<const float *>(kv_from_1 ? p.bias1_ptr : p.bias2_ptr);
    const float *qbias = qbv ? qbv + h * HD : nullptr;
    const float *kbias = kvb ? kvb + C + h * HD : nullptr, *vbias = kvb ? kvb + 2 * C + h * HD : nullptr;

    const int qi = lane & 15, kg = lane >> 4;
    const float qscale = p.scale * kLog2e;                    // scores live in the log2 domain
    // Q^T fragments (B operand) of the wave's QT query tiles: chunk c, slots j = 0..7 <-> e = 32c + 8 kg + j; zeros beyond hd
    int q_tok[QT];
    u4v qh[QT][EC], ql[QT][EC];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        q_tok[t] = qb * (64 * QT) + (wave * QT + t) * kQW + qi;
        const int q_ld = min(q_tok[t], L - 1);
#pragma unroll
        for (int c = 0; c < EC; ++c) {
            float v[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e0 = 32 * c + 8 * kg + 4 * half;
                float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e0 < HD) {
                    tq = *reinterpret_cast<const float4 *>(qsrc + (int64_t)q_ld * ts + e0);
                    if (qbias) { const float4 bq = *reinterpret_cast<const float4 *>(qbias + e0); tq.x += bq.x; tq.y += bq.y; tq.z += bq.z; tq.w += bq.w; }
                }
                v[4 * half + 0] = tq.x * qscale; v[4 * half + 1] = tq.y * qscale; v[4 * half + 2] = tq.z * qscale; v[4 * half + 3] = tq.w * qscale;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) split2(v[2 * i], v[2 * i + 1], qh[t][c].w[i], ql[t][c].w[i]);
        }
    }
    // zero the padding that is never rewritten: K columns e in [hd, EP), V^T rows e in [hd, ET*16)
    if constexpr (EP > HD) {
        for (int i = tid; i < kKT * (EP - HD); i += 256) { const int key = i / (EP - HD), e = (HD + i % (EP - HD)) ^ flip(key); Kh[key * KS + e] = 0; Kl[key * KS + e] = 0; }
    }
    if constexpr (ET * 16 > HD) {
        for (int i = tid; i < (ET * 16 - HD) * kKT; i += 256) { const int e = HD + i / kKT, k = (i % kKT) ^ flip(e); Vh[e * VS + k] = 0; Vl[e * VS + k] = 0; }
    }

    f4 o[QT][ET];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -1e30f; l_run[t] = 0.f;
#pragma unroll
        for (int e = 0; e < ET; ++e) o[t][e] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // K / V rows of a 64-key tile: one work item = 2 keys x 4 e of K and of V (4 float4 loads). With the 256-VGPR budget of
    // the wide-head / two-tile variant the NEXT tile's rows are requested right after the current tile has been staged, so
    // their HBM / L2 latency flies under the tile's MFMA + softmax work (register-staged double buffer, like the dk/dv
    // backward kernel); the 168-VGPR variants (3 workgroups per CU cover the latency) load where they stage.
    constexpr bool kPF = QT == 2 && HD > 64;
    constexpr bool kPFK = false;                  // (K rows only for the 168-VGPR variants: 40-44 B of scratch per lane -- not worth it)
    constexpr int kItems = (kKT / 2) * (HD / 4), kIters = (kItems + 255) / 256;
    float4 rka[kIters], rkb[kIters], rva[kIters], rvb[kIters];
    auto issue_one = [&](int k0, int it, bool want_k, bool want_v) {
        const int i = tid + it * 256;
        if (kItems % 256 == 0 || i < kItems) {
            const int kp = i / (HD / 4), e4 = i - kp * (HD / 4), key = 2 * kp;
            const int tok0 = min(k0 + key, L - 1), tok1 = min(k0 + key + 1, L - 1);
            if (want_k) { rka[it] = *reinterpret_cast<const float4 *>(ksrc + (int64_t)tok0 * ts + e4 * 4); rkb[it] = *reinterpret_cast<const float4 *>(ksrc + (int64_t)tok1 * ts + e4 * 4); }
            if (want_v) { rva[it] = *reinterpret_cast<const float4 *>(vsrc + (int64_t)tok0 * ts + e4 * 4); rvb[it] = *reinterpret_cast<const float4 *>(vsrc + (int64_t)tok1 * ts + e4 * 4); }
        }
    };
    auto issue_kv = [&](int k0, bool want_v) {
#pragma unroll
        for (int it = 0; it < kIters; ++it) issue_one(k0, it, true, want_v);
    };
    if constexpr (kPF || kPFK) issue_kv(0, kPF);

    for (int k0 = 0; k0 < L; k0 += kKT) {
        __syncthreads();
        // ---- stage K [key][e] and V^T [e][pi(key)] (hi / lo bf16 images) for keys k0 .. k0+63; one thread = 2 keys x 4 e -----
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int i = tid + it * 256;
            if (kItems % 256 != 0 && i >= kItems) continue;
            if constexpr (!kPF) issue_one(k0, it, !kPFK, true);
            const int kp = i / (HD / 4), e4 = i - kp * (HD / 4), key = 2 * kp;
            float4 ka = rka[it], kb = rkb[it], va = rva[it], vb = rvb[it];
            if (kbias) {
                const float4 bk = *reinterpret_cast<const float4 *>(kbias + e4 * 4), bv = *reinterpret_cast<const float4 *>(vbias + e4 * 4);
                ka.x += bk.x; ka.y += bk.y; ka.z += bk.z; ka.w += bk.w; kb.x += bk.x; kb.y += bk.y; kb.z += bk.z; kb.w += bk.w;
                va.x += bv.x; va.y += bv.y; va.z += bv.z; va.w += bv.w; vb.x += bv.x; vb.y += bv.y; vb.z += bv.z; vb.w += bv.w;
            }
            unsigned h0, l0, h1, l1;
            split2(ka.x, ka.y, h0, l0); split2(ka.z, ka.w, h1, l1);
            const int ke = (e4 * 4) ^ flip(key);                    // key even: key and key + 1 are rows of the same kind
            *reinterpret_cast<uint2 *>(&Kh[key * KS + ke]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(&Kl[key * KS + ke]) = make_uint2(l0, l1);
            split2(kb.x, kb.y, h0, l0); split2(kb.z, kb.w, h1, l1);
            *reinterpret_cast<uint2 *>(&Kh[(key + 1) * KS + ke]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(&Kl[(key + 1) * KS + ke]) = make_uint2(l0, l1);
            // position of key kappa inside its 32-key chunk: 8 * ((kappa & 15) >> 2) + (kappa & 3) + 4 * ((kappa >> 4) & 1)
            const int kap = key & 31, pos = (key & ~31) + 8 * ((kap & 15) >> 2) + (kap & 3) + 4 * (kap >> 4);   // key even -> pos, pos + 1 = keys key, key + 1
            const float a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                split2(a4[e], b4[e], h0, l0);
                *reinterpret_cast<unsigned *>(&Vh[(e4 * 4 + e) * VS + (pos ^ flip(e4 * 4))]) = h0;     // rows e4*4 .. +3: one kind
                *reinterpret_cast<unsigned *>(&Vl[(e4 * 4 + e) * VS + (pos ^ flip(e4 * 4))]) = l0;
            }
        }
        __syncthreads();
        if constexpr (kPF || kPFK) if (k0 + kKT < L) issue_kv(k0 + kKT, kPF);      // flies under the MFMA + softmax work below

        // ---- S^T = K Q^T for the 4 key tiles of 16: 3 bf16 MFMAs per 32-deep chunk and query tile ------------------------
        f4 s[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int krow = (kt * 16 + qi) * KS + ((8 * kg) ^ flip(qi));   // A operand row: key = kt*16 + (lane&15), slots e = 32c + 8 kg + j
#pragma unroll
            for (int t = 0; t < QT; ++t) s[t][kt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const u4v kh = *reinterpret_cast<const u4v *>(&Kh[krow + 32 * c]), kl = *reinterpret_cast<const u4v *>(&Kl[krow + 32 * c]);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kl), as_bf16x8(qh[t][c]), s[t][kt], 0, 0, 0);
                    s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kh), as_bf16x8(ql[t][c]), s[t][kt], 0, 0, 0);
                    s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kh), as_bf16x8(qh[t][c]), s[t][kt], 0, 0, 0);
                }
            }
        }                                                      // s[t][kt][r]: key k0 + kt*16 + kg*4 + r, query qi of tile t
        // ---- online softmax for this lane's queries (fp32), then P^T (B operand) of the two 32-key chunks: slots j < 4 =
        //      s[2c][j], j >= 4 = s[2c+1][j-4], split hi / lo -----------------------------------------------------------------
        u4v ph[QT][2], pl[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -1e30f;
            if (k0 + kKT > L) {                        // only the last, ragged key tile needs the mask (wave-uniform branch)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + kt * 16 + kg * 4 + r >= L) s[t][kt][r] = -1e30f;
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][kt][r]);
            mx = quad_max(mx);
            const float m_new = fmaxf(m_run[t], mx);
            const float alpha = fast_exp2(m_run[t] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][kt][r] = fast_exp2(s[t][kt][r] - m_new); rs += s[t][kt][r]; }
            rs = quad_sum(rs);
            l_run[t] = l_run[t] * alpha + rs;
            m_run[t] = m_new;
#pragma unroll
            for (int e = 0; e < ET; ++e) o[t][e] *= alpha;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                split2(s[t][2 * c][0], s[t][2 * c][1], ph[t][c].w[0], pl[t][c].w[0]);
                split2(s[t][2 * c][2], s[t][2 * c][3], ph[t][c].w[1], pl[t][c].w[1]);
                split2(s[t][2 * c + 1][0], s[t][2 * c + 1][1], ph[t][c].w[2], pl[t][c].w[2]);
                split2(s[t][2 * c + 1][2], s[t][2 * c + 1][3], ph[t][c].w[3], pl[t][c].w[3]);
            }
        }
        // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------------------
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int vrow = (e * 16 + qi) * VS + ((8 * kg) ^ flip(qi));    // A operand row: e = e*16 + (lane&15), slots = permuted keys
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u4v vh = *reinterpret_cast<const u4v *>(&Vh[vrow + 32 * c]), vl = *reinterpret_cast<const u4v *>(&Vl[vrow + 32 * c]);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    o[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(vl), as_bf16x8(ph[t][c]), o[t][e], 0, 0, 0);
                    o[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(vh), as_bf16x8(pl[t][c]), o[t][e], 0, 0, 0);
                    o[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(vh), as_bf16x8(ph[t][c]), o[t][e], 0, 0, 0);
                }
            }
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const bool valid = q_tok[t] < L;
        const float inv = 1.0f / l_run[t];
        const int64_t row_off = (int64_t)b * p.out_batch_stride + (int64_t)min(q_tok[t], L - 1) * p.out_token_stride;
        if (p.out_split3) {
            // the token's row is the split-bf16 operand image of the proj Linear (3 x ndir x C bf16 [hi | hi | lo]; strides in bf16
            // elements). The 4 lanes of a query (kg = 0..3) each hold 4 of every 16 e: a 4 x 4 block transpose over v_permlane32_swap /
            // v_permlane16_swap gives lane kg the whole e-tile kg, i.e. 16 consecutive e = 32-byte pieces and 128 contiguous bytes per
            // query and plane (8-byte pieces at a 32-byte stride cost the head_dim-64 kernel +86 us).
            unsigned short *img = reinterpret_cast<unsigned short *>(p.out_ptr) + row_off;
            const int N = ndir * C, col0 = dir * C + h * HD;
            constexpr int kTr = ET >= 4 ? 4 : 0;          // e-tiles that go through the transpose
            if constexpr (kTr == 4) {
                float x[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[e][r] = o[t][e][r] * inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    { auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0][r]), __float_as_uint(x[2][r]), false, false); x[0][r] = __uint_as_float(q[0]); x[2][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[1][r]), __float_as_uint(x[3][r]), false, false); x[1][r] = __uint_as_float(q[0]); x[3][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[0][r]), __float_as_uint(x[1][r]), false, false); x[0][r] = __uint_as_float(q[0]); x[1][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[2][r]), __float_as_uint(x[3][r]), false, false); x[2][r] = __uint_as_float(q[0]); x[3][r] = __uint_as_float(q[1]); }
                }
                // lane kg: x[j][r] = e-tile kg, e = 16 kg + 4 j + r
                unsigned hw[8], lw[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { split2(x[j][0], x[j][1], hw[2 * j], lw[2 * j]); split2(x[j][2], x[j][3], hw[2 * j + 1], lw[2 * j + 1]); }
                if (valid) {
                    unsigned short *d0 = img + col0 + 16 * kg;
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const uint4 hv = make_uint4(hw[4 * hlf], hw[4 * hlf + 1], hw[4 * hlf + 2], hw[4 * hlf + 3]);
                        const uint4 lv = make_uint4(lw[4 * hlf], lw[4 * hlf + 1], lw[4 * hlf + 2], lw[4 * hlf + 3]);
                        *reinterpret_cast<uint4 *>(d0 + 8 * hlf) = hv;
                        if (p.out_split3 == 3) {           // the pair [hi | lo]
                            *reinterpret_cast<uint4 *>(d0 + N + 8 * hlf) = lv;
                        } else {
                            *reinterpret_cast<uint4 *>(d0 + N + 8 * hlf) = hv;
                            *reinterpret_cast<uint4 *>(d0 + 2 * N + 8 * hlf) = lv;
                        }
                    }
                }
            }
            if (valid) {
#pragma unroll
                for (int e = kTr; e < ET; ++e) {
                    const int e0 = e * 16 + kg * 4;
                    if (e0 < HD) st_split_left(img, col0 + e0, N, f32x4{{o[t][e][0] * inv, o[t][e][1] * inv, o[t][e][2] * inv, o[t][e][3] * inv}}, p.out_split3 == 3);
                }
            }
        } else if (valid) {
            float *dst = reinterpret_cast<float *>(p.out_ptr) + row_off + dir * C + h * HD;
#pragma unroll
            for (int e = 0; e < ET; ++e) {
                const int e0 = e * 16 + kg * 4;
                if (e0 < HD) *reinterpret_cast<float4 *>(dst + e0) = make_float4(o[t][e][0] * inv, o[t][e][1] * inv, o[t][e][2] * inv, o[t][e][3] * inv);
            }
        }
        if (valid && p.lse_ptr && kg == 0)
            reinterpret_cast<float *>(p.lse_ptr)[(((int64_t)b * ndir + dir) * H + h) * L + q_tok[t]] = (m_run[t] + __builtin_amdgcn_logf(l_run[t])) * kLn2;
    }
}

int launch_xattn_f16(const dimsum_xattn_params_t &p, hipStream_t s);      // xattn_fusion_f16.hip (precision 2)

}  // namespace dimsum

extern "C" int dimsum_xattn_fusion_fwd(const dimsum_xattn_params_t *p, void *stream) {
    using namespace dimsum;
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_xattn_params_t)) return DIMSUM_ERR_ABI;
    const bool self_attn = p->n_dirs == 1;
    if (!p->qkv1_ptr || (!self_attn && !p->qkv2_ptr) || !p->out_ptr) return DIMSUM_ERR_NULL;
    if (p->n_dirs != 0 && p->n_dirs != 1 && p->n_dirs != 2) return DIMSUM_ERR_SHAPE;
    if (p->batch < 0 || p->seqlen <= 0 || p->heads <= 0) return DIMSUM_ERR_SHAPE;
    if (!self_attn && (p->bias1_ptr == nullptr) != (p->bias2_ptr == nullptr)) return DIMSUM_ERR_NULL;
    if ((p->bias1_ptr && !aligned_to<float>(p->bias1_ptr, 16)) || (!self_attn && p->bias2_ptr && !aligned_to<float>(p->bias2_ptr, 16))) return DIMSUM_ERR_STRIDE;
    if (!aligned_to<float>(p->qkv1_ptr, 16) || (!self_attn && !aligned_to<float>(p->qkv2_ptr, 16)) || !aligned_to<float>(p->out_ptr, 16) ||
        p->qkv_batch_stride % 4 != 0 || p->qkv_token_stride % 4 != 0 || p->out_batch_stride % 4 != 0 || p->out_token_stride % 4 != 0)
        return DIMSUM_ERR_STRIDE;
    if (p->batch == 0) return DIMSUM_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t nblk = (int64_t)p->batch * p->heads * (self_attn ? 1 : 2) * ((p->seqlen + 63) / 64);
    if (nblk > 0x7fffffff) return DIMSUM_ERR_SHAPE;
    const dim3 grid((unsigned)nblk), block(256);
    if (p->precision == 2) {
        if (!p->x1_inv_ptr || (!self_attn && !p->x2_inv_ptr) || !p->kv_bound_ptr) return DIMSUM_ERR_NULL;
        if (p->qkv_f16 && (p->qkv_batch_stride % 8 != 0 || p->qkv_token_stride % 8 != 0 || p->head_dim % 8 != 0)) return DIMSUM_ERR_STRIDE;   // 16-byte fp16 loads
        if (p->out_split3 != 0 && p->out_split3 != 2) return DIMSUM_ERR_SHAPE;
        if (p->out_split3 == 2 && (!p->out_inv_ptr || p->out_token_stride < (int64_t)(self_attn ? 1 : 2) * p->heads * p->head_dim ||
                                   p->out_batch_stride % 8 != 0 || p->out_token_stride % 8 != 0 || (p->heads * p->head_dim) % 8 != 0))
            return DIMSUM_ERR_STRIDE;
        return launch_xattn_f16(*p, s);
    }
    if (p->precision != 0 && p->precision != 1) return DIMSUM_ERR_SHAPE;
    // the operand image is written in 16-byte pieces (8 bf16): its base and both strides must keep that alignment
    if (p->out_split3 != 0 && p->out_split3 != 1 && p->out_split3 != 3) return DIMSUM_ERR_SHAPE;
    if (p->out_split3 && (p->precision != 1 || p->out_token_stride < (p->out_split3 == 3 ? 2 : 3) * (int64_t)(self_attn ? 1 : 2) * p->heads * p->head_dim ||
                          reinterpret_cast<uintptr_t>(p->out_ptr) % 16 != 0 || p->out_batch_stride % 8 != 0 || p->out_token_stride % 8 != 0))
        return DIMSUM_ERR_STRIDE;
    if (p->precision == 1) {
        // 2 query tiles per wave (128 queries per workgroup) halve the per-query staging work; short sequences keep 1
        const bool two = p->seqlen >= 128;
        const int64_t nblk2 = (int64_t)p->batch * p->heads * (self_attn ? 1 : 2) * ((p->seqlen + 127) / 128);
        const dim3 grid2((unsigned)nblk2);
#define DIMSUM_XSPLIT(HDV)                                                                                       \
        if (two) hipLaunchKernelGGL((xattn_fusion_fwd_split_kernel<HDV, 2>), grid2, block, 0, s, *p);          \
        else hipLaunchKernelGGL((xattn_fusion_fwd_split_kernel<HDV, 1>), grid, block, 0, s, *p)
        switch (p->head_dim) {
            case 24: DIMSUM_XSPLIT(24); break;
            case 32: DIMSUM_XSPLIT(32); break;
            case 48: DIMSUM_XSPLIT(48); break;
            case 64: DIMSUM_XSPLIT(64); break;
            case 72: DIMSUM_XSPLIT(72); break;
            default: return DIMSUM_ERR_SHAPE;
        }
#undef DIMSUM_XSPLIT
        return launch_status();
    }
    switch (p->head_dim) {
        case 24: hipLaunchKernelGGL(xattn_fusion_fwd_kernel<24>, grid, block, 0, s, *p); break;
        case 32: hipLaunchKernelGGL(xattn_fusion_fwd_kernel<32>, grid, block, 0, s, *p); break;
        case 48: hipLaunchKernelGGL(xattn_fusion_fwd_kernel<48>, grid, block, 0, s, *p); break;
        case 64: hipLaunchKernelGGL(xattn_fusion_fwd_kernel<64>, grid, block, 0, s, *p); break;
        case 72: hipLaunchKernelGGL(xattn_fusion_fwd_kernel<72>, grid, block, 0, s, *p); break;
        default: return DIMSUM_ERR_SHAPE;
    }
    return launch_status();
}
