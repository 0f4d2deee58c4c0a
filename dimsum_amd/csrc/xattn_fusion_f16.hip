// xattn_fusion_f16.hip -- the cross-attention fusion core with ONE fp16 MFMA product per element (precision = 2; gfx950).
//
// The single-product companion of xattn_fusion.hip's split-bf16 kernel for the scaled-fp16 policy: the
// reference evaluates QK^T and PV under TF32 (attention_fusion.py:72-74 with train.py:20-21): 10-bit operand mantissas, fp32
// accumulation. fp16 has that mantissa; the range is handled by construction, operand by operand:
//   Q   per query row, exact: the lane that owns a query holds its whole row (4 lanes x 8 EC values): q16 = fp16(q 2^sq), and 2^-sq goes
//       into the per-lane factor c that the softmax applies anyway (p = exp2(S c - m)): free.
//   K/V per batch element, from a bound that needs no pass over K / V: the qkv GEMM's input is a scaled-fp16 image whose inverse row
//       scales bound its row maxima (max|x_t| < 2^15 inv[t]), so |k|, |v| <= max_t(2^15 inv[b, t]) * max_n sum_c |W_nc| + max|bias|.
//       fp16 keeps its full significand 2^28 below the scaled bound; the bound is loose by ~2^6 on DiM activations.
//   P   in [0, 1]: fp16 as it is (p < 6e-8 flushes: below TF32's own product error against the row's p = 1 term).
//   O   fp32 accumulator; the V scale leaves with the 1 / l normalisation.
// The output is fp32, or the scaled-fp16 operand image of the proj Linear (out_split3 == 2): |o| <= max|v| (a convex combination), so the
// row's scale comes from the same bound (both directions' sources: the row spans them), no reduction over heads.
// Same transposed formulation, LDS layout (rows 4-11 of a fragment keep their 16-byte slot pairs swapped) and workgroup shape as the
// split-bf16 kernel: S^T = K Q^T, O^T = V^T P^T, a lane works for one query per 16-query tile; 1 MFMA where that kernel issues 3, no
// hi / lo splits anywhere (they were ~half of its VALU work).
#include <type_traits>

#include "xattn_common.hpp"

namespace dimsum {

// (f16x8, as_f16x8, pack_h2: xattn_common.hpp)
constexpr int kKT16 = 64;        // keys per tile
constexpr int kQW16 = 16;        // queries per 16-query tile

// kIn16: q | k | v arrive as the scaled fp16 the qkv GEMM's F16_QKV epilogue wrote (include/dimsum_hip.h: q scaled per row, k / v per batch
// element, from the same bound and the same x_inv this kernel holds; biases included) -- half the bytes to read (the fp32 qkv tensors
// made this kernel HBM-bound: 0.94 GB in 0.22 ms at DiM-L/2, batch 256), K goes to LDS as it is, V^T by byte permutes.
template <int HD, int QT, bool kIn16>
__global__ __launch_bounds__(256, (QT == 2 && HD <= 64) ? 3 : 2) void xattn_fusion_fwd_f16_kernel(const dimsum_xattn_params_t p) {
    constexpr int EP = (HD + 31) / 32 * 32;  // reduction length of QK^T, padded with zeros to whole 32-deep chunks
    constexpr int EC = EP / 32;
    constexpr int ET = (HD + 15) / 16;       // 16-row output tiles along e
    constexpr int KS = EP + 8;               // K tile row stride in fp16 elements (16 B of padding: an odd number of 16-byte slots)
    constexpr int VS = kKT16 + 8;
    static_assert(HD % 8 == 0 && (KS / 8) % 2 == 1 && (VS / 8) % 2 == 1, "layout");
    auto flip = [](int row) { return (((row & 15) + 4) & 8); };
    __shared__ __attribute__((aligned(16))) unsigned short Kh[kKT16 * KS];
    __shared__ __attribute__((aligned(16))) unsigned short Vh[ET * 16 * VS];
    __shared__ float red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = p.seqlen, H = p.heads;
    const int qblocks = (L + 64 * QT - 1) / (64 * QT);
    int idx = xcd_group_blocks(blockIdx.x, (int)gridDim.x, qblocks);          // (xattn_common.hpp: a group's q-blocks on ONE XCD: K / V re-reads hit its L2)
    const int qb = idx % qblocks; idx /= qblocks;
    const int ndir = p.n_dirs == 1 ? 1 : 2;
    const int dir = idx % ndir; idx /= ndir;
    const int h = idx % H;
    const int b = idx / H;
    const int C = H * HD;
    const bool kv_from_1 = dir == 1 || ndir == 1;
    using In = typename std::conditional<kIn16, __half, float>::type;
    const In *qsrc = reinterpret_cast<const In *>(dir == 0 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const In *kvsrc = reinterpret_cast<const In *>(kv_from_1 ? p.qkv1_ptr : p.qkv2_ptr) + (int64_t)b * p.qkv_batch_stride + h * HD;
    const In *ksrc = kvsrc + C, *vsrc = kvsrc + 2 * C;
    const int64_t ts = p.qkv_token_stride;
    const float *qbv = reinterpret_cast<const float *>(dir == 0 ? p.bias1_ptr : p.bias2_ptr);
    const float *kvb = reinterpret_cast<const float *>(kv_from_1 ? p.bias1_ptr : p.bias2_ptr);
    const float *qbias = qbv ? qbv + h * HD : nullptr;
    const float *kbias = kvb ? kvb + C + h * HD : nullptr, *vbias = kvb ? kvb + 2 * C + h * HD : nullptr;

    // ---- the K / V (and output) scales of this batch element: max_t inv[b, t] of both qkv inputs, then the bound ----------------------
    float kv_scale, kv_inv, o_scale = 1.f, o_inv = 1.f;
    const float *bnd = reinterpret_cast<const float *>(p.kv_bound_ptr);       // {wl1_1, bmax_1, wl1_2, bmax_2}
    const float *i1 = reinterpret_cast<const float *>(p.x1_inv_ptr) + (int64_t)b * L;
    const float *i2 = ndir == 2 ? reinterpret_cast<const float *>(p.x2_inv_ptr) + (int64_t)b * L : i1;
    {
        float m1 = 0.f, m2 = 0.f;
        for (int t = tid; t < L; t += 256) { m1 = fmaxf(m1, i1[t]); m2 = fmaxf(m2, i2[t]); }
        m1 = wave_allmax(m1); m2 = wave_allmax(m2);
        if (lane == 0) { red[wave] = m1; red[4 + wave] = m2; }
        __syncthreads();
        m1 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        m2 = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        // (factor 2: the fp16 rounding of the GEMM's operands and its fp32 accumulation, many times over)
        const float b1 = 2.0f * (32768.0f * m1 * bnd[0] + bnd[1]), b2 = ndir == 2 ? 2.0f * (32768.0f * m2 * bnd[2] + bnd[3]) : b1;
        f16s_scales(kv_from_1 ? b1 : b2, kv_scale, kv_inv);
        if (p.out_split3 == 2) f16s_scales(fmaxf(b1, b2), o_scale, o_inv);
    }

    const int qi = lane & 15, kg = lane >> 4;
    // Q^T fragments (B operand) of the wave's QT query tiles: chunk c, slots j = 0..7 <-> e = 32c + 8 kg + j; zeros beyond hd.
    // cq[t] = scale log2(e) / (2^sq 2^skv): what turns the scaled accumulator into the log2-domain score of this lane's query
    int q_tok[QT];
    u4v qh[QT][EC];
    float cq[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        q_tok[t] = qb * (64 * QT) + (wave * QT + t) * kQW16 + qi;
        const int q_ld = min(q_tok[t], L - 1);
        if constexpr (kIn16) {
            // already fp16(q 2^sq) with sq from the row's bound: the same expression as in the GEMM epilogue that wrote it
            float qs, qinv;
            const float xinv = (dir == 0 ? i1 : i2)[q_ld];
            f16s_scales(2.0f * (32768.0f * xinv * bnd[dir == 0 ? 0 : 2] + bnd[dir == 0 ? 1 : 3]), qs, qinv);
            cq[t] = fmaxf(p.scale * kLog2e * qinv * kv_inv, 1.17549435e-38f);
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const int e0 = 32 * c + 8 * kg;
                qh[t][c] = u4v{{0u, 0u, 0u, 0u}};
                if (e0 < HD) qh[t][c] = *reinterpret_cast<const u4v *>(qsrc + (int64_t)q_ld * ts + e0);
            }
            continue;
        }
        float v[EC][8], qm = 0.f;
#pragma unroll
        for (int c = 0; c < EC; ++c)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e0 = 32 * c + 8 * kg + 4 * half;
                float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e0 < HD) {
                    tq = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(qsrc) + (int64_t)q_ld * ts + e0);
                    if (qbias) { const float4 bq = *reinterpret_cast<const float4 *>(qbias + e0); tq.x += bq.x; tq.y += bq.y; tq.z += bq.z; tq.w += bq.w; }
                }
                v[c][4 * half + 0] = tq.x; v[c][4 * half + 1] = tq.y; v[c][4 * half + 2] = tq.z; v[c][4 * half + 3] = tq.w;
                qm = fmaxf(fmaxf(qm, fmaxf(fabsf(tq.x), fabsf(tq.y))), fmaxf(fabsf(tq.z), fabsf(tq.w)));
            }
        float qs, qinv;
        f16s_scales(quad_max(qm), qs, qinv);
        // (never 0: a masked score is -inf, and -inf * 0 would be NaN for the whole row -- an all-zero query row with a tiny kv_inv underflows here)
        cq[t] = fmaxf(p.scale * kLog2e * qinv * kv_inv, 1.17549435e-38f);
#pragma unroll
        for (int c = 0; c < EC; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) qh[t][c].w[i] = pack_h2(v[c][2 * i] * qs, v[c][2 * i + 1] * qs);
    }
    // zero the padding that is never rewritten: K columns e in [hd, EP), V^T rows e in [hd, ET*16)
    if constexpr (EP > HD) {
        for (int i = tid; i < kKT16 * (EP - HD); i += 256) { const int key = i / (EP - HD), e = (HD + i % (EP - HD)) ^ flip(key); Kh[key * KS + e] = 0; }
    }
    if constexpr (ET * 16 > HD) {
        for (int i = tid; i < (ET * 16 - HD) * kKT16; i += 256) { const int e = HD + i / kKT16, k = (i % kKT16) ^ flip(e); Vh[e * VS + k] = 0; }
    }

    f4 o[QT][ET];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -1e30f; l_run[t] = 0.f;
#pragma unroll
        for (int e = 0; e < ET; ++e) o[t][e] = f4{0.f, 0.f, 0.f, 0.f};
    }

    constexpr int kItems = (kKT16 / 2) * (HD / 4), kIters = (kItems + 255) / 256;
    constexpr int kItems16 = (kKT16 / 2) * (HD / 8), kIters16 = (kItems16 + 255) / 256;
    // (Round 6, measured and NOT kept: requesting the next key tile's K / V pieces before the current tile is computed -- register-staged double
    // buffering, +16 VGPRs per staging iteration -- changed nothing at head_dim 64 (0.170 ms either way) and cost head_dim 72 its third wave per
    // SIMD (0.63 -> 0.66 ms): with 3 workgroups per CU the loads are hidden already. Counters (profiles/r06_xattn_f16_fwd_pmc.txt): VALU busy 0.60
    // + MFMA busy 0.18 with little overlap -- the softmax's v_exp_f32 (quarter rate) and its 100 VALU instructions per 16 x 64 score tile bound
    // the kernel, not memory: the XCD-aware block order above cut its HBM traffic from 0.81 to 0.54 GB per launch at the same duration.)
    for (int k0 = 0; k0 < L; k0 += kKT16) {
        __syncthreads();
        if constexpr (kIn16) {
            // ---- K [key][e] copied as it is, V^T [e][pi(key)] by byte permutes; one thread = 2 keys x 8 e (16 bytes per key and tensor) -------
#pragma unroll
            for (int it = 0; it < kIters16; ++it) {
                const int i = tid + it * 256;
                if (kItems16 % 256 != 0 && i >= kItems16) continue;
                const int kp = i / (HD / 8), e8 = i - kp * (HD / 8), key = 2 * kp;
                const int tok0 = min(k0 + key, L - 1), tok1 = min(k0 + key + 1, L - 1);
                const u4v ka = *reinterpret_cast<const u4v *>(ksrc + (int64_t)tok0 * ts + e8 * 8), kb = *reinterpret_cast<const u4v *>(ksrc + (int64_t)tok1 * ts + e8 * 8);
                const u4v va = *reinterpret_cast<const u4v *>(vsrc + (int64_t)tok0 * ts + e8 * 8), vb = *reinterpret_cast<const u4v *>(vsrc + (int64_t)tok1 * ts + e8 * 8);
                const int ke = (e8 * 8) ^ flip(key);
                *reinterpret_cast<u4v *>(&Kh[key * KS + ke]) = ka;
                *reinterpret_cast<u4v *>(&Kh[(key + 1) * KS + ke]) = kb;
                const int kap = key & 31, pos = (key & ~31) + 8 * ((kap & 15) >> 2) + (kap & 3) + 4 * (kap >> 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {          // e = 8 e8 + 2 j, + 1: (key, key + 1) pairs out of the low / high halves of word j
                    *reinterpret_cast<unsigned *>(&Vh[(e8 * 8 + 2 * j) * VS + (pos ^ flip(e8 * 8 + 2 * j))]) = __builtin_amdgcn_perm(vb.w[j], va.w[j], 0x05040100u);
                    *reinterpret_cast<unsigned *>(&Vh[(e8 * 8 + 2 * j + 1) * VS + (pos ^ flip(e8 * 8 + 2 * j + 1))]) = __builtin_amdgcn_perm(vb.w[j], va.w[j], 0x07060302u);
                }
            }
        } else {
        // ---- stage K [key][e] and V^T [e][pi(key)] as fp16(x * kv_scale) for keys k0 .. k0+63; one thread = 2 keys x 4 e -----------------
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int i = tid + it * 256;
            if (kItems % 256 != 0 && i >= kItems) continue;
            const int kp = i / (HD / 4), e4 = i - kp * (HD / 4), key = 2 * kp;
            const int tok0 = min(k0 + key, L - 1), tok1 = min(k0 + key + 1, L - 1);
            const float *kf = reinterpret_cast<const float *>(ksrc), *vf = reinterpret_cast<const float *>(vsrc);
            float4 ka = *reinterpret_cast<const float4 *>(kf + (int64_t)tok0 * ts + e4 * 4), kb = *reinterpret_cast<const float4 *>(kf + (int64_t)tok1 * ts + e4 * 4);
            float4 va = *reinterpret_cast<const float4 *>(vf + (int64_t)tok0 * ts + e4 * 4), vb = *reinterpret_cast<const float4 *>(vf + (int64_t)tok1 * ts + e4 * 4);
            if (kbias) {
                const float4 bk = *reinterpret_cast<const float4 *>(kbias + e4 * 4), bv = *reinterpret_cast<const float4 *>(vbias + e4 * 4);
                ka.x += bk.x; ka.y += bk.y; ka.z += bk.z; ka.w += bk.w; kb.x += bk.x; kb.y += bk.y; kb.z += bk.z; kb.w += bk.w;
                va.x += bv.x; va.y += bv.y; va.z += bv.z; va.w += bv.w; vb.x += bv.x; vb.y += bv.y; vb.z += bv.z; vb.w += bv.w;
            }
            const int ke = (e4 * 4) ^ flip(key);                    // key even: key and key + 1 are rows of the same kind
            *reinterpret_cast<uint2 *>(&Kh[key * KS + ke]) = make_uint2(pack_h2(ka.x * kv_scale, ka.y * kv_scale), pack_h2(ka.z * kv_scale, ka.w * kv_scale));
            *reinterpret_cast<uint2 *>(&Kh[(key + 1) * KS + ke]) = make_uint2(pack_h2(kb.x * kv_scale, kb.y * kv_scale), pack_h2(kb.z * kv_scale, kb.w * kv_scale));
            // position of key kappa inside its 32-key chunk: 8 * ((kappa & 15) >> 2) + (kappa & 3) + 4 * ((kappa >> 4) & 1)  (xattn_common.hpp, cslot)
            const int kap = key & 31, pos = (key & ~31) + 8 * ((kap & 15) >> 2) + (kap & 3) + 4 * (kap >> 4);
            const float a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<unsigned *>(&Vh[(e4 * 4 + e) * VS + (pos ^ flip(e4 * 4))]) = pack_h2(a4[e] * kv_scale, b4[e] * kv_scale);
        }
        }
        __syncthreads();

        // ---- S^T = K Q^T for the 4 key tiles of 16: one fp16 MFMA per 32-deep chunk and query tile ----------------------------------
        f4 s[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int krow = (kt * 16 + qi) * KS + ((8 * kg) ^ flip(qi));
#pragma unroll
            for (int t = 0; t < QT; ++t) s[t][kt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                const u4v kh = *reinterpret_cast<const u4v *>(&Kh[krow + 32 * c]);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(kh), as_f16x8(qh[t][c]), s[t][kt], 0, 0, 0);
            }
        }
        // ---- online softmax for this lane's queries (fp32, log2 domain: score = s * cq), then P^T as fp16 ----------------------------------
        u4v ph[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (k0 + kKT16 > L) {                      // only the last, ragged key tile needs the mask (wave-uniform branch)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + kt * 16 + kg * 4 + r >= L) s[t][kt][r] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][kt][r]);
            mx = quad_max(mx) * cq[t];                 // cq > 0: the maximum commutes with the scale
            const float m_new = fmaxf(m_run[t], mx);
            const float alpha = fast_exp2(m_run[t] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][kt][r] = fast_exp2(fmaf(s[t][kt][r], cq[t], -m_new)); rs += s[t][kt][r]; }
            rs = quad_sum(rs);
            l_run[t] = l_run[t] * alpha + rs;
            m_run[t] = m_new;
#pragma unroll
            for (int e = 0; e < ET; ++e) o[t][e] *= alpha;
#pragma unroll
            for (int c = 0; c < 2; ++c) {              // slots j < 4 = s[2c][j], j >= 4 = s[2c+1][j-4]
                ph[t][c].w[0] = pack_h2(s[t][2 * c][0], s[t][2 * c][1]);
                ph[t][c].w[1] = pack_h2(s[t][2 * c][2], s[t][2 * c][3]);
                ph[t][c].w[2] = pack_h2(s[t][2 * c + 1][0], s[t][2 * c + 1][1]);
                ph[t][c].w[3] = pack_h2(s[t][2 * c + 1][2], s[t][2 * c + 1][3]);
            }
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            const int vrow = (e * 16 + qi) * VS + ((8 * kg) ^ flip(qi));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u4v vh = *reinterpret_cast<const u4v *>(&Vh[vrow + 32 * c]);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(vh), as_f16x8(ph[t][c]), o[t][e], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const bool valid = q_tok[t] < L;
        const float inv = kv_inv / l_run[t];                   // the V scale leaves with the normalisation
        const int64_t row_off = (int64_t)b * p.out_batch_stride + (int64_t)min(q_tok[t], L - 1) * p.out_token_stride;
        if (p.out_split3 == 2) {
            // scaled-fp16 operand image of the proj Linear: fp16 rows of ndir x C (strides in fp16 elements) + one inverse scale per token.
            // The 4 lanes of a query each hold 4 of every 16 e: a 4 x 4 block transpose over v_permlane32_swap / v_permlane16_swap gives
            // lane kg the whole e-tile kg = 16 consecutive e = two 16-byte pieces
            __half *img = reinterpret_cast<__half *>(p.out_ptr) + row_off + dir * C + h * HD;
            const float sc = inv * o_scale;
            constexpr int kTr = ET >= 4 ? 4 : 0;
            if constexpr (kTr == 4) {
                float x[4][4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[e][r] = o[t][e][r] * sc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    { auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0][r]), __float_as_uint(x[2][r]), false, false); x[0][r] = __uint_as_float(q[0]); x[2][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[1][r]), __float_as_uint(x[3][r]), false, false); x[1][r] = __uint_as_float(q[0]); x[3][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[0][r]), __float_as_uint(x[1][r]), false, false); x[0][r] = __uint_as_float(q[0]); x[1][r] = __uint_as_float(q[1]); }
                    { auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[2][r]), __float_as_uint(x[3][r]), false, false); x[2][r] = __uint_as_float(q[0]); x[3][r] = __uint_as_float(q[1]); }
                }
                // lane kg: x[j][r] = e-tile kg, e = 16 kg + 4 j + r
                if (valid) {
                    __half *d0 = img + 16 * kg;
                    *reinterpret_cast<uint4 *>(d0) = make_uint4(pack_h2(x[0][0], x[0][1]), pack_h2(x[0][2], x[0][3]), pack_h2(x[1][0], x[1][1]), pack_h2(x[1][2], x[1][3]));
                    *reinterpret_cast<uint4 *>(d0 + 8) = make_uint4(pack_h2(x[2][0], x[2][1]), pack_h2(x[2][2], x[2][3]), pack_h2(x[3][0], x[3][1]), pack_h2(x[3][2], x[3][3]));
                }
            }
            if (valid) {
#pragma unroll
                for (int e = kTr; e < ET; ++e) {
                    const int e0 = e * 16 + kg * 4;
                    if (e0 < HD) *reinterpret_cast<uint2 *>(img + e0) = make_uint2(pack_h2(o[t][e][0] * sc, o[t][e][1] * sc), pack_h2(o[t][e][2] * sc, o[t][e][3] * sc));
                }
                if (dir == 0 && h == 0 && kg == 0) reinterpret_cast<float *>(p.out_inv_ptr)[(int64_t)b * L + q_tok[t]] = o_inv;
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

// called by dimsum_xattn_fusion_fwd (xattn_fusion.hip) for precision == 2
int launch_xattn_f16(const dimsum_xattn_params_t &p, hipStream_t s) {
    const bool self_attn = p.n_dirs == 1;
    const bool two = p.seqlen >= 128;
    const int64_t nblk = (int64_t)p.batch * p.heads * (self_attn ? 1 : 2) * ((p.seqlen + (two ? 127 : 63)) / (two ? 128 : 64));
    if (nblk > 0x7fffffff) return DIMSUM_ERR_SHAPE;
    const dim3 grid((unsigned)nblk), block(256);
#define DIMSUM_XF16(HDV)                                                                                \
    if (p.qkv_f16) {                                                                                    \
        if (two) hipLaunchKernelGGL((xattn_fusion_fwd_f16_kernel<HDV, 2, true>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((xattn_fusion_fwd_f16_kernel<HDV, 1, true>), grid, block, 0, s, p);     \
    } else if (two) hipLaunchKernelGGL((xattn_fusion_fwd_f16_kernel<HDV, 2, false>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((xattn_fusion_fwd_f16_kernel<HDV, 1, false>), grid, block, 0, s, p)
    switch (p.head_dim) {
        case 24: DIMSUM_XF16(24); break;
        case 32: DIMSUM_XF16(32); break;
        case 48: DIMSUM_XF16(48); break;
        case 64: DIMSUM_XF16(64); break;
        case 72: DIMSUM_XF16(72); break;
        default: return DIMSUM_ERR_SHAPE;
    }
#undef DIMSUM_XF16
    return launch_status();
}

}  // namespace dimsum
