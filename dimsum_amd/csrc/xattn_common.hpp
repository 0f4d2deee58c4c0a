// xattn_common.hpp -- operand helpers (split-bf16 pairs, fp16 images) shared by the attention forward and backward kernels (gfx950).
// A fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi); a product a.b is evaluated as hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped lo.lo term is ~2^-16 relative).
#pragma once
#include "common.hpp"

namespace dimsum {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct alignas(16) u4v { unsigned w[4]; };       // 8 bf16 = the A / B operand of one lane for a 32-deep K chunk

// split2 (hi / lo bf16 pair images of two fp32 values): common.hpp
__device__ __forceinline__ bf16x8 as_bf16x8(const u4v &v) { return __builtin_bit_cast(bf16x8, v); }

// max / sum over the 4 lanes {l, l^16, l^32, l^48} that share a query (or key) column of a 16x16 MFMA C tile, on the VALU:
// v_permlane16_swap / v_permlane32_swap exchange half of (x, x) so that op(a, b) is the pair result in every lane. The
// ds_bpermute form of __shfl_xor makes each of these steps an LDS round trip (~100+ cycles of latency, 4 per query tile
// and key tile, in a dependent chain).
__device__ __forceinline__ float quad_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// acc += a . b with a = (ah, al), b = (bh, bl): the small cross terms first
__device__ __forceinline__ f4 mfma_split(const u4v &ah, const u4v &al, const u4v &bh, const u4v &bl, f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), as_bf16x8(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(bh), acc, 0, 0, 0);
    return acc;
}

// Position of index kappa (0..31) inside a 32-deep chunk whose K-slots follow the C layout of two stacked 16-row MFMA
// results: slot 8g + j holds row 4g + j of the first tile (j < 4) or row 4g + j - 4 of the second (j >= 4). An even kappa
// and kappa + 1 land on adjacent positions.
__device__ __forceinline__ int cslot(int kappa) { return 8 * ((kappa & 15) >> 2) + (kappa & 3) + 4 * (kappa >> 4); }

// the 8 slot values of one lane (C registers of two stacked tiles) -> hi / lo B (or A) operand
__device__ __forceinline__ void split_c2(const f4 &t0, const f4 &t1, u4v &hi, u4v &lo) {
    split2(t0[0], t0[1], hi.w[0], lo.w[0]);
    split2(t0[2], t0[3], hi.w[1], lo.w[1]);
    split2(t1[0], t1[1], hi.w[2], lo.w[2]);
    split2(t1[2], t1[3], hi.w[3], lo.w[3]);
}

// ---- one lane's A / B operand of a 32-deep chunk under either carrier ------------------------------------------------------------------
// F16 = false: the split-bf16 pair (hi, lo), three products per element (mfma_split). F16 = true: ONE fp16 image and one
// v_mfma_f32_16x16x32_f16 product -- the TF32-equivalent arithmetic (10-bit mantissas, fp32 accumulation); the caller keeps the values
// inside fp16's exponent range (exact power-of-two row scales where the magnitude is not known a priori).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_f16x8(const u4v &v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return __builtin_bit_cast(unsigned, h);
}
template <bool F16> struct Frag { u4v h, l; };
template <> struct Frag<true> { u4v h; };
template <bool F16> __device__ __forceinline__ void frag_put2(Frag<F16> &f, int i, float x0, float x1) {
    if constexpr (F16) f.h.w[i] = pack_h2(x0, x1);
    else split2(x0, x1, f.h.w[i], f.l.w[i]);
}
template <bool F16> __device__ __forceinline__ Frag<F16> frag_ld(const unsigned short *hi, const unsigned short *lo, int idx) {
    Frag<F16> f;
    f.h = *reinterpret_cast<const u4v *>(&hi[idx]);
    if constexpr (!F16) f.l = *reinterpret_cast<const u4v *>(&lo[idx]);
    return f;
}
template <bool F16> __device__ __forceinline__ f4 frag_mfma(const Frag<F16> &a, const Frag<F16> &b, f4 acc) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(a.h), as_f16x8(b.h), acc, 0, 0, 0);
    else return mfma_split(a.h, a.l, b.h, b.l, acc);
}
// the 8 slot values of one lane (C registers of two stacked tiles) -> operand
template <bool F16> __device__ __forceinline__ Frag<F16> frag_c2(const f4 &t0, const f4 &t1) {
    Frag<F16> f;
    frag_put2<F16>(f, 0, t0[0], t0[1]); frag_put2<F16>(f, 1, t0[2], t0[3]);
    frag_put2<F16>(f, 2, t1[0], t1[1]); frag_put2<F16>(f, 3, t1[2], t1[3]);
    return f;
}
// staging stores into the LDS image(s): 4 consecutive elements / one pair / zero padding
template <bool F16> __device__ __forceinline__ void img_st4(unsigned short *hi, unsigned short *lo, int idx, const float4 &v) {
    if constexpr (F16) *reinterpret_cast<uint2 *>(&hi[idx]) = make_uint2(pack_h2(v.x, v.y), pack_h2(v.z, v.w));
    else {
        unsigned h0, l0, h1, l1;
        split2(v.x, v.y, h0, l0); split2(v.z, v.w, h1, l1);
        *reinterpret_cast<uint2 *>(&hi[idx]) = make_uint2(h0, h1); *reinterpret_cast<uint2 *>(&lo[idx]) = make_uint2(l0, l1);
    }
}
template <bool F16> __device__ __forceinline__ void img_st2(unsigned short *hi, unsigned short *lo, int idx, float x0, float x1) {
    if constexpr (F16) *reinterpret_cast<unsigned *>(&hi[idx]) = pack_h2(x0, x1);
    else {
        unsigned h0, l0;
        split2(x0, x1, h0, l0);
        *reinterpret_cast<unsigned *>(&hi[idx]) = h0; *reinterpret_cast<unsigned *>(&lo[idx]) = l0;
    }
}
template <bool F16> __device__ __forceinline__ void img_zero(unsigned short *hi, unsigned short *lo, int idx) {
    hi[idx] = 0;
    if constexpr (!F16) lo[idx] = 0;
}
// 2^k (exact) that brings a row of absolute maximum `amax` to [2^-5, 2^-4): the fp16 image of the scaled row keeps 11 bits down to
// 2^-10 of its maximum (gradual below), and sums of head_dim products against |v| <= 32 times the 2^8 the probabilities carry
// (kPShift) stay inside fp16. amax = 0 (or denormal) -> 2^122, finite.
__device__ __forceinline__ float row_pow2_scale(float amax) {
    const unsigned e = min((__float_as_uint(amax) >> 23) & 0xffu, 248u);
    return __uint_as_float((249u - e) << 23);
}
constexpr int kPShift = 8;       // fp16 carrier of the backward: P travels as 2^8 P (11 bits down to P = 2^-22; fp16 alone would fade below 2^-14)

// XCD-aware block order. The blocks of one (batch, head, direction) -- its `per_group` query (or key) blocks -- read the same K / V (Q / dO):
// 2 x L x hd values, 64 KB as fp16 at DiM-L/2, 288 KB at XL/2-512. The dispatcher hands block i to XCD i % 8, so consecutive block ids
// fetch that data into `per_group` different L2s (the fp16 forward at DiM-L/2: 0.81 GB from HBM per launch for 0.54 GB of distinct bytes;
// at XL/2-512, 8 blocks per group: 2.7 GB for 0.6). Block 8 j + x takes work item ((j / per_group) 8 + x) per_group + j % per_group: one
// XCD runs all blocks of a group back to back and the re-reads hit its L2. A bijection when the number of groups is a multiple of 8.
__device__ __forceinline__ int xcd_group_blocks(int idx, int nblocks, int per_group) {
    if (((nblocks / per_group) & 7) != 0 || nblocks % per_group != 0) return idx;
    const int x = idx & 7, j = idx >> 3;
    return ((j / per_group) * 8 + x) * per_group + (j % per_group);
}

}  // namespace dimsum
