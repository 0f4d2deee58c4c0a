// xattn_common.hpp -- split-bf16 helpers shared by the attention forward and backward kernels (gfx950).
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
