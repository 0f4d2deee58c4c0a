// ssm_scan_fwd_kernel.hpp -- selective scan (Mamba S6) forward for gfx950: the 64-channels-per-wave kernel and its launcher.
// Instantiated per I/O dtype in ssm_scan_fwd_{f32,f16,bf16}.hip (separate translation units: the scheduled inner
// block makes each instantiation slow to compile); the state-split kernels live in ssm_scan_fwd_split.hpp /
// ssm_scan_fwd_split_{f32,f16,bf16}.hip; checks, kernel choice and the C entry point in ssm_scan_fwd.hip.
//
// Replaces selective_scan_cuda.fwd (mamba/csrc/selective_scan/selective_scan.cpp:226-336; kernel
// selective_scan_fwd_kernel.cuh:67-303). Math per (batch b, channel d) row:
//     dt_t = softplus(delta_t + delta_bias_d)            a_t[n] = exp(dt_t * A[d,n])
//     h_t[n] = a_t[n] h_{t-1}[n] + dt_t B_t[n] u_t       out_t = sum_n C_t[n] h_t[n] + D_d u_t
//     out_z_t = out_t * silu(z_t)                        x[chunk] = (prod a, h) at each 2048 boundary
//
// MI355X design (not the reference's one-row-per-block parallel scan):
//   * lane = channel. One wave64 owns 64 channels of ONE batch element and walks the sequence sequentially with the
//     dstate states of its channel in registers: 5 VALU ops per (t, n) -- the minimum -- instead of the ~3x of a
//     cross-lane parallel scan, and no cross-lane traffic at all.
//   * B_t[n], C_t[n] depend on (batch, group, n, t) only, i.e. they are WAVE-UNIFORM here: staged once per wave and tile
//     in LDS as [n][t] and read back as broadcast ds_read_b128 (4 time steps of one state per read), software-pipelined
//     two states ahead of the FMAs. The reference re-reads the 32 KB (N, L) B/C tile per channel through L2.
//     (Scalar loads were tried: out-of-order SMEM return forces lgkmcnt(0) after every pair -- 0.70 ms, rejected.)
//   * u, delta are streamed HBM -> registers (coalesced 16 B/lane along L: a 64x32 fp32 tile is 64 full 128-B
//     lines) -> LDS transposing tile whose 16-byte slots are XOR-swizzled per row (no padding): conflict-free both as
//     ds_write_b128 in the load layout and as ds_read_b128 in the lane = channel layout. The next tile's global loads are
//     issued before the current tile is computed, so they fly under its ~7k cycles of VALU work (register-staged double
//     buffering; LDS single-buffered: 2 x 8 KB + 2 x 2 KB of B/C = 20 KB per wave -> 8 waves per CU).
//   * out is written back into the u tile in place, re-read in the coalesced layout, gated with silu(z) there
//     (z never goes through LDS) and stored with 16 B/lane.
//   * blockIdx -> tile map keeps the 16 waves of one batch element on one XCD so B/C are fetched into one L2 only.
//
//   * training callers pass ckpt_ptr: the state before every 8th step is stored lane-contiguously (one activation
//     tensor of extra writes) so that the backward (ssm_scan_bwd.hip) needs no sweep of its own to rebuild them.
//
// Algorithmic HBM bytes: 5*B*D*L*s + 2*B*G*N*L*s + B*D*ceil(L/2048)*2N*4 (+ A, D, bias).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace dimsum {

// Wave-uniform 4-element load. Going through the constant address space tells the compiler the data is invariant for
// the kernel's lifetime, so a uniform address selects SMEM (s_load_dwordx4 / x2) instead of a broadcast vector load:
// the values land in SGPRs and feed the FMAs as their scalar operand.
template <typename T> struct UniformLd;
template <> struct UniformLd<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld(const float *p) {
        const v4 r = *(const __attribute__((address_space(4))) v4 *)(p);
        return {{r.x, r.y, r.z, r.w}};
    }
    static __device__ __forceinline__ float ld1(const float *p) { return *(const __attribute__((address_space(4))) float *)(p); }
};
template <> struct UniformLd<__half> {
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 ld(const __half *p) {
        const v2 r = *(const __attribute__((address_space(4))) v2 *)(p);
        Raw4<__half> w; w.r.x = r.x; w.r.y = r.y;
        return widen(w);
    }
    static __device__ __forceinline__ float ld1(const __half *p) {
        const unsigned short r = *(const __attribute__((address_space(4))) unsigned short *)(p);
        return __half2float(__ushort_as_half(r));
    }
};
template <> struct UniformLd<__hip_bfloat16> {
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 ld(const __hip_bfloat16 *p) {
        const v2 r = *(const __attribute__((address_space(4))) v2 *)(p);
        Raw4<__hip_bfloat16> w; w.r.x = r.x; w.r.y = r.y;
        return widen(w);
    }
    static __device__ __forceinline__ float ld1(const __hip_bfloat16 *p) {
        const unsigned short r = *(const __attribute__((address_space(4))) unsigned short *)(p);
        return __uint_as_float((unsigned)r << 16);
    }
};

constexpr int kTC = 32;               // time steps per LDS tile: 128-B row segments, 20 KB of LDS per wave = 2 waves per SIMD.
                                      // (16: 64-B segments, 10 KB, 143 VGPRs = 3 waves per SIMD, was measured SLOWER at the config-2
                                      // shape, 0.40-0.43 ms vs 0.36 ms: twice the per-tile work outweighs the extra occupancy.)
constexpr int kScanWaves = 2;         // waves per SIMD the register allocation aims at
constexpr int kLdsStride = kTC;       // dwords per tile row (unpadded; 16-byte slots are XOR-swizzled instead)
constexpr int kLPR = kTC / 4;         // lanes per tile row in the coalesced load layout (16 B each)
constexpr int kRPP = kWave / kLPR;    // rows per load piece
constexpr int kNP = kWave / kRPP;     // pieces per 64-row tile

// LDS image of a 64-row x 32-column fp32 tile: row r keeps its eight 16-byte slots XOR-permuted by (r >> 1) & 7.
// ds_write_b128 in load layout (kLPR lanes = one row) and ds_read_b128 in lane = row layout are both bank-conflict
// free, with no padding.
__device__ __forceinline__ int tile_off(int row, int col4) {
    return row * kLdsStride + ((col4 ^ ((row >> 1) & 7)) << 2);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// base (wave-uniform, SGPR pair) + 32-bit BYTE offset held in one VGPR: selects the saddr + voffset addressing form.
template <typename T> __device__ __forceinline__ const T *at(const T *base, unsigned elem_off) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (unsigned)(elem_off * (unsigned)sizeof(T)));
}
template <typename T> __device__ __forceinline__ T *at(T *base, unsigned elem_off) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (unsigned)(elem_off * (unsigned)sizeof(T)));
}

// out_z either as a `T` tensor or (planes != nullptr) as its split-bf16 pair: hi plane at `planes`, lo plane lo_off elements behind it.
// Adjacent lanes hold adjacent 4-step groups of one row (all three kernels' epilogue layouts), so a lane pair trades halves (one quad_perm
// DPP each way) and the even lane stores 8 steps of hi, the odd lane 8 steps of lo: ONE 16-byte store per lane like the fp32 form (two
// 8-byte stores per lane cost the 64-channel kernel 12 % -- store issue, not bytes). seqlen % 8 == 0: a pair is live or dead together.
template <typename T> __device__ __forceinline__ void st4_out_z(T *base, unsigned short *planes, int64_t lo_off, unsigned elem_off, const f32x4 &y) {
    if (planes) {
        unsigned h0, l0, h1, l1;
        split2(y.v[0], y.v[1], h0, l0);
        split2(y.v[2], y.v[3], h1, l1);
        const bool odd = (threadIdx.x & 1) != 0;
        const unsigned r0 = __float_as_uint(dpp_mov<0xB1>(__uint_as_float(odd ? h0 : l0)));      // quad_perm [1, 0, 3, 2]: the neighbour's word
        const unsigned r1 = __float_as_uint(dpp_mov<0xB1>(__uint_as_float(odd ? h1 : l1)));
        const uint4 v = odd ? make_uint4(r0, r1, l0, l1) : make_uint4(h0, h1, r0, r1);
        unsigned short *dst = odd ? planes + lo_off : planes;
        *reinterpret_cast<uint4 *>(at(dst, odd ? elem_off - 4u : elem_off)) = v;
    } else {
        st4<T>(at(base, elem_off), y);
    }
}

// kVec : every row base is 4-element aligned and L % 4 == 0 -> 16-byte (fp32) vector I/O, register-staged prefetch.
// kFull: dim/n_groups % 64 == 0 -> all 64 lanes own a live channel, no row masks anywhere (needs kVec).
// kDt  : fused dt_proj (include/dimsum_hip.h, dt_w_ptr): the tile's delta = x_dbl[:, :R] W_dt^T is formed on the matrix cores instead of being
//        read from HBM. Transposed product D[step][channel] (A = 16 steps x 32 r of x_dbl, B = 32 r x 16 channels of W_dt^T): a lane ends up with
//        4 consecutive STEPS of one channel = one 16-byte slot of the delta tile's row. Both operands as bf16 hi / lo pairs, three
//        v_mfma_f32_16x16x32_bf16 per output tile (hi.hi + hi.lo + lo.hi: fp32-class, what the library's GEMM spends on it under
//        allow_tf32); 2 x 4 output tiles = 24 MFMAs per 32-step tile against ~7k cycles of VALU work, issued while the tile is staged (the
//        matrix pipe is otherwise idle and the SIMD's other wave keeps the VALU busy). The x_dbl rows of the next tile are requested a
//        tile ahead (16 VGPRs instead of the 32 of the delta pieces).
typedef __bf16 scan_bf16x8 __attribute__((ext_vector_type(8)));
typedef float scan_f4 __attribute__((ext_vector_type(4)));
struct alignas(16) ScanU4 { unsigned w[4]; };

// kZ16 : out_z leaves as BLOCK-SCALED fp16 (include/dimsum_hip.h, out_z_f16): the wave's 64 channels x 32 steps of a tile share one power-of-two
//        scale from the tile's own maximum (the values are all in registers in the epilogue; one wave_allmax), written as fp16(out_z 2^s) with
//        2^-s in a (tokens / 32, channels / 64) table. out_proj then multiplies ONE fp16 product per element (dimsum_gemm_tn, a_rebase_ptr: the
//        GEMM puts a 32-token group on one scale as it reads the blocks) instead of the library's three bf16 products over fp32 operands, and
//        the scan writes half the bytes. A lane pair trades halves (one quad_perm DPP each way) so that every lane issues ONE 16-byte store
//        for two pieces. The reference feeds out_proj under TF32 (selective_scan_interface.py:954-981 under train.py:20-21): 10-bit mantissas.
template <typename T, int kN, bool kHasZ, bool kVec, bool kFull, bool kCkpt = false, bool kDt = false, bool kZ16 = false>
__global__ __launch_bounds__(kWave, kScanWaves) void ssm_scan_fwd_kernel(const ssm_args_t p) {
    static_assert(!kFull || kVec, "kFull implies kVec");
    static_assert(!kZ16 || (kFull && kHasZ && !kCkpt && std::is_same<T, float>::value), "the fp16 out_z rides on the full fp32 inference path");
    static_assert(!kDt || (kFull && kHasZ && !kCkpt && std::is_same<T, float>::value), "the fused dt_proj rides on the full fp32 inference path");
    __shared__ __attribute__((aligned(16))) float tileU[kWave * kLdsStride];
    __shared__ __attribute__((aligned(16))) float tileD[kWave * kLdsStride];
    __shared__ __attribute__((aligned(16))) float tileB[kN * kTC];   // [n][t] of the current 32 steps (wave-uniform data,
    __shared__ __attribute__((aligned(16))) float tileC[kN * kTC];   //  read back as broadcast ds_read_b128)

    const int lane = threadIdx.x;
    const int L = p.seqlen;
    const int dpg = p.dim / p.n_groups;                    // channels per B/C group
    const int tiles_per_group = (dpg + kWave - 1) / kWave;
    const int tiles_per_batch = p.n_groups * tiles_per_group;
    // XCD-aware remap: consecutive tile ids (same batch element) land on the same XCD (blockIdx % 8).
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int b = wg / tiles_per_batch;
    const int rem = wg - b * tiles_per_batch;
    const int g = rem / tiles_per_group;
    const int d0 = g * dpg + (rem - g * tiles_per_group) * kWave;
    const int nd = kFull ? kWave : min(kWave, (g + 1) * dpg - d0);   // live channels in this tile
    const int d = d0 + (kFull ? lane : min(lane, nd - 1));           // dead lanes shadow the last live channel

    // Wave-uniform tile bases. In-tile offsets are 32-bit (the host checks 64 * d_stride + L < 2^31).
    const T *u_base = reinterpret_cast<const T *>(p.u_ptr) + (int64_t)b * p.u_batch_stride + (int64_t)d0 * p.u_d_stride;
    const T *dl_base = kDt ? nullptr : reinterpret_cast<const T *>(p.delta_ptr) + (int64_t)b * p.delta_batch_stride + (int64_t)d0 * p.delta_d_stride;
    const T *z_base = kHasZ ? reinterpret_cast<const T *>(p.z_ptr) + (int64_t)b * p.z_batch_stride + (int64_t)d0 * p.z_d_stride : nullptr;
    T *out_base = p.out_ptr ? reinterpret_cast<T *>(p.out_ptr) + (int64_t)b * p.out_batch_stride + (int64_t)d0 * p.out_d_stride : nullptr;
    T *oz_base = kHasZ ? reinterpret_cast<T *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride : nullptr;
    unsigned short *oz_planes = (kHasZ && p.out_z_lo_offset) ? reinterpret_cast<unsigned short *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride : nullptr;
    const int u_ds = (int)p.u_d_stride, dl_ds = (int)p.delta_d_stride, z_ds = (int)p.z_d_stride;
    const int out_ds = (int)p.out_d_stride, oz_ds = (int)p.out_z_d_stride;
    // wave-uniform B/C rows of this (batch, group)
    const T *Bp = reinterpret_cast<const T *>(p.B_ptr) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T *Cp = reinterpret_cast<const T *>(p.C_ptr) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const int Bns = (int)p.B_dstate_stride, Cns = (int)p.C_dstate_stride;

    // per-channel constants. A is pre-multiplied by log2(e) so that exp() is a bare v_exp_f32
    // (same trick as selective_scan_fwd_kernel.cuh:169-171).
    float A2[kN], h[kN];
    const float *Ap = reinterpret_cast<const float *>(p.A_ptr) + (int64_t)d * p.A_d_stride;
#pragma unroll
    for (int n = 0; n < kN; ++n) { A2[n] = Ap[n * p.A_dstate_stride] * kLog2e; h[n] = 0.f; }
    const float Dval = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d] : 0.f;
    const float bias = p.delta_bias_ptr ? reinterpret_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
    const bool softplus = p.delta_softplus != 0;
    const bool has_out = out_base != nullptr;
    float sum_dt = 0.f;   // prod_t a_t[n] = exp2(A2[n] * sum_t dt_t)
    // optional saved states for the backward: ckpt[b][t/8][n][d] = h_n before step t, t % 8 == 0
    // (kCkpt is a template parameter: the inference variant carries none of this)
    float *ck_base = (kCkpt && p.ckpt_ptr && (kFull || lane < nd))
                         ? reinterpret_cast<float *>(p.ckpt_ptr) + (int64_t)b * ((L + 7) / 8) * kN * p.dim + d : nullptr;

    const int n_tiles = (L + kTC - 1) / kTC;
    // load layout: piece i of the tile, lane -> (row = i*kRPP + lane/kLPR, 4 columns at (lane%kLPR)*4)
    const int lrow = lane / kLPR, lc4 = lane & (kLPR - 1), lcol = lc4 * 4;

    constexpr int kBCPieces = (kN * kLPR + kWave - 1) / kWave;   // 16-byte pieces per lane of a [kN][kTC] tile
    Raw4<T> ru[kNP], rd[kNP], rz[kNP], rb[kBCPieces], rc[kBCPieces];
    // Branch-free tile loads: rows beyond nd are clamped to the last live row, columns beyond L to the last
    // 4-column group (L % 4 == 0 on this path); the duplicates are never stored.
    // Address = wave-uniform (base + i * 8 * stride) + one per-lane 32-bit offset per tensor, so the 8 pieces of a tile
    // share a single VGPR offset (saddr + voffset addressing) instead of 8 precomputed per-lane addresses.
    auto col_of = [&](int t0) { return min(t0 + lcol, L - 4); };
    auto piece = [&](const T *base, int ds, int i, int col) -> const T * {   // address of piece i of a tile
        if constexpr (kFull) return at(base + i * kRPP * ds, (unsigned)(lrow * ds + col));
        else return at(base, (unsigned)(min(i * kRPP + lrow, nd - 1) * ds + col));   // clamped row: never negative
    };
    // ---- fused dt_proj state: W_dt fragments of this wave's 4 channel blocks (B operand: lane -> channel lane & 15, r = 8 (lane >> 4) .. + 7),
    //      the next tile's x_dbl rows (A operand: lane -> step lane & 15, the same 8 r) and the delta accumulators [step block][channel block]
    ScanU4 wh[kDt ? 4 : 1], wl[kDt ? 4 : 1];
    float xr[2][8];            // [step block][r = 8 fkg + e] of step lane & 15
    scan_f4 dacc[2][4];
    const int fq = lane & 15, fkg = lane >> 4;
    const float *dtx = nullptr;
    int dtx_rs = 0;
    bool r_lo = false, r_hi = false;
    auto split8 = [](const float4 &a, const float4 &c, ScanU4 &hi, ScanU4 &lo) {
        split2(a.x, a.y, hi.w[0], lo.w[0]);
        split2(a.z, a.w, hi.w[1], lo.w[1]);
        split2(c.x, c.y, hi.w[2], lo.w[2]);
        split2(c.z, c.w, hi.w[3], lo.w[3]);
    };
    if constexpr (kDt) {
        r_lo = 8 * fkg < p.dt_rank; r_hi = 8 * fkg + 4 < p.dt_rank;          // (dt_rank % 4 == 0: r beyond it are zeros on both sides)
        dtx = reinterpret_cast<const float *>(p.dt_x_ptr) + (int64_t)(8 * fkg) * p.dt_x_row_stride + (int64_t)b * L;    // row r = 8 fkg of this batch element
        dtx_rs = (int)p.dt_x_row_stride;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const float *w = reinterpret_cast<const float *>(p.dt_w_ptr) + (int64_t)(d0 + 16 * nb + fq) * p.dt_w_row_stride + 8 * fkg;
            split8(r_lo ? *reinterpret_cast<const float4 *>(w) : zero, r_hi ? *reinterpret_cast<const float4 *>(w + 4) : zero, wh[nb], wl[nb]);
        }
    }
    auto issue_x = [&](int t0) {                   // x_dbl^T (r-major, as the inference x_proj writes it) of the tile at t0: 16 consecutive steps per
#pragma unroll                                     // 16 lanes and r row (steps beyond L: the last one; they are never taken)
        for (int mt = 0; mt < 2; ++mt) {
            const float *x = dtx + min(t0 + 16 * mt + fq, L - 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[mt][e] = (e < 4 ? r_lo : r_hi) ? x[(unsigned)(e * dtx_rs)] : 0.f;
        }
    };
    auto mma_delta = [&]() {                       // dacc = the delta tile of the rows in xr
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            ScanU4 ah, al;
            split8(make_float4(xr[mt][0], xr[mt][1], xr[mt][2], xr[mt][3]), make_float4(xr[mt][4], xr[mt][5], xr[mt][6], xr[mt][7]), ah, al);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                scan_f4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(scan_bf16x8, al), __builtin_bit_cast(scan_bf16x8, wh[nb]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(scan_bf16x8, ah), __builtin_bit_cast(scan_bf16x8, wl[nb]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(scan_bf16x8, ah), __builtin_bit_cast(scan_bf16x8, wh[nb]), acc, 0, 0, 0);
                dacc[mt][nb] = acc;
            }
        }
    };
    auto store_delta = [&]() {                     // lane: channel 16 nb + fq, steps 16 mt + 4 fkg .. + 3 = 16-byte slot 4 mt + fkg of its row
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                *reinterpret_cast<scan_f4 *>(&tileD[tile_off(16 * nb + fq, 4 * mt + fkg)]) = dacc[mt][nb];
    };
    auto issue_loads = [&](int t0) {
        const int col = col_of(t0);
#pragma unroll
        for (int i = 0; i < kNP; ++i) {
            ru[i] = ld4<T>(piece(u_base, u_ds, i, col));
            if constexpr (!kDt) rd[i] = ld4<T>(piece(dl_base, dl_ds, i, col));
        }
        if constexpr (kDt) issue_x(t0);
#pragma unroll
        for (int i = 0; i < kBCPieces; ++i) {
            const int n = min(i * kRPP + lrow, kN - 1);
            rb[i] = ld4<T>(at(Bp, (unsigned)(n * Bns + col)));
            rc[i] = ld4<T>(at(Cp, (unsigned)(n * Cns + col)));
        }
    };

    if constexpr (kVec) issue_loads(0);

#pragma unroll 1
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kTC;
        // ---- stage the tile into LDS (transposing layout) ------------------------------------------------------
        if constexpr (kVec) {
#pragma unroll
            for (int i = 0; i < kNP; ++i) {
                const int row = i * kRPP + lrow;
                *reinterpret_cast<f32x4 *>(&tileU[tile_off(row, lc4)]) = widen(ru[i]);
                if constexpr (!kDt) *reinterpret_cast<f32x4 *>(&tileD[tile_off(row, lc4)]) = widen(rd[i]);
            }
            if constexpr (kDt) {                   // this tile's delta from its x_dbl rows (requested a tile ahead), then the next tile's rows
                mma_delta();
                store_delta();
            }
#pragma unroll
            for (int i = 0; i < kBCPieces; ++i) {
                const int n = i * kRPP + lrow;
                if (kN * kLPR % kWave == 0 || n < kN) {
                    *reinterpret_cast<f32x4 *>(&tileB[n * kTC + lcol]) = widen(rb[i]);
                    *reinterpret_cast<f32x4 *>(&tileC[n * kTC + lcol]) = widen(rc[i]);
                }
            }
            if (tile + 1 < n_tiles) issue_loads(t0 + kTC);   // flies under the compute below
            if constexpr (kHasZ) {
                const int col = col_of(t0);
#pragma unroll
                for (int i = 0; i < kNP; ++i) rz[i] = ld4<T>(piece(z_base, z_ds, i, col));
            }
        } else {
            // generic path (unaligned rows or L % 4 != 0): element-wise, still coalesced along L
            for (int i = 0; i < kTC; ++i) {
                const int idx = i * kWave + lane, row = idx / kTC, col = idx & (kTC - 1);
                const bool ok = row < nd && t0 + col < L;
                tileU[tile_off(row, col >> 2) + (col & 3)] = ok ? to_f32<T>(u_base[(unsigned)(row * u_ds + t0 + col)]) : 0.f;
                tileD[tile_off(row, col >> 2) + (col & 3)] = ok ? to_f32<T>(dl_base[(unsigned)(row * dl_ds + t0 + col)]) : 0.f;
            }
            for (int idx = lane; idx < kN * kTC; idx += kWave) {
                const int n = idx / kTC, tc = min(t0 + (idx & (kTC - 1)), L - 1);
                tileB[idx] = to_f32<T>(Bp[(unsigned)(n * Bns + tc)]);
                tileC[idx] = to_f32<T>(Cp[(unsigned)(n * Cns + tc)]);
            }
        }

        // ---- 32 sequential steps, 4 at a time ------------------------------------------------------------------
        // 5 VALU ops per (t, n): mul, v_exp_f32, mul, fma, fma. Measured issue costs on gfx950 (tools/ubench/valu_rates):
        // plain fp32 op 2 cycles per wave64, v_exp_f32 / v_rcp_f32 8, v_pk_*_f32 4 (= two plain ops, so packing buys
        // nothing and its operand shuffles cost extra: the file is built with -fno-slp-vectorize).
#pragma unroll 1
        for (int j = 0; j < kTC / 4; ++j) {
            const int tj = t0 + j * 4;
            if (tj >= L) break;
            if (kCkpt && ck_base && (j & 1) == 0) {
                float *ck = ck_base + (int64_t)(tj >> 3) * kN * p.dim;
#pragma unroll
                for (int n = 0; n < kN; ++n) ck[(int64_t)n * p.dim] = h[n];
            }
            const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tileU[tile_off(lane, j)]);
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tileD[tile_off(lane, j)]);
            // software-pipelined broadcast reads: B/C of states n+1 .. n+kPD are in flight while state n is computed (2 ahead
            // measured 0.306 ms in situ against 0.308-0.321 with 1)
            constexpr int kPD = 2 < kN ? 2 : kN - 1;
            f32x4 bq_pipe[kPD], cq_pipe[kPD];
#pragma unroll
            for (int i = 0; i < kPD; ++i) {
                bq_pipe[i] = *reinterpret_cast<const f32x4 *>(&tileB[i * kTC + j * 4]);
                cq_pipe[i] = *reinterpret_cast<const f32x4 *>(&tileC[i * kTC + j * 4]);
            }
            float dt[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v = softplus_if(d4.v[s] + bias, softplus);
                const bool live = kVec || (tj + s < L);       // dead steps: a = 1, b = 0 -> state untouched
                dt[s] = live ? v : 0.f;
                sum_dt += dt[s];
            }
            float du[4], y[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { du[s] = dt[s] * u4.v[s]; y[s] = Dval * u4.v[s]; }
            __builtin_amdgcn_sched_barrier(0);   // region below = exactly 2*(kN-kPD) ds_read + 20*kN VALU
#pragma unroll
            for (int n = 0; n < kN; ++n) {
                const f32x4 bq = bq_pipe[n % kPD], cq = cq_pipe[n % kPD];
                if (n + kPD < kN) {
                    bq_pipe[n % kPD] = *reinterpret_cast<const f32x4 *>(&tileB[(n + kPD) * kTC + j * 4]);
                    cq_pipe[n % kPD] = *reinterpret_cast<const f32x4 *>(&tileC[(n + kPD) * kTC + j * 4]);
                }
                float hn = h[n];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    hn = fmaf(fast_exp2(dt[s] * A2[n]), hn, bq.v[s] * du[s]);
                    y[s] = fmaf(hn, cq.v[s], y[s]);
                }
                h[n] = hn;
                // keep the issue order: [2 ds_read for n+kPD] then [20 VALU of n]
                if (n + kPD < kN) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 20, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 y4 = {{y[0], y[1], y[2], y[3]}};
            *reinterpret_cast<f32x4 *>(&tileU[tile_off(lane, j)]) = y4;
        }

        // ---- chunk-state store at every 2048 boundary and at the end (selective_scan_fwd_kernel.cuh:251-254) ---
        const int t_end = min(t0 + kTC, L);
        if (p.x_ptr && ((t_end & 2047) == 0 || t_end == L) && (kFull || lane < nd)) {
            float *xr = reinterpret_cast<float *>(p.x_ptr) + (((int64_t)b * p.dim + d) * p.n_chunks + (t_end - 1) / 2048) * (2 * kN);
#pragma unroll
            for (int n = 0; n < kN; n += 2) {
                const f32x4 v = {{fast_exp2(A2[n] * sum_dt), h[n], fast_exp2(A2[n + 1] * sum_dt), h[n + 1]}};
                *reinterpret_cast<f32x4 *>(xr + 2 * n) = v;
            }
        }

        // ---- epilogue: re-read y in the coalesced layout, gate, store -------------------------------------------
        if constexpr (kZ16) {
            f32x4 yz[kNP];
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < kNP; ++i) {
                yz[i] = *reinterpret_cast<const f32x4 *>(&tileU[tile_off(i * kRPP + lrow, lc4)]);
                const f32x4 z4 = widen(rz[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { yz[i].v[e] *= z4.v[e] * sigmoidf_fast(z4.v[e]); mx = fmaxf(mx, fabsf(yz[i].v[e])); }
            }
            float sc, inv;
            f16s_scales(wave_allmax(mx), sc, inv);
            if (lane == 0) reinterpret_cast<float *>(p.out_z_scale_ptr)[(int64_t)(((int64_t)b * L + t0) >> 5) * p.out_z_scale_ld + (d0 >> 6)] = inv;
            __half *z16 = reinterpret_cast<__half *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride + t0;
            const bool odd = (lane & 1) != 0;
#pragma unroll
            for (int i = 0; i < kNP; i += 2) {          // pieces i (even lanes) and i + 1 (odd lanes): 8 steps = 16 bytes per lane
                unsigned w[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const __half2 a = __floats2half2_rn(yz[i + k].v[0] * sc, yz[i + k].v[1] * sc), c = __floats2half2_rn(yz[i + k].v[2] * sc, yz[i + k].v[3] * sc);
                    w[k][0] = __builtin_bit_cast(unsigned, a); w[k][1] = __builtin_bit_cast(unsigned, c);
                }
                // the neighbour needs: (even lane) the odd lane's piece-i words, (odd lane) the even lane's piece-(i + 1) words
                const unsigned r0 = __float_as_uint(dpp_mov<0xB1>(__uint_as_float(odd ? w[0][0] : w[1][0])));
                const unsigned r1 = __float_as_uint(dpp_mov<0xB1>(__uint_as_float(odd ? w[0][1] : w[1][1])));
                const uint4 v = odd ? make_uint4(r0, r1, w[1][0], w[1][1]) : make_uint4(w[0][0], w[0][1], r0, r1);
                const int row = (i + (odd ? 1 : 0)) * kRPP + lrow, col = (lc4 & ~1) * 4;
                *reinterpret_cast<uint4 *>(z16 + (int64_t)row * p.out_z_d_stride + col) = v;
            }
        } else if constexpr (kVec) {
            if (t0 + lcol < L) {
#pragma unroll
                for (int i = 0; i < kNP; ++i) {
                    const int row = i * kRPP + lrow;
                    if (kFull || row < nd) {
                        f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tileU[tile_off(row, lc4)]);
                        if (has_out) st4<T>(at(out_base + i * kRPP * out_ds, (unsigned)(lrow * out_ds + t0 + lcol)), y4);
                        if constexpr (kHasZ) {
                            const f32x4 z4 = widen(rz[i]);
#pragma unroll
                            for (int s = 0; s < 4; ++s) y4.v[s] *= z4.v[s] * sigmoidf_fast(z4.v[s]);
                            st4_out_z<T>(oz_base + i * kRPP * oz_ds, oz_planes ? oz_planes + i * kRPP * oz_ds : nullptr, p.out_z_lo_offset, (unsigned)(lrow * oz_ds + t0 + lcol), y4);
                        }
                    }
                }
            }
        } else {
            for (int i = 0; i < kTC; ++i) {
                const int idx = i * kWave + lane, row = idx / kTC, col = idx & (kTC - 1);
                if (row < nd && t0 + col < L) {
                    const float yv = tileU[tile_off(row, col >> 2) + (col & 3)];
                    if (out_base) out_base[(unsigned)(row * out_ds + t0 + col)] = from_f32<T>(yv);
                    if constexpr (kHasZ) {
                        const float zv = to_f32<T>(z_base[(unsigned)(row * z_ds + t0 + col)]);
                        oz_base[(unsigned)(row * oz_ds + t0 + col)] = from_f32<T>(yv * zv * sigmoidf_fast(zv));
                    }
                }
            }
        }
    }
}

// ---- launchers of this kernel: defined here, explicitly instantiated per I/O dtype in ssm_scan_fwd_{f32,f16,bf16}.hip, called
// by the dispatch in ssm_scan_fwd.hip (the split kernels' launcher lives in ssm_scan_fwd_split.hpp the same way) -----------
template <typename T, int kN>
void ssm_scan_fwd_launch_v0(const ssm_args_t &p, hipStream_t stream, int tiles, bool vec, bool full) {
    const dim3 grid(tiles), block(kWave);
    const hipEvent_t ev0 = reinterpret_cast<hipEvent_t>(p.timing_start_event), ev1 = reinterpret_cast<hipEvent_t>(p.timing_stop_event);
#define DIMSUM_LAUNCH(HASZ, VEC, FULL)                                                                                        \
    do {                                                                                                                       \
        if (p.ckpt_ptr) DIMSUM_LAUNCH_EV((ssm_scan_fwd_kernel<T, kN, HASZ, VEC, FULL, true>), grid, block, stream, ev0, ev1, p); \
        else DIMSUM_LAUNCH_EV((ssm_scan_fwd_kernel<T, kN, HASZ, VEC, FULL, false>), grid, block, stream, ev0, ev1, p);          \
    } while (0)
    if constexpr (std::is_same<T, float>::value && kN == 16) {
        if (p.dt_w_ptr || p.out_z_f16) {       // fused dt_proj / block-scaled fp16 out_z (the caller checked: full vector path, z, no saved states)
            if (p.dt_w_ptr && p.out_z_f16) DIMSUM_LAUNCH_EV((ssm_scan_fwd_kernel<T, kN, true, true, true, false, true, true>), grid, block, stream, ev0, ev1, p);
            else if (p.dt_w_ptr) DIMSUM_LAUNCH_EV((ssm_scan_fwd_kernel<T, kN, true, true, true, false, true, false>), grid, block, stream, ev0, ev1, p);
            else DIMSUM_LAUNCH_EV((ssm_scan_fwd_kernel<T, kN, true, true, true, false, false, true>), grid, block, stream, ev0, ev1, p);
            return;
        }
    }
    if (p.z_ptr) {
        if (full) DIMSUM_LAUNCH(true, true, true);
        else if (vec) DIMSUM_LAUNCH(true, true, false);
        else DIMSUM_LAUNCH(true, false, false);
    } else {
        if (full) DIMSUM_LAUNCH(false, true, true);
        else if (vec) DIMSUM_LAUNCH(false, true, false);
        else DIMSUM_LAUNCH(false, false, false);
    }
#undef DIMSUM_LAUNCH
}

#define DIMSUM_INSTANTIATE_FWD_V0(T)                                                                                \
    template void ssm_scan_fwd_launch_v0<T, 4>(const ssm_args_t &, hipStream_t, int, bool, bool);           \
    template void ssm_scan_fwd_launch_v0<T, 8>(const ssm_args_t &, hipStream_t, int, bool, bool);           \
    template void ssm_scan_fwd_launch_v0<T, 16>(const ssm_args_t &, hipStream_t, int, bool, bool);          \
    template void ssm_scan_fwd_launch_v0<T, 32>(const ssm_args_t &, hipStream_t, int, bool, bool);

}  // namespace dimsum
