// gemm_nt_kernel.hpp -- C (M, N) = A (M, K) . B (N, K)^T on v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulation, with the epilogue of the
// Linear it serves fused into the accumulator registers (gfx950 only).
//
// Both operands are K-contiguous rows ("NT"): A = the activations of a Linear (tokens x features), B = its weight (out x in) --
// dimsum/mlp.py:66-70 (w12, w3), modules/mamba_simple.py in_proj / out_proj, attention_fusion.py qkv / proj. Under the reference's
// allow_tf32 policy (train.py:20-21) the operands are split-bf16 images (operand_split.hip: [hi | hi | lo] x [hi | lo | hi] over 3 K),
// under the scaled-fp16 policy plain fp16 rows: the kernel only sees 16-bit rows of length K.
//
// Structure (one workgroup = 8 waves = one 256 x 256 output tile, 64-deep K tiles, 128 KB of LDS):
//  * half tiles: a K tile is four 16-KB half tiles A0, A1 (128 rows each), B0, B1 (128 weight rows each). A half tile lives in LDS as
//    16 subtiles [16 rows][32 k] of 1024 B; inside a subtile byte p holds logical byte p ^ (((p >> 9) & 1) << 5), which makes the
//    ds_read_b128 of an MFMA operand (lane l: row l & 15, k chunk l >> 4) hit 16 distinct 16-B slots per 16-lane service group.
//  * LDS-DMA staging: every wave fills one row block (16 rows x 64 k = 2 subtiles) of a half tile with two global_load_lds_dwordx4; the
//    LDS image of such a load is lane-linear, so the swizzle sits in the per-lane SOURCE address.
//  * wave tile: wave (wr, wc) owns rows {mi * 128 + wr * 64 + [0, 64)} x cols {ni * 128 + wc * 32 + [0, 32)}, mi, ni in {0, 1}: each
//    half tile feeds exactly one of the four quadrants' operand reads, so a half tile's slot is free one phase after it was read.
//  * phases: a K tile is 4 phases; phase p = {ds_read the operand that becomes live | stage one half tile 7 phases ahead |
//    s_waitcnt vmcnt(12): the half tile read NEXT phase has landed | lgkmcnt(0) | s_barrier | 16 MFMA (one quadrant x K = 64) |
//    s_barrier}. The two wave rows run one barrier apart (the waves 0-3 / 4-7 of a workgroup share SIMDs pairwise): while one
//    wave of a SIMD issues its 16 MFMAs, its partner reads LDS and issues DMA.
//      P1: read B0(t)   stage A0(t+2)   mfma q(0,0) = A0 . B0
//      P2: read B1(t)   stage B0(t+2)   mfma q(0,1) = A0 . B1
//      P3: read A1(t)   stage B1(t+2)   mfma q(1,1) = A1 . B1
//      P4: read A0(t+1) stage A1(t+2)   mfma q(1,0) = A1 . B0
//    Hazards: (RAW) a half tile is read one phase after the counted vmcnt + barrier that retires its DMA in every wave; (WAR) a slot
//    is re-staged one phase after its read, whose lgkmcnt(0) sits BEFORE the reading phase's first barrier.
//  * the MFMA takes the weight fragment as its A operand and the activation fragment as B: D[n][m], so a lane ends up with 4
//    CONSECUTIVE output columns of one row (16-B fp32 / 8-B 16-bit stores), and the two weight halves B0 / B1 can be two different
//    row ranges of the weight: with B0 = W12[c .. c+128) and B1 = W12[F + c ..) a lane holds x1 and x2 of the same output element
//    in acc[.][0] and acc[.][1] -- the gated GeLU of dimsum/mlp.py:66-70 is a per-register epilogue.
//
// TN variant (kVarTN, the weight-gradient shape dW = dY^T X of a Linear: C (P, Q) = sum_r A[r, p] B[r, q], both operands stored as rows over
// the REDUCTION index -- dimsum/mlp.py's w12 / w3, attention_fusion.py's qkv / proj under autograd): same tiles, phases and epilogue; a half
// tile is [64 r][128 columns] (256 B per row, 16 KB), filled by the same two DMA pieces per wave (4 rows each), and an MFMA operand
// (16 columns x 32 r) is assembled by two ds_read_b64_tr_b16 -- the hardware transpose read hands lane (t = l & 15, g = l >> 4) the four
// rows 4 g .. 4 g + 3 of column t from the 8-byte pieces its 16-lane group addresses (tools/ubench/tr_read.hip prints the mapping). The
// r order inside an MFMA differs from the NT kernel's; both operands use the same one, which is all a dot product needs. The 32-byte
// pair u of row r sits at u ^ (r & 7): the 8 rows a 32-lane half of the read touches land in 8 distinct bank ranges. A launch may
// split the reduction (grid = tiles x splits, partial results `c_split_stride` apart: the host adds them in a fixed order).
//
// NN variant (kVarNN: C (P, Q) = sum_r A[p, r] B[r, q] -- A rows contiguous ALONG the reduction like the NT kernel's, B rows OVER the reduction like the
// TN kernel's: the weight gradients of the Mamba projections, whose one operand is a d-major activation (channels x tokens: in_proj's d xz,
// out_proj's out_z) and whose other operand is token-major): A is staged and read exactly like the NT kernel's A, B like the TN kernel's B --
// except that the DMA of B puts source row 8 g + 4 blk + e of every 32-row group into LDS row 16 blk + 4 g + e, so that the transposing read
// hands lane group g the reduction rows 8 g .. 8 g + 7 the NT read gives it for A (an MFMA needs both operands on the same k).
#pragma once
#include "common.hpp"

namespace dimsum {
namespace gemm_nt {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kHalf = 16384;               // bytes of one half tile (128 rows x 64 k x 2 B)
constexpr int kParity = 4 * kHalf;         // bytes of one K tile in LDS
constexpr int kSlotA0 = 0, kSlotA1 = kHalf, kSlotB0 = 2 * kHalf, kSlotB1 = 3 * kHalf;

enum { kOpBf16 = 0, kOpF16 = 1 };
enum { kEpiF32 = 0, kEpiGatedSplit3 = 1, kEpiGatedF16 = 2, kEpiF32Bias = 3, kEpiF32GateRes = 4, kEpiF16Qkv = 5, kEpiF32Conv = 6 };

struct Args {
    const char *A, *B0, *B1;       // B0 / B1: first weight row of the two 128-row halves' matrices (B1 = B0 + 128 rows for a plain GEMM)
    void *C;
    const float *bias0, *bias1;    // per output column (gated: of x1 / x2), may be NULL
    const float *sa, *sb;          // scaled-fp16 operands: inverse scale per A row (M) / per B row (N; gated: 2 F), NULL = 1
    const float *gate_bound;       // kEpiGatedF16 with per-row scales: {max_n sum_k |w_nk| (true units), max |bias|}; NULL = out_scale
    float *inv_out;                // kEpiGatedF16 with per-row scales: (M) inverse scales of the h image
    const float *res, *gate;       // kEpiF32GateRes: C = res + gate[row / rows_per_batch] * (A B^T + bias); res (M, N), gate (M / rows_per_batch, N) or NULL (= 1)
    int64_t ldr, ldg;
    int rows_per_batch;            // a multiple of 256: a tile lies inside one batch element
    int64_t lda, ldb, ldc;         // in elements
    int M, N, K;                   // N = columns per B half matrix row range handled as [0, N) for plain, hidden width F for gated
    int tiles_m, tiles_n;
    float out_scale;               // kEpiGatedF16: h is stored as fp16(h * out_scale)
    int group_m;                   // tile rows per group of the tile order (L2 patch shape)
    int stagger;                   // > 0: the first-round workgroups on odd CUs start `stagger` x 1024 cycles late (see cu_stagger)
    float *x12;                    // kVarKeepX12 (gated epilogues, training): the bias-free fp32 (M, 2 F) [x1 | x2] is stored as well
    int64_t ldx;
    int splits;                    // kVarTN: the reduction is cut into `splits` ranges of K rows each (K = rows per range)
    int64_t c_split_stride;        // elements between the partial results
    int a_alias_tiles;             // != 0 = K tiles kt >= a_alias_tiles of A are the tiles kt - a_alias_tiles: a [hi | lo] pair read as the image [hi | hi | lo]
    int a_alias_from;              // K tiles kt >= a_alias_from are aliased (left order [hi | hi | lo]: = a_alias_tiles; weight order [hi | lo | hi]: 2 a_alias_tiles,
                                   // with a_alias_tiles = twice the piece width: the third piece re-reads the first)
    int tn_pieces;                 // kVarTN: both operands are [hi | lo] pairs: `splits` = 3 x splits_per_piece, split -> (piece, row range); the pieces pair
    int64_t a_pair_cols, b_pair_cols;   //         A (weight order) columns [0, lo, 0] with B (left order) columns [0, 0, lo]
    int b_alias_tiles;             // the same for the B rows (NT only: in_proj, whose activation image is the right operand)
    const float *conv_w, *conv_b;  // kEpiF32Conv: (conv_rows, conv_width) taps (row stride conv_w_ld), (conv_rows) bias or NULL
    int conv_rows, conv_width, conv_seq, conv_w_ld;     //   output rows [0, conv_rows) get conv + SiLU along the columns, sequences of conv_seq columns (256 % conv_seq == 0)
    const float *a_block_inv;      // kVarRebase (TN, fp16): (M / 32, a_block_inv_ld >= K / 64) inverse scales (powers of two): the A values of token group g = m / 32 in K tile kt were
    int a_block_inv_ld;            //   stored with their own scale 1 / a_block_inv[g][kt]; the kernel puts a group on ONE scale (the largest inverse of its row of the table, K <= 4096)
    const _Float16 *k_fac;         // kVarRowFac (TN, fp16): one factor (a power of two <= 1) per REDUCTION row (all splits): A's row r is multiplied by k_fac[r] as it is read --
    const float *c_scale;          //   both operands are scaled-fp16 images with ROW scales (the rows are the reduction index of a weight gradient); *c_scale multiplies the result
    const float *k_inv_a, *k_inv_b; //   ... or the two images' inverse row scales themselves (k_inv_b NULL = 1): the workgroup forms the factors of its range, normalised by the range's maximum
    int q_cols;                    // kEpiF16Qkv: columns [0, q_cols) take the per-row scale, the others the per-batch-element one
    int c_pieces2;                 // kEpiGatedSplit3: the h image is written as the pair [hi | lo] (ldc >= 2 F) for a consumer that reads it with a_alias_tiles
};
// tuning variants (bit mask; 0 = the shipped schedule)
enum { kVarLgkmAfterBarrier = 1, kVarNoEpilogue = 2, kVarNtStores = 4, kVarFullLineStores = 8, kVarNoSetprio = 16, kVarSc1Stores = 32, kVarSc0Stores = 64, kVarKeepX12 = 128, kVarTN = 256, kVarM128 = 512, kVarRebase = 1024, kVarPersist = 2048, kVarRowFac = 4096, kVarNN = 8192 };

__device__ __forceinline__ float dpp_row_ror8(float x) {      // lane l <- lane l ^ 8 (rotation by 8 inside each row of 16 lanes)
    const int v = __builtin_bit_cast(int, x);                  // (old = the source: with a constant `old` hipcc 7.2 merges the calls of an unrolled loop)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
}
template <int kAux> __device__ __forceinline__ void store_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, const f4 &v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, 0, kAux);
}

template <int kOp> __device__ __forceinline__ f4 mma(const u32x4 &a, const u32x4 &b, f4 c) {
    if constexpr (kOp == kOpBf16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// a wave-uniform pointer the compiler lost track of (it went through a struct copy behind a branch): back into SGPRs
__device__ __forceinline__ const char *uniform_ptr(const char *q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
}

// c ? a : b on the bits, with the mask hidden from the optimiser: a plain select whose `a` is a load is turned into a branch around the load
// (CodeGenPrepare), and hipcc then waits vmcnt(0) at every later use of anything loaded -- inside a store sequence that drains the queue
__device__ __forceinline__ unsigned pick_mask(bool c) {
    unsigned m = c ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    return m;
}
__device__ __forceinline__ unsigned pick_mask_lanes(bool c) {          // (a per-lane condition)
    unsigned m = c ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    return m;
}
__device__ __forceinline__ float pick(unsigned m, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
__device__ __forceinline__ f4 pick(unsigned m, const f4 &a, const f4 &b) {
    return f4{pick(m, a[0], b[0]), pick(m, a[1], b[1]), pick(m, a[2], b[2]), pick(m, a[3], b[3])};
}

#define DIMSUM_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define DIMSUM_DS_READ_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define DIMSUM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Two workgroups share a CU under kVarM128: while one is in its epilogue the other's LDS-DMA loads go through the same in-order vector-memory
// path, behind whatever stores are queued there. The epilogue therefore keeps at most DIMSUM_EPI_PACE of its own stores in flight
// (tools/bench_gemm.py --tiles: unpaced, in_proj 180 -> 214 us); the 8-wave kernel has the CU to itself and does not pace.
#ifndef DIMSUM_PERSIST_GS
#define DIMSUM_PERSIST_GS 2
#endif
#ifndef DIMSUM_EPI_PACE
#define DIMSUM_EPI_PACE 0
#endif
#define DIMSUM_STR2(x) #x
#define DIMSUM_STR(x) DIMSUM_STR2(x)
#define DIMSUM_PACE_STORES() do { if constexpr (kM1 && DIMSUM_EPI_PACE >= 0) asm volatile("s_waitcnt vmcnt(" DIMSUM_STR(DIMSUM_EPI_PACE) ")" ::: "memory"); } while (0)
#define DIMSUM_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x sigmoid(2 u): the same form as token_transform.hip's gated GeLU pass
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x * fast_rcp(1.0f + fast_exp2(-2.0f * kLog2e * u));
}

// M128 variant (kVarM128): 128 x 256 output tiles by 4-wave workgroups with an 80-KB ring, TWO workgroups per CU. A 256 x 256 tile's
// epilogue (fp32 stores at the chip's ~5 TB/s shared by all CUs storing at once, or the gated GeLU's VALU work) runs with the matrix pipe idle, and with one
// workgroup per CU nothing else is resident to use it: a third of a K = 1024 launch, half of a K = 512 one (scaled-fp16 operands: one
// product per element, short K). Two independent workgroups per CU overlap one's epilogue with the other's K loop. Same wave tile
// (a wave owns all 128 rows x 64 columns: acc[mi][ni][4][2], mi = row half), same 16-MFMA phases and operand registers; the two A
// halves A_lo / A_hi are 64-row entries (8 KB), B0 / B1 128-row entries (16 KB), consumed in the order
//   A_lo(0) | B0(t) B1(t) A_hi(t) A_lo(t+1) | ...
// out of a ring of ten 8-KB units (positions advance by the entry size, the first entry sits at unit 1 so that a B entry always starts
// at an even unit and never straddles the wrap): while phase p reads entry p + 1 the DMA of entry p + 6 is issued into the units the
// entries <= p have left -- every window of 6 consecutive entries is <= 10 units -- and `s_waitcnt vmcnt(12)` (the DMA pieces of 4
// entries: A 2, B 4 per wave) retires entry p + 2. ONE barrier per phase (lgkmcnt(0) before it frees the slot read, the counted vmcnt
// before it publishes the entry read next); the ping-pong partner of a wave is the other workgroup's wave on its SIMD.
template <int kOp, int kEpi, int kVar>
__device__ __forceinline__ void gemm_body(const Args &p, char *lds) {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool kM1 = (kVar & kVarM128) != 0;
    constexpr int kTileM = kM1 ? 128 : kBM;        // rows of an output tile
    constexpr int kMiRows = kM1 ? 64 : 128;        // rows between the two A halves (mi)
    const int wr = kM1 ? 0 : (w >> 2), wc = w & 3;

    // ---- block -> tile: every XCD (block b runs on XCD b % 8) walks a contiguous range of the tile list, ordered in groups of 8 tile
    // rows (consecutive tiles walk down the rows of a group, then to the next tile column): the 32 workgroups an XCD runs at a time
    // form an 8 x 4 patch that shares its A and B panels through that XCD's L2.
    constexpr bool kTN = (kVar & kVarTN) != 0;
    constexpr bool kNN = (kVar & kVarNN) != 0;            // A like NT, B like TN (rows permuted at staging)
    constexpr bool kBT = kTN || kNN;                      // B is stored over the reduction index
    static_assert(!kBT || kEpi == kEpiF32, "the TN / NN variants have the plain fp32 epilogue");
    static_assert(!(kBT && kM1) && !(kTN && kNN), "the TN / NN variants have 256 x 256 tiles only");
    // kVarPersist: gridDim.x (a multiple of 8, <= the tile count) workgroups walk the tile list, workgroup b taking the list positions b,
    // b + gridDim.x, .. -- the positions the hardware dispatcher would hand to the CU one after the other -- as ONE stream of K tiles: the
    // last two K tiles of an output tile stage the first two of the next one, the epilogue runs with those DMAs in flight, and the K loop
    // resumes on landed data (no pipeline fill, no workgroup launch between tiles). K / 64 must be even (the ring parity follows the stream).
    constexpr bool kPersist = (kVar & kVarPersist) != 0;
    static_assert(!kPersist || (!kTN && !kM1 && (kEpi == kEpiF32 || kEpi == kEpiF32Bias || kEpi == kEpiF32GateRes || kEpi == kEpiGatedF16)), "persistent: the 256-row NT kernel");
    int tile_m, tile_n, split = 0;
    auto map_tile = [&](int bid, int &tile_m, int &tile_n) {
        const int nwg = kPersist ? p.tiles_m * p.tiles_n : (int)gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if constexpr (kBT) {       // (the splits of one tile are `tiles` apart in the list: they run at the same time on different XCDs' ranges)
            const int tiles = p.tiles_m * p.tiles_n;
            split = t / tiles;
            t -= split * tiles;
        }
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int g = t / per_group, within = t - g * per_group;
        const int first_m = g * GM;
        const int gsz = min(p.tiles_m - first_m, GM);
        tile_n = within / gsz;
        tile_m = first_m + (within - tile_n * gsz);
    };
    int bid = blockIdx.x;
    map_tile(bid, tile_m, tile_n);
    constexpr int kTileN = (kEpi == kEpiGatedSplit3 || kEpi == kEpiGatedF16) ? 128 : kBN;
    int m0 = tile_m * kTileM, n0 = tile_n * kTileN;

    // ---- staging addresses: wave w fills row block w (16 rows) of every half tile; lane l -> LDS byte 16 l of a subtile = logical row l >> 2,
    // k chunk (l & 3) ^ (2 if l >= 32)
    const int st_row = w * 16 + (lane >> 2);
    const int st_kc = (lane & 3) ^ ((lane >> 5) << 1);
    const unsigned a_voff = (unsigned)((st_row * p.lda + st_kc * 8) * 2);
    // weight rows beyond N are clamped to the last row (their products land in columns the epilogue masks)
    struct Src { const char *a_base, *b_base[2]; unsigned b_voff[2]; };          // the operand panels of one output tile
    auto make_src = [&](int m0_, int n0_) {
        Src s_;
        if constexpr (kEpi == kEpiGatedSplit3 || kEpi == kEpiGatedF16) {
            s_.b_base[0] = p.B0 + (int64_t)n0_ * p.ldb * 2;
            s_.b_base[1] = p.B1 + (int64_t)n0_ * p.ldb * 2;
            const int row = min(st_row, p.N - 1 - n0_);
            s_.b_voff[0] = s_.b_voff[1] = (unsigned)((row * p.ldb + st_kc * 8) * 2);
        } else {
            s_.b_base[0] = s_.b_base[1] = p.B0 + (int64_t)n0_ * p.ldb * 2;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = min(ni * 128 + st_row, p.N - 1 - n0_);
                s_.b_voff[ni] = (unsigned)((row * p.ldb + st_kc * 8) * 2);
            }
        }
        s_.a_base = p.A + (int64_t)m0_ * p.lda * 2;
        return s_;
    };
    Src cur = make_src(m0, n0);
    const char *(&b_base)[2] = cur.b_base;
    unsigned (&b_voff)[2] = cur.b_voff;
    const char *&a_base = cur.a_base;
    const int64_t a_half = (int64_t)kMiRows * p.lda * 2;
    unsigned b_voff2[2][2];        // M128: a B entry (128 weight rows) is two 64-row pieces per wave pass: rows st_row, st_row + 64
    if constexpr (kM1) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int first = (kEpi == kEpiGatedSplit3 || kEpi == kEpiGatedF16) ? 0 : ni * 128;
                const int row = min(first + h * 64 + st_row, p.N - 1 - n0);
                b_voff2[ni][h] = (unsigned)((row * p.ldb + st_kc * 8) * 2);
            }
    }

    // TN: wave w fills rows 8 w .. 8 w + 7 of every half tile (two pieces of 4 rows x 256 B); lane l -> row l >> 4 of the piece, 16-byte
    // slot l & 15, which holds the source chunk whose 32-byte pair index is (slot >> 1) ^ (row & 7)
    unsigned tn_voff_a[2], tn_voff_b[2];
    const char *tn_a = nullptr, *tn_b = nullptr;
    int64_t tn_a_tile = 0, tn_b_tile = 0;
    int tn_row_split = 0;          // which range of reduction rows this workgroup adds up
    if constexpr (kBT) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int row = w * 8 + pc * 4 + (lane >> 4), key = pc * 4 + (lane >> 4);
            const int chunk = ((((lane & 15) >> 1) ^ key) << 1) | (lane & 1);
            // NN: LDS row 16 blk + 4 g + e of a 32-row group holds SOURCE row 8 g + 4 blk + e (the swizzle key stays the LDS row's)
            const int rl = row & 31, srow = kNN ? ((row & ~31) | (((rl & 15) >> 2) << 3) | ((rl >> 4) << 2) | (rl & 3)) : row;
            tn_voff_a[pc] = (unsigned)(srow * p.lda * 2 + chunk * 16);
            tn_voff_b[pc] = (unsigned)(srow * p.ldb * 2 + chunk * 16);
        }
        int row_split = split;
        int64_t a_col = 0, b_col = 0;
        if (p.tn_pieces) {          // split = piece * splits_per_piece + row range
            const int spp = p.splits / 3, piece = split / spp;
            row_split = split - piece * spp;
            a_col = piece == 1 ? p.a_pair_cols : 0;
            b_col = piece == 2 ? p.b_pair_cols : 0;
        }
        tn_row_split = row_split;
        if constexpr (kNN) cur.a_base += (int64_t)row_split * p.K * 2;          // A rows run along the reduction: the range starts K elements further
        tn_a = p.A + ((int64_t)row_split * p.K * p.lda + m0 + a_col) * 2;
        tn_b = p.B0 + ((int64_t)row_split * p.K * p.ldb + n0 + b_col) * 2;
        tn_a_tile = (int64_t)kBK * p.lda * 2;
        tn_b_tile = (int64_t)kBK * p.ldb * 2;
    }
    // stage(kind, kt): the two DMA pieces (k halves) of this wave's row block of half tile `kind` of K tile kt
    const unsigned st_lds = lds0 + w * 2048;
    auto stage = [&](int slot, const char *base, unsigned voff, int kt, int alias = 0) {      // base: wave-uniform (SGPR pair), voff: this lane's byte offset
        const unsigned dst = st_lds + (kt & 1) * kParity + slot;
        const int src_kt = (alias && kt >= (slot < kSlotB0 ? p.a_alias_from : alias)) ? kt - alias : kt;      // (the LDS parity follows kt, the source tile may be an aliased earlier one)
        const char *s = base + (int64_t)src_kt * (kBK * 2);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + voff), (lds_void_t *)(uintptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + 64 + voff), (lds_void_t *)(uintptr_t)(dst + 1024), 16, 0, 0);
    };
    auto stage_tn = [&](int slot, const char *base, int64_t tile_bytes, const unsigned (&voff)[2], int kt) {
        const unsigned dst = st_lds + (kt & 1) * kParity + slot;
        const char *s = base + (int64_t)kt * tile_bytes;
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + voff[0]), (lds_void_t *)(uintptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + voff[1]), (lds_void_t *)(uintptr_t)(dst + 1024), 16, 0, 0);
    };
    auto stage_tn_a = [&](int slot, const char *base, int kt) {       // (the LDS parity follows kt, the source tile may be an aliased earlier one)
        const unsigned dst = st_lds + (kt & 1) * kParity + slot;
        const int src_kt = (p.a_alias_tiles && kt >= p.a_alias_tiles) ? kt - p.a_alias_tiles : kt;
        const char *s = base + (int64_t)src_kt * tn_a_tile;
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + tn_voff_a[0]), (lds_void_t *)(uintptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(s + tn_voff_a[1]), (lds_void_t *)(uintptr_t)(dst + 1024), 16, 0, 0);
    };
    auto stage_a0 = [&](int kt) { if constexpr (kTN) stage_tn_a(kSlotA0, tn_a, kt); else stage(kSlotA0, a_base, a_voff, kt, p.a_alias_tiles); };
    auto stage_a1 = [&](int kt) { if constexpr (kTN) stage_tn_a(kSlotA1, tn_a + 256, kt); else stage(kSlotA1, a_base + a_half, a_voff, kt, p.a_alias_tiles); };
    auto stage_b0 = [&](int kt) { if constexpr (kBT) stage_tn(kSlotB0, tn_b, tn_b_tile, tn_voff_b, kt); else stage(kSlotB0, b_base[0], b_voff[0], kt, p.b_alias_tiles); };
    auto stage_b1 = [&](int kt) { if constexpr (kBT) stage_tn(kSlotB1, tn_b + 256, tn_b_tile, tn_voff_b, kt); else stage(kSlotB1, b_base[1], b_voff[1], kt, p.b_alias_tiles); };
    // (persistent: half tile `which` of K tile kt of the tile whose panels are s_)
    auto stage_of = [&](const Src &s_, int which, int kt) {
        if (which == 0) stage(kSlotA0, s_.a_base, a_voff, kt, p.a_alias_tiles);
        else if (which == 1) stage(kSlotB0, s_.b_base[0], s_.b_voff[0], kt, p.b_alias_tiles);
        else if (which == 2) stage(kSlotB1, s_.b_base[1], s_.b_voff[1], kt, p.b_alias_tiles);
        else stage(kSlotA1, s_.a_base + a_half, a_voff, kt, p.a_alias_tiles);
    };

    // ---- operand read addresses: lane l reads row l & 15, k chunk l >> 4 of a subtile (swizzled: rows 8-15 swap chunk pairs)
    const unsigned rd = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ (((lane >> 3) & 1) << 1)) * 16));
    const unsigned a_rd = lds0 + wr * 4 * 2048 + rd;       // + slot A0 / A1, + i * 2048 (row block), + kh * 1024
    const unsigned b_rd = lds0 + wc * 2 * 2048 + rd;       // + slot B0 / B1, + j * 2048, + kh * 1024

    // TN: lane (t, g) addresses row 4 g + (t >> 2) of a 16-row block, 8 bytes at (t & 3) * 8 of the 32-byte pair (column block ^ key)
    unsigned a_rd_tn[4], b_rd_tn[2];
    if constexpr (kBT) {
        const int t = lane & 15, g = lane >> 4, key = ((g & 1) << 2) | (t >> 2);
        const unsigned base = lds0 + (unsigned)((4 * g + (t >> 2)) * 256 + (t & 3) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_rd_tn[i] = base + (unsigned)(((wr * 4 + i) ^ key) << 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) b_rd_tn[j] = base + (unsigned)(((wc * 2 + j) ^ key) << 5);
    }

    u32x4 a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    f4 acc[2][2][4][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[mi][ni][i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // one TN operand fragment: rows kh * 32 + [0, 16) and + [16, 32) of the K tile
#define DIMSUM_READ_TN1(dst, addr, slot, kh)                                               \
    do {                                                                                   \
        u32x2 lo_, hi_;                                                                    \
        DIMSUM_DS_READ_TR(lo_, addr, (slot) + (kh) * 8192);                                \
        DIMSUM_DS_READ_TR(hi_, addr, (slot) + (kh) * 8192 + 4096);                         \
        dst = u32x4{lo_[0], lo_[1], hi_[0], hi_[1]};                                       \
    } while (0)
#define DIMSUM_READ_A(dst, slot, par)                                                      \
    do {                                                                                   \
        if constexpr (kTN) {                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                             \
                const unsigned ad_ = a_rd_tn[i_] + (par);                                  \
                DIMSUM_READ_TN1(dst[i_][0], ad_, slot, 0);                                 \
                DIMSUM_READ_TN1(dst[i_][1], ad_, slot, 1);                                 \
            }                                                                              \
            break;                                                                         \
        }                                                                                  \
        const unsigned ad_ = a_rd + (par);                                                 \
        DIMSUM_DS_READ_B128(dst[0][0], ad_, (slot) + 0 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[0][1], ad_, (slot) + 0 * 2048 + 1024);                     \
        DIMSUM_DS_READ_B128(dst[1][0], ad_, (slot) + 1 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[1][1], ad_, (slot) + 1 * 2048 + 1024);                     \
        DIMSUM_DS_READ_B128(dst[2][0], ad_, (slot) + 2 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[2][1], ad_, (slot) + 2 * 2048 + 1024);                     \
        DIMSUM_DS_READ_B128(dst[3][0], ad_, (slot) + 3 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[3][1], ad_, (slot) + 3 * 2048 + 1024);                     \
    } while (0)
#define DIMSUM_READ_B(dst, slot, par)                                                      \
    do {                                                                                   \
        if constexpr (kBT) {                                                               \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                             \
                const unsigned ad_ = b_rd_tn[j_] + (par);                                  \
                DIMSUM_READ_TN1(dst[j_][0], ad_, slot, 0);                                 \
                DIMSUM_READ_TN1(dst[j_][1], ad_, slot, 1);                                 \
            }                                                                              \
            break;                                                                         \
        }                                                                                  \
        const unsigned ad_ = b_rd + (par);                                                 \
        DIMSUM_DS_READ_B128(dst[0][0], ad_, (slot) + 0 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[0][1], ad_, (slot) + 0 * 2048 + 1024);                     \
        DIMSUM_DS_READ_B128(dst[1][0], ad_, (slot) + 1 * 2048);                            \
        DIMSUM_DS_READ_B128(dst[1][1], ad_, (slot) + 1 * 2048 + 1024);                     \
    } while (0)
    // one quadrant x K = 64: 16 MFMAs, the same accumulator recurs after 8
#define DIMSUM_QUADRANT(MI, NI, AF, BF)                                                    \
    do {                                                                                   \
        if constexpr (!(kVar & kVarNoSetprio)) __builtin_amdgcn_s_setprio(1);              \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                              \
                    acc[MI][NI][i][j] = mma<kOp>(BF[j][kh], AF[i][kh], acc[MI][NI][i][j]); \
        if constexpr (!(kVar & kVarNoSetprio)) __builtin_amdgcn_s_setprio(0);              \
    } while (0)
#define DIMSUM_PHASE_SYNC()                                                                \
    do {                                                                                   \
        if constexpr (!(kVar & kVarLgkmAfterBarrier)) DIMSUM_WAIT_LGKM0();                 \
        __builtin_amdgcn_s_barrier();                                                      \
        if constexpr (kVar & kVarLgkmAfterBarrier) DIMSUM_WAIT_LGKM0();                    \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#define DIMSUM_PHASE_END()                                                                 \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        __builtin_amdgcn_s_barrier();                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)

    // kVarRebase: block-scaled fp16 A (the scan's out_z: one power-of-two scale per 32 tokens x 64 channels, section 3.5 of DESIGN.md): a fragment
    // (16 tokens x 32 reduction rows of one K tile) is multiplied by its block's factor right after it was read -- exact (a power of two <= 1)
    // unless the value drops below fp16's normal range, i.e. below 2^-29 of its row group's maximum. 32 v_pk_mul_f16 per fragment set, under the MFMAs.
    constexpr bool kRebase = (kVar & kVarRebase) != 0;
    static_assert(!kRebase || (kTN && kOp == kOpF16), "the rebase serves the fp16 TN variant");
    // The factors: a wave owns 4 token groups (2 per 128-row half); lane t of rb[q] holds group q's factor for K tile t = table[g][t] / top[q], with
    // top[q] = the row's maximum (one load + one wave maximum per group, before the main loop) -- the per-tile factor is then a v_readlane, no memory
    // operation inside the loop (scalar loads would share lgkmcnt with the counted ds_read waits). top[q] is the epilogue's row scale.
    float rb[4] = {1.f, 1.f, 1.f, 1.f}, top[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (kRebase) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = (m0 + (q >> 1) * 128 + wr * 64 + (q & 1) * 32) >> 5;
            const float v = lane < p.K / kBK ? p.a_block_inv[(int64_t)g * p.a_block_inv_ld + lane] : 0.f;
            top[q] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wave_allmax(v))));
            rb[q] = v * __builtin_amdgcn_rcpf(top[q]);                          // (powers of two: exact)
        }
    }
    // kVarRowFac: the weight gradient dW = dY^T X of a Linear under the scaled-fp16 policy. Both operands are the images the forward / the input
    // gradient already use -- fp16 rows with ONE power-of-two scale per row -- but here the rows are the reduction index: term r carries the
    // factor a_inv[r] b_inv[r]. The host hands k_fac[r] = a_inv[r] b_inv[r] / max_r(..) (<= 1, a power of two, as fp16) and the maximum as
    // *c_scale; A's fragment rows are multiplied by their factors as they are read (v_pk_mul_f16, exact unless the value leaves fp16's normal
    // range: a term more than 2^14 below the largest row product -- then it keeps >= 11 - (shift - 14) bits, error <= 2^-25 of the largest term).
    // The factors of this workgroup's reduction range sit in the 32 KB of LDS behind the ring (K <= 16384 rows per range); a lane (t, g) of a
    // transposing read holds reduction rows kh 32 + {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} of the K tile: four 8-byte LDS reads per K tile.
    constexpr bool kRowFac = (kVar & kVarRowFac) != 0;
    static_assert(!kRowFac || (kBT && kOp == kOpF16 && !kRebase), "the row factors serve the fp16 TN / NN variants");
    unsigned fac_rd = 0;
    u32x2 fv[4];
    float c_scale_own = 1.0f;              // (factors formed here: the maximum of this workgroup's range)
    if constexpr (kRowFac) {
        char *fl = lds + 2 * kParity;
        if (p.k_inv_a) {
            // the factors of this range from the images' row scales: products in registers (<= 4 chunks of 8 rows per thread: K <= 16384), the
            // range's maximum through LDS, then fp16(product / maximum) -- powers of two, exact -- into the factor table. One launch less per product
            // (320 per DiM-L/2 training step, ~9 us each as a single-workgroup kernel), and the normalisation is per range instead of per tensor.
            const float *ia = p.k_inv_a + (int64_t)tn_row_split * p.K, *ib = p.k_inv_b ? p.k_inv_b + (int64_t)tn_row_split * p.K : nullptr;
            float pr[4][8], m = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = threadIdx.x + c * 512;
                const bool in = i < p.K / 8;
                const float *a8 = ia + (in ? i : 0) * 8;
                const f4 a0 = *reinterpret_cast<const f4 *>(a8), a1 = *reinterpret_cast<const f4 *>(a8 + 4);
                f4 b0 = f4{1.f, 1.f, 1.f, 1.f}, b1 = b0;
                if (ib) { b0 = *reinterpret_cast<const f4 *>(ib + (in ? i : 0) * 8); b1 = *reinterpret_cast<const f4 *>(ib + (in ? i : 0) * 8 + 4); }
#pragma unroll
                for (int e = 0; e < 4; ++e) { pr[c][e] = in ? a0[e] * b0[e] : 0.f; pr[c][4 + e] = in ? a1[e] * b1[e] : 0.f; }
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, pr[c][e]);
            }
            float *red = reinterpret_cast<float *>(lds);          // (the ring is not in use yet)
            m = wave_allmax(m);
            if (lane == 0) red[threadIdx.x >> 6] = m;
            __syncthreads();
            m = red[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
            c_scale_own = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
            const float r = m > 0.f ? 1.0f / m : 0.f;             // (a power of two: exact; an all-zero range has factors 0)
            typedef _Float16 h8v __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = threadIdx.x + c * 512;
                if (i < p.K / 8) {
                    h8v h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (_Float16)(pr[c][e] * r);
                    *reinterpret_cast<u32x4 *>(fl + i * 16) = __builtin_bit_cast(u32x4, h);
                }
            }
        } else {
            const _Float16 *src = p.k_fac + (int64_t)tn_row_split * p.K;
            for (int i = threadIdx.x; i < p.K / 8; i += 512) *reinterpret_cast<u32x4 *>(fl + i * 16) = *reinterpret_cast<const u32x4 *>(src + i * 8);
        }
        __syncthreads();
        // (NN: A's fragment of lane group g holds reduction rows kh 32 + 8 g .. 8 g + 7: one 16-byte read per k half)
        fac_rd = lds0 + 2 * kParity + (lane >> 4) * (kNN ? 16 : 8);
    }
#define DIMSUM_DS_READ_B64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
    auto read_fac = [&](int kt_) {          // issued with the operand reads of the phase: the phase's lgkmcnt(0) covers them
        if constexpr (kRowFac) {
            const unsigned ad = fac_rd + (unsigned)kt_ * 128u;
            if constexpr (kNN) {
                DIMSUM_DS_READ_B64(fv[0], ad, 0);
                DIMSUM_DS_READ_B64(fv[1], ad, 8);
                DIMSUM_DS_READ_B64(fv[2], ad, 64);
                DIMSUM_DS_READ_B64(fv[3], ad, 72);
            } else {
                DIMSUM_DS_READ_B64(fv[0], ad, 0);
                DIMSUM_DS_READ_B64(fv[1], ad, 32);
                DIMSUM_DS_READ_B64(fv[2], ad, 64);
                DIMSUM_DS_READ_B64(fv[3], ad, 96);
            }
        }
    };
    auto rebase = [&](u32x4 (&frag)[4][2], int mi, int kt_) {
        if constexpr (kRowFac) {
            typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(fv[q]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) asm volatile("" : "+v"(frag[i][kh]));
            const h8_t s0 = __builtin_bit_cast(h8_t, u32x4{fv[0][0], fv[0][1], fv[1][0], fv[1][1]});
            const h8_t s1 = __builtin_bit_cast(h8_t, u32x4{fv[2][0], fv[2][1], fv[3][0], fv[3][1]});
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                frag[i][0] = __builtin_bit_cast(u32x4, __builtin_bit_cast(h8_t, frag[i][0]) * s0);
                frag[i][1] = __builtin_bit_cast(u32x4, __builtin_bit_cast(h8_t, frag[i][1]) * s1);
            }
        }
        if constexpr (kRebase) {
            const _Float16 f0 = (_Float16)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb[mi * 2]), kt_));
            const _Float16 f1 = (_Float16)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rb[mi * 2 + 1]), kt_));
            // (the fragments are outputs of asynchronous ds_read asm statements: the empty asm pins every one of them to its registers HERE, after the
            // phase's lgkmcnt(0), so that whatever copies the multiplies need are made of landed data -- without it the compiler copied the asm outputs
            // right behind the reads, before the wait)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) asm volatile("" : "+v"(frag[i][kh]));
            typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
            const h8_t s0 = {f0, f0, f0, f0, f0, f0, f0, f0}, s1 = {f1, f1, f1, f1, f1, f1, f1, f1};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) frag[i][kh] = __builtin_bit_cast(u32x4, __builtin_bit_cast(h8_t, frag[i][kh]) * (i < 2 ? s0 : s1));
        }
    };

    const int nk = p.K / kBK;      // >= 2

    if constexpr (kM1) {
        constexpr unsigned kU = 8192, kRing = 10 * kU;
        unsigned spos = kU, rpos = kU;          // stage / read positions in the ring (wave-uniform)
        auto adv = [&](unsigned &pos, unsigned sz) { pos += sz; pos = pos >= kRing ? pos - kRing : pos; };
        auto dma = [&](unsigned dst, const char *base, unsigned voff, int kt, int alias, int alias_from) {
            const int src_kt = (alias && kt >= alias_from) ? kt - alias : kt;
            const char *s = base + (int64_t)src_kt * (kBK * 2);
            __builtin_amdgcn_global_load_lds((glb_void_t *)(s + voff), (lds_void_t *)(uintptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t *)(s + 64 + voff), (lds_void_t *)(uintptr_t)(dst + 1024), 16, 0, 0);
        };
        auto st_a = [&](int half, int kt) {      // 64 rows of A: row block w of the entry
            dma(st_lds + spos, a_base + half * a_half, a_voff, kt, p.a_alias_tiles, p.a_alias_from);
            adv(spos, kU);
        };
        auto st_b = [&](int ni, int kt) {        // 128 weight rows: row blocks w and w + 4
            dma(st_lds + spos, b_base[ni], b_voff2[ni][0], kt, p.b_alias_tiles, p.b_alias_tiles);
            dma(st_lds + spos + 4 * 2048, b_base[ni], b_voff2[ni][1], kt, p.b_alias_tiles, p.b_alias_tiles);
            adv(spos, 2 * kU);
        };
#define DIMSUM_M1_SYNC()                                                                   \
    do {                                                                                   \
        DIMSUM_WAIT_LGKM0();                                                               \
        __builtin_amdgcn_s_barrier();                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
        // one phase: read the entry at rpos into FRAG (A: 64 rows = 8 reads, B: 4 reads), stage (STAGE), counted wait, barrier, 16 MFMA
#define DIMSUM_M1_PHASE(READ, SIZE, STAGE, VM, MI, NI, AF, BF)                              \
    do {                                                                                   \
        READ;                                                                              \
        adv(rpos, SIZE);                                                                   \
        STAGE;                                                                             \
        DIMSUM_WAIT_VM(VM);                                                                \
        DIMSUM_M1_SYNC();                                                                  \
        DIMSUM_QUADRANT(MI, NI, AF, BF);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
        // prologue: entries 0 .. 5 = A_lo(0) B0(0) B1(0) A_hi(0) A_lo(1) B0(1) (18 DMA pieces per wave); the first two must have landed
        st_a(0, 0); st_b(0, 0); st_b(1, 0); st_a(1, 0); st_a(0, 1); st_b(0, 1);
        DIMSUM_WAIT_VM(12);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        DIMSUM_READ_A(a0, 0, rpos);               // "phase -1": A_lo(0), retired in every wave before phase 0 stages over its unit
        adv(rpos, kU);
        DIMSUM_M1_SYNC();
        int kt = 0;
        for (; kt < nk - 2; ++kt) {
            DIMSUM_M1_PHASE(DIMSUM_READ_B(b0, 0, rpos), 2 * kU, st_b(1, kt + 1), 12, 0, 0, a0, b0);
            DIMSUM_M1_PHASE(DIMSUM_READ_B(b1, 0, rpos), 2 * kU, st_a(1, kt + 1), 12, 0, 1, a0, b1);
            DIMSUM_M1_PHASE(DIMSUM_READ_A(a1, 0, rpos), kU, st_a(0, kt + 2), 12, 1, 1, a1, b1);
            DIMSUM_M1_PHASE(DIMSUM_READ_A(a0, 0, rpos), kU, st_b(0, kt + 2), 12, 1, 0, a1, b0);
        }
        // K tile nk - 2: its first two phases stage the last two entries, then the counted waits run down
        DIMSUM_M1_PHASE(DIMSUM_READ_B(b0, 0, rpos), 2 * kU, st_b(1, kt + 1), 12, 0, 0, a0, b0);
        DIMSUM_M1_PHASE(DIMSUM_READ_B(b1, 0, rpos), 2 * kU, st_a(1, kt + 1), 12, 0, 1, a0, b1);
        DIMSUM_M1_PHASE(DIMSUM_READ_A(a1, 0, rpos), kU, (void)0, 10, 1, 1, a1, b1);
        DIMSUM_M1_PHASE(DIMSUM_READ_A(a0, 0, rpos), kU, (void)0, 6, 1, 0, a1, b0);
        // K tile nk - 1
        DIMSUM_M1_PHASE(DIMSUM_READ_B(b0, 0, rpos), 2 * kU, (void)0, 2, 0, 0, a0, b0);
        DIMSUM_M1_PHASE(DIMSUM_READ_B(b1, 0, rpos), 2 * kU, (void)0, 0, 0, 1, a0, b1);
        DIMSUM_M1_PHASE(DIMSUM_READ_A(a1, 0, rpos), kU, (void)0, 0, 1, 1, a1, b1);
        DIMSUM_QUADRANT(1, 0, a1, b0);
#undef DIMSUM_M1_PHASE
#undef DIMSUM_M1_SYNC
    } else {
    // ---- prologue: the 8 half tiles of K tiles 0 and 1 in read order; A0(0) and B0(0) must have landed before the first reads
    stage_a0(0); stage_b0(0); stage_b1(0); stage_a1(0);
    stage_a0(1); stage_b0(1); stage_b1(1); stage_a1(1);
    DIMSUM_WAIT_VM(12);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // from here on the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    DIMSUM_READ_A(a0, kSlotA0, 0);                    // "phase -1": A0(0), retired in every wave before anything re-stages its slot
    read_fac(0);
    DIMSUM_PHASE_SYNC();
    rebase(a0, 0, 0);
    DIMSUM_PHASE_END();

    }
    // one K tile = four phases; ST_* stage the four half tiles two K tiles ahead; NEXT_A0: whether P4 reads the next K tile's A0
    // W12 / W34: whether the first / last two phases wait for the steady state's vmcnt(12) (the DMA pieces of 6 half tiles in flight)
#define DIMSUM_KTILE(KT, ST_A0, ST_B0, ST_B1, ST_A1, W12, W34)                                                                  \
    do {                                                                                                                        \
        const unsigned par = ((KT) & 1) * kParity, par_next = kParity - par;                                                    \
        DIMSUM_READ_B(b0, kSlotB0, par); ST_A0; if (W12) DIMSUM_WAIT_VM(12); DIMSUM_PHASE_SYNC();                               \
        DIMSUM_QUADRANT(0, 0, a0, b0); DIMSUM_PHASE_END();                                                                      \
        DIMSUM_READ_B(b1, kSlotB1, par); ST_B0; if (W12) DIMSUM_WAIT_VM(12); DIMSUM_PHASE_SYNC();                               \
        DIMSUM_QUADRANT(0, 1, a0, b1); DIMSUM_PHASE_END();                                                                      \
        DIMSUM_READ_A(a1, kSlotA1, par); read_fac(KT); ST_B1; if (W34) DIMSUM_WAIT_VM(12); DIMSUM_PHASE_SYNC();                 \
        rebase(a1, 1, KT); DIMSUM_QUADRANT(1, 1, a1, b1); DIMSUM_PHASE_END();                                                   \
        DIMSUM_READ_A(a0, kSlotA0, par_next); read_fac((KT) + 1);                                                               \
        ST_A1; if (W34) DIMSUM_WAIT_VM(12); DIMSUM_PHASE_SYNC();                                                                \
        DIMSUM_QUADRANT(1, 0, a1, b0);                                                                                          \
        rebase(a0, 0, (KT) + 1);     /* (under the MFMAs just issued: a0 is next used in P1 of K tile KT + 1) */                \
        DIMSUM_PHASE_END();                                                                                                     \
    } while (0)
    bool resumed = false;      // persistent: this tile's K loop follows an epilogue whose stores may still be in flight
    for (;;) {                 // (one pass unless kPersist)
    bool has_next = false;
    int nbid = 0, ntile_m = 0, ntile_n = 0;
    Src nxt = cur;             // (the last tile of a persistent workgroup streams its own first K tiles again: valid addresses, never read)
    if constexpr (kPersist) {
        nbid = bid + (int)gridDim.x;
        has_next = nbid < p.tiles_m * p.tiles_n;
        if (has_next) {
            map_tile(nbid, ntile_m, ntile_n);
            nxt = make_src(ntile_m * kTileM, ntile_n * kTileN);
        }
    }
    if constexpr (kPersist) {
        // ONE loop body for the whole stream (a second copy of the phases behind a branch -- a drained tail for the last tile, a peeled first K
        // tile -- made hipcc spill ~300 registers at the merges). The last two K tiles of an output tile stage K tiles 0 and 1 of the next
        // one (K / 64 is even: the ring parity of the next tile's K tile j is j & 1). After an epilogue its stores (<= 32 per wave, OLDER
        // than every DMA issued from here on: vmcnt retires in order) may drain under the first six phases, whose operands landed before
        // the epilogue: those phases do not wait at all; P3 of K tile 1 is the first that needs a piece issued after the stores.
        // P4 of the last K tile reads the next tile's A0 like every other (discarded: the epilogue comes first, the read is re-issued after it).
        for (int kt = 0; kt < nk; ++kt) {
            const bool own = kt + 2 < nk;
            const int skt = own ? kt + 2 : kt + 2 - nk;
            Src ss;
            ss.a_base = uniform_ptr(own ? cur.a_base : nxt.a_base);
            ss.b_base[0] = uniform_ptr(own ? cur.b_base[0] : nxt.b_base[0]);
            ss.b_base[1] = uniform_ptr(own ? cur.b_base[1] : nxt.b_base[1]);
            ss.b_voff[0] = cur.b_voff[0];          // (N is a whole number of tiles: no clamped weight rows, the lane offsets are the same in every tile)
            ss.b_voff[1] = cur.b_voff[1];
            const bool w12 = !(resumed && kt < 2), w34 = !(resumed && kt < 1);
            DIMSUM_KTILE(kt, stage_of(ss, 0, skt), stage_of(ss, 1, skt), stage_of(ss, 2, skt), stage_of(ss, 3, skt), w12, w34);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();       // the two wave rows meet
    } else if constexpr (!kM1) {
    // ---- main loop: K tile kt is computed while K tile kt + 2 is staged into the slots kt frees
    int kt = 0;
    for (; kt < nk - 2; ++kt) DIMSUM_KTILE(kt, stage_a0(kt + 2), stage_b0(kt + 2), stage_b1(kt + 2), stage_a1(kt + 2), true, true);
    // ---- the last two K tiles: nothing left to stage, the counted waits run down (5, 4, 3, 2, 1, 0 half tiles behind the one needed next)
    {
        const unsigned par = (kt & 1) * kParity, par_next = kParity - par;
        DIMSUM_READ_B(b0, kSlotB0, par);
        DIMSUM_WAIT_VM(10);
        DIMSUM_PHASE_SYNC();
        DIMSUM_QUADRANT(0, 0, a0, b0);
        DIMSUM_PHASE_END();
        DIMSUM_READ_B(b1, kSlotB1, par);
        DIMSUM_WAIT_VM(8);
        DIMSUM_PHASE_SYNC();
        DIMSUM_QUADRANT(0, 1, a0, b1);
        DIMSUM_PHASE_END();
        DIMSUM_READ_A(a1, kSlotA1, par);
        read_fac(kt);
        DIMSUM_WAIT_VM(6);
        DIMSUM_PHASE_SYNC();
        rebase(a1, 1, kt);
        DIMSUM_QUADRANT(1, 1, a1, b1);
        DIMSUM_PHASE_END();
        DIMSUM_READ_A(a0, kSlotA0, par_next);
        read_fac(kt + 1);
        DIMSUM_WAIT_VM(4);
        DIMSUM_PHASE_SYNC();
        DIMSUM_QUADRANT(1, 0, a1, b0);
        rebase(a0, 0, kt + 1);
        DIMSUM_PHASE_END();
        DIMSUM_READ_B(b0, kSlotB0, par_next);
        DIMSUM_WAIT_VM(2);
        DIMSUM_PHASE_SYNC();
        DIMSUM_QUADRANT(0, 0, a0, b0);
        DIMSUM_PHASE_END();
        DIMSUM_READ_B(b1, kSlotB1, par_next);
        DIMSUM_WAIT_VM(0);
        DIMSUM_PHASE_SYNC();
        DIMSUM_QUADRANT(0, 1, a0, b1);
        DIMSUM_PHASE_END();
        DIMSUM_READ_A(a1, kSlotA1, par_next);
        read_fac(kt + 1);
        DIMSUM_PHASE_SYNC();
        rebase(a1, 1, kt + 1);
        DIMSUM_QUADRANT(1, 1, a1, b1);
        DIMSUM_PHASE_END();
        DIMSUM_QUADRANT(1, 0, a1, b0);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();       // matches the second wave row's extra barrier
    }

    // ---- epilogue: acc[mi][ni][i][j] = rows m0 + mi * 128 + wr * 64 + i * 16 + (lane & 15), 4 columns from ni * 128 + wc * 32 + j * 16 + (lane >> 4) * 4
    // (persistent: the epilogue's lane-derived indices must not be hoisted out of the tile loop -- they would live in registers through the K
    // loop, which has ~10 to spare: everything below derives them from a lane id the optimiser cannot see through)
    int lane_e = lane;
    if constexpr (kPersist) asm volatile("" : "+v"(lane_e));
    {
    const int lane = lane_e;
    const int ecol = wc * 32 + (lane >> 4) * 4;
    if constexpr (kVar & kVarNoEpilogue) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[mi][ni][i][j]));
    } else if constexpr (kEpi == kEpiF32 || kEpi == kEpiF32Bias || kEpi == kEpiF32GateRes || kEpi == kEpiF32Conv) {
        if constexpr (kEpi == kEpiF32Conv) {
            // in_proj of a Mamba mixer with the depthwise causal conv1d + bias + SiLU of its x half in the epilogue (mamba_simple.py in_proj,
            // then causal_conv1d_fn inside mamba_inner_fn: selective_scan_interface.py:616): the product is d-major (rows = channels, columns =
            // tokens), a tile's 256 columns are whole sequences (256 % conv_seq == 0), so the conv runs along a tile row with zero history at
            // every sequence start. A half tile (kMiRows.. rows x 256 tokens, fp32) goes through the ring (free now): accumulators -> LDS
            // (16-byte slot s of row r at s ^ (r & 7): the 8 rows a ds_write_b128 group touches land in 8 bank groups), then one wave per row:
            // 16 bytes per lane along the sequence, the 3-element halo from the neighbouring lane, 1-KB coalesced stores. The conv kernel's
            // pass over x (read + write of B D L 4 bytes and one launch per mixer) is gone; rows >= conv_rows (the z half) take the plain path.
            if (m0 < p.conv_rows) {
                constexpr int kNW = kM1 ? 4 : 8;
                constexpr int kHalfRows = kM1 ? 64 : 128;              // rows of one mi half of the tile
                float *Ct = reinterpret_cast<float *>(p.C) + (int64_t)m0 * p.ldc + n0;
                const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, 0x7fffffff, 0x00020000);
                const int W = p.conv_width;
                // all scale loads before the first store, unconditional (see the plain path below): 8 row scales, 4 column vectors per lane
                const bool has_s = p.sa != nullptr;
                const unsigned ms = pick_mask(has_s);
                const float *sap = has_s ? p.sa + m0 : reinterpret_cast<const float *>(p.A), *sbp = has_s ? p.sb : reinterpret_cast<const float *>(p.B0);
                float csa[2][4];
                f4 csb[2][2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        csa[mi][i] = pick(ms, sap[mi * kMiRows + wr * 64 + i * 16 + (lane & 15)], 1.0f);
                    }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = n0 + ni * 128 + ecol + j * 16;
                        csb[ni][j] = pick(ms, *reinterpret_cast<const f4 *>(sbp + (col < p.N ? col : 0)), f4{1.f, 1.f, 1.f, 1.f});
                    }
                // the taps and the bias of the wave's 16 rows of each half: lane l holds tap l / 16 of row l % 16 (one load per half, before the first
                // store); the row loop below reads them with v_readlane -- loads inside it would each wait behind the previous row's store
                float ctap[2], cbia[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int row = m0 + mi * kMiRows + w * (kHalfRows / kNW) + (lane & 15);
                    ctap[mi] = pick(pick_mask_lanes((lane >> 4) < W), p.conv_w[(int64_t)row * p.conv_w_ld + ((lane >> 4) < W ? (lane >> 4) : 0)], 0.f);
                    cbia[mi] = pick(pick_mask(p.conv_b != nullptr), (p.conv_b ? p.conv_b : reinterpret_cast<const float *>(p.B0))[row], 0.f);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    if (mi) __syncthreads();                            // the first half's rows have been read
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int tcol = ni * 128 + ecol + j * 16;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int rl = wr * 64 + i * 16 + (lane & 15);             // row inside the half
                                const f4 x = acc[mi][ni][i][j] * (csb[ni][j] * csa[mi][i]);
                                *reinterpret_cast<f4 *>(lds + rl * 1024 + (((tcol >> 2) ^ (rl & 7)) << 4)) = x;
                            }
                        }
                    __syncthreads();
                    // wave w: rows 16 w .. 16 w + 15 of the half (kHalfRows / kNW = 16), lane l: tokens 4 l .. 4 l + 3
#pragma unroll 4
                    for (int it = 0; it < kHalfRows / kNW; ++it) {
                        const int rl = w * (kHalfRows / kNW) + it, row = m0 + mi * kMiRows + rl;
                        const f4 v = *reinterpret_cast<const f4 *>(lds + rl * 1024 + ((lane ^ (rl & 7)) << 4));
                        float w4[4];                                    // taps right-aligned into 4 slots (w4[3] multiplies x[t])
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            w4[k] = (k >= 4 - W) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ctap[mi]), it + 16 * (k - (4 - W)))) : 0.f;
                        const float cbias = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cbia[mi]), it));
                        // x[t-1], x[t-2], x[t-3] of the lane's first token: the neighbouring lane's last three, zeros at a sequence start
                        const bool start = ((4 * lane) % p.conv_seq) == 0;
                        // (ds_bpermute: the three wave_shr:1 DPP moves this wants were merged into ONE move of the wrong register by hipcc 7.2 --
                        // ISA checked -- whatever their `old` operand)
                        float p1 = __shfl_up(v[3], 1, 64), p2 = __shfl_up(v[2], 1, 64), p3 = __shfl_up(v[1], 1, 64);
                        p1 = start ? 0.f : p1; p2 = start ? 0.f : p2; p3 = start ? 0.f : p3;
                        f4 y;
                        y[0] = fmaf(w4[0], p3, fmaf(w4[1], p2, fmaf(w4[2], p1, fmaf(w4[3], v[0], cbias))));
                        y[1] = fmaf(w4[0], p2, fmaf(w4[1], p1, fmaf(w4[2], v[0], fmaf(w4[3], v[1], cbias))));
                        y[2] = fmaf(w4[0], p1, fmaf(w4[1], v[0], fmaf(w4[2], v[1], fmaf(w4[3], v[2], cbias))));
                        y[3] = fmaf(w4[0], v[0], fmaf(w4[1], v[1], fmaf(w4[2], v[2], fmaf(w4[3], v[3], cbias))));
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = y[e] * fast_rcp(1.0f + fast_exp(-y[e]));          // SiLU (the conv kernel's sigmoidf_fast)
                        if (n0 + 4 * lane < p.N) store_f4<2>(crsrc, (unsigned)(((mi * kMiRows + rl) * p.ldc + 4 * lane) * 4), y);
                        DIMSUM_PACE_STORES();
                    }
                }
                return;
            }
        }
        // one buffer descriptor per tile (base = the tile's first element: wave-uniform), a 32-bit byte offset per lane
        constexpr int kAux = ((kVar & kVarNtStores) ? 2 : 0) | ((kVar & kVarSc1Stores) ? 16 : 0) | ((kVar & kVarSc0Stores) ? 1 : 0);
        float *Ct = reinterpret_cast<float *>(p.C) + (int64_t)m0 * p.ldc + n0 + (kBT ? (int64_t)split * p.c_split_stride : (int64_t)0);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, 0x7fffffff, 0x00020000);
        if constexpr (kVar & kVarFullLineStores) {
            // 128-byte row segments per store: lanes r and r + 8 of a 16-lane row trade their j = 1 / j = 0 registers (row_ror:8), after
            // which a lane owns (row r, j = lane bit 3) for the first store and (row r + 8, same columns) for the second
            const bool up = (lane & 8) != 0;
            const int frow = wr * 64 + (lane & 7);
            const int fcol = wc * 32 + (lane >> 4) * 4 + (up ? 16 : 0);
            // Every load of the epilogue is issued BEFORE the stores it could otherwise queue behind: loads and stores share vmcnt, so a load issued
            // after a store is waited for with vmcnt(0) -- the round-4 form (two row-scale loads per 16-row step, between the stores) drained
            // the store queue 16 times per tile, ~2 us each with the matrix pipe idle. The per-column vectors and the 16 row scales are
            // loaded up front; the residual rows (kEpiF32GateRes) run one group of four 16-row steps ahead of the stores.
            // (scale loads are unconditional -- without scales they read a valid address and the value is replaced by 1: a branch around a load
            // makes hipcc wait vmcnt(0) at every later use)
            const bool has_s = kRebase || p.sa != nullptr;
            const unsigned ms = pick_mask(has_s);
            const bool has_sb = has_s && !kRowFac;          // (row factors: no column scales -- the B rows' scales are inside the factors)
            const unsigned msb = pick_mask(has_sb);
            const float *sap = p.sa ? p.sa + m0 : reinterpret_cast<const float *>(p.A), *sbp = has_sb ? p.sb : reinterpret_cast<const float *>(p.B0);
            float c_scale = 1.0f;
            if constexpr (kRowFac) c_scale = p.k_inv_a ? c_scale_own : *p.c_scale;           // (wave-uniform: a scalar load, before the stores)
            bool live[2];
            f4 bv[2], sbv[2], gv[2];
            auto load_cols = [&](int ni) {                 // (ni = 1's vectors are loaded under group 1, before its stores: 12 registers less at the peak)
                const int col = ni * 128 + fcol;
                live[ni] = n0 + col < p.N;                 // (the lane exchange below runs in every lane: only the stores are predicated)
                const int lc = live[ni] ? n0 + col : 0;    // dead lanes load a valid column (no branch around the loads)
                bv[ni] = f4{0.f, 0.f, 0.f, 0.f}; sbv[ni] = gv[ni] = f4{1.f, 1.f, 1.f, 1.f};
                if constexpr (kEpi == kEpiF32Bias) bv[ni] = *reinterpret_cast<const f4 *>(p.bias0 + lc);
                if constexpr (kEpi == kEpiF32GateRes) {    // the residual tail of a block: out = res + gate * (x W^T + b), one pass less
                    bv[ni] = pick(pick_mask(p.bias0 != nullptr), *reinterpret_cast<const f4 *>((p.bias0 ? p.bias0 : reinterpret_cast<const float *>(p.B0)) + lc), bv[ni]);
                    gv[ni] = pick(pick_mask(p.gate != nullptr),
                                  *reinterpret_cast<const f4 *>((p.gate ? p.gate + (int64_t)(m0 / p.rows_per_batch) * p.ldg : reinterpret_cast<const float *>(p.B0)) + lc), gv[ni]);
                }
                sbv[ni] = pick(msb, *reinterpret_cast<const f4 *>(sbp + lc), sbv[ni]);
            };
            load_cols(0);
            live[1] = n0 + 128 + fcol < p.N;
            float rs[2][4][2];
            if constexpr (!kPersist || kRebase || kEpi != kEpiF32GateRes)          // (persistent + residual: the row scales ride with the residual groups)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int lrow = frow + mi * kMiRows + i * 16;
                    if constexpr (kRebase) rs[mi][i][0] = rs[mi][i][1] = top[mi * 2 + (i >> 1)];
                    else if constexpr (kRowFac) {           // (NN: A's rows are OUTPUT rows and may carry their own scale)
                        rs[mi][i][0] = c_scale * pick(ms, sap[lrow], 1.0f); rs[mi][i][1] = c_scale * pick(ms, sap[lrow + 8], 1.0f);
                    }
                    else {                                  // scaled-fp16 operands: exact powers of two
                        rs[mi][i][0] = pick(ms, sap[lrow], 1.0f); rs[mi][i][1] = pick(ms, sap[lrow + 8], 1.0f);
                    }
                }
            // kEpiF32GateRes: the residual rows run one group of kGS 16-row steps ahead of the stores (step t = (2 ni + mi) 4 + i; the persistent kernel
            // keeps its loop state in registers through the epilogue and affords 2 steps = 32 registers, the plain one 4)
            constexpr int kGS = kPersist ? DIMSUM_PERSIST_GS : 4, kNG = 16 / kGS;
            f4 rr[2][kGS][2];
            float rg[2][kGS][2];
            auto load_res = [&](int g) {
                if constexpr (kEpi == kEpiF32GateRes) {
                    const int t0 = g * kGS, ni = t0 >> 3, mi = (t0 >> 2) & 1;          // (a group lies inside one (ni, mi) quadrant)
                    if constexpr (kPersist) {
#pragma unroll
                        for (int k = 0; k < kGS; ++k) {
                            const int lrow = frow + mi * kMiRows + ((t0 + k) & 3) * 16;
                            rg[g & 1][k][0] = pick(ms, sap[lrow], 1.0f); rg[g & 1][k][1] = pick(ms, sap[lrow + 8], 1.0f);
                        }
                    }
                    const float *rp = p.res + (int64_t)(m0 + frow + mi * kMiRows) * p.ldr + (live[ni] ? n0 + ni * 128 + fcol : 0);
#pragma unroll
                    for (int k = 0; k < kGS; ++k) {
                        const int i = (t0 + k) & 3;
                        rr[g & 1][k][0] = *reinterpret_cast<const f4 *>(rp + (int64_t)(i * 16) * p.ldr);
                        rr[g & 1][k][1] = *reinterpret_cast<const f4 *>(rp + (int64_t)(i * 16 + 8) * p.ldr);
                    }
                }
            };
            load_res(0);
#pragma unroll
            for (int g = 0; g < kNG; ++g) {
                if (g + 1 < kNG) load_res(g + 1);
                if (g * kGS == 4) load_cols(1);                // (under ni = 0's second half, before its stores: ahead of its first use)
#pragma unroll
                for (int k = 0; k < kGS; ++k) {
                    const int t = g * kGS + k, ni = t >> 3, mi = (t >> 2) & 1, i = t & 3;
                    const int col = ni * 128 + fcol;
                    const f4 x0 = acc[mi][ni][i][0], x1 = acc[mi][ni][i][1];
                    f4 s0, s1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r0 = dpp_row_ror8(x0[e]), r1 = dpp_row_ror8(x1[e]);
                        s0[e] = up ? r1 : x0[e];
                        s1[e] = up ? x1[e] : r0;
                    }
                    const int lrow = frow + mi * kMiRows + i * 16;
                    const unsigned voff = (unsigned)((lrow * p.ldc + col) * 4);
                    constexpr bool kRowsInGroup = kPersist && kEpi == kEpiF32GateRes;
                    s0 = s0 * (sbv[ni] * (kRowsInGroup ? rg[g & 1][k][0] : rs[mi][i][0]));           // (without scales: 1 x 1, exact -- a branch here made hipcc wait vmcnt(0) in every step)
                    s1 = s1 * (sbv[ni] * (kRowsInGroup ? rg[g & 1][k][1] : rs[mi][i][1]));
                    if constexpr (kEpi == kEpiF32GateRes) {
                        s0 = rr[g & 1][k][0] + gv[ni] * (s0 + bv[ni]);
                        s1 = rr[g & 1][k][1] + gv[ni] * (s1 + bv[ni]);
                    } else {
                        s0 = s0 + bv[ni];
                        s1 = s1 + bv[ni];
                    }
                    if (live[ni]) {
                        store_f4<kAux>(rsrc, voff, s0);
                        store_f4<kAux>(rsrc, voff + (unsigned)(8 * p.ldc * 4), s1);
                    }
                    if constexpr (kEpi != kEpiF32GateRes) DIMSUM_PACE_STORES();
                }
            }
        } else {
            const int frow = wr * 64 + (lane & 15);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = ni * 128 + ecol + j * 16;
                    if (n0 + col < p.N) {
                        f4 bv = f4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (kEpi == kEpiF32Bias) bv = *reinterpret_cast<const f4 *>(p.bias0 + n0 + col);
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                store_f4<kAux>(rsrc, (unsigned)(((frow + mi * kMiRows + i * 16) * p.ldc + col) * 4), acc[mi][ni][i][j] + bv);
                    }
                }
        }
    } else if constexpr (kEpi == kEpiF16Qkv) {
        // q | k | v of the attention fusion as scaled fp16 (include/dimsum_hip.h, DIMSUM_GEMM_EPI_F16_QKV): the attention kernel reads half the
        // bytes and stages K / V without a conversion. Scales from the bound |x W^T + b| <= 2^15 a_inv wl1 + bmax: q per row, k / v per
        // batch element (the maximum of a_inv over the element's rows: a tile lies inside one element).
        constexpr int kNW = kM1 ? 4 : 8;
        float *red = reinterpret_cast<float *>(lds);              // (the ring is free: no DMA pending, every operand read retired)
        // every load of the epilogue is issued here, in one batch (the round-4 form paid a memory round trip for the batch maximum, one for the
        // row scales and one per column vector: 6 in a row)
        const float wl1 = p.gate_bound[0], bmax = p.gate_bound[1];
        float row_sa[2][4], row_qs[2][4];
        f4 cbv[2][2], csb[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) row_sa[mi][i] = p.sa[m0 + mi * kMiRows + wr * 64 + i * 16 + (lane & 15)];
        {
            const unsigned mbias = pick_mask(p.bias0 != nullptr);
            const float *b0p = p.bias0 ? p.bias0 : p.sb;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + ni * 128 + ecol + j * 16, lc = col < p.N ? col : 0;
                    cbv[ni][j] = pick(mbias, *reinterpret_cast<const f4 *>(b0p + lc), f4{0.f, 0.f, 0.f, 0.f});
                    csb[ni][j] = *reinterpret_cast<const f4 *>(p.sb + lc);
                }
        }
        {
            const float *sab = p.sa + (int64_t)(m0 / p.rows_per_batch) * p.rows_per_batch;
            float mloc = 0.f;
            for (int t = threadIdx.x; t < p.rows_per_batch; t += kNW * 64) mloc = fmaxf(mloc, sab[t]);
            mloc = wave_allmax(mloc);
            if (lane == 0) red[w] = mloc;
        }
        __syncthreads();
        float mb = red[0];
#pragma unroll
        for (int i = 1; i < kNW; ++i) mb = fmaxf(mb, red[i]);
        __syncthreads();                                            // (the staging below overwrites `red`)
        float kv_scale, kv_inv;
        f16s_scales(2.0f * (32768.0f * mb * wl1 + bmax), kv_scale, kv_inv);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float qinv;
                f16s_scales(2.0f * (32768.0f * row_sa[mi][i] * wl1 + bmax), row_qs[mi][i], qinv);
            }
        // the tile through LDS: [kTileM rows][256 cols] fp16, 512 B per row; the 8-byte unit u (4 columns) of row r sits at u ^ (r & 15)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tcol = ni * 128 + ecol + j * 16, col = n0 + tcol;
                const f4 bv = cbv[ni][j], sbv = csb[ni][j];       // (columns past N hold column 0's values: computed, never stored)
                const bool is_q = col < p.q_cols;
                const int unit = tcol >> 2;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = mi * kMiRows + wr * 64 + i * 16 + (lane & 15);
                        const f4 x = acc[mi][ni][i][j] * (sbv * row_sa[mi][i]) + bv;
                        const float sc = is_q ? row_qs[mi][i] : kv_scale;
                        const __half2 a = __floats2half2_rn(x[0] * sc, x[1] * sc), b = __floats2half2_rn(x[2] * sc, x[3] * sc);
                        *reinterpret_cast<uint2 *>(lds + r * 512 + ((unit ^ (lane & 15)) << 3)) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
                    }
            }
        __syncthreads();
        // wave w stores rows 32 w .. 32 w + 31, two rows (512 B each) per instruction
        char *Ct = reinterpret_cast<char *>(p.C) + ((int64_t)m0 * p.ldc + n0) * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, 0x7fffffff, 0x00020000);
        const int piece = lane & 31;
        const bool live = n0 + piece * 8 < p.N;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = w * 32 + it * 2 + (lane >> 5);
            const char *src = lds + r * 512 + ((piece ^ ((r & 15) >> 1)) << 4);
            u32x4 v = *reinterpret_cast<const u32x4 *>(src);
            v = (r & 1) ? u32x4{v[2], v[3], v[0], v[1]} : v;
            if (live) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (unsigned)((r * p.ldc + piece * 8) * 2), 0, 2);
        }
    } else {
        // gated GeLU: h = gelu_tanh(x1 + b1) (x2 + b2) with x1 = acc[mi][0], x2 = acc[mi][1] of the same hidden column. The 256 x 128 tile of h
        // goes through LDS (free now: every wave's operand reads retired before the barrier it passed last, no DMA is pending) so that
        // the global stores are whole rows: plane 0 = hi (or the fp16 image), plane 1 = lo, [256 rows][128 cols] 16-bit, 256 B per row.
        // The 8-byte unit u of row r sits at u ^ (r & 15): the 16 lanes of a ds_write_b64 service group (same columns, 16 rows)
        // hit 16 distinct units, and a 16-byte read (units 2 v, 2 v + 1) finds both in slot v ^ ((r & 15) >> 1), swapped when r is odd.
        constexpr bool kImg = kEpi == kEpiGatedSplit3;
        // scaled-fp16 operands: the inverse scale of each of this lane's 8 rows, and (fp16 image of h) the row's output scale from the bound
        //   |x1|, |x2| <= max|a_r| * max_n sum_k |w_nk| + max|b|   (max|a_r| < 2^15 sa[r] by the image's construction), |gelu(x)| <= |x|:
        // no reduction over the row -- every column tile derives the same power of two. The factor 4 covers the fp16 rounding of the
        // operands and the fp32 accumulation many times over.
        // ALL loads of the epilogue in one batch, unconditional (dead columns / absent operands read a valid address and the value is replaced:
        // see pick()), before the first store: the round-4 form went to memory 8 times in a row for the row scales (load, wait, store inv_out,
        // load, ..) and twice more for the column vectors -- ten serial round trips per tile with the matrix pipe idle
        const bool has_s = p.sa != nullptr, has_b = p.bias0 != nullptr;
        const unsigned ms = pick_mask(has_s), mb = pick_mask(has_b);
        const float *sap = has_s ? p.sa + m0 : reinterpret_cast<const float *>(p.A);
        const float *sbp = has_s ? p.sb : reinterpret_cast<const float *>(p.B0), *b0p = has_b ? p.bias0 : reinterpret_cast<const float *>(p.B0), *b1p = has_b ? p.bias1 : b0p;
        const int sb2_off = has_s ? p.N : 0;
        float row_sa[2][4], row_hs[2][4];
        f4 cbv1[2], cbv2[2], csb1[2], csb2[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) row_sa[mi][i] = pick(ms, sap[mi * kMiRows + wr * 64 + i * 16 + (lane & 15)], 1.0f);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + ecol + j * 16, lc = col < p.N ? col : 0;
            const f4 zero = f4{0.f, 0.f, 0.f, 0.f}, one = f4{1.f, 1.f, 1.f, 1.f};
            cbv1[j] = pick(mb, *reinterpret_cast<const f4 *>(b0p + lc), zero);
            cbv2[j] = pick(mb, *reinterpret_cast<const f4 *>(b1p + lc), zero);
            csb1[j] = pick(ms, *reinterpret_cast<const f4 *>(sbp + lc), one);
            csb2[j] = pick(ms, *reinterpret_cast<const f4 *>(sbp + sb2_off + lc), one);
        }
        const bool bound = !kImg && has_s && p.gate_bound != nullptr;
        const float gb0 = bound ? p.gate_bound[0] : 0.f, gb1 = bound ? p.gate_bound[1] : 0.f;        // (wave-uniform: scalar loads)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = mi * kMiRows + wr * 64 + i * 16 + (lane & 15);
                row_hs[mi][i] = p.out_scale;
                if constexpr (!kImg) {
                    if (bound) {
                        const float xb = 32768.0f * row_sa[mi][i] * gb0 + gb1;
                        float inv;
                        f16s_scales(4.0f * xb * xb, row_hs[mi][i], inv);
                        if (tile_n == 0 && wc == 0 && lane < 16) p.inv_out[m0 + r] = inv;
                    }
                }
            }
        if constexpr ((kVar & kVarKeepX12) != 0) {
            // training forward: the backward's gated-GeLU adjoint needs x1, x2 -- stored from the accumulators (128-byte row segments, the
            // F32 epilogue's lane exchange) instead of a plain GEMM + a gate pass that reads them back (2.1 GB at 65536 x 8192)
            float *Xt = p.x12 + (int64_t)m0 * p.ldx + n0;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(Xt, 0, 0x7fffffff, 0x00020000);
            const bool up = (lane & 8) != 0;
            const int frow = wr * 64 + (lane & 7);
            const int fcol = wc * 32 + (lane >> 4) * 4 + (up ? 16 : 0);
            const bool xlive = n0 + fcol < p.N;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (scaled-fp16 operands: the true x1 / x2 = accumulator x row scale x column scale -- applied before the lane exchange,
                        // where a lane still holds its own row; 1 x 1 without scales)
                        const f4 x0 = acc[mi][ni][i][0] * ((ni ? csb2[0] : csb1[0]) * row_sa[mi][i]), x1 = acc[mi][ni][i][1] * ((ni ? csb2[1] : csb1[1]) * row_sa[mi][i]);
                        f4 s0, s1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r0 = dpp_row_ror8(x0[e]), r1 = dpp_row_ror8(x1[e]);
                            s0[e] = up ? r1 : x0[e];
                            s1[e] = up ? x1[e] : r0;
                        }
                        const int lrow = frow + mi * kMiRows + i * 16;
                        const unsigned voff = (unsigned)((lrow * p.ldx + ni * p.N + fcol) * 4);
                        if (xlive) {
                            store_f4<2>(xrsrc, voff, s0);
                            store_f4<2>(xrsrc, voff + (unsigned)(8 * p.ldx * 4), s1);
                        }
                    }
        }
        // persistent kernel: the ring holds the next tile's first K tiles -- the h tile goes through the 32 KB behind the ring instead, one 128-row
        // half (mi) at a time: gate, barrier, store, barrier. The plain kernel stages both halves in the (free) ring and stores once.
        static_assert(!(kPersist && kImg), "persistent: the fp16 image of h only");
        constexpr int kPasses = kPersist ? 2 : 1, kPassRows = kTileM / kPasses;
        char *stg = kPersist ? lds + 2 * kParity : lds;
        char *Ct = reinterpret_cast<char *>(p.C) + ((int64_t)m0 * p.ldc + n0) * 2;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Ct, 0, 0x7fffffff, 0x00020000);
        const int piece = lane & 15;
        const bool live = n0 + piece * 8 < p.N;
#pragma unroll
        for (int pass = 0; pass < kPasses; ++pass) {
            if (pass) __syncthreads();                     // (the first half's rows have been read)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f4 bv1 = cbv1[j], bv2 = cbv2[j], sb1 = csb1[j], sb2 = csb2[j];
                const int unit = (ecol + j * 16) >> 2;
#pragma unroll
                for (int mi = kPersist ? pass : 0; mi < (kPersist ? pass + 1 : 2); ++mi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = (kPersist ? 0 : mi * kMiRows) + wr * 64 + i * 16 + (lane & 15);       // row inside the staged rows
                        f4 x1 = acc[mi][0][i][j], x2 = acc[mi][1][i][j];
                        const float hs = row_hs[mi][i];
                        x1 = x1 * (sb1 * row_sa[mi][i]);           // (without scales: 1 x 1, exact)
                        x2 = x2 * (sb2 * row_sa[mi][i]);
                        x1 = x1 + bv1;
                        x2 = x2 + bv2;
                        float h[4];
#pragma unroll
#ifdef DIMSUM_EXP_NO_GELU
                        for (int e = 0; e < 4; ++e) h[e] = x1[e] * x2[e];
#else
                        for (int e = 0; e < 4; ++e) h[e] = gelu_tanh_f(x1[e]) * x2[e];
#endif
                        char *dst = stg + r * 256 + ((unit ^ (lane & 15)) << 3);
                        if constexpr (kImg) {
                            unsigned h0, l0, h1, l1;
                            split2(h[0], h[1], h0, l0);
                            split2(h[2], h[3], h1, l1);
                            *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
                            *reinterpret_cast<uint2 *>(dst + kTileM * 256) = make_uint2(l0, l1);
                        } else {
                            const __half2 a = __floats2half2_rn(h[0] * hs, h[1] * hs), b = __floats2half2_rn(h[2] * hs, h[3] * hs);
                            *reinterpret_cast<uint2 *>(dst) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
                        }
                    }
            }
            __syncthreads();
            // wave w stores its share (kPassRows / number of waves) of the staged rows, four rows (256 B of one plane each) per instruction
#pragma unroll
            for (int it = 0; it < kPassRows / (kM1 ? 4 : 8) / 4; ++it) {
                const int rl = w * (kPassRows / (kM1 ? 4 : 8)) + it * 4 + (lane >> 4), r = pass * kPassRows + rl;
                const char *src = stg + rl * 256 + ((piece ^ ((rl & 15) >> 1)) << 4);
                const bool odd = (rl & 1) != 0;
                const unsigned voff = (unsigned)((r * p.ldc + piece * 8) * 2);
                u32x4 v = *reinterpret_cast<const u32x4 *>(src);
                v = odd ? u32x4{v[2], v[3], v[0], v[1]} : v;
#ifdef DIMSUM_EXP_NO_GSTORE
                if (live && v[0] == 0x12345678u && v[1] == 0x9abcdef1u) {
#else
                if (live) {
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
                    if constexpr (kImg) {
                        if (!p.c_pieces2) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff + (unsigned)(p.N * 2), 0, 2);
                    }
                }
                if constexpr (kImg) {
                    u32x4 l = *reinterpret_cast<const u32x4 *>(src + kTileM * 256);
                    l = odd ? u32x4{l[2], l[3], l[0], l[1]} : l;
                    if (live) __builtin_amdgcn_raw_buffer_store_b128(l, rsrc, voff + (unsigned)(p.N * (p.c_pieces2 ? 2 : 4)), 0, 2);
                }
            }
        }
    }
    }      // (the epilogue's scope)
    if constexpr (!kPersist) break;
    else {
        if (!has_next) {
            DIMSUM_WAIT_VM(0);                            // (the streamed-ahead K tiles nobody reads must have landed before the LDS is released)
            break;
        }
        // the next output tile: its K tiles 0 and 1 are in the ring (the epilogue above did not touch LDS)
        bid = nbid; tile_m = ntile_m; tile_n = ntile_n; m0 = tile_m * kTileM; n0 = tile_n * kTileN; cur = nxt;
        resumed = true;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[mi][ni][i][j] = f4{0.f, 0.f, 0.f, 0.f};
        if (wr == 1) __builtin_amdgcn_s_barrier();       // the second wave row falls one barrier behind again
        __builtin_amdgcn_sched_barrier(0);
        DIMSUM_READ_A(a0, kSlotA0, 0);
        DIMSUM_PHASE_SYNC();
        DIMSUM_PHASE_END();
    }
    }      // for (;;)
}

// Every workgroup of a launch runs the same loop and the same epilogue: started together, all CUs reach their epilogues together and the chip's
// store path (6-7 TB/s) is idle during the K loops and the bound during the epilogues. Delaying the first round on every other CU by part
// of a tile period lets one half of the chip store while the other half multiplies; later rounds inherit the phase (a workgroup starts when
// one ends on its CU). Workgroups sharing a CU get the same delay (their waves pair up on the SIMDs: section 3.5 of DESIGN.md).
__device__ __forceinline__ void cu_stagger(const Args &p, int first_round) {
#ifdef DIMSUM_GEMM_TUNE      // (tuning builds only, like the other schedule variants: measured 0 .. +4 %, section 3.5 of DESIGN.md)
    if (p.stagger > 0 && (int)blockIdx.x < first_round) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if ((hw >> 8) & 1u)                              // CU_ID bit 0
            for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
}

template <int kOp, int kEpi, int kVar = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(const Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * kParity];
    cu_stagger(p, 256);
    gemm_body<kOp, kEpi, kVar>(p, lds);
}

// persistent (kVarPersist): the whole LDS of a CU -- the 128-KB ring and 32 KB behind it for the epilogues that stage their tile
template <int kOp, int kEpi, int kVar = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_persist_kernel(const Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * kParity + 32768];
    gemm_body<kOp, kEpi, kVar | kVarPersist>(p, lds);
}

// TN with per-reduction-row factors (kVarRowFac): the ring + 32 KB for the factors of the workgroup's reduction range
template <int kOp, int kEpi, int kVar = 0>
__global__ __launch_bounds__(512, 2) void gemm_tn_rowfac_kernel(const Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * kParity + 32768];
    gemm_body<kOp, kEpi, kVar | kVarRowFac>(p, lds);
}

template <int kOp, int kEpi, int kVar = 0>
__global__ __launch_bounds__(512, 2) void gemm_nn_rowfac_kernel(const Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * kParity + 32768];
    gemm_body<kOp, kEpi, kVar | kVarRowFac | kVarNN>(p, lds);
}

// 4 waves, 80 KB: two workgroups per CU (2 waves per SIMD: the same 256-VGPR budget)
template <int kOp, int kEpi, int kVar = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_m128_kernel(const Args p) {
    __shared__ __attribute__((aligned(1024))) char lds[10 * 8192];
    cu_stagger(p, 512);
    gemm_body<kOp, kEpi, kVar | kVarM128>(p, lds);
}

#undef DIMSUM_KTILE
#undef DIMSUM_READ_A
#undef DIMSUM_READ_TN1
#undef DIMSUM_DS_READ_TR
#undef DIMSUM_READ_B
#undef DIMSUM_QUADRANT
#undef DIMSUM_PHASE_SYNC
#undef DIMSUM_PHASE_END
#undef DIMSUM_DS_READ_B128
#undef DIMSUM_DS_READ_B64
#undef DIMSUM_WAIT_VM
#undef DIMSUM_WAIT_LGKM0

}  // namespace gemm_nt
}  // namespace dimsum
