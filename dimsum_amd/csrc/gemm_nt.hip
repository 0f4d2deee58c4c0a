// gemm_nt.hip -- C ABI of the hand-written NT GEMM (gemm_nt_kernel.hpp): plain fp32 output, and the w12 GEMM of the gated MLP with
// bias + tanh-GeLU + gate as its epilogue (dimsum/mlp.py:66-70), written directly as the operand image of the w3 GEMM.
#include "gemm_nt_kernel.hpp"

namespace dimsum {
namespace gemm_nt {

template <int kOp, int kEpi, int kVar = 0> int launch(const Args &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int *probe) {
    if (probe) { *probe = 0; return DIMSUM_OK; }
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(512);
    if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_kernel<kOp, kEpi, kVar>), grid, block, 0, s, e0, e1, 0, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<kOp, kEpi, kVar>), grid, block, 0, s, a);
    return launch_status();
}

// the 128 x 256-tile variant (4-wave workgroups, two per CU): launches whose epilogue is a large share of a tile's time (short K)
template <int kOp, int kEpi, int kVar = 0> int launch_m128(const Args &a0, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int *probe) {
    if (probe) { *probe = 1; return DIMSUM_OK; }
    Args a = a0;
    a.tiles_m = a.M / 128;
    a.group_m = a.tiles_m <= 32 ? a.tiles_m : 2 * a0.group_m;     // the same L2 patch in rows
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(256);
    if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_m128_kernel<kOp, kEpi, kVar>), grid, block, 0, s, e0, e1, 0, a);
    else hipLaunchKernelGGL((gemm_nt_m128_kernel<kOp, kEpi, kVar>), grid, block, 0, s, a);
    return launch_status();
}

// persistent workgroups (kVarPersist): one per CU (a multiple of 8: the tile walk's XCD ranges), each walking the tile list as one K stream
inline int persist_grid() {        // queried per call for the CURRENT device (no cached state: several devices per process, any thread)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus >= 8 ? cus / 8 * 8 : 8;
}
// the persistent stream needs an even number (>= 4) of K tiles, more tiles than workgroups and plain (un-aliased) operands; tune_variant 512 / 513
// forbid it (A / B runs), 514 asks for it where a launch would not take it by default (the fp32 gate + residual epilogue) -- under the same conditions
inline bool persist_ok(const Args &a, int tune_variant) {
    const int nk = a.K / kBK;
    return tune_variant != 513 && tune_variant != 512 && nk >= 4 && nk % 2 == 0 && a.tiles_m * a.tiles_n > persist_grid() && a.a_alias_tiles == 0 && a.b_alias_tiles == 0;
}
template <int kOp, int kEpi, int kVar = 0> int launch_persist(const Args &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int *probe) {
    if (probe) { *probe = 2; return DIMSUM_OK; }
    const dim3 grid((unsigned)persist_grid()), block(512);
    if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_persist_kernel<kOp, kEpi, kVar>), grid, block, 0, s, e0, e1, 0, a);
    else hipLaunchKernelGGL((gemm_nt_persist_kernel<kOp, kEpi, kVar>), grid, block, 0, s, a);
    return launch_status();
}

// scaled-fp16 operands (one product per element): both tile shapes are built
template <int kEpi, int kVar = 0> int launch_f16(const Args &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1, bool m128, int *probe) {
    return m128 ? launch_m128<kOpF16, kEpi, kVar>(a, s, e0, e1, probe) : launch<kOpF16, kEpi, kVar>(a, s, e0, e1, probe);
}

// base + extension of the public parameter struct, flattened (include/dimsum_hip.h "Versioning"): what the entry points below read
struct gemm_flat_t : dimsum_gemm_ext_t {
    int32_t m, n, k, operand_dtype, epilogue;
    float out_scale;
    int64_t lda, ldb, ldc;
    const void *a_ptr, *b_ptr, *bias_ptr;
    void *c_ptr;
    const void *a_inv_scale_ptr, *b_inv_scale_ptr;
};
static int gemm_flat_from(const dimsum_gemm_params_t *p, gemm_flat_t &f) {
    if (!p) return DIMSUM_ERR_NULL;
    if (p->struct_size != sizeof(dimsum_gemm_params_t)) return DIMSUM_ERR_ABI;
    const int rc = ext_from<dimsum_gemm_ext_t>(p->ext, f);
    if (rc != DIMSUM_OK) return rc;
    f.m = p->m; f.n = p->n; f.k = p->k; f.operand_dtype = p->operand_dtype; f.epilogue = p->epilogue; f.out_scale = p->out_scale;
    f.lda = p->lda; f.ldb = p->ldb; f.ldc = p->ldc;
    f.a_ptr = p->a_ptr; f.b_ptr = p->b_ptr; f.bias_ptr = p->bias_ptr; f.c_ptr = p->c_ptr;
    f.a_inv_scale_ptr = p->a_inv_scale_ptr; f.b_inv_scale_ptr = p->b_inv_scale_ptr;
    return DIMSUM_OK;
}

}  // namespace gemm_nt
}  // namespace dimsum

// `probe` != NULL: nothing is launched, *probe receives the kernel family (0 = 256-row tiles, 1 = 128-row tiles, 2 = persistent stream)
static int gemm_nt_run(const dimsum_gemm_params_t *pub, void *stream, int *probe) {
    using namespace dimsum;
    using namespace dimsum::gemm_nt;
    gemm_flat_t flat;
    {
        const int frc = gemm_flat_from(pub, flat);
        if (frc != DIMSUM_OK) return frc;
    }
    const gemm_flat_t *p = &flat;
    if (!p->a_ptr || !p->b_ptr || !p->c_ptr) return DIMSUM_ERR_NULL;
    if (p->operand_dtype != DIMSUM_F16 && p->operand_dtype != DIMSUM_BF16) return DIMSUM_ERR_DTYPE;
    if (p->m <= 0 || p->n <= 0 || p->k < 2 * kBK || p->m % kBM != 0 || p->k % kBK != 0 || p->n % 4 != 0) return DIMSUM_ERR_SHAPE;
    // a_alias_rows = C: the A rows are [hi | lo] pairs (2 C columns) read as the left image [hi | hi | lo] over k = 3 C
    if (p->a_alias_rows != 0 && (p->a_alias_rows < 0 || p->a_alias_rows % kBK != 0 || p->k != 3 * p->a_alias_rows)) return DIMSUM_ERR_SHAPE;
    if (p->b_alias_rows != 0 && (p->b_alias_rows < 0 || p->b_alias_rows % kBK != 0 || p->k != 3 * p->b_alias_rows || (p->epilogue != DIMSUM_GEMM_EPI_F32 && p->epilogue != DIMSUM_GEMM_EPI_F32_CONV))) return DIMSUM_ERR_SHAPE;
    if (p->lda % 8 != 0 || p->ldb % 8 != 0 || p->lda < (p->a_alias_rows ? 2 * p->a_alias_rows : p->k) || p->ldb < (p->b_alias_rows ? 2 * p->b_alias_rows : p->k) ||
        !aligned_to<char>(p->a_ptr, 16) ||
        !aligned_to<char>(p->b_ptr, 16))
        return DIMSUM_ERR_STRIDE;
    // one 32-bit byte offset per lane inside a 256-row panel
    if ((int64_t)256 * p->lda * 2 >= ((int64_t)1 << 31) || (int64_t)256 * p->ldb * 2 >= ((int64_t)1 << 31) || (int64_t)257 * p->ldc * 4 >= ((int64_t)1 << 31))
        return DIMSUM_ERR_STRIDE;
    Args a{};
    a.A = reinterpret_cast<const char *>(p->a_ptr);
    a.C = p->c_ptr;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc;
    a.M = p->m; a.K = p->k;
    a.tiles_m = p->m / kBM;
    a.out_scale = p->out_scale;
    a.a_alias_tiles = (int)(p->a_alias_rows / kBK);
    a.a_alias_from = a.a_alias_tiles;
    if (p->a_alias_weight_order) {       // the pair read as [hi | lo | hi]: the third piece re-reads the first (shift = two pieces)
        if (!p->a_alias_rows) return DIMSUM_ERR_SHAPE;
        a.a_alias_tiles *= 2;
        a.a_alias_from = a.a_alias_tiles;
    }
    a.b_alias_tiles = (int)(p->b_alias_rows / kBK);
    // tile order: groups of 4 tile rows; a matrix of few tile rows (in_proj's d-major product: the weight is the left operand) walks whole tile
    // columns, so that every streamed right-operand panel is loaded once (tools/scratch/gm_sweep.py: 187 -> 178 us at 2048 x 65536 x 512)
    a.group_m = p->tune_group_m > 0 ? p->tune_group_m : (a.tiles_m <= 16 ? a.tiles_m : 4);
    a.stagger = p->tune_reserved > 0 ? p->tune_reserved : 0;
    if ((p->a_inv_scale_ptr == nullptr) != (p->b_inv_scale_ptr == nullptr)) return DIMSUM_ERR_NULL;
    a.sa = reinterpret_cast<const float *>(p->a_inv_scale_ptr);
    a.sb = reinterpret_cast<const float *>(p->b_inv_scale_ptr);
    if (a.sb && !aligned_to<char>(a.sb, 16)) return DIMSUM_ERR_STRIDE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0 = reinterpret_cast<hipEvent_t>(p->timing_start_event), e1 = reinterpret_cast<hipEvent_t>(p->timing_stop_event);
    const bool bf = p->operand_dtype == DIMSUM_BF16;
    // tile shape: tune_variant 512 / 513 force the 128-row / 256-row tiles (A / B runs, tools/bench_gemm.py); 0 = by shape: scaled-fp16
    // operands (one product per element) with K <= 1024 spend a third to a half of a 256 x 256 tile's time in its epilogue
    const bool m128 = p->tune_variant == 512 || (p->tune_variant == 0 && !bf && p->k <= 576 && (p->epilogue == DIMSUM_GEMM_EPI_F32 || p->epilogue == DIMSUM_GEMM_EPI_F32_BIAS));
#ifndef DIMSUM_GEMM_TUNE
    if (p->tune_variant != 0 && p->tune_variant != 512 && p->tune_variant != 513 && p->tune_variant != 514) return DIMSUM_ERR_UNSUPPORTED;
#endif
    if (p->tune_variant == 512 && bf) return DIMSUM_ERR_UNSUPPORTED;       // (the 128-row tiles are built for the fp16 operands only)
    if (p->epilogue == DIMSUM_GEMM_EPI_F32_GATE_RESIDUAL) {
        if (!p->residual_ptr) return DIMSUM_ERR_NULL;
        if (p->ldc % 4 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16) || p->residual_ld % 4 != 0 || p->residual_ld < p->n ||
            !aligned_to<char>(p->residual_ptr, 16) || (p->bias_ptr && !aligned_to<char>(p->bias_ptr, 16)) ||
            (p->gate_ptr && (p->gate_ld % 4 != 0 || !aligned_to<char>(p->gate_ptr, 16))))
            return DIMSUM_ERR_STRIDE;
        if (p->gate_ptr && (p->rows_per_batch <= 0 || p->rows_per_batch % kBM != 0 || p->m % p->rows_per_batch != 0)) return DIMSUM_ERR_SHAPE;
        a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
        a.bias0 = reinterpret_cast<const float *>(p->bias_ptr);
        a.res = reinterpret_cast<const float *>(p->residual_ptr);
        a.gate = reinterpret_cast<const float *>(p->gate_ptr);
        a.ldr = p->residual_ld; a.ldg = p->gate_ld;
        a.rows_per_batch = p->gate_ptr ? p->rows_per_batch : p->m;
        a.N = p->n;
        a.tiles_n = (p->n + kBN - 1) / kBN;
        constexpr int kShipR = kVarFullLineStores | kVarNtStores;
        if (!bf && !m128 && p->tune_variant == 514 && p->n % kBN == 0 && persist_ok(a, p->tune_variant)) return launch_persist<kOpF16, kEpiF32GateRes, kShipR>(a, s, e0, e1, probe);     // (A / B only: see DESIGN 3.5)
        return bf ? launch<kOpBf16, kEpiF32GateRes, kShipR>(a, s, e0, e1, probe) : launch_f16<kEpiF32GateRes, kShipR>(a, s, e0, e1, m128, probe);
    }
    if (p->epilogue == DIMSUM_GEMM_EPI_F16_QKV) {
        if (bf || !a.sa || !p->gate_bound_ptr) return DIMSUM_ERR_NULL;
        if (p->rows_per_batch <= 0 || p->rows_per_batch % kBM != 0 || p->m % p->rows_per_batch != 0 || p->qkv_q_cols <= 0 || p->qkv_q_cols % 16 != 0 ||
            p->qkv_q_cols > p->n || p->n % 8 != 0)
            return DIMSUM_ERR_SHAPE;
        if (p->ldc % 8 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16) || (p->bias_ptr && !aligned_to<char>(p->bias_ptr, 16)) ||
            (int64_t)257 * p->ldc * 2 >= ((int64_t)1 << 31))
            return DIMSUM_ERR_STRIDE;
        a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
        a.bias0 = reinterpret_cast<const float *>(p->bias_ptr);
        a.gate_bound = reinterpret_cast<const float *>(p->gate_bound_ptr);
        a.rows_per_batch = p->rows_per_batch;
        a.q_cols = p->qkv_q_cols;
        a.N = p->n;
        a.tiles_n = (p->n + kBN - 1) / kBN;
        return launch_f16<kEpiF16Qkv>(a, s, e0, e1, p->tune_variant == 512 || (p->tune_variant == 0 && p->k <= 576), probe);
    }
    if (p->epilogue == DIMSUM_GEMM_EPI_F32_CONV) {
        if (!p->conv_weight_ptr) return DIMSUM_ERR_NULL;
        if (p->conv_rows <= 0 || p->conv_rows % kBM != 0 || p->conv_rows > p->m || p->conv_width < 2 || p->conv_width > 4 || p->conv_seq <= 0 || 256 % p->conv_seq != 0 ||
            p->conv_seq % 4 != 0 || p->n % p->conv_seq != 0 || p->conv_weight_ld < p->conv_width)
            return DIMSUM_ERR_SHAPE;
        if (p->ldc % 4 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16)) return DIMSUM_ERR_STRIDE;
        a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
        a.conv_w = reinterpret_cast<const float *>(p->conv_weight_ptr);
        a.conv_b = reinterpret_cast<const float *>(p->conv_bias_ptr);
        a.conv_rows = p->conv_rows; a.conv_width = p->conv_width; a.conv_seq = p->conv_seq; a.conv_w_ld = p->conv_weight_ld;
        a.N = p->n;
        a.tiles_n = (p->n + kBN - 1) / kBN;
        constexpr int kShipC = kVarFullLineStores | kVarNtStores;
        return bf ? launch<kOpBf16, kEpiF32Conv, kShipC>(a, s, e0, e1, probe) : launch_f16<kEpiF32Conv, kShipC>(a, s, e0, e1, p->tune_variant == 512 || (p->tune_variant == 0 && p->k <= 576), probe);
    }
    if (p->epilogue == DIMSUM_GEMM_EPI_F32 || p->epilogue == DIMSUM_GEMM_EPI_F32_BIAS) {
        if (p->ldc % 4 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16)) return DIMSUM_ERR_STRIDE;
        const bool bias = p->epilogue == DIMSUM_GEMM_EPI_F32_BIAS;
        if (bias && (!p->bias_ptr || !aligned_to<char>(p->bias_ptr, 16))) return DIMSUM_ERR_NULL;
        a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
        a.bias0 = reinterpret_cast<const float *>(p->bias_ptr);
        a.N = p->n;
        a.tiles_n = (p->n + kBN - 1) / kBN;
        constexpr int kShip = kVarFullLineStores | kVarNtStores;      // 128-byte row segments, streaming stores (tools/bench_gemm.py --tune)
        if (bias) return bf ? launch<kOpBf16, kEpiF32Bias, kShip>(a, s, e0, e1, probe) : launch_f16<kEpiF32Bias, kShip>(a, s, e0, e1, m128, probe);
#ifdef DIMSUM_GEMM_TUNE      // tuning builds only (tools/scratch/build_variant.sh ... -DDIMSUM_GEMM_TUNE): schedule / store-policy variants of the plain kernel
        if (bf) switch (p->tune_variant) {
            case 0: case 512: case 513: break;
            case 100: return launch<kOpBf16, kEpiF32, 0>(a, s, e0, e1, probe);
            case 1: return launch<kOpBf16, kEpiF32, 1>(a, s, e0, e1, probe);
            case 2: return launch<kOpBf16, kEpiF32, 2>(a, s, e0, e1, probe);
            case 4: return launch<kOpBf16, kEpiF32, 4>(a, s, e0, e1, probe);
            case 8: return launch<kOpBf16, kEpiF32, 8>(a, s, e0, e1, probe);
            case 12: return launch<kOpBf16, kEpiF32, 12>(a, s, e0, e1, probe);
            case 16: return launch<kOpBf16, kEpiF32, 16>(a, s, e0, e1, probe);
            case 3: return launch<kOpBf16, kEpiF32, 3>(a, s, e0, e1, probe);
            case 40: return launch<kOpBf16, kEpiF32, 40>(a, s, e0, e1, probe);
            case 44: return launch<kOpBf16, kEpiF32, 44>(a, s, e0, e1, probe);
            case 72: return launch<kOpBf16, kEpiF32, 72>(a, s, e0, e1, probe);
            case 104: return launch<kOpBf16, kEpiF32, 104>(a, s, e0, e1, probe);
            default: return DIMSUM_ERR_UNSUPPORTED;
        }
#endif
        return bf ? launch<kOpBf16, kEpiF32, kShip>(a, s, e0, e1, probe) : launch_f16<kEpiF32, kShip>(a, s, e0, e1, m128, probe);
    }
    if (p->epilogue == DIMSUM_GEMM_EPI_GATED_GELU_SPLIT3 || p->epilogue == DIMSUM_GEMM_EPI_GATED_GELU_F16) {
        // b_ptr: the (2 F, K) weight of w12; n = 2 F; output: (M, 3 F) bf16 left image [hi | hi | lo] or (M, F) fp16
        if (p->n % 16 != 0) return DIMSUM_ERR_SHAPE;
        const int F = p->n / 2;
        const bool img = p->epilogue == DIMSUM_GEMM_EPI_GATED_GELU_SPLIT3;
        if (p->c_image_pieces != 0 && p->c_image_pieces != 3 && !(img && p->c_image_pieces == 2)) return DIMSUM_ERR_UNSUPPORTED;
        a.c_pieces2 = p->c_image_pieces == 2;
        if (p->ldc % 8 != 0 || p->ldc < (img ? (a.c_pieces2 ? 2 : 3) : 1) * (int64_t)F || !aligned_to<char>(p->c_ptr, 16) || (int64_t)257 * p->ldc * 2 + 6 * (int64_t)F >= ((int64_t)1 << 31))
            return DIMSUM_ERR_STRIDE;
        if (p->bias_ptr && !aligned_to<char>(p->bias_ptr, 16)) return DIMSUM_ERR_STRIDE;
        a.B0 = reinterpret_cast<const char *>(p->b_ptr);
        a.B1 = a.B0 + (int64_t)F * p->ldb * 2;
        a.bias0 = reinterpret_cast<const float *>(p->bias_ptr);
        a.bias1 = a.bias0 ? a.bias0 + F : nullptr;
        a.N = F;
        a.tiles_n = (F + 127) / 128;
        if (p->gate_bound_ptr) {
            if (img || !a.sa || !p->h_inv_scale_ptr) return DIMSUM_ERR_NULL;
            a.gate_bound = reinterpret_cast<const float *>(p->gate_bound_ptr);
            a.inv_out = reinterpret_cast<float *>(p->h_inv_scale_ptr);
        }
        if (p->x12_ptr) {       // training forward: keep the fp32 [x1 | x2] for the backward: split-bf16 images, or scaled-fp16 operands with the bound-derived h scale
            const bool f16_train = !img && !bf && a.sa && p->gate_bound_ptr;
            if (!f16_train && (!img || !bf || a.sa)) return DIMSUM_ERR_UNSUPPORTED;
            if (p->x12_ld % 4 != 0 || p->x12_ld < p->n || !aligned_to<char>(p->x12_ptr, 16) || (int64_t)257 * p->x12_ld * 4 + (int64_t)p->n * 4 >= ((int64_t)1 << 31))
                return DIMSUM_ERR_STRIDE;
            a.x12 = reinterpret_cast<float *>(p->x12_ptr);
            a.ldx = p->x12_ld;
            if (f16_train) return launch<kOpF16, kEpiGatedF16, kVarKeepX12>(a, s, e0, e1, probe);
            return launch<kOpBf16, kEpiGatedSplit3, kVarKeepX12>(a, s, e0, e1, probe);
        }
        if (img) return bf ? launch<kOpBf16, kEpiGatedSplit3>(a, s, e0, e1, probe) : launch<kOpF16, kEpiGatedSplit3>(a, s, e0, e1, probe);
        if (!bf && !m128 && F % 128 == 0 && persist_ok(a, p->tune_variant)) return launch_persist<kOpF16, kEpiGatedF16>(a, s, e0, e1, probe);
        return bf ? launch<kOpBf16, kEpiGatedF16>(a, s, e0, e1, probe) : launch_f16<kEpiGatedF16>(a, s, e0, e1, m128, probe);
    }
    return DIMSUM_ERR_UNSUPPORTED;
}

extern "C" int dimsum_gemm_nt(const dimsum_gemm_params_t *p, void *stream) { return gemm_nt_run(p, stream, nullptr); }

extern "C" int dimsum_gemm_nt_kernel_for(const dimsum_gemm_params_t *p) {
    int which = -1;
    const int rc = gemm_nt_run(p, nullptr, &which);
    return rc == DIMSUM_OK ? which : -rc;
}

// dW-shaped product: C[s] (m, n) = sum over rows r in split s of A[r, :m]^T B[r, :n]; A (k, m) and B (k, n) 16-bit rows over the reduction index
extern "C" int dimsum_gemm_tn(const dimsum_gemm_params_t *pub, int32_t splits, int64_t c_split_stride, void *stream) {
    using namespace dimsum;
    using namespace dimsum::gemm_nt;
    gemm_flat_t flat;
    {
        const int frc = gemm_flat_from(pub, flat);
        if (frc != DIMSUM_OK) return frc;
    }
    const gemm_flat_t *p = &flat;
    if (!p->a_ptr || !p->b_ptr || !p->c_ptr) return DIMSUM_ERR_NULL;
    if (p->operand_dtype != DIMSUM_F16 && p->operand_dtype != DIMSUM_BF16) return DIMSUM_ERR_DTYPE;
    if (p->epilogue != DIMSUM_GEMM_EPI_F32 || p->bias_ptr) return DIMSUM_ERR_UNSUPPORTED;
    if (((p->a_inv_scale_ptr || p->a_block_inv_ptr) == 0) != (p->b_inv_scale_ptr == nullptr)) return DIMSUM_ERR_NULL;
    if ((p->a_inv_scale_ptr || p->a_block_inv_ptr) && (p->operand_dtype != DIMSUM_F16 || splits != 1 || p->tn_pair_a_cols != 0 || p->a_alias_rows != 0)) return DIMSUM_ERR_UNSUPPORTED;
    if (p->k_inv_b_ptr && !p->k_inv_a_ptr) return DIMSUM_ERR_NULL;
    if (p->k_scale_ptr || p->k_inv_a_ptr) {          // per-reduction-row factors: fp16 operands, plain rows, the factors of one range fit the 32 KB behind the ring
        if (p->k_inv_a_ptr ? (p->k_scale_ptr || p->c_scale_ptr) : !p->c_scale_ptr) return DIMSUM_ERR_NULL;      // the table + its maximum, OR the row scales (formed in the kernel)
        if (p->operand_dtype != DIMSUM_F16 || p->a_inv_scale_ptr || p->a_block_inv_ptr || p->tn_pair_a_cols != 0 || p->a_alias_rows != 0) return DIMSUM_ERR_UNSUPPORTED;
        if (splits < 1 || p->k / splits > 16384) return DIMSUM_ERR_SHAPE;
        if ((p->k_scale_ptr && (!aligned_to<char>(p->k_scale_ptr, 16) || !aligned_to<char>(p->c_scale_ptr, 4))) ||
            (p->k_inv_a_ptr && (!aligned_to<char>(p->k_inv_a_ptr, 16) || (p->k_inv_b_ptr && !aligned_to<char>(p->k_inv_b_ptr, 16)) || p->k / splits % 8 != 0)))
            return DIMSUM_ERR_STRIDE;
    }
    if (p->a_block_inv_ptr && (p->a_inv_scale_ptr || p->a_block_inv_ld < p->k / kBK || p->k > 64 * kBK)) return DIMSUM_ERR_SHAPE;
    if (p->b_inv_scale_ptr && !aligned_to<char>(p->b_inv_scale_ptr, 16)) return DIMSUM_ERR_STRIDE;
    const int row_splits = (p->tn_pair_a_cols != 0) ? splits / 3 : splits;       // (pairs: the three pieces share the row ranges)
    // (n % 256 != 0: the caller zero-pads B's rows to whole 256-column tiles -- ldb says so -- and only columns < n are stored)
    const int64_t n_pad = (p->n + kBN - 1) / kBN * kBN;
    if (splits < 1 || row_splits < 1 || p->m <= 0 || p->n <= 0 || p->m % kBM != 0 || p->n % 4 != 0 || (p->n % kBN != 0 && (p->ldb < n_pad || p->tn_pair_a_cols != 0)) ||
        p->k % ((int64_t)row_splits * kBK) != 0 || p->k / row_splits < 2 * kBK)
        return DIMSUM_ERR_SHAPE;
    if (p->lda % 8 != 0 || p->ldb % 8 != 0 || p->lda < p->m || p->ldb < p->n || !aligned_to<char>(p->a_ptr, 16) || !aligned_to<char>(p->b_ptr, 16) ||
        p->ldc % 4 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16) || (splits > 1 && (c_split_stride % 4 != 0 || c_split_stride < (int64_t)p->m * p->ldc)))
        return DIMSUM_ERR_STRIDE;
    // one 32-bit byte offset per lane inside a 64-row K tile / a 256-row output panel
    if ((int64_t)64 * p->lda * 2 + 512 >= ((int64_t)1 << 31) || (int64_t)64 * p->ldb * 2 + 512 >= ((int64_t)1 << 31) || (int64_t)257 * p->ldc * 4 >= ((int64_t)1 << 31))
        return DIMSUM_ERR_STRIDE;
    Args a{};
    a.A = reinterpret_cast<const char *>(p->a_ptr);
    a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
    a.C = p->c_ptr;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc;
    a.M = p->m; a.N = p->n; a.K = p->k / row_splits;
    a.tiles_m = p->m / kBM;
    a.tiles_n = (int)(n_pad / kBN);
    a.group_m = p->tune_group_m > 0 ? p->tune_group_m : (a.tiles_m <= 16 ? a.tiles_m : 4);
    if (p->b_alias_rows != 0 || p->a_alias_weight_order) return DIMSUM_ERR_UNSUPPORTED;
    if (p->tn_pair_a_cols != 0 || p->tn_pair_b_cols != 0) {
        // both operands as [hi | lo] pairs: k = the rows of ONE piece, splits = 3 x (row ranges); lda >= 2 m-ish is the caller's business
        if (p->tn_pair_a_cols <= 0 || p->tn_pair_b_cols <= 0 || splits % 3 != 0 || p->a_alias_rows != 0 || p->tn_pair_a_cols % 8 != 0 || p->tn_pair_b_cols % 8 != 0)
            return DIMSUM_ERR_SHAPE;
        if (p->lda < p->tn_pair_a_cols + p->m || p->ldb < p->tn_pair_b_cols + p->n) return DIMSUM_ERR_STRIDE;
        a.tn_pieces = 1;
        a.a_pair_cols = p->tn_pair_a_cols;
        a.b_pair_cols = p->tn_pair_b_cols;
    }
    if (p->a_alias_rows != 0) {
        if (splits != 1 || p->a_alias_rows < 0 || p->a_alias_rows % kBK != 0 || p->k != 3 * p->a_alias_rows) return DIMSUM_ERR_SHAPE;
        a.a_alias_tiles = (int)(p->a_alias_rows / kBK);
    }
    a.splits = splits;
    a.c_split_stride = c_split_stride;
    a.out_scale = 1.0f;
    a.sa = reinterpret_cast<const float *>(p->a_inv_scale_ptr);
    a.sb = reinterpret_cast<const float *>(p->b_inv_scale_ptr);
    a.a_block_inv = reinterpret_cast<const float *>(p->a_block_inv_ptr);
    a.a_block_inv_ld = (int)p->a_block_inv_ld;
    a.k_fac = reinterpret_cast<const _Float16 *>(p->k_scale_ptr);
    a.c_scale = reinterpret_cast<const float *>(p->c_scale_ptr);
    a.k_inv_a = reinterpret_cast<const float *>(p->k_inv_a_ptr);
    a.k_inv_b = reinterpret_cast<const float *>(p->k_inv_b_ptr);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0 = reinterpret_cast<hipEvent_t>(p->timing_start_event), e1 = reinterpret_cast<hipEvent_t>(p->timing_stop_event);
    constexpr int kShipT = kVarFullLineStores | kVarNtStores | kVarTN;
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * splits)), block(512);
    if (p->operand_dtype == DIMSUM_BF16) {
        if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_kernel<kOpBf16, kEpiF32, kShipT>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((gemm_nt_kernel<kOpBf16, kEpiF32, kShipT>), grid, block, 0, s, a);
    } else if (a.k_fac || a.k_inv_a) {
        if (e0 || e1) hipExtLaunchKernelGGL((gemm_tn_rowfac_kernel<kOpF16, kEpiF32, kShipT>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((gemm_tn_rowfac_kernel<kOpF16, kEpiF32, kShipT>), grid, block, 0, s, a);
    } else if (a.a_block_inv) {
        if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_kernel<kOpF16, kEpiF32, kShipT | kVarRebase>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((gemm_nt_kernel<kOpF16, kEpiF32, kShipT | kVarRebase>), grid, block, 0, s, a);
    } else {
        if (e0 || e1) hipExtLaunchKernelGGL((gemm_nt_kernel<kOpF16, kEpiF32, kShipT>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((gemm_nt_kernel<kOpF16, kEpiF32, kShipT>), grid, block, 0, s, a);
    }
    return launch_status();
}

// C[s] (m, n) = sum over the rows r of range s of A[0..m, r] B[r, 0..n): A rows contiguous along the reduction (NT-style), B rows over it (TN-style)
extern "C" int dimsum_gemm_nn(const dimsum_gemm_params_t *pub, int32_t splits, int64_t c_split_stride, void *stream) {
    using namespace dimsum;
    using namespace dimsum::gemm_nt;
    gemm_flat_t flat;
    {
        const int frc = gemm_flat_from(pub, flat);
        if (frc != DIMSUM_OK) return frc;
    }
    const gemm_flat_t *p = &flat;
    if (!p->a_ptr || !p->b_ptr || !p->c_ptr) return DIMSUM_ERR_NULL;
    if (p->k_inv_a_ptr ? (p->k_scale_ptr || p->c_scale_ptr) : (!p->k_scale_ptr || !p->c_scale_ptr)) return DIMSUM_ERR_NULL;      // the factor table + maximum, OR the row scales
    if (p->k_inv_b_ptr && !p->k_inv_a_ptr) return DIMSUM_ERR_NULL;
    if (p->operand_dtype != DIMSUM_F16) return DIMSUM_ERR_DTYPE;
    if (p->epilogue != DIMSUM_GEMM_EPI_F32 || p->bias_ptr || p->b_inv_scale_ptr || p->a_block_inv_ptr || p->tn_pair_a_cols != 0 || p->a_alias_rows != 0 ||
        p->b_alias_rows != 0 || p->a_alias_weight_order)
        return DIMSUM_ERR_UNSUPPORTED;
    if (splits < 1 || p->m <= 0 || p->n <= 0 || p->m % kBM != 0 || p->n % kBN != 0 || p->k % ((int64_t)splits * kBK) != 0 || p->k / splits < 2 * kBK ||
        p->k / splits > 16384)
        return DIMSUM_ERR_SHAPE;
    if (p->lda % 8 != 0 || p->ldb % 8 != 0 || p->lda < p->k || p->ldb < p->n || !aligned_to<char>(p->a_ptr, 16) || !aligned_to<char>(p->b_ptr, 16) ||
        p->ldc % 4 != 0 || p->ldc < p->n || !aligned_to<char>(p->c_ptr, 16) || (splits > 1 && (c_split_stride % 4 != 0 || c_split_stride < (int64_t)p->m * p->ldc)) ||
        (p->k_scale_ptr && (!aligned_to<char>(p->k_scale_ptr, 16) || !aligned_to<char>(p->c_scale_ptr, 4))) ||
        (p->k_inv_a_ptr && (!aligned_to<char>(p->k_inv_a_ptr, 16) || (p->k_inv_b_ptr && !aligned_to<char>(p->k_inv_b_ptr, 16)))))
        return DIMSUM_ERR_STRIDE;
    if ((int64_t)256 * p->lda * 2 >= ((int64_t)1 << 31) || (int64_t)64 * p->ldb * 2 + 512 >= ((int64_t)1 << 31) || (int64_t)257 * p->ldc * 4 >= ((int64_t)1 << 31))
        return DIMSUM_ERR_STRIDE;
    Args a{};
    a.A = reinterpret_cast<const char *>(p->a_ptr);
    a.B0 = a.B1 = reinterpret_cast<const char *>(p->b_ptr);
    a.C = p->c_ptr;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc;
    a.M = p->m; a.N = p->n; a.K = p->k / splits;
    a.tiles_m = p->m / kBM;
    a.tiles_n = p->n / kBN;
    a.group_m = p->tune_group_m > 0 ? p->tune_group_m : (a.tiles_m <= 16 ? a.tiles_m : 4);
    a.splits = splits;
    a.c_split_stride = c_split_stride;
    a.out_scale = 1.0f;
    a.sa = reinterpret_cast<const float *>(p->a_inv_scale_ptr);
    a.k_fac = reinterpret_cast<const _Float16 *>(p->k_scale_ptr);
    a.c_scale = reinterpret_cast<const float *>(p->c_scale_ptr);
    a.k_inv_a = reinterpret_cast<const float *>(p->k_inv_a_ptr);
    a.k_inv_b = reinterpret_cast<const float *>(p->k_inv_b_ptr);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0 = reinterpret_cast<hipEvent_t>(p->timing_start_event), e1 = reinterpret_cast<hipEvent_t>(p->timing_stop_event);
    constexpr int kShipN = kVarFullLineStores | kVarNtStores;
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * splits)), block(512);
    if (e0 || e1) hipExtLaunchKernelGGL((gemm_nn_rowfac_kernel<kOpF16, kEpiF32, kShipN>), grid, block, 0, s, e0, e1, 0, a);
    else hipLaunchKernelGGL((gemm_nn_rowfac_kernel<kOpF16, kEpiF32, kShipN>), grid, block, 0, s, a);
    return launch_status();
}
