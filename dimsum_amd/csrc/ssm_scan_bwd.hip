// ssm_scan_bwd.hip -- selective scan (Mamba S6) backward for gfx950.
//
// Replaces selective_scan_cuda.bwd (mamba/csrc/selective_scan/selective_scan.cpp:338-492; kernel
// selective_scan_bwd_kernel.cuh:75-489 + reverse_scan.cuh). With dt = softplus(delta + bias), a_t = exp(dt_t A),
// b_t = dt_t B_t u_t, h_t = a_t h_{t-1} + b_t, y_t = C_t.h_t + D u_t, out_z = y silu(z):
//     dy_t  = dout_t silu(z_t)                dz_t = dout_t y_t sig(z_t) (1 + z_t (1 - sig(z_t)))     (bwd_kernel.cuh:171-207)
//     dh_t  = a_{t+1} dh_{t+1} + C_t dy_t     (reverse recurrence)
//     dC_t[n] = sum_d dy_t h_t[n]             dB_t[n] = sum_d dh_t[n] dt_t u_t
//     dA[n]  += dh_t[n] dt_t (a_t h_{t-1})[n]                       with a_t h_{t-1} = h_t - b_t     (bwd_kernel.cuh:289)
//     ddt_t  = u_t s1_t + s2_t,  s1 = sum_n dh B,  s2 = sum_n dh A (h_t - b_t);   ddelta = ddt * sigmoid(delta+bias)  (:439-452)
//     du_t   = dt_t s1_t + D dy_t             dD += dy_t u_t        ddelta_bias += ddelta_t
//
// MI355X design. Same mapping as the forward (lane = channel, one wave64 = 64 channels of one batch element, sequence
// walked in registers), which needs the forward states in REVERSE order. Instead of the reference's per-row block-wide
// forward + reverse parallel scans (about 3x the arithmetic, plus 1024-way global atomic contention on dB/dC):
//   phase A  one forward sweep over the sequence stores the state at every 16-step tile boundary in a workspace
//            (B*D*L*N/16 floats = 1/2 of one activation tensor; lane-contiguous so the traffic is fully coalesced);
//   phase B  tiles are walked backwards. The N state recurrences are independent, so each tile is processed 4 states
//            at a time: forward sweep (16 steps) keeping h_t[4] in registers (64 VGPRs), then the reverse sweep over
//            the same registers. Nothing per-(t,n) ever touches memory.
//   dB / dC  are sums over the wave's 64 channels of per-lane values: they are reduced with a TRANSPOSED butterfly --
//            64 values per lane go in, one fully reduced value per lane comes out, in 6 levels of
//            v_permlane32_swap / v_permlane16_swap / DPP adds (~2 VALU ops per value instead of ~12 for 64
//            independent wave reductions) -- then ONE coalesced atomicAdd per (n, t) per wave (16 waves contend per
//            address instead of 1024 rows).
//   u, delta, dy tiles go through an XOR-swizzled LDS transpose like in the forward; dz / out_z are computed in the
//   coalesced load layout and never touch LDS.
#include "common.hpp"

namespace dimsum {

constexpr int kBT = 16;   // time steps per tile
constexpr int kBG = 4;    // states processed together

// 64 rows x 16 columns fp32, row = 4 slots of 16 B, slots XOR-swizzled by (row >> 2) & 3 (conflict-free b128 both ways)
__device__ __forceinline__ int btile_off(int row, int col4) { return row * kBT + ((col4 ^ ((row >> 2) & 3)) << 2); }

// ---- transposed butterfly: v[0..63] per lane in -> sum over the 64 lanes of v[lane] out ---------------------------------
__device__ __forceinline__ void swap32(float &x, float &y) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &x, float &y) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// NV values per lane (64 or 32). After the call v[0] holds, in lane l, the 64-lane sum of value (l % NV).
template <int NV> __device__ __forceinline__ float transposed_reduce(float *v, int lane) {
    static_assert(NV == 64 || NV == 32, "");
    if constexpr (NV == 64) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { swap32(v[i], v[i + 32]); v[i] += v[i + 32]; }     // lane bit 5 <-> value bit 5
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap16(v[i], v[i + 16]); v[i] += v[i + 16]; }         // lane bit 4 <-> value bit 4
    // in-row levels: a lane keeps the value its bit selects and receives the partner's copy of that same value, i.e. the
    // partner sends the value it does NOT keep
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                                   // row_ror:8 pairs l <-> l ^ 8
        const bool hi = lane & 8;
        const float keep = hi ? v[i + 8] : v[i], send = hi ? v[i] : v[i + 8];
        v[i] = keep + dpp<0x128>(send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                                   // row_half_mirror pairs l <-> 7 - l (bit 2 differs)
        const bool hi = lane & 4;
        const float keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
        v[i] = keep + dpp<0x141>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                                   // quad_perm [2,3,0,1] pairs l <-> l ^ 2
        const bool hi = lane & 2;
        const float keep = hi ? v[i + 2] : v[i], send = hi ? v[i] : v[i + 2];
        v[i] = keep + dpp<0x4E>(send);
    }
    const bool hi1 = lane & 1;
    const float keep1 = hi1 ? v[1] : v[0], send1 = hi1 ? v[0] : v[1];
    float r = keep1 + dpp<0xB1>(send1);                                             // quad_perm [1,0,3,2] pairs l <-> l ^ 1
    if constexpr (NV == 32) r += __shfl_xor(r, 32, kWave);     // the two half-waves hold partial sums of the same value
    return r;
}

template <typename T, int kN, bool kHasZ, bool kVec>
__global__ __launch_bounds__(kWave, 1) void ssm_scan_bwd_kernel(const dimsum_ssm_bwd_params_t q, float *__restrict__ ws) {
    const dimsum_ssm_params_t &p = q.fwd;
    __shared__ __attribute__((aligned(16))) float tU[kWave * kBT], tD[kWave * kBT], tY[kWave * kBT];   // u, dt (softplus'ed), dy
    __shared__ __attribute__((aligned(16))) float tB[kN * kBT], tC[kN * kBT];
    // per-(state, lane) persistent values live in LDS ([n][lane], conflict-free) so that the 4-state group loop can stay
    // rolled: A, the running dA sum and the reverse-recurrence carry dh
    __shared__ float sA[kN * kWave], sdA[kN * kWave], sdh[kN * kWave];

    const int lane = threadIdx.x;
    const int L = p.seqlen;
    const int dpg = p.dim / p.n_groups;
    const int tiles_per_group = (dpg + kWave - 1) / kWave;
    const int tiles_per_batch = p.n_groups * tiles_per_group;
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int b = wg / tiles_per_batch;
    const int rem = wg - b * tiles_per_batch;
    const int g = rem / tiles_per_group;
    const int d0 = g * dpg + (rem - g * tiles_per_group) * kWave;
    const int nd = min(kWave, (g + 1) * dpg - d0);
    const bool live = lane < nd;
    const int d = d0 + min(lane, nd - 1);

    const T *u_base = reinterpret_cast<const T *>(p.u_ptr) + (int64_t)b * p.u_batch_stride + (int64_t)d0 * p.u_d_stride;
    const T *dl_base = reinterpret_cast<const T *>(p.delta_ptr) + (int64_t)b * p.delta_batch_stride + (int64_t)d0 * p.delta_d_stride;
    const T *do_base = reinterpret_cast<const T *>(q.dout_ptr) + (int64_t)b * q.dout_batch_stride + (int64_t)d0 * q.dout_d_stride;
    const T *z_base = kHasZ ? reinterpret_cast<const T *>(p.z_ptr) + (int64_t)b * p.z_batch_stride + (int64_t)d0 * p.z_d_stride : nullptr;
    const T *y_base = kHasZ ? reinterpret_cast<const T *>(p.out_ptr) + (int64_t)b * p.out_batch_stride + (int64_t)d0 * p.out_d_stride : nullptr;
    T *oz_base = (kHasZ && p.out_z_ptr) ? reinterpret_cast<T *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride : nullptr;
    T *dz_base = kHasZ ? reinterpret_cast<T *>(q.dz_ptr) + (int64_t)b * q.dz_batch_stride + (int64_t)d0 * q.dz_d_stride : nullptr;
    T *du_base = reinterpret_cast<T *>(q.du_ptr) + (int64_t)b * q.du_batch_stride + (int64_t)d0 * q.du_d_stride;
    T *dd_base = reinterpret_cast<T *>(q.ddelta_ptr) + (int64_t)b * q.ddelta_batch_stride + (int64_t)d0 * q.ddelta_d_stride;
    const T *Bp = reinterpret_cast<const T *>(p.B_ptr) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T *Cp = reinterpret_cast<const T *>(p.C_ptr) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    float *dBp = reinterpret_cast<float *>(q.dB_ptr) + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride;
    float *dCp = reinterpret_cast<float *>(q.dC_ptr) + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride;
    const int u_ds = (int)p.u_d_stride, dl_ds = (int)p.delta_d_stride, do_ds = (int)q.dout_d_stride, z_ds = (int)p.z_d_stride;
    const int y_ds = (int)p.out_d_stride, oz_ds = (int)p.out_z_d_stride, dz_ds = (int)q.dz_d_stride, du_ds = (int)q.du_d_stride;
    const int dd_ds = (int)q.ddelta_d_stride;
    const int Bns = (int)p.B_dstate_stride, Cns = (int)p.C_dstate_stride;

    float A[kN];
    const float *Ap = reinterpret_cast<const float *>(p.A_ptr) + (int64_t)d * p.A_d_stride;
#pragma unroll
    for (int n = 0; n < kN; ++n) {
        A[n] = Ap[n * p.A_dstate_stride];
        sA[n * kWave + lane] = A[n]; sdA[n * kWave + lane] = 0.f; sdh[n * kWave + lane] = 0.f;
    }
    const float Dval = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d] : 0.f;
    const float bias = p.delta_bias_ptr ? reinterpret_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
    const bool softplus = p.delta_softplus != 0;
    float dD = 0.f, dbias = 0.f;
    float dtl_next = 0.f;   // dt * log2(e) of the first step of the tile processed before (later in time): a_{t+1} at the seam

    const int n_tiles = (L + kBT - 1) / kBT;
    float *wsw = ws + (int64_t)wg * n_tiles * kN * kWave;     // [tile][n][lane]
    // coalesced tile layout: 64 rows x 16 columns = 4 pieces of (16 rows x 4 lanes-per-row x 4 columns)
    const int lrow = lane >> 2, lc4 = lane & 3, lcol = lc4 * 4;

    // stage a 64 x 16 tile of `src` (rows = channels) into the swizzled LDS image `dst`; out-of-range -> 0
    auto stage = [&](const T *base, int ds, int t0, float *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            f32x4 v = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (kVec) {
                if (row < nd && t0 + lcol < L) v = widen(ld4<T>(base + (int64_t)row * ds + t0 + lcol));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (row < nd && t0 + lcol + e < L) v.v[e] = to_f32<T>(base[(int64_t)row * ds + t0 + lcol + e]);
            }
            *reinterpret_cast<f32x4 *>(&dst[btile_off(row, lc4)]) = v;
        }
    };
    auto stage_bc = [&](int t0) {
        for (int idx = lane; idx < kN * 4; idx += kWave) {
            const int n = idx >> 2, c4 = idx & 3;
            f32x4 vb = {{0.f, 0.f, 0.f, 0.f}}, vc = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (kVec) {
                if (t0 + c4 * 4 < L) { vb = widen(ld4<T>(Bp + (int64_t)n * Bns + t0 + c4 * 4)); vc = widen(ld4<T>(Cp + (int64_t)n * Cns + t0 + c4 * 4)); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t0 + c4 * 4 + e < L) { vb.v[e] = to_f32<T>(Bp[(int64_t)n * Bns + t0 + c4 * 4 + e]); vc.v[e] = to_f32<T>(Cp[(int64_t)n * Cns + t0 + c4 * 4 + e]); }
            }
            *reinterpret_cast<f32x4 *>(&tB[n * kBT + c4 * 4]) = vb;
            *reinterpret_cast<f32x4 *>(&tC[n * kBT + c4 * 4]) = vc;
        }
    };
    // in-place: tD <- softplus(delta + bias) (0 beyond L so that dead steps are identities: a = 1, b = 0)
    auto finish_dt = [&](int t0) {
#pragma unroll
        for (int j = 0; j < kBT / 4; ++j) {
            f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v = d4.v[s] + bias;
                if (softplus) v = softplus_ref(v);
                d4.v[s] = (t0 + j * 4 + s < L) ? v : 0.f;
            }
            *reinterpret_cast<f32x4 *>(&tD[btile_off(lane, j)]) = d4;
        }
    };

    // =============================== phase A: tile-boundary states ===============================================
    {
        float h[kN];
#pragma unroll
        for (int n = 0; n < kN; ++n) h[n] = 0.f;
        for (int tile = 0; tile < n_tiles; ++tile) {
            const int t0 = tile * kBT;
#pragma unroll
            for (int n = 0; n < kN; ++n) wsw[((int64_t)tile * kN + n) * kWave + lane] = h[n];
            if (tile == n_tiles - 1) break;                    // the last tile's end state is not needed
            stage(u_base, u_ds, t0, tU);
            stage(dl_base, dl_ds, t0, tD);
            stage_bc(t0);
            finish_dt(t0);
#pragma unroll 1
            for (int j = 0; j < kBT / 4; ++j) {
                const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, j)]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
                float dtl[4], du[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { dtl[s] = d4.v[s] * kLog2e; du[s] = d4.v[s] * u4.v[s]; }
#pragma unroll
                for (int n = 0; n < kN; ++n) {
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(&tB[n * kBT + j * 4]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) h[n] = fmaf(fast_exp2(dtl[s] * A[n]), h[n], bq.v[s] * du[s]);
                }
            }
        }
    }

    // =============================== phase B: tiles in reverse ===================================================
    for (int tile = n_tiles - 1; tile >= 0; --tile) {
        const int t0 = tile * kBT;
        stage(u_base, u_ds, t0, tU);
        stage(dl_base, dl_ds, t0, tD);
        stage_bc(t0);
        // dy = dout * silu(z), dz, optional out_z -- in the coalesced layout; dy goes to LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            f32x4 dy = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + lcol + e;
                if (row < nd && t < L) {
                    const float go = to_f32<T>(do_base[(int64_t)row * do_ds + t]);
                    if constexpr (kHasZ) {
                        const float zv = to_f32<T>(z_base[(int64_t)row * z_ds + t]), yv = to_f32<T>(y_base[(int64_t)row * y_ds + t]);
                        const float sg = sigmoidf_fast(zv), silu = zv * sg;
                        dz_base[(int64_t)row * dz_ds + t] = from_f32<T>(go * yv * sg * (1.0f + zv * (1.0f - sg)));
                        if (oz_base) oz_base[(int64_t)row * oz_ds + t] = from_f32<T>(yv * silu);
                        dy.v[e] = go * silu;
                    } else {
                        dy.v[e] = go;
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(&tY[btile_off(row, lc4)]) = dy;
        }
        finish_dt(t0);

        float s1[kBT], s2[kBT];
#pragma unroll
        for (int t = 0; t < kBT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
        float dtl_first = 0.f;

#pragma unroll 1
        for (int n0 = 0; n0 < kN; n0 += kBG) {
            float H[kBT * kBG];       // [t][k]: h_t of state n0+k, later overwritten by the dB terms
            float hk[kBG], Ak[kBG], dAk[kBG], dhk[kBG];
#pragma unroll
            for (int k = 0; k < kBG; ++k) {
                const int n = min(n0 + k, kN - 1);
                hk[k] = wsw[((int64_t)tile * kN + n) * kWave + lane];
                Ak[k] = sA[n * kWave + lane]; dAk[k] = sdA[n * kWave + lane]; dhk[k] = sdh[n * kWave + lane];
            }
            // ---- forward sweep -------------------------------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < kBT / 4; ++j) {
                const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, j)]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
#pragma unroll
                for (int k = 0; k < kBG; ++k) {
                    const int n = min(n0 + k, kN - 1);
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(&tB[n * kBT + j * 4]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float dtl = d4.v[s] * kLog2e, du = d4.v[s] * u4.v[s];
                        hk[k] = fmaf(fast_exp2(dtl * Ak[k]), hk[k], bq.v[s] * du);
                        H[(j * 4 + s) * kBG + k] = hk[k];
                    }
                }
            }
            // ---- dC[n, t] = sum_d dy_t h_t[n]: two transposed reductions of 32 values (k pair x 16 steps) ----------
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float v[32];
#pragma unroll
                for (int j = 0; j < kBT / 4; ++j) {
                    const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, j)]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            v[kk * 16 + j * 4 + s] = live ? y4.v[s] * H[(j * 4 + s) * kBG + half * 2 + kk] : 0.f;
                }
                const float r = transposed_reduce<32>(v, lane);
                const int n = n0 + half * 2 + ((lane >> 4) & 1), t = t0 + (lane & 15);
                if (lane < 32 && n < kN && t < L) atomicAdd(dCp + (int64_t)n * q.dC_dstate_stride + t, r);
            }
            // ---- reverse sweep -------------------------------------------------------------------------------------
#pragma unroll
            for (int j = kBT / 4 - 1; j >= 0; --j) {
                const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, j)]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
                const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, j)]);
                float dnext[4];       // dt * log2e of step t+1
#pragma unroll
                for (int s = 0; s < 3; ++s) dnext[s] = d4.v[s + 1] * kLog2e;
                if (j == kBT / 4 - 1) dnext[3] = dtl_next;
                else dnext[3] = (*reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j + 1)])).v[0] * kLog2e;
                if (j == 0) dtl_first = d4.v[0] * kLog2e;
#pragma unroll
                for (int k = 0; k < kBG; ++k) {
                    const int n = min(n0 + k, kN - 1);
                    const bool nlive = n0 + k < kN;
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(&tB[n * kBT + j * 4]);
                    const f32x4 cq = *reinterpret_cast<const f32x4 *>(&tC[n * kBT + j * 4]);
#pragma unroll
                    for (int s = 3; s >= 0; --s) {
                        const int t = j * 4 + s;
                        const float du = d4.v[s] * u4.v[s];
                        const float a_next = fast_exp2(dnext[s] * Ak[k]);
                        const float dhn = nlive ? fmaf(a_next, dhk[k], cq.v[s] * y4.v[s]) : 0.f;
                        dhk[k] = dhn;
                        const float ah = H[t * kBG + k] - bq.v[s] * du;           // a_t h_{t-1}
                        const float gterm = dhn * ah;
                        dAk[k] = fmaf(gterm, d4.v[s], dAk[k]);
                        s2[t] = fmaf(gterm, Ak[k], s2[t]);
                        s1[t] = fmaf(dhn, bq.v[s], s1[t]);
                        H[t * kBG + k] = dhn * du;                                 // dB term
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kBG; ++k)
                if (n0 + k < kN) { sdA[(n0 + k) * kWave + lane] = dAk[k]; sdh[(n0 + k) * kWave + lane] = dhk[k]; }
            // ---- dB[n, t] = sum_d dh_t[n] dt_t u_t -----------------------------------------------------------------
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float v[32];
#pragma unroll
                for (int t = 0; t < kBT; ++t)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) v[kk * 16 + t] = live ? H[t * kBG + half * 2 + kk] : 0.f;
                const float r = transposed_reduce<32>(v, lane);
                const int n = n0 + half * 2 + ((lane >> 4) & 1), t = t0 + (lane & 15);
                if (lane < 32 && n < kN && t < L) atomicAdd(dBp + (int64_t)n * q.dB_dstate_stride + t, r);
            }
        }
        dtl_next = dtl_first;

        // ---- per-(d, t) results: du, ddelta (softplus chain), dD, ddelta_bias; through LDS for coalesced stores ----
#pragma unroll
        for (int j = 0; j < kBT / 4; ++j) {
            const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, j)]);
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
            const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, j)]);
            f32x4 du4, dd4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int t = j * 4 + s;
                du4.v[s] = fmaf(d4.v[s], s1[t], Dval * y4.v[s]);
                dD = fmaf(y4.v[s], u4.v[s], dD);
                dd4.v[s] = fmaf(u4.v[s], s1[t], s2[t]);
            }
            *reinterpret_cast<f32x4 *>(&tU[btile_off(lane, j)]) = du4;
            *reinterpret_cast<f32x4 *>(&tY[btile_off(lane, j)]) = dd4;
        }
        // softplus derivative needs the raw delta again: re-stage it (bwd_kernel.cuh:439-452 reloads it too)
        stage(dl_base, dl_ds, t0, tD);
#pragma unroll
        for (int j = 0; j < kBT / 4; ++j) {
            const f32x4 r4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, j)]);
            f32x4 dd4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, j)]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float raw = r4.v[s] + bias;
                if (softplus && raw <= 20.0f) dd4.v[s] *= sigmoidf_fast(raw);
                if (t0 + j * 4 + s < L) dbias += dd4.v[s];
            }
            *reinterpret_cast<f32x4 *>(&tY[btile_off(lane, j)]) = dd4;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&tU[btile_off(row, lc4)]);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(&tY[btile_off(row, lc4)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + lcol + e;
                if (row < nd && t < L) {
                    du_base[(int64_t)row * du_ds + t] = from_f32<T>(a.v[e]);
                    dd_base[(int64_t)row * dd_ds + t] = from_f32<T>(c.v[e]);
                }
            }
        }
    }

    if (live) {
        float *dAp = reinterpret_cast<float *>(q.dA_ptr) + (int64_t)d * q.dA_d_stride;
#pragma unroll
        for (int n = 0; n < kN; ++n) atomicAdd(dAp + n * q.dA_dstate_stride, sdA[n * kWave + lane]);
        if (q.dD_ptr) atomicAdd(reinterpret_cast<float *>(q.dD_ptr) + d, dD);
        if (q.ddelta_bias_ptr) atomicAdd(reinterpret_cast<float *>(q.ddelta_bias_ptr) + d, dbias);
    }
}

template <typename T, int kN>
static int launch_bwd(const dimsum_ssm_bwd_params_t &q, float *ws, hipStream_t stream) {
    const dimsum_ssm_params_t &p = q.fwd;
    const int dpg = p.dim / p.n_groups;
    const int tiles = p.batch * p.n_groups * ((dpg + kWave - 1) / kWave);
    const size_t va = 4 * sizeof(T);
    const bool vec = (p.seqlen % 4 == 0) && aligned_to<T>(p.u_ptr, va) && aligned_to<T>(p.delta_ptr, va) && aligned_to<T>(p.B_ptr, va) &&
                     aligned_to<T>(p.C_ptr, va) && p.u_batch_stride % 4 == 0 && p.u_d_stride % 4 == 0 && p.delta_batch_stride % 4 == 0 &&
                     p.delta_d_stride % 4 == 0 && p.B_batch_stride % 4 == 0 && p.B_group_stride % 4 == 0 && p.B_dstate_stride % 4 == 0 &&
                     p.C_batch_stride % 4 == 0 && p.C_group_stride % 4 == 0 && p.C_dstate_stride % 4 == 0;
    dim3 grid(tiles), block(kWave);
    if (p.z_ptr) {
        if (vec) hipLaunchKernelGGL((ssm_scan_bwd_kernel<T, kN, true, true>), grid, block, 0, stream, q, ws);
        else hipLaunchKernelGGL((ssm_scan_bwd_kernel<T, kN, true, false>), grid, block, 0, stream, q, ws);
    } else {
        if (vec) hipLaunchKernelGGL((ssm_scan_bwd_kernel<T, kN, false, true>), grid, block, 0, stream, q, ws);
        else hipLaunchKernelGGL((ssm_scan_bwd_kernel<T, kN, false, false>), grid, block, 0, stream, q, ws);
    }
    return launch_status();
}

template <typename T>
static int dispatch_bwd(const dimsum_ssm_bwd_params_t &q, float *ws, hipStream_t stream) {
    switch (q.fwd.dstate) {
        case 4: return launch_bwd<T, 4>(q, ws, stream);
        case 8: return launch_bwd<T, 8>(q, ws, stream);
        case 16: return launch_bwd<T, 16>(q, ws, stream);
        case 32: return launch_bwd<T, 32>(q, ws, stream);
        default: return DIMSUM_ERR_SHAPE;
    }
}

int ssm_check(const dimsum_ssm_params_t *p, bool forward);

}  // namespace dimsum

extern "C" int64_t dimsum_ssm_scan_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t seqlen, int32_t dstate, int32_t n_groups) {
    if (batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0 || n_groups <= 0 || dim % n_groups != 0) return 0;
    const int64_t dpg = dim / n_groups;
    const int64_t waves = (int64_t)batch * n_groups * ((dpg + dimsum::kWave - 1) / dimsum::kWave);
    const int64_t n_tiles = (seqlen + dimsum::kBT - 1) / dimsum::kBT;
    return waves * n_tiles * dstate * dimsum::kWave * (int64_t)sizeof(float);
}

extern "C" int dimsum_ssm_scan_bwd(const dimsum_ssm_bwd_params_t *q, void *stream) {
    using namespace dimsum;
    if (!q) return DIMSUM_ERR_NULL;
    const int rc = ssm_check(&q->fwd, false);
    if (rc != DIMSUM_OK) return rc;
    if (!q->dout_ptr || !q->dA_ptr || !q->dB_ptr || !q->dC_ptr || !q->du_ptr || !q->ddelta_ptr || !q->workspace_ptr) return DIMSUM_ERR_NULL;
    if (q->fwd.z_ptr && (!q->dz_ptr || !q->fwd.out_ptr)) return DIMSUM_ERR_NULL;
    const dimsum_ssm_params_t &p = q->fwd;
    if (q->workspace_bytes < dimsum_ssm_scan_bwd_workspace_bytes(p.batch, p.dim, p.seqlen, p.dstate, p.n_groups)) return DIMSUM_ERR_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float *ws = reinterpret_cast<float *>(q->workspace_ptr);
    switch (p.dtype) {
        case DIMSUM_F32: return dispatch_bwd<float>(*q, ws, s);
        case DIMSUM_F16: return dispatch_bwd<__half>(*q, ws, s);
        case DIMSUM_BF16: return dispatch_bwd<__hip_bfloat16>(*q, ws, s);
        default: return DIMSUM_ERR_DTYPE;
    }
}
