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
// MI355X design. lane = (channel, state half): one wave64 owns 32 channels of one batch element, lanes 0-31 carry the
// first dstate/2 states of their channel and lanes 32-63 the second half; the sequence is walked BACKWARDS in registers.
// Instead of the reference's per-row block-wide forward + reverse parallel scans (about 3x the arithmetic, plus 1024-way
// global atomic contention on dB/dC):
//   * the state before every 8-step half tile comes from the forward kernel (ckpt_ptr; training callers keep it) or, for
//     callers with the reference's exact interface, from one extra state-only forward sweep into the workspace. A half's
//     states are fetched one whole half (~2 us of VALU work) before they are needed;
//   * 32-step tiles (128-B row segments: every HBM line is fetched exactly once) are walked backwards as four 8-step
//     halves; a half is processed 4 states at a time: forward sweep keeping h_t[4 states][8 steps] in 32 VGPRs, then the
//     reverse sweep over the same registers (a_t h_{t-1} = h_t - b_t: no division). Nothing per-(t, n) touches memory.
//   * sums over the states of a channel (s1, s2) are lane-local over dstate/2 states + ONE v_permlane32_swap per step;
//   * dB / dC are sums over the wave's 32 channels of per-lane values: a TRANSPOSED butterfly inside each half wave -- 32
//     values per lane go in, one fully reduced value per lane comes out, in 5 levels of v_permlane16_swap / DPP adds
//     (~2.5 VALU ops per value instead of ~10 for independent reductions). The wave's (n, t) sums of a tile are
//     collected in LDS and stored once, 16 B per lane, as this wave's PARTIAL dB / dC; a second small kernel adds the
//     partials of the dim/32 waves of a (batch, group) in a fixed order. No atomics on dB / dC at all (the reference
//     does 1024-way contended atomicAdds per address, selective_scan_bwd_kernel.cuh:297-316) -> bitwise reproducible.
//   * u, delta, dy tiles go through an XOR-swizzled LDS transpose like in the forward; dz / out_z are computed in the
//     coalesced load layout (16 B per lane in and out) and never touch LDS. 20 KB of LDS per wave = 8 waves per CU.
#include <type_traits>

#include "common.hpp"

namespace dimsum {

constexpr int kBC = 32;   // channels per wave
constexpr int kBT = 32;   // time steps per LDS tile (128 B per row and tensor: whole HBM lines)
constexpr int kBS = 8;    // time steps per register sweep (= distance of the saved states)

// 32 rows x 32 columns fp32, row = 8 slots of 16 B, slots XOR-swizzled by (row >> 1) & 7: ds_write_b128 in the load
// layout (8 lanes = one row) and ds_read_b128 in the lane = row layout are both bank-conflict free, no padding
__device__ __forceinline__ int btile_off(int row, int col4) { return row * kBT + ((col4 ^ ((row >> 1) & 7)) << 2); }

template <typename T> __device__ __forceinline__ const T *at(const T *base, unsigned elem_off) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (unsigned)(elem_off * (unsigned)sizeof(T)));
}
template <typename T> __device__ __forceinline__ T *at(T *base, unsigned elem_off) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (unsigned)(elem_off * (unsigned)sizeof(T)));
}

// ---- transposed butterfly: value i of every lane in -> lane l returns the 64-lane sum of value l ------------------------
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
// gen(i), i = 0..NV-1, produces this lane's value i (NV = 32 or 16); the first level is fused with the generation so
// that only 16 temporaries are ever live. Returns, in lane l, the sum of value l % NV over the 32 lanes of l's half wave.
template <int NV, typename Gen> __device__ __forceinline__ float transposed_reduce_half(Gen gen, int lane) {
    static_assert(NV == 32 || NV == 16, "");
    float v[16];
    if constexpr (NV == 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { float a = gen(i), b = gen(i + 16); swap16(a, b); v[i] = a + b; }   // lane bit 4 <-> value bit 4
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = gen(i);
    }
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
    if constexpr (NV == 16) { float a = r, b = r; swap16(a, b); r = a + b; }        // the two rows hold partial sums of the same value
    return r;
}

// Hides a value's origin from the optimiser: the reverse sweep recomputes a_{t+1} = exp2(dt_{t+1} A) instead of keeping
// the forward sweep's 32 exponentials alive across both transposed reductions (register budget; see DIMSUM_BWD_KEEP_A).
__device__ __forceinline__ float opaque(float x) {
#ifndef DIMSUM_BWD_KEEP_A
    asm volatile("" : "+v"(x));
#endif
    return x;
}

// sigmoid(x) for dt = softplus(x) = log(1 + e^x):  sigmoid(x) = 1 - exp(-dt)  (for x > 20 the reference takes dt = x and
// a derivative of 1: 1 - exp(-20) rounds to 1). Small dt: alternating series (the direct form cancels).
__device__ __forceinline__ float dsoftplus_from_dt(float dt) {
    const float ser = dt * (1.0f - dt * (0.5f - dt * (1.0f / 6 - dt * (1.0f / 24 - dt * (1.0f / 120 - dt * (1.0f / 720))))));
    float big = 1.0f - fast_exp(-dt);
    asm volatile("" : "+v"(big));          // both forms in straight-line code: no per-element branch around the v_exp_f32
    return dt < 0.25f ? ser : big;
}

// timing experiments (tools/scratch): DIMSUM_BWD_X_NOMEM drops the kernel's global loads / stores of tile data
#ifdef DIMSUM_BWD_X_NOMEM
#define XLD(T, ptr) (Raw4<T>{})   /* loads vanish */
#define XST_ON (L < 0)
#define XCK(expr) (1e-3f * (float)(lane + 1))
#else
#define XLD(T, ptr) ld4<T>(ptr)
#define XST_ON true
#define XCK(expr) (expr)
#endif

// kVec : every row base 4-element aligned and L % 4 == 0 -> 16-byte vector I/O.   kFull: all 32 channel slots are live.
template <typename T, int kN, bool kHasZ, bool kVec, bool kFull>
__global__ __launch_bounds__(kWave, 2) void ssm_scan_bwd_kernel(const dimsum_ssm_bwd_params_t q, const float *__restrict__ ckpt, float *__restrict__ part) {
    constexpr int kNL = kN / 2;                   // states per lane (lane = channel + 32 * state half)
    #ifndef DIMSUM_BWD_BG
#define DIMSUM_BWD_BG 4
#endif
    constexpr int kBG = kNL < DIMSUM_BWD_BG ? kNL : DIMSUM_BWD_BG;        // states per register sweep
    constexpr int NV = kBG * kBS;                 // (state, step) values per transposed reduction
    constexpr int kNG = kNL / kBG;                // register sweeps (state groups) per half tile
    static_assert(kN >= 4 && kN % 2 == 0 && kNL % kBG == 0 && (NV == 32 || NV == 16), "dstate must be 4 or a multiple of 8");
    const dimsum_ssm_params_t &p = q.fwd;
    __shared__ __attribute__((aligned(16))) float tU[kBC * kBT], tD[kBC * kBT], tY[kBC * kBT];   // u, dt (softplus'ed), dy
    __shared__ __attribute__((aligned(16))) float tB[kN * kBT], tC[kN * kBT];
    __shared__ __attribute__((aligned(16))) float tdB[kN * kBT], tdC[kN * kBT];      // this wave's dB / dC sums of the tile

#ifdef DIMSUM_BWD_X_PADLDS         // occupancy experiment: extra LDS per wave
    __shared__ float xpad[DIMSUM_BWD_X_PADLDS];
    if (q.fwd.seqlen < 0) xpad[threadIdx.x] = 1.f;
#endif
    const int lane = threadIdx.x, c = lane & (kBC - 1), sh = lane >> 5;
    const int ns0 = sh * kNL;                     // first state of this lane
    const int L = p.seqlen;
    const int dpg = p.dim / p.n_groups;
    const int tiles_per_group = (dpg + kBC - 1) / kBC;
    const int tiles_per_batch = p.n_groups * tiles_per_group;
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);      // a batch element's waves share an XCD (one L2 for B / C)
    const int b = wg / tiles_per_batch;
    const int rem = wg - b * tiles_per_batch;
    const int g = rem / tiles_per_group;
    const int d0 = g * dpg + (rem - g * tiles_per_group) * kBC;
    const int nd = kFull ? kBC : min(kBC, (g + 1) * dpg - d0);
    const bool live = kFull || c < nd;
    const int d = d0 + (kFull ? c : min(c, nd - 1));

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
    // partial sums of this wave: part[wave][dB | dC][n][L]
    float *pB = part + (int64_t)wg * 2 * kN * L, *pC = pB + (int64_t)kN * L;
    const int u_ds = (int)p.u_d_stride, dl_ds = (int)p.delta_d_stride, do_ds = (int)q.dout_d_stride, z_ds = (int)p.z_d_stride;
    const int y_ds = (int)p.out_d_stride, oz_ds = (int)p.out_z_d_stride, dz_ds = (int)q.dz_d_stride, du_ds = (int)q.du_d_stride;
    const int dd_ds = (int)q.ddelta_d_stride;
    const int Bns = (int)p.B_dstate_stride, Cns = (int)p.C_dstate_stride;

    // per-lane constants and carries, all in registers (only ever indexed with compile-time constants)
    float A2[kNL], rdh[kNL], rdA[kNL];
    {
        const float *Ap = reinterpret_cast<const float *>(p.A_ptr) + (int64_t)d * p.A_d_stride;
#pragma unroll
        for (int k = 0; k < kNL; ++k) { A2[k] = Ap[(ns0 + k) * p.A_dstate_stride] * kLog2e; rdh[k] = 0.f; rdA[k] = 0.f; }   // exp(dt A) = exp2(dt A log2 e)
    }
    const float Dval = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d] : 0.f;
    const float *bias_p = reinterpret_cast<const float *>(p.delta_bias_ptr);
    const bool softplus = p.delta_softplus != 0;
    float dD = 0.f, dbias = 0.f;
    float dt_next = 0.f;   // dt of the first step of the tile processed before (later in time): a_{t+1} at the seam

    const int n_tiles = (L + kBT - 1) / kBT;
    const int n_halves = (L + kBS - 1) / kBS;
    // saved states: [b][half tile][n][d]
    const float *ck_lane = ckpt + (int64_t)b * n_halves * kN * p.dim + (int64_t)ns0 * p.dim + d;
    const int64_t ck_ns = p.dim;                                // stride between states
    // coalesced tile layout: 32 rows x 32 columns = 4 pieces of (8 rows x 8 lanes-per-row x 4 columns): whole 128-B lines
    const int lrow = lane >> 3, lc4 = lane & 7, lcol = lc4 * 4;
    float brow[4];                                               // delta_bias of the rows this lane stages
#pragma unroll
    for (int i = 0; i < 4; ++i) brow[i] = bias_p ? bias_p[d0 + min(i * 8 + lrow, nd - 1)] : 0.f;

    // states of the last half tile (the first one processed); later halves are prefetched one half ahead
    float hpre[kNL];
#pragma unroll
    for (int k = 0; k < kNL; ++k) hpre[k] = ck_lane[((int64_t)(n_halves - 1) * kN + k) * ck_ns];

    // Register-staged prefetch (vector path): the next tile's u / delta (kPF >= 1), dout (>= 2), z / out (>= 3) rows are
    // requested right after the current tile has been staged, so they fly under the tile's sweeps. Branch-free: rows
    // beyond nd are clamped to the last live row, columns beyond L to the last 4-column group; the masks are applied
    // when the registers are staged.
#ifndef DIMSUM_BWD_PF
#define DIMSUM_BWD_PF 1
#endif
    constexpr int kPF = kVec ? DIMSUM_BWD_PF : 0;
    Raw4<T> pu[4], pd[4], pg[4], pz[4], py[4];
    auto tile_addr = [&](const T *base, int ds, int i, int col) -> const T * {
        if constexpr (kFull) return at(base + i * 8 * ds, (unsigned)(lrow * ds + col));
        else return at(base, (unsigned)(min(i * 8 + lrow, nd - 1) * ds + col));
    };
    auto issue = [&](int t0n, int lo, int hi) {          // requests the tensors with prefetch rank in (lo, hi]
        const int col = min(t0n + lcol, L - 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (lo < 1 && 1 <= hi) { pu[i] = XLD(T, tile_addr(u_base, u_ds, i, col)); pd[i] = XLD(T, tile_addr(dl_base, dl_ds, i, col)); }
            if (lo < 2 && 2 <= hi) pg[i] = XLD(T, tile_addr(do_base, do_ds, i, col));
            if constexpr (kHasZ)
                if (lo < 3 && 3 <= hi) { pz[i] = XLD(T, tile_addr(z_base, z_ds, i, col)); py[i] = XLD(T, tile_addr(y_base, y_ds, i, col)); }
        }
    };
    if constexpr (kVec) issue((n_tiles - 1) * kBT, 0, kPF);

#pragma unroll 1
    for (int tile = n_tiles - 1; tile >= 0; --tile) {
        const int t0 = tile * kBT;
        // ---- stage u, dt = softplus(delta + bias) (0 beyond L: dead steps are identities, a = 1, b = 0), B, C ------------
        if constexpr (kVec) {
            issue(t0, kPF, 3);                                   // whatever is not prefetched is requested now
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + lrow;
                f32x4 vu = {{0.f, 0.f, 0.f, 0.f}}, vd = {{0.f, 0.f, 0.f, 0.f}};
                if ((kFull || row < nd) && t0 + lcol < L) {
                    vu = widen(pu[i]);
                    vd = widen(pd[i]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float v = vd.v[s] + brow[i];
                        vd.v[s] = softplus_if(v, softplus);
                    }
                }
                *reinterpret_cast<f32x4 *>(&tU[btile_off(row, lc4)]) = vu;
                *reinterpret_cast<f32x4 *>(&tD[btile_off(row, lc4)]) = vd;
            }
        } else {
            for (int i = 0; i < kBC * kBT / kWave; ++i) {
                const int idx = i * kWave + lane, row = idx / kBT, col = idx & (kBT - 1);
                const bool ok = row < nd && t0 + col < L;
                float vu = 0.f, vd = 0.f;
                if (ok) {
                    vu = to_f32<T>(u_base[(unsigned)(row * u_ds + t0 + col)]);
                    const float v = to_f32<T>(dl_base[(unsigned)(row * dl_ds + t0 + col)]) + (bias_p ? bias_p[d0 + row] : 0.f);
                    vd = softplus_if(v, softplus);
                }
                tU[btile_off(row, col >> 2) + (col & 3)] = vu;
                tD[btile_off(row, col >> 2) + (col & 3)] = vd;
            }
        }
        for (int idx = lane; idx < kN * (kBT / 4); idx += kWave) {
            const int n = idx >> 3, c4 = idx & 7;
            f32x4 vb = {{0.f, 0.f, 0.f, 0.f}}, vc = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (kVec) {
                if (t0 + c4 * 4 < L) { vb = widen(ld4<T>(at(Bp, (unsigned)(n * Bns + t0 + c4 * 4)))); vc = widen(ld4<T>(at(Cp, (unsigned)(n * Cns + t0 + c4 * 4)))); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t0 + c4 * 4 + e < L) { vb.v[e] = to_f32<T>(Bp[(unsigned)(n * Bns + t0 + c4 * 4 + e)]); vc.v[e] = to_f32<T>(Cp[(unsigned)(n * Cns + t0 + c4 * 4 + e)]); }
            }
            *reinterpret_cast<f32x4 *>(&tB[n * kBT + c4 * 4]) = vb;
            *reinterpret_cast<f32x4 *>(&tC[n * kBT + c4 * 4]) = vc;
        }
        // ---- dy = dout * silu(z), dz, optional out_z -- in the coalesced layout; only dy goes to LDS -------------------
        if constexpr (kVec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + lrow;
                f32x4 dy = {{0.f, 0.f, 0.f, 0.f}};
                if ((kFull || row < nd) && t0 + lcol < L) {
                    const unsigned col = (unsigned)(t0 + lcol);
                    const f32x4 go = widen(pg[i]);
                    if constexpr (kHasZ) {
                        const f32x4 zv = widen(pz[i]);
                        const f32x4 yv = widen(py[i]);
                        f32x4 dz, oz;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sgz = sigmoidf_fast(zv.v[e]), silu = zv.v[e] * sgz;
                            dz.v[e] = go.v[e] * yv.v[e] * sgz * (1.0f + zv.v[e] * (1.0f - sgz));
                            oz.v[e] = yv.v[e] * silu;
                            dy.v[e] = go.v[e] * silu;
                        }
                        if (XST_ON) st4<T>(at(dz_base + i * 8 * dz_ds, (unsigned)(lrow * dz_ds) + col), dz);
                        if (XST_ON && oz_base) st4<T>(at(oz_base + i * 8 * oz_ds, (unsigned)(lrow * oz_ds) + col), oz);
                    } else {
                        dy = go;
                    }
                }
                *reinterpret_cast<f32x4 *>(&tY[btile_off(row, lc4)]) = dy;
            }
            if (tile > 0) issue(t0 - kBT, 0, kPF);               // flies under the sweeps below
        } else {
            for (int i = 0; i < kBC * kBT / kWave; ++i) {
                const int idx = i * kWave + lane, row = idx / kBT, col = idx & (kBT - 1), t = t0 + col;
                float dy = 0.f;
                if (row < nd && t < L) {
                    const float go = to_f32<T>(do_base[(unsigned)(row * do_ds + t)]);
                    if constexpr (kHasZ) {
                        const float zv = to_f32<T>(z_base[(unsigned)(row * z_ds + t)]), yv = to_f32<T>(y_base[(unsigned)(row * y_ds + t)]);
                        const float sgz = sigmoidf_fast(zv), silu = zv * sgz;
                        dz_base[(unsigned)(row * dz_ds + t)] = from_f32<T>(go * yv * sgz * (1.0f + zv * (1.0f - sgz)));
                        if (oz_base) oz_base[(unsigned)(row * oz_ds + t)] = from_f32<T>(yv * silu);
                        dy = go * silu;
                    } else {
                        dy = go;
                    }
                }
                tY[btile_off(row, col >> 2) + (col & 3)] = dy;
            }
        }
        const float dt_first = (*reinterpret_cast<const f32x4 *>(&tD[btile_off(c, 0)])).v[0];

#pragma unroll 1
        for (int half = kBT / kBS - 1; half >= 0; --half) {
            const int hidx = tile * (kBT / kBS) + half;          // index of this half tile's saved state
            const int jb = half * (kBS / 4);                     // first 4-step slot of the half
            if (hidx >= n_halves) continue;                      // a trailing half entirely beyond L
            // dt of the step after the half: next slot of the tile, or the seam to the tile processed before
            const float dt_after = (half == kBT / kBS - 1) ? dt_next : (*reinterpret_cast<const f32x4 *>(&tD[btile_off(c, jb + kBS / 4)])).v[0];
            // this half's saved states were fetched one half ago; every group re-issues the fetch of its states for the NEXT
            // half as soon as it has consumed them (a whole half of VALU work ahead of their use)
            const float *ck_next = ck_lane + (int64_t)max(hidx - 1, 0) * kN * ck_ns;
            float s1[kBS], s2[kBS];
#pragma unroll
            for (int t = 0; t < kBS; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

            // one register sweep over kBG states x kBS steps per iteration. The group loop stays rolled (one copy of the body;
            // real control flow between groups keeps the scheduler from interleaving them and blowing the register budget).
#pragma unroll 1
#ifdef DIMSUM_BWD_X_NOSWEEP
            for (int G = 0; G < 0; ++G) {
#else
            for (int G = 0; G < kNG; ++G) {
#endif
                const int n0 = G * kBG;                   // first state (lane-local) of the group
                float H[kBG * kBS];                       // [k][t]: h_t of state n0+k, later overwritten by the dB terms
                float hk[kBG], dhk[kBG], dAk[kBG], Ak[kBG];
                // uniform selects on static indices (a dynamically indexed private array would live in scratch memory)
#pragma unroll
                for (int k = 0; k < kBG; ++k) {
                    hk[k] = hpre[k]; dhk[k] = rdh[k]; dAk[k] = rdA[k]; Ak[k] = A2[k];
#pragma unroll
                    for (int gq = 1; gq < kNG; ++gq) {
                        hk[k] = (G == gq) ? hpre[gq * kBG + k] : hk[k];
                        dhk[k] = (G == gq) ? rdh[gq * kBG + k] : dhk[k];
                        dAk[k] = (G == gq) ? rdA[gq * kBG + k] : dAk[k];
                        Ak[k] = (G == gq) ? A2[gq * kBG + k] : Ak[k];
                    }
                }
                {
                    float hn[kBG];
#pragma unroll
                    for (int k = 0; k < kBG; ++k) hn[k] = XCK(ck_next[(n0 + k) * ck_ns]);
#pragma unroll
                    for (int gq = 0; gq < kNG; ++gq)
#pragma unroll
                        for (int k = 0; k < kBG; ++k) hpre[gq * kBG + k] = (G == gq) ? hn[k] : hpre[gq * kBG + k];
                }
                const int nrow = (ns0 + n0) * kBT;        // LDS row of the group's first state in tB / tC / tdB / tdC
                // ---- forward sweep: h_t for the 8 steps of the half ------------------------------------------------------
                {
                    f32x4 bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + jb * 4]);
#pragma unroll
                    for (int jj = 0; jj < kBS / 4; ++jj) {
                        const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(c, jb + jj)]);
                        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(c, jb + jj)]);
                        float du[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) du[s] = d4.v[s] * u4.v[s];
#pragma unroll
                        for (int k = 0; k < kBG; ++k) {
                            const f32x4 bq = bq_nxt;
                            if (k + 1 < kBG) bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + (k + 1) * kBT + (jb + jj) * 4]);
                            else if (jj + 1 < kBS / 4) bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + (jb + jj + 1) * 4]);
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                hk[k] = fmaf(fast_exp2(d4.v[s] * Ak[k]), hk[k], bq.v[s] * du[s]);
                                H[k * kBS + jj * 4 + s] = hk[k];
                            }
                        }
                    }
                }
                // ---- dC[n, t] = sum_d dy_t h_t[n]: transposed reduction of the (k, t) products over the 32 channels --------
                {
                    float y8[kBS];
#pragma unroll
                    for (int jj = 0; jj < kBS / 4; ++jj) {
                        const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(c, jb + jj)]);
#pragma unroll
                        for (int s = 0; s < 4; ++s) y8[jj * 4 + s] = live ? y4.v[s] : 0.f;
                    }
#ifdef DIMSUM_BWD_X_NOREDUCE
                    float r = 0.f;
#pragma unroll
                    for (int i = 0; i < NV; ++i) r = fmaf(y8[i & (kBS - 1)], H[i], r);
#else
                    const float r = transposed_reduce_half<NV>([&](int i) { return y8[i & (kBS - 1)] * H[i]; }, lane);
#endif
                    const int vi = lane & (NV - 1);
                    if (NV == 32 || (lane & 16) == 0) tdC[nrow + (vi >> 3) * kBT + half * kBS + (vi & 7)] = r;
                }
                // ---- reverse sweep ----------------------------------------------------------------------------------------
                {
                    f32x4 bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + (jb + kBS / 4 - 1) * 4]);
                    f32x4 cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[nrow + (jb + kBS / 4 - 1) * 4]);
                    float dt_succ = dt_after;       // dt of the step after the current slot
#pragma unroll
                    for (int jj = kBS / 4 - 1; jj >= 0; --jj) {
                        const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(c, jb + jj)]);
                        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(c, jb + jj)]);
                        const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(c, jb + jj)]);
                        float dnext[4], du[4];       // dt of step t+1; dt u
#pragma unroll
                        for (int s = 0; s < 3; ++s) dnext[s] = d4.v[s + 1];
                        dnext[3] = dt_succ;
                        dt_succ = d4.v[0];
#pragma unroll
                        for (int s = 0; s < 4; ++s) du[s] = d4.v[s] * u4.v[s];
#pragma unroll
                        for (int k = 0; k < kBG; ++k) {
                            const f32x4 bq = bq_nxt, cq = cq_nxt;
                            if (k + 1 < kBG) {
                                bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + (k + 1) * kBT + (jb + jj) * 4]);
                                cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[nrow + (k + 1) * kBT + (jb + jj) * 4]);
                            } else if (jj > 0) {
                                bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[nrow + (jb + jj - 1) * 4]);
                                cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[nrow + (jb + jj - 1) * 4]);
                            }
#pragma unroll
                            for (int s = 3; s >= 0; --s) {
                                const int t = jj * 4 + s;
                                const float a_next = fast_exp2(opaque(dnext[s]) * Ak[k]);
                                const float dhn = fmaf(a_next, dhk[k], cq.v[s] * y4.v[s]);
                                dhk[k] = dhn;
                                const float ah = fmaf(-bq.v[s], du[s], H[k * kBS + t]);     // a_t h_{t-1} = h_t - b_t
                                const float gterm = dhn * ah;
                                dAk[k] = fmaf(gterm, d4.v[s], dAk[k]);
                                s2[t] = fmaf(gterm, Ak[k], s2[t]);
                                s1[t] = fmaf(dhn, bq.v[s], s1[t]);
                                H[k * kBS + t] = dhn * du[s];                               // dB term
                            }
                        }
                    }
                }
#pragma unroll
                for (int gq = 0; gq < kNG; ++gq)
#pragma unroll
                    for (int k = 0; k < kBG; ++k) {
                        rdh[gq * kBG + k] = (G == gq) ? dhk[k] : rdh[gq * kBG + k];
                        rdA[gq * kBG + k] = (G == gq) ? dAk[k] : rdA[gq * kBG + k];
                    }
                // ---- dB[n, t] = sum_d dh_t[n] dt_t u_t -----------------------------------------------------------------------
                {
#ifdef DIMSUM_BWD_X_NOREDUCE
                    float r = 0.f;
#pragma unroll
                    for (int i = 0; i < NV; ++i) r += H[i];
#else
                    const float r = transposed_reduce_half<NV>([&](int i) { return live ? H[i] : 0.f; }, lane);
#endif
                    const int vi = lane & (NV - 1);
                    if (NV == 32 || (lane & 16) == 0) tdB[nrow + (vi >> 3) * kBT + half * kBS + (vi & 7)] = r;
                }
            }

            // ---- per-(d, t) results of the half. The two lanes of a channel hold the sums over their 8 states; ONE swap gives
            //      the low lane s1 = sum_n dh B and the high lane ddt = u s1 + s2 (both linear in the per-lane partials): the
            //      low lane finishes du (into the u tile), the high lane ddelta (softplus chain, into the dy tile).
            //      (s2 was accumulated with A * log2 e) ----------------------------------------------------------------------
#pragma unroll
            for (int jj = 0; jj < kBS / 4; ++jj) {
                const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(c, jb + jj)]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(c, jb + jj)]);
                const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(c, jb + jj)]);
                f32x4 o4;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int t = jj * 4 + s;
                    float pa = s1[t], qa = fmaf(u4.v[s], s1[t], s2[t] * kLn2);
                    swap32(pa, qa);
                    const float r = pa + qa;                     // low lane: s1 of the channel; high lane: ddt of the channel
                    const float du_v = fmaf(d4.v[s], r, Dval * y4.v[s]);
                    // dead steps (t >= L) have dt = 0 -> factor 0 with softplus; without it they carry u = dy = 0 -> ddt = 0
                    float dsp = dsoftplus_from_dt(d4.v[s]);
                    asm volatile("" : "+v"(dsp));
                    const float dd_v = softplus ? r * dsp : r;
                    dD = fmaf(y4.v[s], u4.v[s], dD);
                    dbias += dd_v;                               // meaningful in the high lane only
                    o4.v[s] = sh ? dd_v : du_v;
                }
                float *dst = sh ? tY : tU;
                *reinterpret_cast<f32x4 *>(&dst[btile_off(c, jb + jj)]) = o4;
            }
        }
        dt_next = dt_first;

        // ---- this wave's partial dB / dC of the tile: 16 B per lane, rows of 128 B ----------------------------------------
        for (int idx = lane; idx < kN * (kBT / 4); idx += kWave) {
            const int n = idx >> 3, cc = (idx & 7) * 4, t = t0 + cc;
            const f32x4 vb = *reinterpret_cast<const f32x4 *>(&tdB[n * kBT + cc]);
            const f32x4 vc = *reinterpret_cast<const f32x4 *>(&tdC[n * kBT + cc]);
            if (XST_ON && t + 3 < L) {
                st4<float>(pB + (int64_t)n * L + t, vb);
                st4<float>(pC + (int64_t)n * L + t, vc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < L) { pB[(int64_t)n * L + t + e] = vb.v[e]; pC[(int64_t)n * L + t + e] = vc.v[e]; }
            }
        }
        // ---- coalesced stores of du, ddelta -------------------------------------------------------------------------------
        if constexpr (kVec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + lrow;
                const f32x4 a = *reinterpret_cast<const f32x4 *>(&tU[btile_off(row, lc4)]);
                const f32x4 cv = *reinterpret_cast<const f32x4 *>(&tY[btile_off(row, lc4)]);
                if (XST_ON && (kFull || row < nd) && t0 + lcol < L) {
                    st4<T>(at(du_base + i * 8 * du_ds, (unsigned)(lrow * du_ds + t0 + lcol)), a);
                    st4<T>(at(dd_base + i * 8 * dd_ds, (unsigned)(lrow * dd_ds + t0 + lcol)), cv);
                }
            }
        } else {
            for (int i = 0; i < kBC * kBT / kWave; ++i) {
                const int idx = i * kWave + lane, row = idx / kBT, col = idx & (kBT - 1), t = t0 + col;
                if (row < nd && t < L) {
                    du_base[(unsigned)(row * du_ds + t)] = from_f32<T>(tU[btile_off(row, col >> 2) + (col & 3)]);
                    dd_base[(unsigned)(row * dd_ds + t)] = from_f32<T>(tY[btile_off(row, col >> 2) + (col & 3)]);
                }
            }
        }
    }

    if (live) {
        float *dAp = reinterpret_cast<float *>(q.dA_ptr) + (int64_t)d * q.dA_d_stride;
#pragma unroll
        for (int k = 0; k < kNL; ++k) atomicAdd(dAp + (ns0 + k) * q.dA_dstate_stride, rdA[k]);
        if (q.dD_ptr && sh == 0) atomicAdd(reinterpret_cast<float *>(q.dD_ptr) + d, dD);
        if (q.ddelta_bias_ptr && sh == 1) atomicAdd(reinterpret_cast<float *>(q.ddelta_bias_ptr) + d, dbias);
    }
}

// dB[b, g, n, t] = sum over the waves w of (b, g), in index order, of part[b, g, w][0][n][t]  (same for dC).
// One thread owns 4 consecutive steps (16-byte loads when L % 4 == 0); the partial rows of a (b, g) are 2 N L floats apart.
template <bool kVec4>
__global__ __launch_bounds__(256) void ssm_scan_bwd_reduce_kernel(const float *__restrict__ part, const dimsum_ssm_bwd_params_t q, int waves_per_group) {
    const dimsum_ssm_params_t &p = q.fwd;
    const int L = p.seqlen, N = p.dstate, L4 = (L + 3) / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over (b, g, which, n, t / 4)
    const int64_t total = (int64_t)p.batch * p.n_groups * 2 * N * L4;
    if (i >= total) return;
    const int t = (int)(i % L4) * 4;
    int64_t r = i / L4;
    const int n = (int)(r % N); r /= N;
    const int which = (int)(r & 1); r >>= 1;
    const int g = (int)(r % p.n_groups);
    const int b = (int)(r / p.n_groups);
    const float *src = part + (((int64_t)(b * p.n_groups + g) * waves_per_group) * 2 + which) * N * L + (int64_t)n * L + t;
    const int64_t ws = (int64_t)2 * N * L;
    float *dst = which == 0 ? reinterpret_cast<float *>(q.dB_ptr) + (int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride + (int64_t)n * q.dB_dstate_stride + t
                            : reinterpret_cast<float *>(q.dC_ptr) + (int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride + (int64_t)n * q.dC_dstate_stride + t;
    if constexpr (kVec4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int w = 0;
        for (; w + 4 <= waves_per_group; w += 4) {       // 4 independent 16-byte loads in flight per thread
            const float4 v0 = *reinterpret_cast<const float4 *>(src + (int64_t)w * ws), v1 = *reinterpret_cast<const float4 *>(src + (int64_t)(w + 1) * ws);
            const float4 v2 = *reinterpret_cast<const float4 *>(src + (int64_t)(w + 2) * ws), v3 = *reinterpret_cast<const float4 *>(src + (int64_t)(w + 3) * ws);
            acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x; acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
            acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z; acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; w < waves_per_group; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)w * ws);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4 *>(dst) = acc;
    } else {
        for (int e = 0; e < 4 && t + e < L; ++e) {
            float acc = 0.f;
            for (int w = 0; w < waves_per_group; ++w) acc += src[(int64_t)w * ws + e];
            dst[e] = acc;
        }
    }
}

template <typename T, int kN>
static int launch_bwd(const dimsum_ssm_bwd_params_t &q, const float *ckpt, float *part, hipStream_t stream) {
    const dimsum_ssm_params_t &p = q.fwd;
    const int dpg = p.dim / p.n_groups;
    const int tiles = p.batch * p.n_groups * ((dpg + kBC - 1) / kBC);
    const size_t va = 4 * sizeof(T);
    auto ok4 = [&](const void *ptr, int64_t bs, int64_t ds) { return aligned_to<T>(ptr, va) && bs % 4 == 0 && ds % 4 == 0; };
    bool vec = (p.seqlen % 4 == 0) && ok4(p.u_ptr, p.u_batch_stride, p.u_d_stride) && ok4(p.delta_ptr, p.delta_batch_stride, p.delta_d_stride) &&
               ok4(q.dout_ptr, q.dout_batch_stride, q.dout_d_stride) && ok4(q.du_ptr, q.du_batch_stride, q.du_d_stride) &&
               ok4(q.ddelta_ptr, q.ddelta_batch_stride, q.ddelta_d_stride) &&
               ok4(p.B_ptr, p.B_batch_stride, p.B_dstate_stride) && ok4(p.C_ptr, p.C_batch_stride, p.C_dstate_stride) &&
               p.B_group_stride % 4 == 0 && p.C_group_stride % 4 == 0;
    if (p.z_ptr) {
        vec = vec && ok4(p.z_ptr, p.z_batch_stride, p.z_d_stride) && ok4(p.out_ptr, p.out_batch_stride, p.out_d_stride) &&
              ok4(q.dz_ptr, q.dz_batch_stride, q.dz_d_stride);
        if (p.out_z_ptr) vec = vec && ok4(p.out_z_ptr, p.out_z_batch_stride, p.out_z_d_stride);
    }
    // 32-bit in-tile offsets
    const int64_t lim = (int64_t)1 << 31, Ls = p.seqlen;
    const int64_t dss[] = {p.u_d_stride, p.delta_d_stride, q.dout_d_stride, q.du_d_stride, q.ddelta_d_stride,
                           p.z_ptr ? p.z_d_stride : 0, p.z_ptr ? p.out_d_stride : 0, p.z_ptr ? q.dz_d_stride : 0,
                           (p.z_ptr && p.out_z_ptr) ? p.out_z_d_stride : 0};
    for (int64_t ds : dss)
        if (ds < 0 || 64 * ds + Ls >= lim) return DIMSUM_ERR_STRIDE;
    const int64_t nss[] = {p.B_dstate_stride, p.C_dstate_stride};
    for (int64_t ns : nss)
        if (ns < 0 || (int64_t)p.dstate * ns + Ls >= lim) return DIMSUM_ERR_STRIDE;
    const bool full = vec && (dpg % kBC == 0);
    dim3 grid(tiles), block(kWave);
#define DIMSUM_LAUNCH(HASZ, VEC, FULL) \
    hipLaunchKernelGGL((ssm_scan_bwd_kernel<T, kN, HASZ, VEC, FULL>), grid, block, 0, stream, q, ckpt, part)
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
    if (launch_status() != DIMSUM_OK) return DIMSUM_ERR_LAUNCH;
    const int64_t total = (int64_t)p.batch * p.n_groups * 2 * kN * ((p.seqlen + 3) / 4);
    const bool vec4 = p.seqlen % 4 == 0 && aligned_to<float>(q.dB_ptr, 16) && aligned_to<float>(q.dC_ptr, 16) && q.dB_batch_stride % 4 == 0 &&
                      q.dB_group_stride % 4 == 0 && q.dB_dstate_stride % 4 == 0 && q.dC_batch_stride % 4 == 0 && q.dC_group_stride % 4 == 0 &&
                      q.dC_dstate_stride % 4 == 0;
    const dim3 rgrid((unsigned)((total + 255) / 256)), rblock(256);
    if (vec4) hipLaunchKernelGGL(ssm_scan_bwd_reduce_kernel<true>, rgrid, rblock, 0, stream, part, q, (dpg + kBC - 1) / kBC);
    else hipLaunchKernelGGL(ssm_scan_bwd_reduce_kernel<false>, rgrid, rblock, 0, stream, part, q, (dpg + kBC - 1) / kBC);
    return launch_status();
}

template <typename T>
static int dispatch_bwd(const dimsum_ssm_bwd_params_t &q, const float *ckpt, float *part, hipStream_t stream) {
    switch (q.fwd.dstate) {
#ifndef DIMSUM_DEV_ONE      // development builds instantiate the headline variant only
        case 4: return launch_bwd<T, 4>(q, ckpt, part, stream);
        case 8: return launch_bwd<T, 8>(q, ckpt, part, stream);
        case 32: return launch_bwd<T, 32>(q, ckpt, part, stream);
#endif
        case 16: return launch_bwd<T, 16>(q, ckpt, part, stream);
        default: return DIMSUM_ERR_SHAPE;
    }
}

int ssm_check(const dimsum_ssm_params_t *p, bool forward);

}  // namespace dimsum

static int64_t partial_bytes(int32_t batch, int32_t dim, int32_t seqlen, int32_t dstate, int32_t n_groups) {
    const int64_t dpg = dim / n_groups;
    const int64_t waves = (int64_t)batch * n_groups * ((dpg + dimsum::kBC - 1) / dimsum::kBC);
    return waves * 2 * dstate * seqlen * (int64_t)sizeof(float);                    // (waves, dB | dC, dstate, seqlen)
}
static int64_t ckpt_bytes(int32_t batch, int32_t dim, int32_t seqlen, int32_t dstate) {
    const int64_t n_halves = (seqlen + dimsum::kBS - 1) / dimsum::kBS;
    return (int64_t)batch * n_halves * dstate * dim * (int64_t)sizeof(float);       // (batch, half tiles, dstate, dim)
}

extern "C" int64_t dimsum_ssm_scan_bwd_workspace_bytes(int32_t batch, int32_t dim, int32_t seqlen, int32_t dstate, int32_t n_groups) {
    if (batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0 || n_groups <= 0 || dim % n_groups != 0) return 0;
    return partial_bytes(batch, dim, seqlen, dstate, n_groups) + ckpt_bytes(batch, dim, seqlen, dstate);
}

extern "C" int dimsum_ssm_scan_bwd(const dimsum_ssm_bwd_params_t *q, void *stream) {
    using namespace dimsum;
    if (!q) return DIMSUM_ERR_NULL;
    const int rc = ssm_check(&q->fwd, false);
    if (rc != DIMSUM_OK) return rc;
    if (!q->dout_ptr || !q->dA_ptr || !q->dB_ptr || !q->dC_ptr || !q->du_ptr || !q->ddelta_ptr || !q->workspace_ptr) return DIMSUM_ERR_NULL;
    if (q->fwd.z_ptr && (!q->dz_ptr || !q->fwd.out_ptr)) return DIMSUM_ERR_NULL;
    const dimsum_ssm_params_t &p = q->fwd;
    if (!aligned_to<float>(q->workspace_ptr, 16)) return DIMSUM_ERR_STRIDE;
    // workspace = [per-wave partial dB / dC | saved states (only when the caller did not keep the forward's)]
    const int64_t pbytes = partial_bytes(p.batch, p.dim, p.seqlen, p.dstate, p.n_groups);
    const float *ckpt = reinterpret_cast<const float *>(p.ckpt_ptr);
    if (q->workspace_bytes < pbytes + (ckpt ? 0 : ckpt_bytes(p.batch, p.dim, p.seqlen, p.dstate))) return DIMSUM_ERR_SHAPE;
    float *part = reinterpret_cast<float *>(q->workspace_ptr);
    if (!ckpt) {
        // reference-shaped call (no saved states): one state-only forward sweep rebuilds them in the workspace
        dimsum_ssm_params_t f = p;
        f.z_ptr = nullptr; f.out_ptr = nullptr; f.out_z_ptr = nullptr; f.x_ptr = nullptr; f.D_ptr = nullptr;
        f.ckpt_ptr = reinterpret_cast<char *>(q->workspace_ptr) + pbytes;
        const int frc = dimsum_ssm_scan_fwd(&f, stream);
        if (frc != DIMSUM_OK) return frc;
        ckpt = reinterpret_cast<const float *>(f.ckpt_ptr);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (p.dtype) {
        case DIMSUM_F32: return dispatch_bwd<float>(*q, ckpt, part, s);
#ifndef DIMSUM_DEV_ONE
        case DIMSUM_F16: return dispatch_bwd<__half>(*q, ckpt, part, s);
        case DIMSUM_BF16: return dispatch_bwd<__hip_bfloat16>(*q, ckpt, part, s);
#endif
        default: return DIMSUM_ERR_DTYPE;
    }
}
