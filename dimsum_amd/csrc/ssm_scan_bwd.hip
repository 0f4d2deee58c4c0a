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
//   * the state before every 8-step half tile comes from the forward kernel (ckpt_ptr; training callers keep it) or, for
//     callers with the reference's exact interface, from one extra state-only forward sweep into the workspace;
//   * 16-step tiles are walked backwards, each as two 8-step halves. The N state recurrences are independent, so a half
//     is processed 8 states at a time: forward sweep keeping h_t[8 states][8 steps] in 64 VGPRs, then the reverse
//     sweep over the same registers (a_t h_{t-1} = h_t - b_t: no division). Nothing per-(t, n) ever touches memory.
//     ~17 VALU ops + 2 v_exp_f32 per (t, n): the kernel is VALU-bound (the forward needs 4 + 1), HBM traffic is the
//     algorithmic minimum + the states.
//   * dB / dC are sums over the wave's 64 channels of per-lane values: a TRANSPOSED butterfly -- 64 values per lane go
//     in, one fully reduced value per lane comes out, in 6 levels of v_permlane32_swap / v_permlane16_swap / DPP adds
//     (~2.5 VALU ops per value instead of ~12 for independent wave reductions). The wave's (n, t) sums of a tile are
//     collected in LDS and stored once, 16 B per lane, as this wave's PARTIAL dB / dC; a second small kernel adds the
//     partials of the dim/64 waves of a (batch, group) in a fixed order. No atomics on dB / dC at all (the reference
//     does 1024-way contended atomicAdds per address, selective_scan_bwd_kernel.cuh:297-316) -> bitwise reproducible.
//   * u, delta, dy tiles go through an XOR-swizzled LDS transpose like in the forward; dz / out_z are computed in the
//     coalesced load layout (16 B per lane in and out) and never touch LDS.
//   * register budget <= 256 VGPRs (2 waves per SIMD cover each other's tile-staging latency).
#include "common.hpp"

namespace dimsum {

constexpr int kBT = 16;   // time steps per LDS tile (64 B per row and tensor: one HBM burst)
constexpr int kBS = 8;    // time steps per register sweep (= distance of the saved states)

// 64 rows x 16 columns fp32, row = 4 slots of 16 B, slots XOR-swizzled by (row >> 2) & 3 (conflict-free b128 both ways)
__device__ __forceinline__ int btile_off(int row, int col4) { return row * kBT + ((col4 ^ ((row >> 2) & 3)) << 2); }

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
// gen(i), i = 0..NV-1, produces this lane's value i (NV = 64 or 32); the first level is fused with the generation so
// that only NV/2 temporaries are ever live. Returns, in lane l, the 64-lane sum of value l % NV.
template <int NV, typename Gen> __device__ __forceinline__ float transposed_reduce(Gen gen, int lane) {
    static_assert(NV == 64 || NV == 32, "");
    float v[32];
    if constexpr (NV == 64) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { float a = gen(i), b = gen(i + 32); swap32(a, b); v[i] = a + b; }   // lane bit 5 <-> value bit 5
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = gen(i);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { swap16(v[i], v[i + 16]); v[i] += v[i + 16]; }                       // lane bit 4 <-> value bit 4
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

// sigmoid(x) for dt = softplus(x) = log(1 + e^x):  sigmoid(x) = 1 - exp(-dt)  (for x > 20 the reference takes dt = x and
// a derivative of 1: 1 - exp(-20) rounds to 1). Small dt: alternating series (the direct form cancels).
__device__ __forceinline__ float dsoftplus_from_dt(float dt) {
    const float ser = dt * (1.0f - dt * (0.5f - dt * (1.0f / 6 - dt * (1.0f / 24 - dt * (1.0f / 120 - dt * (1.0f / 720))))));
    return dt < 0.25f ? ser : 1.0f - fast_exp(-dt);
}

// kVec : every row base 4-element aligned and L % 4 == 0 -> 16-byte vector I/O.   kFull: all 64 lanes own a live channel.
template <typename T, int kN, bool kHasZ, bool kVec, bool kFull>
__global__ __launch_bounds__(kWave, 2) void ssm_scan_bwd_kernel(const dimsum_ssm_bwd_params_t q, const float *__restrict__ ckpt, float *__restrict__ part) {
    constexpr int kBG = 4;                        // states processed together
    constexpr int NV = kBG * kBS;                 // (state, step) values per transposed reduction
    static_assert(kN % kBG == 0 && (NV == 64 || NV == 32), "dstate must be a multiple of 4");
    const dimsum_ssm_params_t &p = q.fwd;
    __shared__ __attribute__((aligned(16))) float tU[kWave * kBT], tD[kWave * kBT], tY[kWave * kBT];   // u, dt (softplus'ed), dy
    __shared__ __attribute__((aligned(16))) float tB[kN * kBT], tC[kN * kBT];
    __shared__ __attribute__((aligned(16))) float tdB[kN * kBT], tdC[kN * kBT];      // this wave's dB / dC sums of the tile
    // Per-(state, lane) values that persist over the whole walk: the running dA sum lives in LDS ([n][lane], conflict
    // free); the reverse-recurrence carry dh in registers for dstate <= 16 (-> 20 KB of LDS per wave = 8 waves per CU),
    // in LDS otherwise; A is re-read from L1/L2 one group ahead. The state-group loop stays rolled (one copy of the body;
    // real control flow between groups keeps the scheduler from interleaving them and blowing the register budget);
    // a uniform switch moves the group's dh in and out of the static register array.
    constexpr bool kRegState = kN <= 16;
    __shared__ float sdA[kN * kWave], sdh[kRegState ? 1 : kN * kWave];
    float rdh[kRegState ? kN : 1];      // only ever indexed with compile-time constants (see the group loop)

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
    const int nd = kFull ? kWave : min(kWave, (g + 1) * dpg - d0);
    const bool live = kFull || lane < nd;
    const int d = d0 + (kFull ? lane : min(lane, nd - 1));

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

    const float *Ap = reinterpret_cast<const float *>(p.A_ptr) + (int64_t)d * p.A_d_stride;
    const int A_ns = (int)p.A_dstate_stride;
#pragma unroll
    for (int n = 0; n < kN; ++n) {
        sdA[n * kWave + lane] = 0.f;
        if constexpr (kRegState) rdh[n] = 0.f;
        else sdh[n * kWave + lane] = 0.f;
    }
    const float Dval = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d] : 0.f;
    const float bias = p.delta_bias_ptr ? reinterpret_cast<const float *>(p.delta_bias_ptr)[d] : 0.f;
    const bool softplus = p.delta_softplus != 0;
    float dD = 0.f, dbias = 0.f;
    float dt_next = 0.f;   // dt of the first step of the tile processed before (later in time): a_{t+1} at the seam

    const int n_tiles = (L + kBT - 1) / kBT;
    const int n_halves = (L + kBS - 1) / kBS;
    // saved states: [b][half tile][n][d]
    const float *ck_lane = ckpt + (int64_t)b * n_halves * kN * p.dim + d;
    const int ck_ns = p.dim;                                    // stride between states
    // coalesced tile layout: 64 rows x 16 columns = 4 pieces of (16 rows x 4 lanes-per-row x 4 columns)
    const int lrow = lane >> 2, lc4 = lane & 3, lcol = lc4 * 4;

    // stage a 64 x 16 tile of `src` (rows = channels) into the swizzled LDS image `dst`; out-of-range -> 0
    auto stage = [&](const T *base, int ds, int t0, float *dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            f32x4 v = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (kVec) {
                if ((kFull || row < nd) && t0 + lcol < L) v = widen(ld4<T>(at(base + i * 16 * ds, (unsigned)(lrow * ds + t0 + lcol))));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (row < nd && t0 + lcol + e < L) v.v[e] = to_f32<T>(base[(unsigned)(row * ds + t0 + lcol + e)]);
            }
            *reinterpret_cast<f32x4 *>(&dst[btile_off(row, lc4)]) = v;
        }
    };
    auto stage_bc = [&](int t0) {
        for (int idx = lane; idx < kN * 4; idx += kWave) {
            const int n = idx >> 2, c4 = idx & 3;
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
    };

    // states of the first group to be processed (last half tile, first kBG states); later groups are prefetched one ahead
    float h_pre[kBG], a_pre[kBG];
#pragma unroll
    for (int k = 0; k < kBG; ++k) { h_pre[k] = ck_lane[((int64_t)(n_halves - 1) * kN + k) * ck_ns]; a_pre[k] = Ap[k * A_ns]; }

#pragma unroll 1
    for (int tile = n_tiles - 1; tile >= 0; --tile) {
        const int t0 = tile * kBT;
        stage(u_base, u_ds, t0, tU);
        stage(dl_base, dl_ds, t0, tD);
        stage_bc(t0);
        // ---- dy = dout * silu(z), dz, optional out_z -- in the coalesced layout; dy goes to LDS ------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            f32x4 dy = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (kVec) {
                if ((kFull || row < nd) && t0 + lcol < L) {
                    const unsigned col = (unsigned)(t0 + lcol);
                    const f32x4 go = widen(ld4<T>(at(do_base + i * 16 * do_ds, (unsigned)(lrow * do_ds) + col)));
                    if constexpr (kHasZ) {
                        const f32x4 zv = widen(ld4<T>(at(z_base + i * 16 * z_ds, (unsigned)(lrow * z_ds) + col)));
                        const f32x4 yv = widen(ld4<T>(at(y_base + i * 16 * y_ds, (unsigned)(lrow * y_ds) + col)));
                        f32x4 dz, oz;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sgz = sigmoidf_fast(zv.v[e]), silu = zv.v[e] * sgz;
                            dz.v[e] = go.v[e] * yv.v[e] * sgz * (1.0f + zv.v[e] * (1.0f - sgz));
                            oz.v[e] = yv.v[e] * silu;
                            dy.v[e] = go.v[e] * silu;
                        }
                        st4<T>(at(dz_base + i * 16 * dz_ds, (unsigned)(lrow * dz_ds) + col), dz);
                        if (oz_base) st4<T>(at(oz_base + i * 16 * oz_ds, (unsigned)(lrow * oz_ds) + col), oz);
                    } else {
                        dy = go;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = t0 + lcol + e;
                    if (row < nd && t < L) {
                        const float go = to_f32<T>(do_base[(unsigned)(row * do_ds + t)]);
                        if constexpr (kHasZ) {
                            const float zv = to_f32<T>(z_base[(unsigned)(row * z_ds + t)]), yv = to_f32<T>(y_base[(unsigned)(row * y_ds + t)]);
                            const float sgz = sigmoidf_fast(zv), silu = zv * sgz;
                            dz_base[(unsigned)(row * dz_ds + t)] = from_f32<T>(go * yv * sgz * (1.0f + zv * (1.0f - sgz)));
                            if (oz_base) oz_base[(unsigned)(row * oz_ds + t)] = from_f32<T>(yv * silu);
                            dy.v[e] = go * silu;
                        } else {
                            dy.v[e] = go;
                        }
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(&tY[btile_off(row, lc4)]) = dy;
        }
        // ---- in place: tD <- softplus(delta + bias) (0 beyond L: dead steps are identities, a = 1, b = 0) ----------------
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
        const float dt_first = (*reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, 0)])).v[0];

#pragma unroll 1
        for (int half = kBT / kBS - 1; half >= 0; --half) {
            const int hidx = tile * (kBT / kBS) + half;          // index of this half tile's saved state
            const int jb = half * (kBS / 4);                     // first 4-step slot of the half
            if (hidx >= n_halves) continue;                      // a trailing half entirely beyond L
            // dt of the step after the half: next slot of the tile, or the seam to the tile processed before
            const float dt_after = (half == kBT / kBS - 1) ? dt_next : (*reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, jb + kBS / 4)])).v[0];
            float s1[kBS], s2[kBS];
#pragma unroll
            for (int t = 0; t < kBS; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

#pragma unroll 1
            for (int n0 = 0; n0 < kN; n0 += kBG) {
                float H[kBG * kBS];       // [k][t]: h_t of state n0+k, later overwritten by the dB terms
                float hk[kBG], Ak[kBG], dAk[kBG], dhk[kBG];
#pragma unroll
                for (int k = 0; k < kBG; ++k) {
                    hk[k] = h_pre[k];
                    Ak[k] = a_pre[k] * kLog2e;            // exp(dt A) = exp2(dt A log2 e)
                    dAk[k] = sdA[(n0 + k) * kWave + lane];
                    if constexpr (!kRegState) dhk[k] = sdh[(n0 + k) * kWave + lane];
                }
                if constexpr (kRegState) {
                    // uniform selects on static indices (a `switch` gets merged into a dynamically indexed private array,
                    // i.e. scratch memory = HBM round trips in the hot loop)
#pragma unroll
                    for (int k = 0; k < kBG; ++k) {
                        float v = rdh[k];
#pragma unroll
                        for (int gq = 1; gq < kN / kBG; ++gq) v = (n0 == gq * kBG) ? rdh[gq * kBG + k] : v;
                        dhk[k] = v;
                    }
                }
                {   // prefetch the saved states of the next group (next kBG states of this half, or the previous half's first)
                    const bool wrap = n0 + kBG >= kN;
                    const int nh = wrap ? max(hidx - 1, 0) : hidx, nn = wrap ? 0 : n0 + kBG;
#pragma unroll
                    for (int k = 0; k < kBG; ++k) { h_pre[k] = ck_lane[((int64_t)nh * kN + nn + k) * ck_ns]; a_pre[k] = Ap[(nn + k) * A_ns]; }
                }
                // ---- forward sweep: h_t for the 8 steps of the half (B row of the next (slot, state) fetched one ahead) --------
                {
                    f32x4 bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[n0 * kBT + jb * 4]);
#pragma unroll
                    for (int jj = 0; jj < kBS / 4; ++jj) {
                        const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, jb + jj)]);
                        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, jb + jj)]);
                        float du[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) du[s] = d4.v[s] * u4.v[s];
#pragma unroll
                        for (int k = 0; k < kBG; ++k) {
                            const f32x4 bq = bq_nxt;
                            if (k + 1 < kBG) bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[(n0 + k + 1) * kBT + (jb + jj) * 4]);
                            else if (jj + 1 < kBS / 4) bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[n0 * kBT + (jb + jj + 1) * 4]);
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                hk[k] = fmaf(fast_exp2(d4.v[s] * Ak[k]), hk[k], bq.v[s] * du[s]);
                                H[k * kBS + jj * 4 + s] = hk[k];
                            }
                        }
                    }
                }
                // ---- dC[n, t] = sum_d dy_t h_t[n]: transposed reduction of the (k, t) products -----------------------------
                {
                    float y8[kBS];
#pragma unroll
                    for (int jj = 0; jj < kBS / 4; ++jj) {
                        const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, jb + jj)]);
#pragma unroll
                        for (int s = 0; s < 4; ++s) y8[jj * 4 + s] = live ? y4.v[s] : 0.f;
                    }
                    const float r = transposed_reduce<NV>([&](int i) { return y8[i & (kBS - 1)] * H[i]; }, lane);
                    const int vi = lane & (NV - 1);
                    if (lane < NV) tdC[(n0 + (vi >> 3)) * kBT + half * kBS + (vi & 7)] = r;
                }
                // ---- reverse sweep (B / C rows of the next (slot, state) are fetched one iteration ahead) -------------------
                {
                    f32x4 bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[n0 * kBT + (jb + kBS / 4 - 1) * 4]);
                    f32x4 cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[n0 * kBT + (jb + kBS / 4 - 1) * 4]);
                    float dt_succ = dt_after;       // dt of the step after the current slot
#pragma unroll
                    for (int jj = kBS / 4 - 1; jj >= 0; --jj) {
                        const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, jb + jj)]);
                        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, jb + jj)]);
                        const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, jb + jj)]);
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
                                bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[(n0 + k + 1) * kBT + (jb + jj) * 4]);
                                cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[(n0 + k + 1) * kBT + (jb + jj) * 4]);
                            } else if (jj > 0) {
                                bq_nxt = *reinterpret_cast<const f32x4 *>(&tB[n0 * kBT + (jb + jj - 1) * 4]);
                                cq_nxt = *reinterpret_cast<const f32x4 *>(&tC[n0 * kBT + (jb + jj - 1) * 4]);
                            }
#pragma unroll
                            for (int s = 3; s >= 0; --s) {
                                const int t = jj * 4 + s;
                                const float a_next = fast_exp2(dnext[s] * Ak[k]);
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
                for (int k = 0; k < kBG; ++k) {
                    sdA[(n0 + k) * kWave + lane] = dAk[k];
                    if constexpr (!kRegState) sdh[(n0 + k) * kWave + lane] = dhk[k];
                }
                if constexpr (kRegState) {
#pragma unroll
                    for (int gq = 0; gq < kN / kBG; ++gq)
#pragma unroll
                        for (int k = 0; k < kBG; ++k) rdh[gq * kBG + k] = (n0 == gq * kBG) ? dhk[k] : rdh[gq * kBG + k];
                }
                // ---- dB[n, t] = sum_d dh_t[n] dt_t u_t -----------------------------------------------------------------------
                {
                    const float r = transposed_reduce<NV>([&](int i) { return live ? H[i] : 0.f; }, lane);
                    const int vi = lane & (NV - 1);
                    if (lane < NV) tdB[(n0 + (vi >> 3)) * kBT + half * kBS + (vi & 7)] = r;
                }
            }

            // ---- per-(d, t) results of the half: du, ddelta (softplus chain), dD, ddelta_bias; parked in LDS over u / dy
            //      (s2 was accumulated with A * log2 e) ----------------------------------------------------------------------
#pragma unroll
            for (int jj = 0; jj < kBS / 4; ++jj) {
                const f32x4 u4 = *reinterpret_cast<const f32x4 *>(&tU[btile_off(lane, jb + jj)]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&tD[btile_off(lane, jb + jj)]);
                const f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tY[btile_off(lane, jb + jj)]);
                f32x4 du4, dd4;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int t = jj * 4 + s;
                    du4.v[s] = fmaf(d4.v[s], s1[t], Dval * y4.v[s]);
                    dD = fmaf(y4.v[s], u4.v[s], dD);
                    const float ddt = fmaf(u4.v[s], s1[t], s2[t] * kLn2);
                    // dead steps (t >= L) have dt = 0 -> factor 0 with softplus; without it they carry u = dy = 0 -> ddt = 0
                    dd4.v[s] = softplus ? ddt * dsoftplus_from_dt(d4.v[s]) : ddt;
                    dbias += dd4.v[s];
                }
                *reinterpret_cast<f32x4 *>(&tU[btile_off(lane, jb + jj)]) = du4;
                *reinterpret_cast<f32x4 *>(&tY[btile_off(lane, jb + jj)]) = dd4;
            }
        }
        dt_next = dt_first;

        // ---- this wave's partial dB / dC of the tile: 16 B per lane, rows of 64 B ------------------------------------------
        for (int idx = lane; idx < kN * 4; idx += kWave) {
            const int n = idx >> 2, c = (idx & 3) * 4, t = t0 + c;
            const f32x4 vb = *reinterpret_cast<const f32x4 *>(&tdB[n * kBT + c]);
            const f32x4 vc = *reinterpret_cast<const f32x4 *>(&tdC[n * kBT + c]);
            if (t + 3 < L) {
                st4<float>(pB + (int64_t)n * L + t, vb);
                st4<float>(pC + (int64_t)n * L + t, vc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (t + e < L) { pB[(int64_t)n * L + t + e] = vb.v[e]; pC[(int64_t)n * L + t + e] = vc.v[e]; }
            }
        }
        // ---- coalesced stores of du, ddelta -------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + lrow;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&tU[btile_off(row, lc4)]);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(&tY[btile_off(row, lc4)]);
            if constexpr (kVec) {
                if ((kFull || row < nd) && t0 + lcol < L) {
                    st4<T>(at(du_base + i * 16 * du_ds, (unsigned)(lrow * du_ds + t0 + lcol)), a);
                    st4<T>(at(dd_base + i * 16 * dd_ds, (unsigned)(lrow * dd_ds + t0 + lcol)), c);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = t0 + lcol + e;
                    if (row < nd && t < L) {
                        du_base[(unsigned)(row * du_ds + t)] = from_f32<T>(a.v[e]);
                        dd_base[(unsigned)(row * dd_ds + t)] = from_f32<T>(c.v[e]);
                    }
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

// dB[b, g, n, t] = sum over the waves w of (b, g), in index order, of part[b, g, w][0][n][t]  (same for dC)
__global__ __launch_bounds__(256) void ssm_scan_bwd_reduce_kernel(const float *__restrict__ part, const dimsum_ssm_bwd_params_t q, int waves_per_group) {
    const dimsum_ssm_params_t &p = q.fwd;
    const int L = p.seqlen, N = p.dstate;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over (b, g, which, n, t)
    const int64_t total = (int64_t)p.batch * p.n_groups * 2 * N * L;
    if (i >= total) return;
    const int t = (int)(i % L);
    int64_t r = i / L;
    const int n = (int)(r % N); r /= N;
    const int which = (int)(r & 1); r >>= 1;
    const int g = (int)(r % p.n_groups);
    const int b = (int)(r / p.n_groups);
    const float *src = part + (((int64_t)(b * p.n_groups + g) * waves_per_group) * 2 + which) * N * L + (int64_t)n * L + t;
    float acc = 0.f;
    for (int w = 0; w < waves_per_group; ++w) acc += src[(int64_t)w * 2 * N * L];
    if (which == 0) reinterpret_cast<float *>(q.dB_ptr)[(int64_t)b * q.dB_batch_stride + (int64_t)g * q.dB_group_stride + (int64_t)n * q.dB_dstate_stride + t] = acc;
    else reinterpret_cast<float *>(q.dC_ptr)[(int64_t)b * q.dC_batch_stride + (int64_t)g * q.dC_group_stride + (int64_t)n * q.dC_dstate_stride + t] = acc;
}

template <typename T, int kN>
static int launch_bwd(const dimsum_ssm_bwd_params_t &q, const float *ckpt, float *part, hipStream_t stream) {
    const dimsum_ssm_params_t &p = q.fwd;
    const int dpg = p.dim / p.n_groups;
    const int tiles = p.batch * p.n_groups * ((dpg + kWave - 1) / kWave);
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
    const bool full = vec && (dpg % kWave == 0);
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
    const int64_t total = (int64_t)p.batch * p.n_groups * 2 * kN * p.seqlen;
    hipLaunchKernelGGL(ssm_scan_bwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, part, q, (dpg + kWave - 1) / kWave);
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
    const int64_t waves = (int64_t)batch * n_groups * ((dpg + dimsum::kWave - 1) / dimsum::kWave);
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
