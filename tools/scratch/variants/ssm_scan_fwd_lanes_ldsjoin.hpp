// ssm_scan_fwd_lanes.hpp -- selective scan forward, lane = (channel, state): 16 lanes per channel, 4 channels per wave.
//
// Same math and interface as ssm_scan_fwd_kernel (ssm_scan_fwd_kernel.hpp; reference selective_scan_fwd_kernel.cuh:67-303),
// dstate 16 only. The end of the widening along the STATE axis that ssm_scan_fwd_split.hpp starts: a launch with so few
// channels that even 16 per wave leave the SIMDs with one wave each (16 x 1152 channels x 4096 steps = 1152 such waves) is
// bound by the latency of a single wave's dependent chain. Here every lane carries ONE state, a wave 4 channels, and the
// launch has 16x the waves of the 64-channel kernel (4608 for that shape) at the same VALU cost per (t, n):
//   * per step a lane issues mul, v_exp_f32, mul, fma, mul -- the per-channel products dt * u are formed once per element in
//     the coalesced load layout, D u is added in the coalesced epilogue, sum(dt) (for the chunk state's prod a) is a DPP row
//     sum per tile in the load layout;
//   * y_t = sum_n C_t[n] h_t[n] crosses 16 lanes. The kernel is VALU-bound and cross-lane VALU moves are dear on gfx950
//     (v_permlane32_swap / v_permlane16_swap ~3.5 plain ops, an add with a DPP source ~2: DESIGN.md section 3), so the sum goes
//     through the otherwise idle LDS pipe: each lane writes its h c of 8 steps into a [channel][step][state] buffer
//     (ds_write_b32 with immediate offsets, conflict-free), reads back 8 states of ONE step as 2 x ds_read_b128, adds them
//     (7 adds) and joins the two halves with one quad_perm add: 1.25 VALU ops per step instead of 4.2 for a butterfly;
//   * u / dt tiles are 4 channels x 64 steps (256-B row segments: two whole HBM lines per row), B / C are staged per 32 steps
//     as [n][32] with the next half requested in registers; all images XOR-swizzled on address bits the 4-step immediates
//     do not touch, so the scan's reads are bank-conflict free (ds_read_b128 is serviced in four fixed 16-lane groups,
//     MI355X_MICROARCH.md LDS) and need two VALU address ops per 8 steps; 8 KB of LDS per wave = 20 waves per CU.
#pragma once
#include "ssm_scan_fwd_split.hpp"   // the helpers of ssm_scan_fwd_kernel.hpp

namespace dimsum {

constexpr int kLT = 64;    // time steps per u / dt tile
constexpr int kLH = 32;    // time steps per B / C half tile
constexpr int kLC = 4;     // channels per wave
constexpr int kLG = 8;     // time steps per join group

// u / dt tile: rows = the wave's 4 channels (one 256-B bank row each), 16-byte slots XOR-ed with 4 x row: a read group holds
// 4 channels. B / C half tile: rows = the 16 states (two rows per bank row), slots XOR-ed with 2 x ((row / 2) % 4): a read group
// holds 8 consecutive states. Bit 0 of the slot index -- which of a group's two 4-step quads -- is never touched.
__device__ __forceinline__ int lt_off(int row, int col4) { return row * kLT + ((col4 ^ ((row & 3) << 2)) << 2); }
__device__ __forceinline__ int bc_off(int row, int col4) { return row * kLH + ((col4 ^ (((row >> 1) & 3) << 1)) << 2); }

template <int CTRL> __device__ __forceinline__ float lanes_dpp(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}

template <typename T, bool kHasZ, bool kVec, bool kFull, bool kCkpt = false>
__global__ __launch_bounds__(kWave, 5) void ssm_scan_fwd_lanes_kernel(const dimsum_ssm_params_t p) {
    static_assert(!kFull || kVec, "kFull implies kVec");
    constexpr int kN = 16;
    __shared__ __attribute__((aligned(16))) float tileU[kLC * kLT];       // dt * u, then y in place
    __shared__ __attribute__((aligned(16))) float tileD[kLC * kLT];       // dt
    __shared__ __attribute__((aligned(16))) float tileB[kN * kLH];
    __shared__ __attribute__((aligned(16))) float tileC[kN * kLH];
    __shared__ __attribute__((aligned(16))) float tileY[kLC * kLG * kN];  // h c of one join group, [channel][step][state]

    // lane = s0 + 2 s1 + 4 c + 16 s2 + 32 s3, state n = s0 + 2 s1 + 4 s2 + 8 s3

    const int lane = threadIdx.x;
    const int c = (lane >> 2) & 3;                         // channel of the wave's 4
    const int n = (lane & 3) | ((lane >> 4) << 2);         // state
    const int L = p.seqlen;
    const int dpg = p.dim / p.n_groups;
    const int tiles_per_group = (dpg + kLC - 1) / kLC;
    const int tiles_per_batch = p.n_groups * tiles_per_group;
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);   // a batch element's waves share an XCD (one L2 for B / C)
    const int b = wg / tiles_per_batch;
    const int rem = wg - b * tiles_per_batch;
    const int g = rem / tiles_per_group;
    const int d0 = g * dpg + (rem - g * tiles_per_group) * kLC;
    const int nd = kFull ? kLC : min(kLC, (g + 1) * dpg - d0);
    const bool live = kFull || c < nd;
    const int d = d0 + (kFull ? c : min(c, nd - 1));

    const T *u_base = reinterpret_cast<const T *>(p.u_ptr) + (int64_t)b * p.u_batch_stride + (int64_t)d0 * p.u_d_stride;
    const T *dl_base = reinterpret_cast<const T *>(p.delta_ptr) + (int64_t)b * p.delta_batch_stride + (int64_t)d0 * p.delta_d_stride;
    const T *z_base = kHasZ ? reinterpret_cast<const T *>(p.z_ptr) + (int64_t)b * p.z_batch_stride + (int64_t)d0 * p.z_d_stride : nullptr;
    T *out_base = p.out_ptr ? reinterpret_cast<T *>(p.out_ptr) + (int64_t)b * p.out_batch_stride + (int64_t)d0 * p.out_d_stride : nullptr;
    T *oz_base = kHasZ ? reinterpret_cast<T *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride : nullptr;
    const int u_ds = (int)p.u_d_stride, dl_ds = (int)p.delta_d_stride, z_ds = (int)p.z_d_stride;
    const int out_ds = (int)p.out_d_stride, oz_ds = (int)p.out_z_d_stride;
    const T *Bp = reinterpret_cast<const T *>(p.B_ptr) + (int64_t)b * p.B_batch_stride + (int64_t)g * p.B_group_stride;
    const T *Cp = reinterpret_cast<const T *>(p.C_ptr) + (int64_t)b * p.C_batch_stride + (int64_t)g * p.C_group_stride;
    const int Bns = (int)p.B_dstate_stride, Cns = (int)p.C_dstate_stride;

    const float A2 = reinterpret_cast<const float *>(p.A_ptr)[(int64_t)d * p.A_d_stride + (int64_t)n * p.A_dstate_stride] * kLog2e;
    float h = 0.f;
    const float *bias_p = reinterpret_cast<const float *>(p.delta_bias_ptr);
    const bool softplus = p.delta_softplus != 0;
    const bool has_out = out_base != nullptr;
    float *ck_base = (kCkpt && p.ckpt_ptr && live) ? reinterpret_cast<float *>(p.ckpt_ptr) + (int64_t)b * ((L + 7) / 8) * kN * p.dim + (int64_t)n * p.dim + d : nullptr;

    // join buffer, float index = (state % 8) + 8 ((state / 8) ^ (c & 1)) + 16 (c / 2) + 32 (c & 1) + 64 step: the 64 writes of one
    // step fall on 64 different words of two bank rows; lane (c, n) reads back step n / 2, states 8 (n & 1) .. + 7.
    float *const y_wr = &tileY[(n & 7) | (((n >> 3) ^ (c & 1)) << 3) | ((c >> 1) << 4) | ((c & 1) << 5)];
    const float *const y_rd = &tileY[(((n & 1) ^ (c & 1)) << 3) | ((c >> 1) << 4) | ((c & 1) << 5) | ((n >> 1) << 6)];

    const int n_tiles = (L + kLT - 1) / kLT;
    // load layouts. u / dt / z / out: lane -> (row = lane / 16, 4 columns at (lane % 16) * 4).  B / C: piece i of a half holds
    // rows 8 i + lane / 8, 4 columns at (lane % 8) * 4.
    const int lrow = lane >> 4, lc4 = lane & 15, lcol = lc4 * 4;
    const int ldrow = kFull ? lrow : min(lrow, nd - 1);
    const int brow8 = lane >> 3, bc4 = lane & 7;
    const float brow = bias_p ? bias_p[d0 + ldrow] : 0.f;
    const float Drow = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d0 + ldrow] : 0.f;
    float sum_dt = 0.f;   // load layout: sum of dt of row lrow so far (all 16 lanes of the DPP row hold it)

    Raw4<T> ru, rd, rz, rb[2], rc[2];
    auto issue_ud = [&](int t0) {
        const int col = min(t0 + lcol, L - 4);
        ru = ld4<T>(at(u_base, (unsigned)(ldrow * u_ds + col)));
        rd = ld4<T>(at(dl_base, (unsigned)(ldrow * dl_ds + col)));
    };
    auto issue_bc = [&](int th) {
        const int col = min(th + bc4 * 4, L - 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rb[i] = ld4<T>(at(Bp, (unsigned)((i * 8 + brow8) * Bns + col)));
            rc[i] = ld4<T>(at(Cp, (unsigned)((i * 8 + brow8) * Cns + col)));
        }
    };

    if constexpr (kVec) { issue_ud(0); issue_bc(0); }

#pragma unroll 1
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kLT;
        f32x4 uk = {{0.f, 0.f, 0.f, 0.f}};
        // ---- stage u, dt = softplus(delta + bias) (0 beyond L: a = 1, b = 0, the state is untouched) and dt * u ----------------
        if constexpr (kVec) {
            const bool col_ok = t0 + lcol < L;
            uk = widen(ru);
            f32x4 vd = widen(rd), vdu;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vd.v[k] = col_ok ? softplus_if(vd.v[k] + brow, softplus) : 0.f;
                vdu.v[k] = vd.v[k] * uk.v[k];
                s += vd.v[k];
            }
            *reinterpret_cast<f32x4 *>(&tileU[lt_off(lrow, lc4)]) = vdu;
            *reinterpret_cast<f32x4 *>(&tileD[lt_off(lrow, lc4)]) = vd;
            if (p.x_ptr) {   // row sum over the 16 lanes of the DPP row
                s += lanes_dpp<0x128>(s);   // row_ror:8
                s += lanes_dpp<0x124>(s);   // row_ror:4
                s += lanes_dpp<0x4E>(s);    // quad_perm [2,3,0,1]
                s += lanes_dpp<0xB1>(s);    // quad_perm [1,0,3,2]
                sum_dt += s;
            }
            if (tile + 1 < n_tiles) issue_ud(t0 + kLT);   // flies under the compute below
            if constexpr (kHasZ) rz = ld4<T>(at(z_base, (unsigned)(ldrow * z_ds + min(t0 + lcol, L - 4))));
        } else {
            for (int i = 0; i < kLC * kLT / kWave; ++i) {
                const int idx = i * kWave + lane, row = idx / kLT, col = idx & (kLT - 1);
                const bool ok = row < nd && t0 + col < L;
                float vu = 0.f, vd = 0.f;
                if (ok) {
                    vu = to_f32<T>(u_base[(unsigned)(row * u_ds + t0 + col)]);
                    vd = softplus_if(to_f32<T>(dl_base[(unsigned)(row * dl_ds + t0 + col)]) + (bias_p ? bias_p[d0 + row] : 0.f), softplus);
                }
                tileU[lt_off(row, col >> 2) + (col & 3)] = vd * vu;
                tileD[lt_off(row, col >> 2) + (col & 3)] = vd;
            }
            if (p.x_ptr) {   // all lanes of DPP row r end up with the row's sum of dt (LDS is in order within the wave)
                float s = 0.f;
                for (int col = lc4; col < kLT; col += 16) s += tileD[lt_off(ldrow, col >> 2) + (col & 3)];
                s += lanes_dpp<0x128>(s);
                s += lanes_dpp<0x124>(s);
                s += lanes_dpp<0x4E>(s);
                s += lanes_dpp<0xB1>(s);
                sum_dt += s;
            }
        }

#pragma unroll 1
        for (int half = 0; half < kLT / kLH; ++half) {
            const int th = t0 + half * kLH;
            if (th >= L) break;
            // ---- stage B, C of these 32 steps; request the next 32 ----------------------------------------------------------
            if constexpr (kVec) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    *reinterpret_cast<f32x4 *>(&tileB[bc_off(i * 8 + brow8, bc4)]) = widen(rb[i]);
                    *reinterpret_cast<f32x4 *>(&tileC[bc_off(i * 8 + brow8, bc4)]) = widen(rc[i]);
                }
                if (th + kLH < L) issue_bc(th + kLH);
            } else {
                for (int idx = lane; idx < kN * kLH; idx += kWave) {
                    const int r = idx / kLH, col = idx & (kLH - 1), tc = min(th + col, L - 1);
                    tileB[bc_off(r, col >> 2) + (col & 3)] = to_f32<T>(Bp[(unsigned)(r * Bns + tc)]);
                    tileC[bc_off(r, col >> 2) + (col & 3)] = to_f32<T>(Cp[(unsigned)(r * Cns + tc)]);
                }
            }

            // ---- 32 sequential steps in 4 groups of 8; per step and lane: mul, v_exp_f32, mul, fma, mul, ds_write_b32. The operands
            //      of the next 4 steps are requested from LDS before the current 4 are computed (also across the join) ----------
            struct Ops { f32x4 du, dt, b, c; };
            // LDS offsets of a group's first quad; the second quad is the next 16 bytes (the swizzles leave slot bit 0 alone)
            // (offsets in 16-byte units so that the reads stay ds_read_b128)
            auto ud_of = [&](int gq) { return lt_off(c, half * (kLH / 4) + gq * 2) >> 2; };
            auto bc_of = [&](int gq) { return bc_off(n, gq * 2) >> 2; };
            auto fetch = [&](int ud, int bc) {
                Ops o;
                o.du = reinterpret_cast<const f32x4 *>(tileU)[ud];
                o.dt = reinterpret_cast<const f32x4 *>(tileD)[ud];
                o.b = reinterpret_cast<const f32x4 *>(tileB)[bc];
                o.c = reinterpret_cast<const f32x4 *>(tileC)[bc];
                return o;
            };
            int ud = ud_of(0), bc = bc_of(0);
            Ops nxt = fetch(ud, bc);
            const int y_st = ((n >> 3) << 2) | ((n >> 1) & 3);   // step n / 2 of the group, relative to the group's first quad
#pragma unroll 1
            for (int gq = 0; gq < kLH / kLG; ++gq) {
                const int tg = th + gq * kLG;
                if (tg >= L) break;
                if (kCkpt && ck_base) ck_base[(int64_t)(tg >> 3) * kN * p.dim] = h;
                const int ud_n = ud_of(min(gq + 1, kLH / kLG - 1)), bc_n = bc_of(min(gq + 1, kLH / kLG - 1));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const Ops o = nxt;
                    nxt = q == 0 ? fetch(ud + 1, bc + 1) : fetch(ud_n, bc_n);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        h = fmaf(fast_exp2(o.dt.v[s] * A2), h, o.b.v[s] * o.du.v[s]);
                        y_wr[(q * 4 + s) * 64] = h * o.c.v[s];
                    }
                }
                // lane (c, n): states 8 (n & 1) .. + 7 of step n / 2, then the other half of the states from lane n ^ 1
                const f32x4 ya = *reinterpret_cast<const f32x4 *>(y_rd), yb = *reinterpret_cast<const f32x4 *>(y_rd + 4);
                float yt = ((ya.v[0] + ya.v[1]) + (ya.v[2] + ya.v[3])) + ((yb.v[0] + yb.v[1]) + (yb.v[2] + yb.v[3]));
                yt += lanes_dpp<0xB1>(yt);                                   // quad_perm [1,0,3,2]
                tileU[ud * 4 + y_st] = yt;                                       // both lanes of a pair store the same value
                ud = ud_n; bc = bc_n;
            }
        }

        // ---- chunk-state store at every 2048 boundary and at the end (selective_scan_fwd_kernel.cuh:251-254) ---
        const int t_end = min(t0 + kLT, L);
        if (p.x_ptr && ((t_end & 2047) == 0 || t_end == L)) {
            // sum(dt) lives in the load layout (DPP row r = channel r): hand it to the scan layout through LDS. tileD is
            // dead here; the wave's LDS operations execute in order.
            if (lc4 == 0) tileD[lrow] = sum_dt;
            const float sd = tileD[c];
            if (live) {
                float *xr = reinterpret_cast<float *>(p.x_ptr) + (((int64_t)b * p.dim + d) * p.n_chunks + (t_end - 1) / 2048) * (2 * kN) + 2 * n;
                v2f v; v.x = fast_exp2(A2 * sd); v.y = h;
                *reinterpret_cast<v2f *>(xr) = v;
            }
        }

        // ---- epilogue: y in the coalesced layout, + D u, gate, store -------------------------------------------------
        if constexpr (kVec) {
            if (t0 + lcol < L && (kFull || lrow < nd)) {
                f32x4 y4 = *reinterpret_cast<const f32x4 *>(&tileU[lt_off(lrow, lc4)]);
#pragma unroll
                for (int k = 0; k < 4; ++k) y4.v[k] = fmaf(Drow, uk.v[k], y4.v[k]);
                if (has_out) st4<T>(at(out_base, (unsigned)(lrow * out_ds + t0 + lcol)), y4);
                if constexpr (kHasZ) {
                    const f32x4 z4 = widen(rz);
#pragma unroll
                    for (int k = 0; k < 4; ++k) y4.v[k] *= z4.v[k] * sigmoidf_fast(z4.v[k]);
                    st4<T>(at(oz_base, (unsigned)(lrow * oz_ds + t0 + lcol)), y4);
                }
            }
        } else {
            for (int i = 0; i < kLC * kLT / kWave; ++i) {
                const int idx = i * kWave + lane, row = idx / kLT, col = idx & (kLT - 1);
                if (row < nd && t0 + col < L) {
                    float yv = tileU[lt_off(row, col >> 2) + (col & 3)];
                    if (p.D_ptr) yv = fmaf(reinterpret_cast<const float *>(p.D_ptr)[d0 + row], to_f32<T>(u_base[(unsigned)(row * u_ds + t0 + col)]), yv);
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

// ---- launcher: explicitly instantiated per I/O dtype in ssm_scan_fwd_split_{f32,f16,bf16}.hip ---------------------------------
template <typename T>
void ssm_scan_fwd_launch_lanes(const dimsum_ssm_params_t &p, hipStream_t stream, int tiles, bool vec, bool full) {
    const dim3 grid(tiles), block(kWave);
#define DIMSUM_LAUNCH(HASZ, VEC, FULL)                                                                                        \
    do {                                                                                                                       \
        if (p.ckpt_ptr) hipLaunchKernelGGL((ssm_scan_fwd_lanes_kernel<T, HASZ, VEC, FULL, true>), grid, block, 0, stream, p);   \
        else hipLaunchKernelGGL((ssm_scan_fwd_lanes_kernel<T, HASZ, VEC, FULL, false>), grid, block, 0, stream, p);             \
    } while (0)
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

#define DIMSUM_INSTANTIATE_FWD_LANES(T) \
    template void ssm_scan_fwd_launch_lanes<T>(const dimsum_ssm_params_t &, hipStream_t, int, bool, bool);

}  // namespace dimsum
