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
//   * lane = s0 + 2 s1 + 4 c + 16 s2 + 32 s3 (state n = s0 + 2 s1 + 4 s2 + 8 s3): y_t = sum_n C_t[n] h_t[n] of 16 steps is a
//     transposed butterfly -- v_permlane32_swap (8 values), v_permlane16_swap (4), quad_perm DPP with selects (2 + 1) --
//     33 VALU ops per 16 steps, after which lane (c, n) holds the finished y of step n of the group;
//   * tiles are 4 channels x 64 steps (256-B row segments: two whole HBM lines per row), B / C as [n][64]; 16-byte slots
//     XOR-swizzled by 4 x row so that the load-layout writes, the broadcast reads of the scan and the scattered y writes
//     are all bank-conflict free without padding; 10 KB of LDS per wave = 16 waves per CU.
#pragma once
#include "ssm_scan_fwd_split.hpp"   // swap_halves / swap_rows and the helpers of ssm_scan_fwd_kernel.hpp

namespace dimsum {

constexpr int kLT = 64;    // time steps per tile
constexpr int kLC = 4;     // channels per wave

// u / dt tile: rows = the wave's 4 channels, 16-byte slots XOR-ed with 4 x row. B / C tile: rows = the 16 states, slots XOR-ed with
// 2 x (row % 8): a ds_read_b128 is serviced in four fixed groups of 16 lanes (MI355X_MICROARCH.md, LDS) and with
// lane = s0 + 2 s1 + 4 c + 16 s2 + 32 s3 every group holds 4 channels or 8 states, each on its own slot of the 256-B bank row.
__device__ __forceinline__ int lt_off(int row, int col4) { return row * kLT + ((col4 ^ ((row & 3) << 2)) << 2); }
__device__ __forceinline__ int bc_off(int row, int col4) { return row * kLT + ((col4 ^ ((row & 7) << 1)) << 2); }

template <int CTRL> __device__ __forceinline__ float lanes_dpp(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}

template <typename T, bool kHasZ, bool kVec, bool kFull, bool kCkpt = false>
__global__ __launch_bounds__(kWave, 4) void ssm_scan_fwd_lanes_kernel(const ssm_args_t p) {
    static_assert(!kFull || kVec, "kFull implies kVec");
    constexpr int kN = 16;
    // one LDS block [dt * u (then y in place) | dt | B | C]: the sequential loop addresses it with byte offsets formed by ONE
    // v_xor per operand pair (the 16-byte slot index enters lt_off / bc_off by XOR and the row bases have no bits below 256)
    __shared__ __attribute__((aligned(16))) float smem[2 * kLC * kLT + 2 * kN * kLT];
    float *const tileU = smem, *const tileD = smem + kLC * kLT, *const tileB = smem + 2 * kLC * kLT, *const tileC = tileB + kN * kLT;

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
    unsigned short *oz_planes = (kHasZ && p.out_z_lo_offset) ? reinterpret_cast<unsigned short *>(p.out_z_ptr) + (int64_t)b * p.out_z_batch_stride + (int64_t)d0 * p.out_z_d_stride : nullptr;
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

    const int n_tiles = (L + kLT - 1) / kLT;
    // load layout: lane -> (row = lane / 16, 4 columns at (lane % 16) * 4); B / C: piece i holds rows 4 i + lane / 16
    const int lrow = lane >> 4, lc4 = lane & 15, lcol = lc4 * 4;
    const int ldrow = kFull ? lrow : min(lrow, nd - 1);
    const float brow = bias_p ? bias_p[d0 + ldrow] : 0.f;
    const float Drow = p.D_ptr ? reinterpret_cast<const float *>(p.D_ptr)[d0 + ldrow] : 0.f;
    float sum_dt = 0.f;   // load layout: sum of dt of row lrow so far (all 16 lanes of the DPP row hold it)

    Raw4<T> ru, rd, rz, rb[4], rc[4];
    auto col_of = [&](int t0) { return min(t0 + lcol, L - 4); };
    auto issue_loads = [&](int t0) {
        const int col = col_of(t0);
        ru = ld4<T>(at(u_base, (unsigned)(ldrow * u_ds + col)));
        rd = ld4<T>(at(dl_base, (unsigned)(ldrow * dl_ds + col)));
    };
    // only the two HBM streams are requested a tile ahead in registers; B / C (L2-resident: shared by all waves of a batch
    // element) are requested where they are staged -- the other waves of the SIMD cover that latency
    auto issue_bc = [&](int t0) {
        const int col = col_of(t0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rb[i] = ld4<T>(at(Bp, (unsigned)((i * 4 + lrow) * Bns + col)));
            rc[i] = ld4<T>(at(Cp, (unsigned)((i * 4 + lrow) * Cns + col)));
        }
    };

    if constexpr (kVec) issue_loads(0);

#pragma unroll 1
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int t0 = tile * kLT;
        f32x4 uk = {{0.f, 0.f, 0.f, 0.f}};
        // ---- stage the tile: dt = softplus(delta + bias) (0 beyond L: a = 1, b = 0, the state is untouched), dt * u, B, C ----
        if constexpr (kVec) {
            const bool col_ok = t0 + lcol < L;
            issue_bc(t0);
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
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4 *>(&tileB[bc_off(i * 4 + lrow, lc4)]) = widen(rb[i]);
                *reinterpret_cast<f32x4 *>(&tileC[bc_off(i * 4 + lrow, lc4)]) = widen(rc[i]);
            }
            if (p.x_ptr) {   // row sum over the 16 lanes of the DPP row
                s += lanes_dpp<0x128>(s);   // row_ror:8
                s += lanes_dpp<0x124>(s);   // row_ror:4
                s += lanes_dpp<0x4E>(s);    // quad_perm [2,3,0,1]
                s += lanes_dpp<0xB1>(s);    // quad_perm [1,0,3,2]
                sum_dt += s;
            }
            if (tile + 1 < n_tiles) issue_loads(t0 + kLT);   // flies under the compute below
            if constexpr (kHasZ) rz = ld4<T>(at(z_base, (unsigned)(ldrow * z_ds + col_of(t0))));
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
            for (int idx = lane; idx < kN * kLT; idx += kWave) {
                const int r = idx / kLT, col = idx & (kLT - 1), tc = min(t0 + col, L - 1);
                tileB[bc_off(r, col >> 2) + (col & 3)] = to_f32<T>(Bp[(unsigned)(r * Bns + tc)]);
                tileC[bc_off(r, col >> 2) + (col & 3)] = to_f32<T>(Cp[(unsigned)(r * Cns + tc)]);
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

        // ---- 64 sequential steps in 4 groups of 16; per step and lane: mul, v_exp_f32, mul, fma, mul. The operands of the next 4
        //      steps are requested from LDS before the current 4 are computed (also across the join below) ----------------------
        struct Ops { f32x4 du, dt, b, c; };
        auto lds4 = [&](unsigned off) -> const f32x4 & { return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(smem) + off); };
        const unsigned ub = 4u * (unsigned)lt_off(c, 0), bb = 4u * (unsigned)(2 * kLC * kLT + bc_off(n, 0));
        auto fetch = [&](int col4) {
            const unsigned cx = (unsigned)col4 << 4;
            Ops o;
            o.du = lds4(ub ^ cx);
            o.dt = lds4((ub ^ cx) + 4u * kLC * kLT);
            o.b = lds4(bb ^ cx);
            o.c = lds4((bb ^ cx) + 4u * kN * kLT);
            return o;
        };
        // the slot lane (c, n) writes its finished y of a 16-step group to: step n of the group = column 4 (gq * 4 + n / 4) + n % 4
        const unsigned yb = 4u * (unsigned)(lt_off(c, n >> 2) + (n & 3));
        Ops nxt = fetch(0);
#pragma unroll 1
        for (int gq = 0; gq < kLT / 16; ++gq) {
            const int tg = t0 + gq * 16;
            if (tg >= L) break;
            float y[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (kCkpt && ck_base && (q & 1) == 0 && tg + q * 4 < L) ck_base[(int64_t)((tg + q * 4) >> 3) * kN * p.dim] = h;
                const Ops o = nxt;
                nxt = fetch(min(gq * 4 + q + 1, kLT / 4 - 1));
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    h = fmaf(fast_exp2(o.dt.v[s] * A2), h, o.b.v[s] * o.du.v[s]);
                    y[q * 4 + s] = h * o.c.v[s];
                }
            }
            // transposed butterfly over the channel's 16 lanes: lane (c, n) <- sum over the states of step n of the group
#pragma unroll
            for (int k = 0; k < 8; ++k) { swap_halves(y[k], y[k + 8]); y[k] += y[k + 8]; }     // state bit 3
#pragma unroll
            for (int k = 0; k < 4; ++k) { swap_rows(y[k], y[k + 4]); y[k] += y[k + 4]; }       // state bit 2
            {
                const bool hi = lane & 2;                                                       // state bit 1
#pragma unroll
                for (int k = 0; k < 2; ++k) y[k] = (hi ? y[k + 2] : y[k]) + lanes_dpp<0x4E>(hi ? y[k] : y[k + 2]);
            }
            {
                const bool hi = lane & 1;                                                       // state bit 0
                y[0] = (hi ? y[1] : y[0]) + lanes_dpp<0xB1>(hi ? y[0] : y[1]);
            }
            *reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + (yb ^ ((unsigned)gq << 6))) = y[0];
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
                    st4_out_z<T>(oz_base, oz_planes, p.out_z_lo_offset, (unsigned)(lrow * oz_ds + t0 + lcol), y4);
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
void ssm_scan_fwd_launch_lanes(const ssm_args_t &p, hipStream_t stream, int tiles, bool vec, bool full) {
    const dim3 grid(tiles), block(kWave);
    const hipEvent_t ev0 = reinterpret_cast<hipEvent_t>(p.timing_start_event), ev1 = reinterpret_cast<hipEvent_t>(p.timing_stop_event);
#define DIMSUM_LAUNCH(HASZ, VEC, FULL)                                                                                        \
    do {                                                                                                                       \
        if (p.ckpt_ptr) DIMSUM_LAUNCH_EV((ssm_scan_fwd_lanes_kernel<T, HASZ, VEC, FULL, true>), grid, block, stream, ev0, ev1, p); \
        else DIMSUM_LAUNCH_EV((ssm_scan_fwd_lanes_kernel<T, HASZ, VEC, FULL, false>), grid, block, stream, ev0, ev1, p);          \
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
    template void ssm_scan_fwd_launch_lanes<T>(const ssm_args_t &, hipStream_t, int, bool, bool);

}  // namespace dimsum
