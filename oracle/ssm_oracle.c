/*
 * oracle/ssm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's math for the denoiser hot path. It is the checker that the HIP
 * kernels are compared against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg). Nothing under
 * dimsum_amd/ may import, link or call it.
 *
 * Parity status: PINNED. Every function here is checked in tests/test_oracle_golden.py against golden vectors
 * captured from the reference's own pure-PyTorch paths (tools/gen_golden.py, run in the build container):
 *   selective scan  <- selective_scan_ref            mamba/mamba_ssm/ops/selective_scan_interface.py:104-171
 *   chunk states x  <- selective_scan_fwd_kernel     mamba/csrc/selective_scan/selective_scan_fwd_kernel.cuh:239-254
 *                      host layout                   mamba/csrc/selective_scan/selective_scan.cpp:307-313
 *   scan backward   <- autograd of selective_scan_ref; formulas cross-read with
 *                      selective_scan_bwd_kernel.cuh:171-207 (dz/out_z), :284-329 (per-state grads), :439-452 (softplus)
 *   causal conv1d   <- causal_conv1d_ref             causal-conv1d/causal_conv1d/causal_conv1d_interface.py:48-64
 *                      kernels                       causal-conv1d/csrc/causal_conv1d_fwd.cu:103-127, _bwd.cu:153-239
 *   rms/layer norm  <- rms_norm_ref/layer_norm_ref   mamba/mamba_ssm/ops/triton/layernorm.py:19-45 (+ bwd :190-285)
 *
 * All tensors are dense row-major float32; internal arithmetic is double so the oracle sits closer to the exact
 * value than either implementation (tolerances are stated in the tests). Parallel over (batch,channel) rows with
 * OpenMP when compiled with -fopenmp (used by the cpu_baseline timing; results do not depend on thread count
 * except for the fp32 rounding of the final cross-row sums, which are accumulated in double and reduced in a fixed
 * order).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SCAN_CHUNK 2048 /* selective_scan.cpp:307  n_chunks = ceil(L / 2048) */

static inline double softplus_d(double x) { return x <= 20.0 ? log1p(exp(x)) : x; } /* fwd_kernel.cuh:153-155 */
static inline double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------------------
 * selective scan forward.
 *   u, delta, z, y, out_z : (B, D, L)   A : (D, N)   Bm, Cm : (B, G, N, L)   Dv, delta_bias : (D) or NULL
 *   x : (B, D, n_chunks, 2N)  [2n] = prod of a over the prefix, [2n+1] = state h at the end of each 2048-chunk
 *   y     = C.h + D*u                       (the kernel's `out`)
 *   out_z = y * silu(z)                     (only if z != NULL)
 * ------------------------------------------------------------------------------------------------------------ */
int oracle_selective_scan_fwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                              const float *Dv, const float *z, const float *delta_bias, int delta_softplus,
                              float *y, float *out_z, float *x, int B, int D, int L, int N, int G) {
    if (N > 256 || D % G != 0) return 1;
    const int n_chunks = (L + SCAN_CHUNK - 1) / SCAN_CHUNK;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int d = 0; d < D; ++d) {
            const int g = d / (D / G);
            const float *ur = u + ((size_t)b * D + d) * L;
            const float *dr = delta + ((size_t)b * D + d) * L;
            const float *zr = z ? z + ((size_t)b * D + d) * L : NULL;
            const float *Bg = Bm + ((size_t)b * G + g) * N * L;
            const float *Cg = Cm + ((size_t)b * G + g) * N * L;
            float *yr = y + ((size_t)b * D + d) * L;
            float *ozr = out_z ? out_z + ((size_t)b * D + d) * L : NULL;
            double h[256], ap[256];
            for (int n = 0; n < N; ++n) { h[n] = 0.0; ap[n] = 1.0; }
            const double bias = delta_bias ? delta_bias[d] : 0.0;
            const double Dd = Dv ? Dv[d] : 0.0;
            for (int t = 0; t < L; ++t) {
                double dt = (double)dr[t] + bias;
                if (delta_softplus) dt = softplus_d(dt);
                const double uu = ur[t];
                double acc = Dd * uu;
                for (int n = 0; n < N; ++n) {
                    const double a = exp(dt * (double)A[d * N + n]);
                    h[n] = a * h[n] + dt * (double)Bg[(size_t)n * L + t] * uu;
                    ap[n] *= a;
                    acc += h[n] * (double)Cg[(size_t)n * L + t];
                }
                yr[t] = (float)acc;
                if (ozr) { const double zz = zr[t]; ozr[t] = (float)(acc * zz * sigmoid_d(zz)); }
                if (x && ((t + 1) % SCAN_CHUNK == 0 || t == L - 1)) {
                    float *xr = x + (((size_t)b * D + d) * n_chunks + t / SCAN_CHUNK) * 2 * N;
                    for (int n = 0; n < N; ++n) { xr[2 * n] = (float)ap[n]; xr[2 * n + 1] = (float)h[n]; }
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * selective scan backward. dout is the gradient of the FINAL output (out_z if z != NULL else y).
 * Outputs (any may be NULL to skip): du, ddelta, dz : (B,D,L); dA : (D,N); dB, dC : (B,G,N,L); dD, ddelta_bias : (D)
 * ------------------------------------------------------------------------------------------------------------ */
int oracle_selective_scan_bwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                              const float *Dv, const float *z, const float *delta_bias, int delta_softplus,
                              const float *dout, float *du, float *ddelta, float *dA, float *dB, float *dC, float *dD,
                              float *dz, float *ddelta_bias, int B, int D, int L, int N, int G) {
    if (N > 256 || D % G != 0) return 1;
    /* cross-row reductions are accumulated in double, per row, then reduced in a fixed order */
    double *dA_acc = (double *)calloc((size_t)D * N, sizeof(double));
    double *dD_acc = (double *)calloc((size_t)D, sizeof(double));
    double *db_acc = (double *)calloc((size_t)D, sizeof(double));
    double *dB_acc = (double *)calloc((size_t)B * G * N * L, sizeof(double));
    double *dC_acc = (double *)calloc((size_t)B * G * N * L, sizeof(double));
    if (!dA_acc || !dD_acc || !db_acc || !dB_acc || !dC_acc) return 2;
    const int rows_per_group = D / G;
    /* parallel over (b, g): rows of one group are walked serially so dB/dC need no atomics */
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        for (int g = 0; g < G; ++g) {
            double *hs = (double *)malloc((size_t)L * N * sizeof(double));   /* h_t[n]   */
            double *as = (double *)malloc((size_t)L * N * sizeof(double));   /* a_t[n]   */
            double *dts = (double *)malloc((size_t)L * sizeof(double));      /* softplus'd delta */
            double *dys = (double *)malloc((size_t)L * sizeof(double));      /* grad wrt y */
            const float *Bg = Bm + ((size_t)b * G + g) * N * L;
            const float *Cg = Cm + ((size_t)b * G + g) * N * L;
            double *dBg = dB_acc + ((size_t)b * G + g) * N * L;
            double *dCg = dC_acc + ((size_t)b * G + g) * N * L;
            for (int d = g * rows_per_group; d < (g + 1) * rows_per_group; ++d) {
                const size_t ro = ((size_t)b * D + d) * L;
                const double bias = delta_bias ? delta_bias[d] : 0.0;
                const double Dd = Dv ? Dv[d] : 0.0;
                /* forward recompute */
                double h[256];
                for (int n = 0; n < N; ++n) h[n] = 0.0;
                for (int t = 0; t < L; ++t) {
                    double dt = (double)delta[ro + t] + bias;
                    if (delta_softplus) dt = softplus_d(dt);
                    dts[t] = dt;
                    const double uu = u[ro + t];
                    double acc = Dd * uu;
                    for (int n = 0; n < N; ++n) {
                        const double a = exp(dt * (double)A[d * N + n]);
                        h[n] = a * h[n] + dt * (double)Bg[(size_t)n * L + t] * uu;
                        as[(size_t)t * N + n] = a;
                        hs[(size_t)t * N + n] = h[n];
                        acc += h[n] * (double)Cg[(size_t)n * L + t];
                    }
                    double dy = dout[ro + t];
                    if (z) { /* bwd_kernel.cuh:171-207 */
                        const double zz = z[ro + t], sg = sigmoid_d(zz);
                        if (dz) dz[ro + t] = (float)(dy * acc * sg * (1.0 + zz * (1.0 - sg)));
                        dy *= zz * sg;
                    }
                    dys[t] = dy;
                }
                /* reverse sweep */
                double dh[256], an[256];
                for (int n = 0; n < N; ++n) { dh[n] = 0.0; an[n] = 0.0; }
                double dDd = 0.0, dbias = 0.0;
                for (int t = L - 1; t >= 0; --t) {
                    const double dy = dys[t], dt = dts[t], uu = u[ro + t];
                    double ddt = 0.0, duu = Dd * dy;
                    dDd += dy * uu;
                    for (int n = 0; n < N; ++n) {
                        const double a_next = an[n];                   /* a_{t+1}, 0 past the end */
                        const double a_t = as[(size_t)t * N + n];
                        const double Cn = Cg[(size_t)n * L + t], Bn = Bg[(size_t)n * L + t];
                        dh[n] = a_next * dh[n] + Cn * dy;              /* dL/dh_t */
                        const double hprev = t > 0 ? hs[(size_t)(t - 1) * N + n] : 0.0;
                        const double ah = a_t * hprev;                 /* = h_t - dt*B*u   (bwd_kernel.cuh:289) */
                        const double An = A[d * N + n];
                        dCg[(size_t)n * L + t] += dy * hs[(size_t)t * N + n];
                        dBg[(size_t)n * L + t] += dh[n] * dt * uu;
                        duu += dh[n] * dt * Bn;
                        ddt += dh[n] * (Bn * uu + An * ah);
                        /* dA is per row d and rows of different b race -> per-(t,n) terms are parked in as[]
                           and summed under a critical section below */
                        as[(size_t)t * N + n] = dh[n] * dt * ah;
                        an[n] = a_t;
                    }
                    if (delta_softplus) { /* bwd_kernel.cuh:439-452 */
                        const double raw = (double)delta[ro + t] + bias;
                        if (raw <= 20.0) ddt *= sigmoid_d(raw);
                    }
                    dbias += ddt;
                    if (du) du[ro + t] = (float)duu;
                    if (ddelta) ddelta[ro + t] = (float)ddt;
                }
#pragma omp critical
                {
                    for (int n = 0; n < N; ++n) {
                        double s = 0.0;
                        for (int t = 0; t < L; ++t) s += as[(size_t)t * N + n];
                        dA_acc[(size_t)d * N + n] += s;
                    }
                    dD_acc[d] += dDd;
                    db_acc[d] += dbias;
                }
            }
            free(hs); free(as); free(dts); free(dys);
        }
    }
    if (dA) for (size_t i = 0; i < (size_t)D * N; ++i) dA[i] = (float)dA_acc[i];
    if (dD) for (int d = 0; d < D; ++d) dD[d] = (float)dD_acc[d];
    if (ddelta_bias) for (int d = 0; d < D; ++d) ddelta_bias[d] = (float)db_acc[d];
    if (dB) for (size_t i = 0; i < (size_t)B * G * N * L; ++i) dB[i] = (float)dB_acc[i];
    if (dC) for (size_t i = 0; i < (size_t)B * G * N * L; ++i) dC[i] = (float)dC_acc[i];
    free(dA_acc); free(dD_acc); free(db_acc); free(dB_acc); free(dC_acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * causal depthwise conv1d:  out[b,d,t] = act(bias[d] + sum_w W[d,w] * x[b,d,t-(width-1-w)]), zero left pad
 * x rows may be strided (x_batch_stride, x_d_stride in elements; last stride 1) like x = xz.chunk(2,1)[0].
 * ------------------------------------------------------------------------------------------------------------ */
int oracle_causal_conv1d_fwd(const float *x, long x_bs, long x_ds, const float *w, const float *bias, int silu,
                             float *out, int B, int D, int L, int width) {
    if (width < 2 || width > 4) return 1; /* causal_conv1d.cpp:248 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            const float *xr = x + (size_t)b * x_bs + (size_t)d * x_ds;
            float *o = out + ((size_t)b * D + d) * L;
            for (int t = 0; t < L; ++t) {
                double acc = bias ? bias[d] : 0.0;
                for (int k = 0; k < width; ++k) {
                    const int s = t - (width - 1 - k);
                    if (s >= 0) acc += (double)w[d * width + k] * (double)xr[s];
                }
                o[t] = (float)(silu ? acc * sigmoid_d(acc) : acc);
            }
        }
    return 0;
}

int oracle_causal_conv1d_bwd(const float *x, long x_bs, long x_ds, const float *w, const float *bias, int silu,
                             const float *dout, float *dx, float *dw, float *dbias, int B, int D, int L, int width) {
    if (width < 2 || width > 4) return 1;
    double *dw_acc = (double *)calloc((size_t)D * width, sizeof(double));
    double *db_acc = (double *)calloc((size_t)D, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int d = 0; d < D; ++d) {
        double *g = (double *)malloc((size_t)L * sizeof(double));
        for (int b = 0; b < B; ++b) {
            const float *xr = x + (size_t)b * x_bs + (size_t)d * x_ds;
            const float *go = dout + ((size_t)b * D + d) * L;
            for (int t = 0; t < L; ++t) {
                double gg = go[t];
                if (silu) { /* causal_conv1d_bwd.cu:153-164 */
                    double acc = bias ? bias[d] : 0.0;
                    for (int k = 0; k < width; ++k) {
                        const int s = t - (width - 1 - k);
                        if (s >= 0) acc += (double)w[d * width + k] * (double)xr[s];
                    }
                    const double sg = sigmoid_d(acc);
                    gg *= sg * (1.0 + acc * (1.0 - sg));
                }
                g[t] = gg;
                db_acc[d] += gg;
                for (int k = 0; k < width; ++k) {
                    const int s = t - (width - 1 - k);
                    if (s >= 0) dw_acc[d * width + k] += gg * (double)xr[s];
                }
            }
            if (dx) {
                float *dxr = dx + ((size_t)b * D + d) * L;
                for (int s = 0; s < L; ++s) {
                    double acc = 0.0;
                    for (int k = 0; k < width; ++k) {
                        const int t = s + (width - 1 - k);
                        if (t < L) acc += (double)w[d * width + k] * g[t];
                    }
                    dxr[s] = (float)acc;
                }
            }
        }
        free(g);
    }
    if (dw) for (size_t i = 0; i < (size_t)D * width; ++i) dw[i] = (float)dw_acc[i];
    if (dbias) for (int d = 0; d < D; ++d) dbias[d] = (float)db_acc[d];
    free(dw_acc); free(db_acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * fused residual-add + RMSNorm / LayerNorm, prenorm semantics (layernorm.py:19-45, upcast=True):
 *   r = x (+ residual);  y = r * rstd * w (+ b)   [RMS: rstd = 1/sqrt(mean(r^2)+eps)]
 *                        y = (r-mean) * rstd * w + b   [LN]
 * ------------------------------------------------------------------------------------------------------------ */
int oracle_norm_fwd(const float *x, const float *residual, const float *w, const float *b, double eps, int is_rms,
                    float *y, float *res_out, float *rstd_out, float *mean_out, int M, int N) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float *xr = x + (size_t)m * N;
        const float *rr = residual ? residual + (size_t)m * N : NULL;
        double s = 0.0, ss = 0.0;
        for (int i = 0; i < N; ++i) {
            /* the reference adds in fp32 then stores: (x + residual).to(x.dtype) */
            const float r = rr ? (float)(xr[i] + rr[i]) : xr[i];
            if (res_out) res_out[(size_t)m * N + i] = r;
            s += r; ss += (double)r * r;
        }
        const double mean = is_rms ? 0.0 : s / N;
        const double var = is_rms ? ss / N : ss / N - mean * mean;
        const double rstd = 1.0 / sqrt(var + eps);
        if (rstd_out) rstd_out[m] = (float)rstd;
        if (mean_out) mean_out[m] = (float)mean;
        for (int i = 0; i < N; ++i) {
            const float r = rr ? (float)(xr[i] + rr[i]) : xr[i];
            y[(size_t)m * N + i] = (float)(((double)r - mean) * rstd * (double)w[i] + (b ? (double)b[i] : 0.0));
        }
    }
    return 0;
}

/* grads wrt r (= x and residual alike) given dy and (optionally) dres_out flowing into the prenorm output */
int oracle_norm_bwd(const float *r, const float *w, double eps, int is_rms, const float *dy, const float *dres_out,
                    float *dr, float *dw, float *db, int M, int N) {
    double *dw_acc = (double *)calloc((size_t)N, sizeof(double));
    double *db_acc = (double *)calloc((size_t)N, sizeof(double));
    for (int m = 0; m < M; ++m) {
        const float *rr = r + (size_t)m * N;
        const float *g = dy + (size_t)m * N;
        double s = 0.0, ss = 0.0;
        for (int i = 0; i < N; ++i) { s += rr[i]; ss += (double)rr[i] * rr[i]; }
        const double mean = is_rms ? 0.0 : s / N;
        const double var = is_rms ? ss / N : ss / N - mean * mean;
        const double rstd = 1.0 / sqrt(var + eps);
        double c1 = 0.0, c2 = 0.0; /* layernorm.py:246-262 */
        for (int i = 0; i < N; ++i) {
            const double xhat = ((double)rr[i] - mean) * rstd, wdy = (double)w[i] * g[i];
            c1 += xhat * wdy; c2 += wdy;
            dw_acc[i] += g[i] * xhat; db_acc[i] += g[i];
        }
        c1 /= N; c2 /= N;
        for (int i = 0; i < N; ++i) {
            const double xhat = ((double)rr[i] - mean) * rstd, wdy = (double)w[i] * g[i];
            double v = is_rms ? (wdy - xhat * c1) * rstd : (wdy - (xhat * c1 + c2)) * rstd;
            if (dres_out) v += dres_out[(size_t)m * N + i];
            dr[(size_t)m * N + i] = (float)v;
        }
    }
    if (dw) for (int i = 0; i < N; ++i) dw[i] = (float)dw_acc[i];
    if (db) for (int i = 0; i < N; ++i) db[i] = (float)db_acc[i];
    free(dw_acc); free(db_acc);
    return 0;
}
