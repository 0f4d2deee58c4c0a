// Micro-benchmark of the scan kernel's inner block: 16 states x 4 steps = 320 VALU (64 v_exp_f32).
// Variants: ORDER 0 = state-by-state (compiler order), 1 = two states interleaved, 2 = four states interleaved,
// 3 = all exps of a state first (4 mul, 4 exp, then 4 mul + 8 fma)
#include <hip/hip_runtime.h>
#include <cstdio>
#define EXP2(x) __builtin_amdgcn_exp2f(x)
template <int ORDER> __global__ __launch_bounds__(64) void k(float *out, const float *in, int iters) {
    float h[16], A2[16], dt[4], du[4], y[4], bq[4], cq[4];
    for (int n = 0; n < 16; ++n) { h[n] = in[n] * 0.01f; A2[n] = -in[16 + n] - 0.1f * threadIdx.x; }
    for (int s = 0; s < 4; ++s) { dt[s] = in[32 + s] * 0.1f + 0.01f; du[s] = in[36 + s]; y[s] = 0.f; bq[s] = in[40 + s]; cq[s] = in[44 + s]; }
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                float hn = h[n];
#pragma unroll
                for (int s = 0; s < 4; ++s) { hn = fmaf(EXP2(dt[s] * A2[n]), hn, bq[s] * du[s]); y[s] = fmaf(hn, cq[s], y[s]); }
                h[n] = hn;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (ORDER == 1 || ORDER == 2) {
            constexpr int G = ORDER == 1 ? 2 : 4;
#pragma unroll
            for (int n0 = 0; n0 < 16; n0 += G) {
                float a[G][4], b[G][4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < G; ++g) { a[g][s] = dt[s] * A2[n0 + g]; b[g][s] = bq[s] * du[s]; }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < G; ++g) a[g][s] = EXP2(a[g][s]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < G; ++g) { h[n0 + g] = fmaf(a[g][s], h[n0 + g], b[g][s]); y[s] = fmaf(h[n0 + g], cq[s], y[s]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                float a[4], b[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = dt[s] * A2[n];
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = EXP2(a[s]);
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = bq[s] * du[s];
                float hn = h[n];
#pragma unroll
                for (int s = 0; s < 4; ++s) { hn = fmaf(a[s], hn, b[s]); y[s] = fmaf(hn, cq[s], y[s]); }
                h[n] = hn;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // keep inputs "changing" so nothing is hoisted
#pragma unroll
        for (int s = 0; s < 4; ++s) { dt[s] += 1e-6f; asm volatile("" : "+v"(bq[s]), "+v"(cq[s]), "+v"(du[s])); }
    }
    float acc = 0;
    for (int n = 0; n < 16; ++n) acc += h[n];
    out[blockIdx.x * 64 + threadIdx.x] = acc + y[0] + y[1] + y[2] + y[3];
}
template <int ORDER> void run(int waves_per_simd) {
    float *out, *in; float hin[64];
    for (int i = 0; i < 64; ++i) hin[i] = 0.3f + 0.01f * i;
    hipMalloc(&out, 256 * 16 * 64 * 4); hipMalloc(&in, 256);
    hipMemcpy(in, hin, 256, hipMemcpyHostToDevice);
    const int iters = 4000, blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(64), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(64), 0, 0, out, in, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves each doing iters blocks of 320 VALU
    printf("order %d, %d waves/SIMD: %.1f ns per 320-VALU block per SIMD  (%.2f ns/instr)\n", ORDER, waves_per_simd,
           ms * 1e6 / (iters * waves_per_simd), ms * 1e6 / (iters * waves_per_simd) / 320);
}
int main() {
    for (int w : {1, 2, 4}) { run<0>(w); run<1>(w); run<2>(w); run<3>(w); }
    return 0;
}
