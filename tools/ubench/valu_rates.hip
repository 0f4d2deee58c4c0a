// Micro-benchmark: issue cost (cycles per wave64 instruction on one SIMD) of the VALU ops the scan kernel uses.
// One wave per SIMD (256 threads/block, 1 block/CU) and 2 waves per SIMD variants. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP> __global__ void k(float *out, float seed, int iters, unsigned long long *cyc) {
    float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 0.999f, c = 1e-3f;
    const v2f mm = {m, m}, cc = {c, c};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) { a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c); a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c); }
            if (OP == 1) { a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3); a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7); }
            if (OP == 2) { p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc); p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
                           p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc); p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc); }
            if (OP == 3) { a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m; }
            if (OP == 4) { p0 *= mm; p1 *= mm; p2 *= mm; p3 *= mm; p0 *= mm; p1 *= mm; p2 *= mm; p3 *= mm; }
            if (OP == 5) { a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3); a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7); }
            // mix like the scan: 1 exp + 2 fma + 2 mul
            if (OP == 6) { a0 = __builtin_amdgcn_exp2f(a0 * m); a1 = fmaf(a0, a1, a2 * c); a3 = fmaf(a1, m, a3);
                           a4 = __builtin_amdgcn_exp2f(a4 * m); a5 = fmaf(a4, a5, a6 * c); a7 = fmaf(a5, m, a7); }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char *name, int nops_per_rep8, int threads) {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, 0.5f, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, 0.5f, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double ninstr = (double)iters * (REP / 8) * nops_per_rep8;
    const int waves_per_simd = threads / 256;
    printf("%-28s threads=%4d  %.2f s_memtime-ticks/instr/wave   wall: %.2f ns per instr per SIMD (x%d waves)\n", name, threads,
           (double)h / ninstr, ms * 1e6 / (ninstr * waves_per_simd), waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("v_fma_f32", 8, th); run<1>("v_exp_f32", 8, th); run<2>("v_pk_fma_f32", 8, th); run<3>("v_mul_f32", 8, th);
        run<4>("v_pk_mul_f32", 8, th); run<5>("v_rcp_f32", 8, th); run<6>("mix exp+2fma+2mul (x2)", 10, th);
    }
    return 0;
}
