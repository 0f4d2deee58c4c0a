// dependent-chain latency vs independent throughput of v_fma_f32 / v_exp_f32 on gfx950, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int CH> __global__ void k(float *out, float seed, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f * (i + 1);
    const float m = 0.999f, c = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = OP == 0 ? fmaf(a[i], m, c) : __builtin_amdgcn_exp2f(a[i]);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int CH> void run(const char *name, int threads) {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, out, 0.5f, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, out, 0.5f, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s chains=%d waves/SIMD=%d : %.3f ns per instr per wave, %.3f ns per instr per SIMD\n", name, CH, threads / 256, ms * 1e6 / (iters * 64.0),
           ms * 1e6 / (iters * 64.0 * (threads / 256)));
    hipFree(out);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0, 1>("fma", th); run<0, 2>("fma", th); run<0, 4>("fma", th); run<0, 8>("fma", th);
        run<1, 1>("exp", th); run<1, 2>("exp", th); run<1, 8>("exp", th);
    }
    return 0;
}
