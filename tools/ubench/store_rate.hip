// tools/ubench/store_rate.hip: store throughput of n workgroups (one per CU: 512 threads, 128 KB of LDS), each writing 256-KB tiles the way the GEMM
// epilogue does (16-byte nt stores, 128-byte row segments of rows 1 KB... 4 KB apart). Is the epilogue's ~12 B/clk per CU a CU limit or the chip's?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_rate tools/ubench/store_rate.hip ; ./tools/ubench/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool kNt> __global__ __launch_bounds__(512) void k(float *c, int64_t ldc, int tiles_per_wg, int tiles_n) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x * tiles_per_wg + t;
        float *ct = c + (int64_t)(tile / tiles_n) * 256 * ldc + (tile % tiles_n) * 256;
        const f4 v = {(float)t, 1.f, 2.f, 3.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) {        // wave w: rows 32 w .. 32 w + 31, 1 KB per row per store instruction
            float *p = ct + (int64_t)(w * 32 + i) * ldc + lane * 4;
            if (kNt) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p)); else *reinterpret_cast<f4 *>(p) = v;
        }
    }
}
int main() {
    const int64_t M = 65536, N = 2048;                 // 537 MB like in_proj's output
    float *c; hipMalloc(&c, M * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = (M / 256) * (N / 256);
    for (int nt = 0; nt < 2; ++nt)
        for (int wgs : {16, 32, 64, 128, 256, 512, 2048}) {
            const int per = tiles / wgs;
            auto run = [&]() { if (nt) hipLaunchKernelGGL(k<true>, dim3(wgs), dim3(512), 128 * 1024 - 64, 0, c, N, per, (int)(N / 256));
                               else hipLaunchKernelGGL(k<false>, dim3(wgs), dim3(512), 128 * 1024 - 64, 0, c, N, per, (int)(N / 256)); };
            hipFuncSetAttribute((const void *)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipFuncSetAttribute((const void *)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double gb = (double)per * wgs * 256 * 1024 / 1e9;
            printf("{\"nt\": %d, \"workgroups\": %d, \"GBps\": %.0f, \"GBps_per_wg\": %.1f, \"ms\": %.4f}\n", nt, wgs, gb / (ms * 1e-3), gb / (ms * 1e-3) / (wgs > 256 ? 256 : wgs), ms);
        }
    return 0;
}
