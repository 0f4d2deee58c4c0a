// issue cost of the VALU op forms the scan backward uses (gfx950): VGPR-operand FMAs, selects, DPP adds, permlane swaps,
// broadcast LDS reads. 8 independent chains per wave; 1, 2 and 4 waves per SIMD. ns per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float dpp_ror8(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xF, 0xF, true)); }
template <int OP> __global__ void k(float *out, const float *in, int iters) {
    __shared__ float lds[1024];
    float a[8], b[8], c[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + i]; b[i] = in[threadIdx.x + 8 + i] * 0.999f; c[i] = in[threadIdx.x + 16 + i] * 1e-3f; }
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    const bool hi = threadIdx.x & 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = fmaf(a[i], 0.999f, 1e-3f);                 // 1 VGPR source
                if (OP == 1) a[i] = fmaf(a[i], b[i], 1e-3f);                   // 2 VGPR sources
                if (OP == 2) a[i] = fmaf(a[i], b[i], c[i]);                    // 3 VGPR sources
                if (OP == 3) a[i] = a[i] * b[i];                               // mul 2 VGPR
                if (OP == 4) a[i] = hi ? a[i] : b[i];                          // cndmask
                if (OP == 5) a[i] = b[i] + dpp_ror8(a[i]);                     // add with DPP source
                if (OP == 6) { auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(b[i]), false, false); a[i] = __uint_as_float(q[0]); b[i] = __uint_as_float(q[1]); }
                if (OP == 7) { auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(b[i]), false, false); a[i] = __uint_as_float(q[0]); b[i] = __uint_as_float(q[1]); }
                if (OP == 8) { const float4 v = *reinterpret_cast<const float4 *>(&lds[((it + r * 8 + i) & 63) * 4]); a[i] += v.x + v.y + v.z + v.w; }   // broadcast b128 + 4 adds
                if (OP == 9) { const float4 v = *reinterpret_cast<const float4 *>(&lds[((it + r * 8 + i) & 3) * 256 + (threadIdx.x & 63) * 4]); a[i] += v.x + v.y + v.z + v.w; }   // per-lane b128 + 4 adds
                if (OP == 10) a[i] = fmaf(__builtin_amdgcn_exp2f(b[i] * c[i]), a[i], c[i]);   // mul + exp + fma (the recurrence step)
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int threads, int per) {
    float *out, *in; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 8192); (void)hipMemset(in, 0, 8192);
    const int iters = 2000;
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(threads), 0, 0, out, in, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(threads), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d : %.3f ns per op-group per SIMD (%d instr per group)\n", name, threads / 256, ms * 1e6 / (iters * 64.0 * (threads / 256)), per);
    (void)hipFree(out); (void)hipFree(in);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("fma 1 vgpr src", th, 1); run<1>("fma 2 vgpr src", th, 1); run<2>("fma 3 vgpr src", th, 1); run<3>("mul 2 vgpr", th, 1);
        run<4>("cndmask", th, 1); run<5>("add dpp row_ror8", th, 1); run<6>("permlane16_swap", th, 1); run<7>("permlane32_swap", th, 1);
        run<8>("ds_read_b128 broadcast + 4 add", th, 5); run<9>("ds_read_b128 per-lane + 4 add", th, 5); run<10>("mul+exp+fma", th, 3);
    }
    return 0;
}
