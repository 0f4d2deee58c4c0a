// tools/ubench/stream_pattern.hip -- which tile shape does HBM like? (GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/sp
// tools/ubench/stream_pattern.hip && /tmp/sp). The scan kernels stream (rows x 128 B) tiles of d-major tensors: rows of one
// tile are B*L*4 bytes apart. This probe moves the same bytes (3 tensors read, 2 written, like scan fwd: u, delta, z -> out,
// out_z) with NO compute, one wave per row group walking the sequence tile by tile, for several (rows, segment) shapes of
// equal tile size, with the wave count per CU of the scan kernel (8) and more.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ float4 ldv(const float *p) {
    if constexpr (NT & 1) { const v4f r = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(r.x, r.y, r.z, r.w); }
    else return *reinterpret_cast<const float4 *>(p);
}
template <int NT> __device__ __forceinline__ void stv(float *p, float4 v) {
    if constexpr (NT & 2) { v4f r = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(r, reinterpret_cast<v4f *>(p)); }
    else *reinterpret_cast<float4 *>(p) = v;
}

template <int ROWS, int SEG4, int NT = 0>      // rows per wave tile, 16-byte pieces per row segment; NT: 1 = nontemporal loads, 2 = stores
__global__ __launch_bounds__(64) void stream_kernel(const float *u, const float *dl, const float *z, float *o, float *oz, int B, int D, int L) {
    constexpr int PIECES = ROWS * SEG4 / 64;
    const int lane = threadIdx.x;
    const int tiles_per_batch = D / ROWS;
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int b = wg / tiles_per_batch, d0 = (wg - b * tiles_per_batch) * ROWS;
    const size_t ds = (size_t)B * L;
    const size_t base = (size_t)b * L + (size_t)d0 * ds;
    const int rpp = 64 / SEG4;                       // rows per piece
    const int lrow = lane / SEG4, lcol = (lane % SEG4) * 4;
    float4 a[PIECES], c[PIECES], e[PIECES];
    auto issue = [&](int t0) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const size_t off = base + (size_t)(i * rpp + lrow) * ds + t0 + lcol;
            a[i] = ldv<NT>(u + off);
            c[i] = ldv<NT>(dl + off);
            e[i] = ldv<NT>(z + off);
        }
    };
    const int seg = SEG4 * 4;
    issue(0);
    for (int t0 = 0; t0 < L; t0 += seg) {
        float4 x[PIECES], y[PIECES];
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            x[i] = make_float4(a[i].x + c[i].x, a[i].y + c[i].y, a[i].z + c[i].z, a[i].w + c[i].w);
            y[i] = make_float4(x[i].x * e[i].x, x[i].y * e[i].y, x[i].z * e[i].z, x[i].w * e[i].w);
        }
        const int tn = t0 + seg < L ? t0 + seg : t0;
        issue(tn);                                   // next tile in flight while this one is stored (register double buffer)
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const size_t off = base + (size_t)(i * rpp + lrow) * ds + t0 + lcol;
            stv<NT>(o + off, x[i]);
            stv<NT>(oz + off, y[i]);
        }
    }
}

template <int ROWS, int SEG4, int NT = 0> float run(const float *u, const float *dl, const float *z, float *o, float *oz, int B, int D, int L, int iters) {
    const int grid = B * D / ROWS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<ROWS, SEG4, NT>), dim3(grid), dim3(64), 0, 0, u, dl, z, o, oz, B, D, L);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stream_kernel<ROWS, SEG4, NT>), dim3(grid), dim3(64), 0, 0, u, dl, z, o, oz, B, D, L);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, D = argc > 2 ? atoi(argv[2]) : 1024, L = argc > 3 ? atoi(argv[3]) : 256;
    const size_t n = (size_t)B * D * L;
    float *buf[5];
    for (auto &p : buf) { hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); }
    const double gb = 5.0 * n * 4 / 1e9;
    printf("shape (%d, %d, %d) d-major, %.3f GB per launch (3 reads + 2 writes)\n", B, D, L, gb);
#define RUN(R, S4) { float ms = run<R, S4>(buf[0], buf[1], buf[2], buf[3], buf[4], B, D, L, 20); \
    printf("rows %3d x %4d B segments (%d waves): %.3f ms = %.2f TB/s\n", R, S4 * 16, B * D / R, ms, gb / ms); }
#define RUNNT(R, S4, NT) { float ms = run<R, S4, NT>(buf[0], buf[1], buf[2], buf[3], buf[4], B, D, L, 20); \
    printf("rows %3d x %4d B segments, nontemporal %s: %.3f ms = %.2f TB/s\n", R, S4 * 16, NT == 1 ? "loads" : NT == 2 ? "stores" : "loads + stores", ms, gb / ms); }
    RUNNT(64, 8, 1) RUNNT(64, 8, 2) RUNNT(64, 8, 3) RUNNT(16, 8, 3)
    RUN(64, 8) RUN(32, 8) RUN(16, 8) RUN(32, 16) RUN(16, 16) RUN(16, 32) RUN(8, 32) RUN(8, 64) RUN(4, 64)
    return 0;
}
