// Micro-test: what does ds_read_b64_tr_b16 deliver to each lane? LDS holds u16 value = row * 256 + col ([64 rows][128 cols], 256 B per row);
// lane l = 16 g + t supplies the address of row 4 g + (t >> 2), columns 4 (t & 3) .. + 3. Prints (row, col) of the 4 elements each lane receives.
// Build: hipcc --offload-arch=gfx950 -O2 tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned short *out) {
    __shared__ __attribute__((aligned(1024))) unsigned short lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)((i / 128) * 256 + (i % 128));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short *)lds + (unsigned)((4 * g + (t >> 2)) * 256 + (t & 3) * 8);
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d (g %d t %2d):", l, l >> 4, l & 15);
        for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
        printf("\n");
    }
    return 0;
}
