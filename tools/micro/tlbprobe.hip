// Micro-benchmark: random 64-byte line reads over growing footprints (1 .. 128 GiB of ONE allocation).  If a box's rate falls off
// with the footprint while another box's does not, the difference between boxes that bench.py sees at level 2 (96.6 GiB of state
// slots: encode 122 vs 142 ms on different fresh boxes) is address translation (page fragments of the big allocation), not the kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_rand(const unsigned char *base, unsigned long long nlines, int iters, unsigned *out)
{
    unsigned long long rng = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 99;
    unsigned acc = 0;
    for (int k = 0; k < iters; k++) {
        u32x4 v[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            v[d] = *reinterpret_cast<const u32x4 *>(base + ((rng >> 20) % nlines) * 64ull);
        }
#pragma unroll
        for (int d = 0; d < 4; d++) acc += v[d].x + v[d].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char **argv)
{
    const unsigned long long gib = argc > 1 ? strtoull(argv[1], 0, 0) : 128;
    unsigned char *buf; unsigned *d;
    if (hipMalloc(&buf, gib << 30) != hipSuccess) { printf("alloc of %llu GiB failed\n", gib); return 1; }
    hipMemset(buf, 1, gib << 30);
    hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned long long f = 1; f <= gib; f *= 2) {
        const unsigned long long nlines = (f << 30) / 64;
        const int grid = 256 * 4, iters = 1000;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rand, dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("random 64-byte lines over the first %3llu GiB of a %llu GiB allocation: %6.2f G lines/s\n", f, gib, (double)grid * 256 * iters * 4 / (ms * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
