// Micro-benchmark: how does a SIMD share its VALU issue between waves?  W waves per SIMD (workgroup of 4*W waves on
// one CU) each run the same VALU loop: a dependent chain, or four independent chains.  Cycles per iteration per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(1024) k(unsigned *out, int mode, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned a = lane * 7u + 1u, b = lane + 3u, c = lane ^ 5u, d = lane + 11u;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) a = (a ^ (a >> 3)) + b;
        }
    } else {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) { a = (a ^ (a >> 3)) + 1u; b = (b ^ (b >> 5)) + 3u; c = (c ^ (c >> 7)) + 5u; d = (d ^ (d >> 9)) + 7u; }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[(threadIdx.x >> 6) * 2] = (unsigned)(t1 - t0); out[(threadIdx.x >> 6) * 2 + 1] = a + b + c + d; }
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 4096);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++) {
        printf("%s, 32 VALU instructions per iteration\n", mode == 0 ? "one dependent chain" : "four independent chains");
        for (int w = 1; w <= 4; w++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256 * w), 0, 0, d, mode, iters);
            (void)hipDeviceSynchronize();
            unsigned h[32]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double mx = 0; for (int i = 0; i < 4 * w; i++) mx = h[2 * i] > mx ? h[2 * i] : mx;
            printf("  %d wave(s) per SIMD: %.1f cycles per iteration per wave (slowest wave) = %.2f per instruction per wave, %.2f per instruction per SIMD\n",
                   w, mx / iters, mx / iters / 32, mx / iters / 32 / w);
        }
    }
    return 0;
}
