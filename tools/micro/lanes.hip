// Micro-benchmark: does the speed of a lone wave depend on HOW MANY of its lanes are active?
// (k_pipe's component waves ran at half speed with <= 8 active lanes, EXPERIMENTS.md 4.4.)
// One workgroup of 4 waves (one per SIMD).  Modes 0-2: lanes >= N are switched off for the whole loop (a dependent
// VALU chain, dependent LDS reads, four independent VALU chains).  Mode 3: every iteration runs 96 VALU instructions
// on all 64 lanes and then 32 on the first N lanes only -- the shape of a coder section inside a full-width bit step.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k(unsigned *out, int nact, int mode, int iters)
{
    __shared__ unsigned tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = (i * 2654435761u >> 7) & 4095u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned a = lane * 7u + 1u, b = lane + 3u, c = lane ^ 5u, d = lane + 11u;
    unsigned long long t0 = 0, t1 = 0;
    if (mode == 3) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 48; k++) a = (a ^ (a >> 3)) + b;
            if (lane < nact) {
#pragma unroll
                for (int k = 0; k < 16; k++) c = (c ^ (c >> 3)) + d;
            }
        }
        t1 = __builtin_readcyclecounter();
    } else if (lane < nact) {
        t0 = __builtin_readcyclecounter();
        if (mode == 0) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) a = (a ^ (a >> 3)) + b;
            }
        } else if (mode == 1) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) a = tab[(a + b) & 4095u];
            }
        } else {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int k = 0; k < 4; k++) { a = (a ^ (a >> 3)) + 1u; b = (b ^ (b >> 5)) + 3u; c = (c ^ (c >> 7)) + 5u; d = (d ^ (d >> 9)) + 7u; }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    if (lane == 0) { out[(threadIdx.x >> 6) * 2] = (unsigned)(t1 - t0); out[(threadIdx.x >> 6) * 2 + 1] = a + b + c + d; }
}
int main()
{
    unsigned *d; hipMalloc(&d, 64);
    const int iters = 20000;
    const char *names[4] = {"dependent VALU chain, 32 instructions per iteration, lanes >= N off throughout",
                            "dependent LDS reads, 16 per iteration, lanes >= N off throughout",
                            "4 independent VALU chains, 32 instructions per iteration, lanes >= N off throughout",
                            "96 instructions on all lanes + 32 on the first N lanes, per iteration"};
    for (int mode = 0; mode < 4; mode++) {
        printf("%s\n", names[mode]);
        for (int n : {1, 2, 4, 8, 9, 12, 16, 32, 64}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, n, mode, iters);
            hipDeviceSynchronize();
            unsigned h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("  N = %2d: %.1f cycles per iteration\n", n, (double)h[0] / iters);
        }
    }
    return 0;
}
