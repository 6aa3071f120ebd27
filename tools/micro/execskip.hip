// Micro-benchmark: does a VALU instruction of a wave whose EXEC mask covers only 16 / 32 / 48 lanes issue faster than a full
// one (i.e. does gfx950 skip the quarter-wave passes whose lanes are all inactive)?  One wave per SIMD, a dependent chain and
// four independent chains, L active lanes (lanes 0 .. L-1).  s_memtime ticks per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k(unsigned *out, int mode, int iters, int L)
{
    const int lane = threadIdx.x & 63;
    unsigned a = lane * 7u + 1u, b = lane + 3u, c = lane ^ 5u, d = lane + 11u;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < L) {
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
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[(threadIdx.x >> 6) * 2] = (unsigned)(t1 - t0);
    out[64 + threadIdx.x] = a + b + c + d;
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 8192);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++) {
        printf("%s, 32 VALU instructions per iteration, one wave per SIMD\n", mode == 0 ? "one dependent chain" : "four independent chains");
        for (int L = 16; L <= 64; L += 16) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, mode, iters, L);
            (void)hipDeviceSynchronize();
            unsigned h[8]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double mx = 0; for (int i = 0; i < 4; i++) mx = h[2 * i] > mx ? h[2 * i] : mx;
            printf("  %2d active lanes: %.2f ticks per instruction\n", L, mx / iters / 32);
        }
    }
    return 0;
}
