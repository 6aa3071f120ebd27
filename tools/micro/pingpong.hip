// Micro-benchmark: how long does a value take from one wave to another of the same workgroup through LDS?
// Wave 0 writes seq k to mailbox A and spins on mailbox B until it reads k; wave 1 spins on A and echoes into B.
// Reports cycles per round trip (two hops), with the waves on different SIMDs (4 waves per workgroup, waves 2/3 idle).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) pingpong(unsigned *out, int iters, int work)
{
    __shared__ volatile unsigned A[64], B[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) { A[lane] = 0; B[lane] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = lane;
    if (wave == 0) {
        for (int k = 1; k <= iters; k++) {
            for (int w = 0; w < work; w++) acc = acc * 1664525u + 1013904223u;
            A[lane] = (unsigned)k + (acc & 0u);
            while (B[lane] != (unsigned)k) {}
        }
    } else if (wave == 1) {
        for (int k = 1; k <= iters; k++) {
            while (A[lane] != (unsigned)k) {}
            for (int w = 0; w < work; w++) acc = acc * 1664525u + 1013904223u;
            B[lane] = (unsigned)k + (acc & 0u);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave < 2) { out[blockIdx.x * 4 + wave * 2] = (unsigned)((t1 - t0) / iters); out[blockIdx.x * 4 + wave * 2 + 1] = acc; }
}
int main()
{
    unsigned *d; hipMalloc(&d, 4096 * 4);
    for (int work = 0; work <= 40; work += 20) {
        for (int nwg = 1; nwg <= 256; nwg *= 256) {
            hipLaunchKernelGGL(pingpong, dim3(nwg), dim3(256), 0, 0, d, 20000, work);
            hipDeviceSynchronize();
            unsigned h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("work %d instrs/side, %d workgroups: %u clock ticks per round trip (wave0), %u (wave1)\n", work, nwg, h[0], h[2]);
        }
    }
    // s_memtime ticks at 100 MHz on gfx9; readcyclecounter -> s_memtime.  Also report wall time.
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work = 0; work <= 40; work += 20) {
        hipEventRecord(e0); hipLaunchKernelGGL(pingpong, dim3(256), dim3(256), 0, 0, d, 200000, work); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("work %d: %.1f ns per round trip (wall)\n", work, ms * 1e6 / 200000);
    }
    return 0;
}
