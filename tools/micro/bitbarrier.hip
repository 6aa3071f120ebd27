// Micro-benchmark for a wave-split DECODER (EXPERIMENTS.md 4.5): what does one hand-off per coded bit cost when it is
// an LDS mailbox + s_barrier instead of a spin (tools/micro/pingpong.hip: ~210 cycles per hop)?
// NW waves of one workgroup; per iteration every wave
//   reads the other waves' mailbox words of the previous iteration (one ds_read per producer),
//   runs `work` dependent VALU instructions on them,
//   looks one value up in an LDS table (a second, DEPENDENT LDS round trip, like squash()),
//   writes its own mailbox word, then  s_waitcnt lgkmcnt(0); s_barrier.
// Reports clock ticks (s_memtime, 100 MHz) and wall time per iteration; the decoder's bit step has this shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NW>
__global__ void __launch_bounds__(64 * NW) k_bitbarrier(unsigned *out, int iters, int work, int lookups)
{
    __shared__ unsigned mail[2][NW][64];
    __shared__ unsigned short table[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) table[i] = (unsigned short)(i * 2654435761u >> 20);
    for (int i = threadIdx.x; i < 2 * NW * 64; i += blockDim.x) (&mail[0][0][0])[i] = i;
    __syncthreads();
    unsigned acc = lane * 2654435761u + wave;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; k++) {
        const int cur = k & 1, prv = cur ^ 1;
        unsigned v = acc;
#pragma unroll
        for (int w = 0; w < NW; w++) v += mail[prv][w][lane];
        for (int w = 0; w < work; w++) v = v * 1664525u + 1013904223u;
        for (int l = 0; l < lookups; l++) v += table[(v >> 7) & 4095u];
        for (int w = 0; w < work; w++) v = v * 1664525u + 1013904223u;
        acc = v;
        mail[cur][wave][lane] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[(blockIdx.x * NW + wave) * 2] = (unsigned)((t1 - t0) * 100 / iters); out[(blockIdx.x * NW + wave) * 2 + 1] = acc; }
}

template <int NW>
static void run(unsigned *d, int work, int lookups)
{
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_bitbarrier<NW>, dim3(256), dim3(64 * NW), 0, 0, d, 1000, work, lookups);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bitbarrier<NW>, dim3(256), dim3(64 * NW), 0, 0, d, iters, work, lookups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("waves %2d  valu %3d  dependent lookups %d : %7.1f ns per iteration (wall) = %5.0f cycles at 2.4 GHz; s_memtime x100: %u\n",
           NW, 2 * work, lookups, ms * 1e6 / iters, ms * 1e6 / iters * 2.4, h[0]);
}

int main()
{
    unsigned *d; hipMalloc(&d, 256 * 16 * 2 * 4);
    for (int lookups = 0; lookups <= 2; lookups++)
        for (int work = 0; work <= 60; work += 20) {
            run<4>(d, work, lookups);
            if (work == 40) { run<3>(d, work, lookups); run<6>(d, work, lookups); run<8>(d, work, lookups); run<10>(d, work, lookups); }
        }
    return 0;
}
