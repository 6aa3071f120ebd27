// Micro-benchmark: the chip's ceiling for RANDOM 64-byte lines (16 bytes used of each), the access pattern of every
// kernel in this repository (a bit-history row, a CM / SSE / MIX entry: one small piece of a line that no cache holds).
// Every lane draws random line addresses over a BUF_GIB buffer, keeps `depth` independent 16-byte loads in flight and
// (mode 1) writes 16 bytes back to every line it read.  Reported: lines per second chip-wide and the bytes that moves at
// 64 bytes per line (read) or 128 (read + write-back).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// CHUNK lines (1, 2 or 4) that are neighbours -- one aligned 64 / 128 / 256-byte chunk per random draw, a 16-byte load from each line
template <int DEPTH, int CHUNK>
__global__ void __launch_bounds__(256) k_chunks(unsigned char *base, unsigned long long nlines, int iters, unsigned *out)
{
    unsigned long long rng = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned acc = 0;
    for (int k = 0; k < iters; k++) {
        u32x4 v[DEPTH][CHUNK];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned long long off = ((rng >> 20) % (nlines / CHUNK)) * 64ull * CHUNK;
#pragma unroll
            for (int c = 0; c < CHUNK; c++) v[d][c] = *reinterpret_cast<const u32x4 *>(base + off + 64 * c);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int c = 0; c < CHUNK; c++) acc += v[d][c].x + v[d][c].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// Round 4: four lines per random draw that are TILE-MATES -- the same 64-byte position in the four 256-byte
// quarters of one aligned 1-KiB tile (what a mixed-radix relabelling of the level-2 tables would make of a byte boundary's four
// outcomes) -- against four unrelated lines and against one aligned 256-byte quad.  STRIDE = 256: tile-mates; 64: the quad.
template <int DEPTH, int STRIDE>
__global__ void __launch_bounds__(256) k_tile(unsigned char *base, unsigned long long nlines, int iters, unsigned *out)
{
    unsigned long long rng = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    unsigned acc = 0;
    for (int k = 0; k < iters; k++) {
        u32x4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned long long tile = ((rng >> 20) % (nlines / 16)) * 1024ull;       // an aligned 1-KiB tile (16 lines)
            const unsigned long long sub = STRIDE == 256 ? ((rng >> 9) & 3ull) * 64ull : ((rng >> 9) & 3ull) * 256ull;
#pragma unroll
            for (int c = 0; c < 4; c++) v[d][c] = *reinterpret_cast<const u32x4 *>(base + tile + sub + (unsigned long long)STRIDE * c);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc += v[d][c].x + v[d][c].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH>
__global__ void __launch_bounds__(256) k_lines(unsigned char *base, unsigned long long nlines, int iters, int do_store, unsigned *out)
{
    unsigned long long rng = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned acc = 0;
    for (int k = 0; k < iters; k++) {
        u32x4 v[DEPTH];
        unsigned long long off[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            off[d] = ((rng >> 20) % nlines) * 64ull;
            v[d] = *reinterpret_cast<const u32x4 *>(base + off[d]);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc += v[d].x + v[d].w;
            if (do_store) *reinterpret_cast<u32x4 *>(base + off[d] + 16) = u32x4{acc, v[d].y, v[d].z, v[d].w};
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv)
{
    const unsigned long long gib = argc > 1 ? strtoull(argv[1], 0, 0) : 128;
    const unsigned long long bytes = gib << 30, nlines = bytes / 64;
    unsigned char *buf; unsigned *d;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc of %llu GiB failed\n", gib); return 1; }
    hipMemset(buf, 1, bytes);
    hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("random 64-byte lines over %llu GiB, 256 CUs\n", gib);
    for (int st = 0; st <= 1; st++)
    for (int wpc = 8; wpc <= 32; wpc *= 2)
    for (int depth = 2; depth <= 8; depth *= 2) {
        const int grid = 256 * wpc / 4, iters = 4000 / depth;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (depth == 2) hipLaunchKernelGGL(k_lines<2>, dim3(grid), dim3(256), 0, 0, buf, nlines, iters, st, d);
            else if (depth == 4) hipLaunchKernelGGL(k_lines<4>, dim3(grid), dim3(256), 0, 0, buf, nlines, iters, st, d);
            else hipLaunchKernelGGL(k_lines<8>, dim3(grid), dim3(256), 0, 0, buf, nlines, iters, st, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)grid * 256 * iters * depth;
        printf("%s waves/CU %2d loads in flight per lane %d: %6.2f G lines/s = %5.2f TB/s at %d B per line\n", st ? "read+write-back" : "read only      ",
               wpc, depth, lines / (ms * 1e-3) / 1e9, lines * (st ? 128 : 64) / (ms * 1e-3) / 1e12, st ? 128 : 64);
        fflush(stdout);
    }
    for (int chunk = 1; chunk <= 4; chunk *= 2) {
        const int wpc = 16, grid = 256 * wpc / 4, iters = 1000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (chunk == 1) hipLaunchKernelGGL((k_chunks<2, 1>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
            else if (chunk == 2) hipLaunchKernelGGL((k_chunks<2, 2>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
            else hipLaunchKernelGGL((k_chunks<2, 4>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double chunks = (double)grid * 256 * iters * 2;
        printf("read only, %d neighbouring line(s) per random draw (aligned %3d-byte chunk): %6.2f G chunks/s = %6.2f G lines/s = %5.2f TB/s\n",
               chunk, 64 * chunk, chunks / (ms * 1e-3) / 1e9, chunks * chunk / (ms * 1e-3) / 1e9, chunks * chunk * 64 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    // ---- round 4: four lines per draw -- unrelated, tile-mates (256-byte stride inside an aligned 1-KiB tile), one aligned quad
    {
        const int wpc = 16, grid = 256 * wpc / 4, iters = 1000;
        double base_rate = 0;
        for (int mode = 0; mode < 3; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((k_lines<8>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, 0, d);   // 8 unrelated lines per iteration
                else if (mode == 1) hipLaunchKernelGGL((k_tile<2, 256>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
                else hipLaunchKernelGGL((k_tile<2, 64>), dim3(grid), dim3(256), 0, 0, buf, nlines, iters, d);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double lines = (double)grid * 256 * iters * 8;                     // every mode: 8 lines per lane and iteration
            const double rate = lines / (ms * 1e-3) / 1e9;
            if (mode == 0) base_rate = rate;
            printf("read only, four lines per draw, %-58s: %6.2f G lines/s = %5.2f TB/s; a line costs %.2f of an unrelated line\n",
                   mode == 0 ? "unrelated lines" : mode == 1 ? "tile-mates (256-byte stride inside one aligned 1-KiB tile)" : "one aligned 256-byte quad",
                   rate, lines * 64 / (ms * 1e-3) / 1e12, base_rate / rate);
            fflush(stdout);
        }
    }
    return 0;
}
