// Micro-benchmark: LOADED latency of the decoder's hash-row fetch, against what is in flight.
// The decoder's access per nibble, hashed component and block: three 16-byte candidate rows of one random 64-byte line
// (find_ht: h0, h0^16, h0^32), later one 16-byte row written back.  All blocks of a CU hit their nibble boundary in the
// same cycles, so the requests of a CU arrive together.
//
// One workgroup of WPC waves per CU; every lane owns a private TABLE_BYTES region ("hash table").  Per iteration:
//   t0; issue the loads of pattern P; `lead` dependent VALU instructions (work that overlaps the fetch);
//   s_waitcnt vmcnt(0); t1          -> latency beyond the lead = t1 - t0 (s_memtime, averaged)
//   one 16-byte store to the line just read (the row write-back), `gap` dependent VALU instructions (the nibble's
//   bit steps), workgroup barrier (the blocks of a CU stay in lockstep).
// Patterns (lines requested per lane and nibble):
//   0: one random line                                        (decoder that asks after the bit is known)
//   1: two random lines, one per lane PAIR member             (two hypotheses, unrelated addresses: byte boundary)
//   2: two ADJACENT lines of a 128-byte pair, one per neighbouring lane    (two hypotheses with address bits 8<->6 swapped)
//   3: four lines of one aligned 256-byte chunk, two per neighbouring lane (four hypotheses, bits 8,9 <-> 6,7)
//   4: four random lines, two per lane                         (four hypotheses, unrelated addresses)
// In patterns 1-4 a lane pair (2k, 2k+1) stands for one block and the pair shares one table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k_rowlat(unsigned char *base, unsigned long long table_bytes, int pattern, int iters,
                                                int lead, int gap, int do_store, unsigned *out)
{
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned long long gl = ((unsigned long long)blockIdx.x * nw + wave) * 64 + (pattern ? (lane & ~1) : lane);
    unsigned char *tab = base + gl * table_bytes;
    const unsigned mask = (unsigned)(table_bytes - 1);
    unsigned rng = (unsigned)(gl * 2654435761u + 12345u + (pattern == 1 || pattern == 4 ? lane : 0));
    unsigned acc = lane;
    unsigned long long lat = 0;
    for (int k = 0; k < iters; k++) {
        rng = rng * 1664525u + 1013904223u;
        unsigned off = (rng >> 4) & mask & ~63u;
        unsigned off2 = 0;
        if (pattern == 1) { /* own random line per lane */ }
        if (pattern == 2) off = (off & ~127u) | ((lane & 1) << 6);
        if (pattern == 3) { off = (off & ~255u) | ((lane & 1) << 6); off2 = off | 128u; }
        if (pattern == 4) { off2 = ((rng * 0x9E3779B1u) >> 4) & mask & ~63u; }
        const unsigned long long t0 = __builtin_readcyclecounter();
        u32x4 A = *reinterpret_cast<const u32x4 *>(tab + off);
        u32x4 B = *reinterpret_cast<const u32x4 *>(tab + (off ^ 16u));
        u32x4 C = *reinterpret_cast<const u32x4 *>(tab + (off ^ 32u));
        u32x4 D = {0, 0, 0, 0}, E = D, F = D;
        if (pattern >= 3) {
            D = *reinterpret_cast<const u32x4 *>(tab + off2);
            E = *reinterpret_cast<const u32x4 *>(tab + (off2 ^ 16u));
            F = *reinterpret_cast<const u32x4 *>(tab + (off2 ^ 32u));
        }
        for (int w = 0; w < lead; w++) acc = acc * 1664525u + 1013904223u;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F), "+v"(acc));
        const unsigned long long t1 = __builtin_readcyclecounter();
        lat += t1 - t0;
        acc += A.x + B.y + C.z + D.x + E.y + F.z;
        if (do_store && (pattern == 0 || (lane & 1) == 0)) *reinterpret_cast<u32x4 *>(tab + off) = u32x4{acc, A.y, A.z, A.w};
        for (int w = 0; w < gap; w++) acc = acc * 1664525u + 1013904223u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = (unsigned)(lat / iters); out[(blockIdx.x * nw + wave) * 2 + 1] = acc; }
}

int main(int argc, char **argv)
{
    const unsigned long long table = argc > 1 ? strtoull(argv[1], 0, 0) : (1ull << 20);
    const int ncu = 256;
    unsigned char *buf; unsigned *d;
    const unsigned long long total = table * 64ull * 8 * ncu;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc of %llu MiB failed\n", total >> 20); return 1; }
    hipMemset(buf, 1, total);
    hipMalloc(&d, ncu * 8 * 2 * 4);
    hipFuncSetAttribute((const void *)k_rowlat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("table %llu KiB per lane(-pair), buffer %llu MiB; latency = s_memtime ticks (shader cycles on gfx950), wave 0 of CU 0\n", table >> 10, total >> 20);
    const int iters = 4000;
    for (int st = 0; st <= 1; st++)
    for (int wpc = 1; wpc <= 8; wpc *= 2)
    for (int pattern = 0; pattern <= 4; pattern++)
    for (int gap = 200; gap <= 800; gap += 600) {
        hipLaunchKernelGGL(k_rowlat, dim3(ncu), dim3(64 * wpc), 100 * 1024, 0, buf, table, pattern, 200, 0, gap, st, d);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rowlat, dim3(ncu), dim3(64 * wpc), 100 * 1024, 0, buf, table, pattern, iters, 0, gap, st, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned h[2]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        const double per_it = ms * 1e6 / iters * 2.4;
        const int lines = pattern == 0 ? 64 : (pattern <= 2 ? 64 : 128);
        printf("store %d waves/CU %d pattern %d gap %3d valu: latency %5u cyc, iteration %6.0f cyc, %6.2f G lines/s chip-wide\n",
               st, wpc, pattern, gap, h[0], per_it, (double)lines * wpc * ncu * iters / (ms * 1e-3) / 1e9);
    }
    return 0;
}
