// zpq_sha1.hip -- SHA-1 of many independent byte ranges, one range per lane.
//
// The reference hashes every uncompressed byte of a segment on the host, one put() per byte
// (compressor.v:284, decompressor.v:493,505; sha1.v:6-146) and stores the 20-byte digest in the
// segment trailer (compressor.v:389-395).  Once the coder runs on the GPU that scalar loop is
// the slowest stage of an archive pipeline, and the bytes are already in HBM -- so the digests
// of a whole batch are computed here next to the coder.  SHA-1 is serial inside one message
// (each 64-byte chunk chains on the previous state), so the parallel axis is the batch:
// lane = message.  The 80 rounds run out of a 16-word rolling schedule in registers; the next
// chunk's four 16-byte loads are issued before the current chunk's rounds so that their latency
// hides behind ~1300 ALU instructions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zpaq_hip.h"

namespace zpqs {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 rol(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

struct State { u32 a, b, c, d, e; };

// one 64-byte chunk, w[] big-endian words (sha1.v:38-96)
__device__ __forceinline__ void rounds(State &h, u32 (&w)[16])
{
    u32 a = h.a, b = h.b, c = h.c, d = h.d, e = h.e;
#pragma unroll
    for (int t = 0; t < 80; t++) {
        u32 wt;
        if (t < 16) wt = w[t];
        else {
            wt = rol(w[(t + 13) & 15] ^ w[(t + 8) & 15] ^ w[(t + 2) & 15] ^ w[t & 15], 1);
            w[t & 15] = wt;
        }
        u32 f, k;
        if (t < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
        else if (t < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
        else if (t < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
        else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
        const u32 tmp = rol(a, 5) + f + e + k + wt;
        e = d; d = c; c = rol(b, 30); b = a; a = tmp;
    }
    h.a += a; h.b += b; h.c += c; h.d += d; h.e += e;
}

typedef u32x4 __attribute__((aligned(4))) u32x4_a4;   // global dwordx4 loads only need dword alignment

// 64 message bytes starting `mis` bytes into the dword-aligned address p4.  The 17th dword is
// read only when mis != 0; it lies in the dword that holds the chunk's last byte, so it never
// touches a page the message does not.
struct Chunk { u32x4 q[4]; u32 x; };
__device__ __forceinline__ void load_chunk(const u8 *p4, u32 mis, Chunk &k)
{
    const u32x4_a4 *v = reinterpret_cast<const u32x4_a4 *>(p4);
    k.q[0] = v[0]; k.q[1] = v[1]; k.q[2] = v[2]; k.q[3] = v[3];
    k.x = mis ? reinterpret_cast<const u32 *>(p4)[16] : 0u;
}
__device__ __forceinline__ void chunk_words(const Chunk &k, u32 mis, u32 (&w)[16])
{
    const u32 d[17] = {k.q[0].x, k.q[0].y, k.q[0].z, k.q[0].w, k.q[1].x, k.q[1].y, k.q[1].z, k.q[1].w,
                       k.q[2].x, k.q[2].y, k.q[2].z, k.q[2].w, k.q[3].x, k.q[3].y, k.q[3].z, k.q[3].w, k.x};
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32(__builtin_amdgcn_alignbyte(d[i + 1], d[i], mis));
}

__global__ void __launch_bounds__(64) k_sha1(const u8 *in, const u64 *beg, const u64 *end, int n, u8 *out20)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    const u8 *p = in + beg[b];
    const u64 len = end[b] - beg[b];
    const u32 mis = (u32)(reinterpret_cast<uintptr_t>(p) & 3u);
    const u8 *p4 = p - mis;
    State h = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    const u64 nfull = len >> 6;
    Chunk ck = {};
    if (nfull) load_chunk(p4, mis, ck);
    for (u64 c = 0; c < nfull; c++) {
        u32 w[16];
        chunk_words(ck, mis, w);
        if (c + 1 < nfull) load_chunk(p4 + (c + 1) * 64, mis, ck);     // in flight during the rounds
        rounds(h, w);
    }
    // padding: 0x80, zeros, 64-bit big-endian bit length (sha1.v:98-140): one or two more chunks
    const u32 rem = (u32)(len & 63u);
    const u8 *tail = p + (nfull << 6);
    const int nch = rem < 56 ? 1 : 2;
    const u64 bits = len * 8u;
    for (int k = 0; k < nch; k++) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            u32 v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 pos = (u32)(k * 64 + i * 4 + j);
                const u32 byte = pos < rem ? (u32)tail[pos] : (pos == rem ? 0x80u : 0u);
                v = (v << 8) | byte;
            }
            w[i] = v;
        }
        if (k == nch - 1) { w[14] = (u32)(bits >> 32); w[15] = (u32)bits; }
        rounds(h, w);
    }
    u8 *o = out20 + (u64)b * 20;
    const u32 hv[5] = {h.a, h.b, h.c, h.d, h.e};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        o[4 * i + 0] = (u8)(hv[i] >> 24); o[4 * i + 1] = (u8)(hv[i] >> 16);
        o[4 * i + 2] = (u8)(hv[i] >> 8); o[4 * i + 3] = (u8)hv[i];
    }
}

}  // namespace zpqs

// message b = in[beg[b] .. end[b]); the contiguous form passes (in_off, in_off + 1)
extern "C" int zpq_launch_sha1(const uint8_t *in, const uint64_t *beg, const uint64_t *end, int n, uint8_t *out20, hipStream_t stream)
{
    if (n <= 0) return ZPQ_OK;
    hipLaunchKernelGGL(zpqs::k_sha1, dim3((n + 63) / 64), dim3(64), 0, stream, in, beg, end, n, out20);
    return hipGetLastError() == hipSuccess ? ZPQ_OK : ZPQ_E_INTERNAL;
}
