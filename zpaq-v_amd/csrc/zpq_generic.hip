// zpq_generic.hip -- header-driven device interpreter for ANY ZPAQ model: all nine
// component types (CONST CM ICM MATCH AVG MIX2 MIX ISSE SSE), the full ZPAQL
// opcode set, and the 32-bit arithmetic coder bit loop.
//
// One ZPAQ block per 64-lane workgroup.  The bit loop is a strictly serial
// dependency chain, so lane 0 walks it; all 64 lanes cooperate on what is
// parallel: (re)initialising the block's HBM state slot with wide stores and
// the m-term MIX dot product / weight update (lane l owns weight l, partial
// sums reduced with wavefront shuffles).  This kernel is the coverage path --
// models made only of ICM/ISSE/MIX2 (every shipped level) take the LDS-resident
// chain kernel in zpq_chain.hip instead.
//
// Reference behaviour being reproduced (file:line under the reference's zpaq/):
//   Predictor.predict predictor.v:536-668   Predictor.update predictor.v:672-824
//   find_ht predictor.v:495-532             ZPAQL.run/execute zpaql.v:167-954
//   Encoder.encode/compress/flush encoder.v:48-139
//   Decoder.init/decode/decompress decoder.v:29-145
// V `int` wraps at 32 bits and `>>` on it is arithmetic; every signed op below is
// done on uint32_t and reinterpreted, so there is no UB to differ on.
#include <hip/hip_runtime.h>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_vm.h"

namespace zpqg {

typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
__device__ __forceinline__ i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }

constexpr int GEN_LDS_COMP = 32;   // component descriptors cached in LDS (the rest are read from HBM)

// Per-block scratch in LDS, carved from dynamic shared memory and sized by the model (about
// 1.5 KiB for a 9-component model), so that residency is set by registers, not by LDS.
struct Lds {
    i32 *p;              // Predictor.p   (stretch domain)            [n]
    u32 *h;              // Predictor.h   (contexts copied from the VM) [n]
    i32 *row;            // Component.c for ICM/ISSE: current hash row byte index [n]
    u32 *cxt;            // Component.cxt                              [n]
    DCompScal *cs;       // MATCH scalars                              [n]
    u8 *header;          // COMP+HCOMP bytes                           [hdr_len]
    DComp *comp;         // descriptors of the first components        [min(n, GEN_LDS_COMP)]
    i32 *mix;            // [0] err, [1] idx: broadcast slots for the cooperative MIX
};

__host__ __device__ inline size_t gen_lds_bytes(int n, int hdr_len)
{
    const int nc = n < GEN_LDS_COMP ? n : GEN_LDS_COMP;
    return (size_t)nc * sizeof(DComp) + (size_t)n * sizeof(DCompScal) + (size_t)n * 16 + 16 +
           (((size_t)hdr_len + 15) & ~(size_t)15) + 16;
}

struct Tab {
    const int16_t *squash, *stretch, *dt2k;
    const u32 *dt;
    const u8 *ns;
};

// predictor.v:193-202
__device__ __forceinline__ i32 squash(const Tab &T, i32 d)
{
    i32 idx = wadd(d, 2047);
    if (idx < 0) idx = 0;
    if (idx >= 4094) idx = 4093;
    return T.squash[idx];
}
// predictor.v:205-214
__device__ __forceinline__ i32 stretch(const Tab &T, i32 p)
{
    if (p < 1) p = 1;
    if (p >= 32768) p = 32767;
    return T.stretch[p];
}
__device__ __forceinline__ i32 clamp2k(i32 x) { return x < -2048 ? -2048 : (x > 2047 ? 2047 : x); }
__device__ __forceinline__ i32 clamp512k(i32 x) { return x < -262144 ? -262144 : (x > 262143 ? 262143 : x); }
// statetable.v:75-84 (state is always a byte here)
__device__ __forceinline__ u8 ns_next(const Tab &T, u32 s, i32 y) { return T.ns[(s & 255) * 4 + y]; }

using zpqvm::Vm;
using zpqvm::vm_run;

// ------------------------------------------------------------------ predictor
struct Pred {
    const DModel *M;
    u8 *slot;
    Lds *S;
    Tab T;
    u32 c8, hmap4;
    i32 n;
};

// predictor.v:495-532
__device__ i32 find_ht(u8 *ht, u32 ht_len, int sizebits, u32 cxt)
{
    const u32 chk = (cxt >> sizebits) & 255;
    const i32 h0 = (i32)((cxt * 16u) & (ht_len - 16u));
    if (ht[h0] == chk) return h0;
    const i32 h1 = h0 ^ 16;
    if (ht[h1] == chk) return h1;
    const i32 h2 = h0 ^ 32;
    if (ht[h2] == chk) return h2;
    const u32 q0 = ht[h0 + 1], q1 = ht[h1 + 1], q2 = ht[h2 + 1];
    i32 r;
    if (q0 <= q1 && q0 <= q2) r = h0;
    else if (q1 < q2) r = h1;
    else r = h2;
    uint4 *row = reinterpret_cast<uint4 *>(ht + r);
    *row = make_uint4(chk, 0, 0, 0);
    return r;
}

// predictor.v:536-668.  Lane 0 only, except the MIX dot product which every lane
// joins (callers keep the wave converged around predict()).
__device__ i32 predict(Pred &P, const int lane)
{
    const i32 n = P.n;
    if (n == 0) return 16384;
    Lds &S = *P.S;
    for (i32 i = 0; i < n; i++) {
        const DComp &c = (i < GEN_LDS_COMP) ? P.S->comp[i] : P.M->comp[i];
        if (c.type == ZT_MIX) {
            // cooperative: lane l accumulates terms l, l+64, ...; shuffle-reduce
            const i32 j = c.b, m = c.limit;
            i32 idx = 0;
            if (lane == 0) {
                const u32 cx = (u32)(wadd((i32)S.h[i], (i32)P.c8 & c.mask) & (c.c - 1));
                S.cxt[i] = cx;
                S.mix[1] = (i32)cx * m;
            }
            __syncthreads();
            idx = S.mix[1];
            const u32 *cm = reinterpret_cast<const u32 *>(P.slot + c.cm_off);
            i32 part = 0;
            for (i32 l = lane; l < m && (j + l) < n; l += 64)
                part = wadd(part, wmul((i32)cm[idx + l] >> 8, S.p[j + l]));
            for (int off = 32; off > 0; off >>= 1) part = wadd(part, __shfl_xor(part, off));
            if (lane == 0) S.p[i] = clamp2k(part >> 8);
            __syncthreads();
            continue;
        }
        if (lane != 0) continue;
        i32 pi = 0;
        switch (c.type) {
        case ZT_CONST: pi = (c.a - 128) * 16; break;
        case ZT_CM: {
            const u32 cx = S.h[i] ^ P.hmap4;
            S.cxt[i] = cx;
            const u32 *cm = reinterpret_cast<const u32 *>(P.slot + c.cm_off);
            pi = stretch(P.T, (i32)(cm[(i32)cx & (i32)(c.cm_len - 1)] >> 17));
            break;
        }
        case ZT_ICM: {
            u8 *ht = P.slot + c.ht_off;
            if (P.c8 == 1 || (P.c8 & 0xf0) == 16)
                S.row[i] = find_ht(ht, c.ht_len, c.a + 2, S.h[i] + 16u * P.c8);
            const u32 st = ht[S.row[i] + (i32)(P.hmap4 & 15)];
            S.cxt[i] = st;
            const u32 *cm = reinterpret_cast<const u32 *>(P.slot + c.cm_off);
            pi = stretch(P.T, (i32)(cm[st] >> 8));
            break;
        }
        case ZT_MATCH: {
            DCompScal &s = S.cs[i];
            if (s.a == 0) pi = 0;
            else {
                const u8 *ht = P.slot + c.ht_off;
                const i32 idx = wsub(s.limit, s.b) & (i32)(c.ht_len - 1);
                s.c = (ht[idx] >> (7 - (i32)s.cxt)) & 1;
                const i32 w = P.T.dt2k[s.a & 255];
                pi = stretch(P.T, (w * (s.c * -2 + 1)) & 32767);
            }
            break;
        }
        case ZT_AVG:
            pi = (c.a < n && c.b < n)
                     ? (wadd(wmul(S.p[c.a], c.c), wmul(S.p[c.b], 256 - c.c)) >> 8) : 0;
            break;
        case ZT_MIX2: {
            const u32 cx = (S.h[i] + (P.c8 & (u32)c.mask)) & (u32)(c.c - 1);
            S.cxt[i] = cx;
            const u16 *a16 = reinterpret_cast<const u16 *>(P.slot + c.a16_off);
            const i32 w = a16[cx];
            pi = (c.j < n && c.k < n)
                     ? clamp2k(wadd(wmul(w, S.p[c.j]), wmul(65536 - w, S.p[c.k])) >> 16) : 0;
            break;
        }
        case ZT_ISSE: {
            u8 *ht = P.slot + c.ht_off;
            if (P.c8 == 1 || (P.c8 & 0xf0) == 16)
                S.row[i] = find_ht(ht, c.ht_len, c.a + 2, S.h[i] + 16u * P.c8);
            const u32 st = ht[S.row[i] + (i32)(P.hmap4 & 15)];
            S.cxt[i] = st;
            const u32 *cm = reinterpret_cast<const u32 *>(P.slot + c.cm_off);
            const i32 w0 = (i32)cm[st * 2], w1 = (i32)cm[st * 2 + 1];
            if (c.b < n) pi = clamp2k(wadd(wmul(w0, S.p[c.b]), wmul(w1, 64)) >> 16);
            else pi = clamp2k(w1 >> 10);
            break;
        }
        case ZT_SSE: {
            const u32 cx = (S.h[i] + P.c8) * 32u;
            i32 pq = 992;
            if (c.b < n) pq = wadd(S.p[c.b], 992);
            if (pq < 0) pq = 0;
            if (pq > 1983) pq = 1983;
            const i32 wt = pq & 63;
            pq >>= 6;
            const i32 idx = wadd((i32)cx, pq), idx2 = wadd(idx, 1);
            if (idx >= 0 && idx2 < (i32)c.cm_len) {
                const u32 *cm = reinterpret_cast<const u32 *>(P.slot + c.cm_off);
                const i32 p1 = (i32)(cm[idx] >> 10), p2 = (i32)(cm[idx2] >> 10);
                pi = stretch(P.T, wadd(wmul(p1, 64 - wt), wmul(p2, wt)) >> 13);
            } else pi = 0;
            S.cxt[i] = (u32)idx + (u32)(wt >> 5);
            break;
        }
        default: pi = 0; break;
        }
        S.p[i] = pi;
    }
    i32 res = 0;
    if (lane == 0) res = squash(P.T, S.p[n - 1]);
    return res;
}

// predictor.v:672-805 (component training); the c8/hmap4/VM step is in advance().
__device__ void update(Pred &P, const i32 y, const int lane)
{
    const i32 n = P.n;
    Lds &S = *P.S;
    for (i32 i = 0; i < n; i++) {
        const DComp &c = (i < GEN_LDS_COMP) ? P.S->comp[i] : P.M->comp[i];
        if (c.type == ZT_MIX) {
            const i32 jj = c.b, m = c.limit;
            if (lane == 0) {
                S.mix[0] = wmul(y * 32767 - squash(P.T, S.p[i]), c.rate) >> 4;
                S.mix[1] = (i32)S.cxt[i] * m;
            }
            __syncthreads();
            const i32 err = S.mix[0], idx = S.mix[1];
            u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
            for (i32 l = lane; l < m && (jj + l) < n; l += 64)
                cm[idx + l] = (u32)clamp512k(
                    wadd((i32)cm[idx + l], wadd(wmul(err, S.p[jj + l]), 1 << 12) >> 13));
            __syncthreads();
            continue;
        }
        if (lane != 0) continue;
        switch (c.type) {
        case ZT_CM: {
            u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
            const i32 idx = (i32)S.cxt[i] & (i32)(c.cm_len - 1);
            const u32 pn = cm[idx];
            const i32 count = (i32)(pn & 0x3ff);
            const i32 err = y * 32767 - (i32)(pn >> 17);
            const i32 upd = wmul(err, (i32)P.T.dt[count]) & -1024;
            cm[idx] = (u32)wadd(wadd((i32)pn, upd), count < c.limit ? 1 : 0);
            break;
        }
        case ZT_ICM: {
            u8 *ht = P.slot + c.ht_off;
            const i32 at = S.row[i] + (i32)(P.hmap4 & 15);
            ht[at] = ns_next(P.T, ht[at], y);
            u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
            const u32 v = cm[S.cxt[i]];
            cm[S.cxt[i]] = (u32)wadd((i32)v, (y * 32767 - (i32)(v >> 8)) >> 2);
            break;
        }
        case ZT_MATCH: {
            DCompScal &s = S.cs[i];
            u8 *ht = P.slot + c.ht_off;
            u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
            const i32 mask = (i32)(c.ht_len - 1);
            if (s.c != y) s.a = 0;
            const i32 idx = s.limit & mask;
            ht[idx] = (u8)(((u32)ht[idx] << 1) | (u32)y);
            s.cxt++;
            if (s.cxt >= 8) {
                s.cxt = 0;
                s.limit = wadd(s.limit, 1) & mask;
                const i32 ci = (i32)S.h[i] & (i32)(c.cm_len - 1);
                if (s.a == 0) {
                    s.b = wsub(s.limit, (i32)cm[ci]);
                    if ((s.b & mask) != 0) {
                        while (s.a < 255) {
                            const i32 i1 = wsub(wsub(s.limit, s.a), 1) & mask;
                            const i32 i2 = wsub(wsub(wsub(s.limit, s.a), s.b), 1) & mask;
                            if (ht[i1] != ht[i2]) break;
                            s.a++;
                        }
                    }
                } else if (s.a < 255) s.a++;
                cm[ci] = (u32)s.limit;
            }
            break;
        }
        case ZT_MIX2: {
            const i32 err = wmul(y * 32767 - squash(P.T, S.p[i]), c.rate) >> 5;
            if (c.j < n && c.k < n) {
                u16 *a16 = reinterpret_cast<u16 *>(P.slot + c.a16_off);
                i32 w = a16[S.cxt[i]];
                w = wadd(w, wadd(wmul(err, wsub(S.p[c.j], S.p[c.k])), 1 << 12) >> 13);
                if (w < 0) w = 0;
                if (w > 65535) w = 65535;
                a16[S.cxt[i]] = (u16)w;
            }
            break;
        }
        case ZT_ISSE: {
            const i32 err = y * 32767 - squash(P.T, S.p[i]);
            const u32 st = S.cxt[i];
            if (c.b < n) {
                u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
                const i32 w0 = clamp512k(wadd((i32)cm[st * 2], wadd(wmul(err, S.p[c.b]), 1 << 12) >> 13));
                const i32 w1 = clamp512k(wadd((i32)cm[st * 2 + 1], (err + 16) >> 5));
                cm[st * 2] = (u32)w0;
                cm[st * 2 + 1] = (u32)w1;
            }
            u8 *ht = P.slot + c.ht_off;
            ht[S.row[i] + (i32)(P.hmap4 & 15)] = ns_next(P.T, st, y);
            break;
        }
        case ZT_SSE: {
            u32 *cm = reinterpret_cast<u32 *>(P.slot + c.cm_off);
            const i32 idx = (i32)S.cxt[i] & (i32)(c.cm_len - 1);
            u32 v = cm[idx];
            const i32 err = y * 32767 - (i32)(v >> 17);
            const i32 count = (i32)v & 1023;
            if (count < c.limit)
                v = (u32)wadd(wadd((i32)v, wadd(wmul(err, c.limit - count), 1 << 12) >> 13), 1);
            cm[idx] = v;
            break;
        }
        default: break;
        }
    }
}

// predictor.v:807-823 (lane 0).  Returns false on VM step-cap overflow.
__device__ bool advance(Pred &P, Vm &z, const i32 y)
{
    P.c8 = (P.c8 << 1) | (u32)y;
    bool ok = true;
    if (P.c8 >= 256) {
        ok = vm_run(z, P.c8 - 256);
        for (i32 i = 0; i < P.n && (u32)i < z.hlen; i++) P.S->h[i] = z.h[i];
        P.hmap4 = 1;
        P.c8 = 1;
    } else if (P.c8 >= 16 && P.c8 < 32) {
        P.hmap4 = ((P.hmap4 & 0xf) << 5) | ((u32)y << 4) | 1;
    } else {
        P.hmap4 = (P.hmap4 & 0x1f0) | (((P.hmap4 & 0xf) * 2 + (u32)y) & 0xf);
    }
    return ok;
}

// ------------------------------------------------------------------ coder
struct Enc {
    u32 low, high;
    u8 *out; u32 cap, pos;
};
// encoder.v:48-89
__device__ __forceinline__ void enc_bit(Enc &e, i32 y, u32 p16)
{
    const u32 mid = e.low + (u32)(((u64)(e.high - e.low) * p16) >> 16);
    if (y) e.high = mid; else e.low = mid + 1;
    while ((e.high ^ e.low) < 0x1000000u) {
        if (e.pos < e.cap) e.out[e.pos] = (u8)(e.high >> 24);
        e.pos++;
        e.low <<= 8;
        e.high = (e.high << 8) | 255u;
        if (e.low == 0) e.low = 1;
    }
}

struct Dec {
    u32 low, high, code;
    const u8 *in; u32 n, pos;
};
__device__ __forceinline__ void dec_shift(Dec &d)
{
    u32 c = 0;                       // Reader.get() == -1 shifts in 0 (decoder.v:39-45,109-114)
    if (d.pos < d.n) c = d.in[d.pos++];
    d.code = (d.code << 8) | c;
}
// decoder.v:73-118
__device__ __forceinline__ i32 dec_bit(Dec &d, u32 p16)
{
    const u32 mid = d.low + (u32)(((u64)(d.high - d.low) * p16) >> 16);
    i32 y;
    if (d.code <= mid) { y = 1; d.high = mid; }
    else { y = 0; d.low = mid + 1; }
    while ((d.high ^ d.low) < 0x1000000u) {
        d.low <<= 8;
        d.high = (d.high << 8) | 255u;
        if (d.low == 0) d.low = 1;
        dec_shift(d);
    }
    return y;
}

// ------------------------------------------------------------------ slot init
// Predictor.init's allocation + fill (predictor.v:325-470) and ZPAQL.clear/inith/
// initp (zpaql.v:54-95) as wide stores by the whole workgroup.
__device__ void init_slot(const DBatch &B, u8 *slot, const int lane)
{
    const DModel &M = *B.model;
    uint4 *z4 = reinterpret_cast<uint4 *>(slot);
    const u64 n16 = M.zero_bytes / 16;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (u64 i = lane; i < n16; i += 64) z4[i] = zero;
    __syncthreads();
    for (i32 ci = 0; ci < M.n; ci++) {
        const DComp &c = M.comp[ci];
        if (c.cm_len && c.cm_fill != ZF_ZERO) {
            u32 *cm = reinterpret_cast<u32 *>(slot + c.cm_off);
            if (c.cm_fill == ZF_CONST) {
                for (u32 i = lane; i < c.cm_len; i += 64) cm[i] = c.cm_fill_val;
            } else {
                const u32 *img = B.img + c.cm_fill_val;
                for (u32 i = lane; i < c.cm_len; i += 64) cm[i] = img[i % c.cm_pat_len];
            }
        }
        if (c.a16_len && c.a16_fill) {
            u16 *a16 = reinterpret_cast<u16 *>(slot + c.a16_off);
            for (u32 i = lane; i < c.a16_len; i += 64) a16[i] = (u16)c.a16_fill;
        }
        if (c.type == ZT_MATCH && lane == 0) {
            // Quirk: Predictor.init leaves sizebits/bufbits in cr.a/cr.b (predictor.v:372-373),
            // which predict/update then read as the initial match length and offset (:566-572).
            DCompScal *gs = reinterpret_cast<DCompScal *>(slot + M.scal_off);
            gs[ci].a = c.a;
            gs[ci].b = c.b;
        }
    }
    __syncthreads();
}

template <bool DEC>
__global__ void __launch_bounds__(64) k_generic(const DBatch B)
{
    extern __shared__ __align__(16) u8 gen_lds[];
    Lds S;
    {
        const DModel &M0 = *B.model;
        const int nc = M0.n < GEN_LDS_COMP ? M0.n : GEN_LDS_COMP;
        u8 *q = gen_lds;
        S.comp = reinterpret_cast<DComp *>(q); q += (size_t)nc * sizeof(DComp);
        S.cs = reinterpret_cast<DCompScal *>(q); q += (size_t)M0.n * sizeof(DCompScal);
        S.p = reinterpret_cast<i32 *>(q); q += (size_t)M0.n * 4;
        S.h = reinterpret_cast<u32 *>(q); q += (size_t)M0.n * 4;
        S.row = reinterpret_cast<i32 *>(q); q += (size_t)M0.n * 4;
        S.cxt = reinterpret_cast<u32 *>(q); q += (size_t)M0.n * 4;
        S.mix = reinterpret_cast<i32 *>(q); q += 16;
        S.header = q;
    }
    const int lane = threadIdx.x;
    const DModel &M = *B.model;
    for (int i = lane; i < M.hdr_len && i < ZPQ_MAX_HDR; i += 64) S.header[i] = M.header[i];
    {
        const int nd = (M.n < GEN_LDS_COMP ? M.n : GEN_LDS_COMP) * (int)(sizeof(DComp) / 4);
        const u32 *src = reinterpret_cast<const u32 *>(&M.comp[0]);
        u32 *dst = reinterpret_cast<u32 *>(&S.comp[0]);
        for (int i = lane; i < nd; i += 64) dst[i] = src[i];
    }
    __syncthreads();

    for (int blk = blockIdx.x; blk < B.nblocks; blk += gridDim.x) {
        u8 *slot = B.slots + (u64)blockIdx.x * M.slot_bytes;
        if (!(B.flags & ZB_KEEP_STATE)) init_slot(B, slot, lane);
        DCompScal *gs = reinterpret_cast<DCompScal *>(slot + M.scal_off);
        DVmRegs *gr = reinterpret_cast<DVmRegs *>(slot + M.regs_off);
        for (int i = lane; i < M.n; i += 64) {
            S.cs[i] = gs[i];
            S.h[i] = 0;                              // Predictor.reset() (predictor.v:827-833)
            S.p[i] = 0;
            S.row[i] = 0;
            S.cxt[i] = 0;
        }
        __syncthreads();

        Pred P;
        P.M = &M; P.slot = slot; P.S = &S; P.n = M.n;
        P.T.squash = B.squash; P.T.stretch = B.stretch; P.T.dt2k = B.dt2k; P.T.dt = B.dt; P.T.ns = B.ns;
        P.c8 = 1; P.hmap4 = 1;
        // p[] persists across segments in the reference (stale-p quirk Q13) but every
        // shipped use predicts before reading; slots keep it zeroed like a fresh Predictor.
        Vm z;
        z.a = gr->a; z.b = gr->b; z.c = gr->c; z.d = gr->d; z.f = gr->f; z.pc = gr->pc;
        z.m = slot + M.m_off; z.mlen = M.mlen;
        z.h = reinterpret_cast<u32 *>(slot + M.h_off); z.hlen = M.hlen;
        z.r = reinterpret_cast<u32 *>(slot + M.r_off);
        z.hdr = S.header; z.hdr_len = M.hdr_len; z.hbegin = M.hbegin; z.hend = M.hend;
        z.out = nullptr;

        const u8 *src = B.in + B.in_off[blk];
        const u32 nin = (u32)(B.in_off[blk + 1] - B.in_off[blk]);
        u8 *dst = B.out + B.out_off[blk];
        const u32 cap = (u32)(B.out_off[blk + 1] - B.out_off[blk]);
        i32 st = ZPQ_OK;

        if (B.flags & ZB_CTX_ONLY) {
            if (lane == 0) {
                for (u32 i = 0; i < nin; i++) {
                    if (!vm_run(z, src[i])) st = ZPQ_E_VMSTEPS;
                    for (i32 k = 0; k < M.n; k++)
                        B.ctx_out[(u64)i * M.n + k] = ((u32)k < z.hlen) ? z.h[k] : 0u;
                }
                B.status[blk] = st;
            }
        } else if (!DEC) {
            Enc e; e.low = 1; e.high = 0xFFFFFFFFu; e.out = dst; e.cap = cap; e.pos = 0;
            const u32 total = nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u);
            u32 tpos = 0;
            for (u32 i = 0; i < total; i++) {
                i32 ch = 0;
                if (B.flags & ZPQ_FLAG_PP) ch = (i == 0) ? 0 : src[i - 1];
                else ch = src[i];
                if (lane == 0) enc_bit(e, 0, 0);                // EOF flag, p=0 (encoder.v:108)
                for (int bit = 7; bit >= 0; bit--) {
                    const i32 y = (ch >> bit) & 1;
                    const i32 p = predict(P, lane);
                    if (lane == 0) {
                        if (B.trace && blk == 0 && tpos < B.ntrace) B.trace[tpos++] = p;
                        enc_bit(e, y, (u32)(p * 2 + 1));
                    }
                    update(P, y, lane);
                    if (lane == 0 && !advance(P, z, y)) st = ZPQ_E_VMSTEPS;
                }
            }
            if (lane == 0) {
                if (!(B.flags & ZPQ_FLAG_NOEOF)) {
                    enc_bit(e, 1, 0);                           // compress(-1) (encoder.v:101-105)
                    for (int s = 24; s >= 0; s -= 8) {          // flush() (encoder.v:130-139)
                        if (e.pos < e.cap) e.out[e.pos] = (u8)(e.high >> s);
                        e.pos++;
                    }
                }
                if (e.pos > e.cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
                B.out_len[blk] = e.pos;
                B.status[blk] = st;
            }
        } else {
            Dec d; d.low = 1; d.high = 0xFFFFFFFFu; d.code = 0; d.in = src; d.n = nin; d.pos = 0;
            if (lane == 0) for (int i = 0; i < 4; i++) dec_shift(d);    // decoder.v:38-46
            u32 opos = 0, first = 0xFFFFFFFFu;
            bool got_first = false;
            for (;;) {
                i32 eof = 0;
                if (lane == 0) eof = dec_bit(d, 0);                     // decoder.v:128
                eof = __shfl(eof, 0);
                if (eof) break;
                u32 c = 1;
                for (int bit = 0; bit < 8; bit++) {
                    const i32 p = predict(P, lane);
                    i32 y = 0;
                    if (lane == 0) y = dec_bit(d, (u32)(p * 2 + 1));
                    y = __shfl(y, 0);
                    update(P, y, lane);
                    if (lane == 0 && !advance(P, z, y)) st = ZPQ_E_VMSTEPS;
                    c = (c << 1) | (u32)y;
                }
                if ((B.flags & ZPQ_FLAG_PP) && !got_first) { first = c - 256; got_first = true; }
                else {
                    if (opos < cap) { if (lane == 0) dst[opos] = (u8)(c - 256); }
                    opos++;
                    if (opos > cap) break;          // slab exhausted: stop (status = overflow)
                }
            }
            if (lane == 0) {
                if (opos > cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
                B.out_len[blk] = opos;
                if (B.consumed) B.consumed[blk] = d.pos;
                if (B.final_code) B.final_code[blk] = d.code;
                if (B.first_byte) B.first_byte[blk] = first;
                B.status[blk] = st;
            }
        }
        __syncthreads();
        // persist what outlives a segment
        if (lane == 0) { gr->a = z.a; gr->b = z.b; gr->c = z.c; gr->d = z.d; gr->f = z.f; gr->pc = z.pc; }
        for (int i = lane; i < M.n; i += 64) gs[i] = S.cs[i];
        __syncthreads();
    }
}

}  // namespace zpqg

extern "C" size_t zpq_generic_lds_bytes(const DModel *M) { return zpqg::gen_lds_bytes(M->n, M->hdr_len); }

// resident one-wave workgroups per CU for this model (registers and LDS decide)
extern "C" int zpq_generic_blocks_per_cu(const DModel *M)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)zpqg::k_generic<false>, 64,
                                                     zpq_generic_lds_bytes(M)) != hipSuccess || nb < 1)
        nb = 8;
    return nb > 32 ? 32 : nb;
}

extern "C" void zpq_launch_generic(const DBatch *B, const DModel *hostM, int decode, int grid, hipStream_t stream)
{
    const size_t lds = zpq_generic_lds_bytes(hostM);
    if (decode) hipLaunchKernelGGL(zpqg::k_generic<true>, dim3(grid), dim3(64), lds, stream, *B);
    else hipLaunchKernelGGL(zpqg::k_generic<false>, dim3(grid), dim3(64), lds, stream, *B);
}
