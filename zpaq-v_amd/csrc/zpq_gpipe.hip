// zpq_gpipe.hip -- the ENCODER of general models (any mix of the nine component types, predictor.v:536-824) as a pipeline
// of WAVES: wave i = component i of every block of the workgroup, lane = block, the last wave = the arithmetic coder.
//
// zpq_lanes.hip maps a block to a row of lanes (lane = component) and so issues every component TYPE's code on every
// lane, one type after the other: ~190 instructions per coded bit and block for the all-nine-types model.  When
// ENCODING, every context, bit and bit-history state is a function of the input alone; only predictions flow from a
// component to the components that name it as an input -- and those are EARLIER components in every model this kernel
// takes (an input index >= the consumer's own reads last bit's value, predictor.v:536-668: such models stay with
// zpq_lanes.hip).  So, as in zpq_pipe.hip:
//   * in iteration `it` wave i works on byte it - i of every block; it leaves its eight predictions of that byte (16 bits
//     each, the stretch domain) in an LDS ring of its own, deep enough for its farthest consumer;
//   * a consumer reads its inputs' predictions of ITS byte from the rings; the coder (wave n, byte it - n) reads the
//     last component's, squashes and codes;
//   * one s_barrier per byte keeps the waves in step.
// A wave issues only its own type's instructions, for 64 blocks at once.  Component state lives in the block's HBM slot
// exactly as zpq_lanes.hip lays it out (zpq_model.cpp); every table access is a per-lane load / store in program order
// (a lane touches only its own block's tables, so the order of the reference's reads and writes is the lane's own).
// The HCOMP program must be the shipped hash chain (contexts in registers); coded bytes are identical to zpq_lanes.hip's,
// zpq_generic.hip's and the CPU oracle's.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_host.h"

namespace zpqg {

typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BPW = 64;                          // blocks per workgroup: lane = block
constexpr int L_STRETCH = 0;                     // u32[2048+128]
constexpr int L_SQUASH = (2048 + 128) * 4;       // u16[4096]
constexpr int L_NS = L_SQUASH + 4096 * 2;        // u8[1024]
constexpr int L_DT = L_NS + 1024;                // u32[1024]
constexpr int L_DT2K = L_DT + 4096;              // i16[256]
constexpr int L_LINK = L_DT2K + 512;             // uint4 link[n][D][BPW], then u32 misc

struct GCfg {
    int32_t n;
    int32_t ring;                // bytes of predictions kept, all components together (units of BPW uint4)
    uint16_t roff[16];           // component j's ring starts here ...
    uint16_t rmask[16];          // ... and keeps rmask[j] + 1 bytes: a power of two > the distance to its farthest consumer
    int32_t hashes;              // links of the HCOMP hash chain: H[i], i >= hashes, stays 0
    uint32_t types;              // bit t: some component has type t
};

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
__device__ __forceinline__ i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
__device__ __forceinline__ i32 clamp2k(i32 x) { return min(max(x, -2048), 2047); }
__device__ __forceinline__ i32 clamp512k(i32 x) { return min(max(x, -262144), 262143); }
__device__ __forceinline__ uint32_t mul_shr16(uint32_t range, uint32_t p16)   // see zpq_chain.hip
{
    return (uint32_t)__umul24(range >> 16, p16) + ((uint32_t)__umul24(range & 0xFFFFu, p16) >> 16);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// everything a stage needs
struct Stage {
    const DBatch *B;
    const DModel *M;
    u8 *lds;
    int ci, lane, n, hashes;
    const GCfg *cfg;
    bool active;
    u8 *slot;
    const u8 *src;
    u32 nin, total, iters;
    u8 *dst;
    u32 cap;
};

// the eight predictions of one byte, packed
struct P8 {
    u32 a, b, c, d;
    __device__ __forceinline__ i32 get(const int k) const
    {
        const u32 w = (k >> 1) == 0 ? a : ((k >> 1) == 1 ? b : ((k >> 1) == 2 ? c : d));
        return (i32)(int16_t)(w >> ((k & 1) * 16));
    }
    __device__ __forceinline__ void set(const int k, const i32 v)
    {
        const u32 ov = ((u32)v & 0xFFFFu) << ((k & 1) * 16);
        u32 &w = (k >> 1) == 0 ? a : ((k >> 1) == 1 ? b : ((k >> 1) == 2 ? c : d));
        w = (k & 1) ? (w | ov) : ov;
    }
};

// One component wave.  The per-bit code is the reference's predict() + update() for this component's type, restated on
// the block's tables in HBM (predictor.v:536-824; the CPU oracle's pred_predict / pred_update are the same text).
template <int TYPE>
__device__ __forceinline__ void comp_stage(const Stage &S)
{
    const DBatch &B = *S.B;
    const DModel &M = *S.M;
    u8 *const lds = S.lds;
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + L_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const u8 *s_ns = lds + L_NS;
    const u32 *s_dt = reinterpret_cast<const u32 *>(lds + L_DT);
    const int16_t *s_dt2k = reinterpret_cast<const int16_t *>(lds + L_DT2K);
    auto squash = [&](i32 d) -> i32 { return s_squash[min(max(wadd(d, 2047), 0), 4093)]; };      // predictor.v:193-202
    auto stretch = [&](i32 pr) -> i32 {                                                           // predictor.v:205-214
        const u32 q = (u32)min(max(pr, 1), 32767);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
    };
    (void)s_ns; (void)s_dt; (void)s_dt2k; (void)squash; (void)stretch;
    const int ci = S.ci;
    const GCfg &G = *S.cfg;
    const DComp &C = M.comp[ci];
    const i32 ca = C.a, cb = C.b, cc = C.c, climit = C.limit, cj = C.j, ck = C.k, crate = C.rate, cmask = C.mask;
    const u32 cm_len = C.cm_len, ht_len = C.ht_len;
    u32 *const cm = reinterpret_cast<u32 *>(S.slot + C.cm_off);
    u8 *const ht = S.slot + C.ht_off;
    u16 *const a16 = reinterpret_cast<u16 *>(S.slot + C.a16_off);
    (void)ca; (void)cb; (void)cc; (void)climit; (void)cj; (void)ck; (void)crate; (void)cmask; (void)cm_len; (void)ht_len; (void)cm; (void)ht; (void)a16;
    uint4 *const link = reinterpret_cast<uint4 *>(lds + L_LINK);
    auto ring_at = [&](const int comp, const u32 byte_index) -> uint4 * { return link + ((u32)G.roff[comp] + (byte_index & (u32)G.rmask[comp])) * BPW + S.lane; };
    auto ring_get = [&](const int comp, const u32 byte_index) -> P8 { const uint4 x = *ring_at(comp, byte_index); return P8{x.x, x.y, x.z, x.w}; };
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u32 total = S.total;
    const u32 hash_steps = (u32)ci < (u32)S.hashes ? (u32)ci + 1u : 0u;   // H[ci] = hash^(ci+1)(byte, previous byte), or never written

    u32 prev = 0, hctx = 0;
    // MATCH state: a = len, b = offset, c = predicted bit, cxt = bit position, limit = buffer position; Predictor.init leaves
    // sizebits / bufbits in a / b (quirk Q17, predictor.v:372-373)
    i32 ma = (TYPE == ZT_MATCH) ? ca : 0, mb = (TYPE == ZT_MATCH) ? cb : 0, mc = 0, mlimit = 0;
    u32 mcxt = 0;
    u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0;               // ICM / ISSE: the nibble's bit-history row
    u32 roff = 0;
    (void)ma; (void)mb; (void)mc; (void)mlimit; (void)mcxt; (void)r0; (void)r1; (void)r2; (void)r3; (void)roff;

    auto byte_at = [&](const u32 bi) -> u32 {
        const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
        const u32 v = (bi < total) ? (u32)S.src[pos] : 0u;
        return (pp && bi == 0) ? 0u : v;
    };
    u32 ch_next = S.active ? byte_at(0) : 0u;

    for (u32 it = 0; it < S.iters; it++) {
        const u32 bi = it - (u32)ci;
        if (S.active && bi < total) {
            const u32 ch = ch_next;
            ch_next = byte_at(bi + 1u);                // on its way while this byte is coded
            // inputs: the predictions of this byte by the components this one names (all earlier ones)
            P8 in0 = {0, 0, 0, 0}, in1 = {0, 0, 0, 0};
            P8 inm[8];
            if (TYPE == ZT_AVG) { in0 = ring_get(ca, bi); in1 = ring_get(cb, bi); }
            else if (TYPE == ZT_MIX2) { in0 = ring_get(cj, bi); in1 = ring_get(ck, bi); }
            else if (TYPE == ZT_ISSE || TYPE == ZT_SSE) in0 = ring_get(cb, bi);
            else if (TYPE == ZT_MIX) {
#pragma unroll
                for (int l = 0; l < 8; l++) inm[l] = l < climit ? ring_get(cb + l, bi) : P8{0, 0, 0, 0};
            }
            (void)in0; (void)in1; (void)inm;
            P8 out = {0, 0, 0, 0};
            u32 c8 = 1, hmap4 = 1;
#pragma unroll 1
            for (int kb = 0; kb < 8; kb++) {
                const i32 y = (i32)((ch >> (7 - kb)) & 1u);
                const i32 t32767 = y ? 32767 : 0;
                i32 p = 0;
                if (TYPE == ZT_CONST) p = (ca - 128) * 16;
                else if (TYPE == ZT_CM) {                        // predictor.v:549-554,681-700
                    const u32 cxt = hctx ^ hmap4;
                    const i32 idx = (i32)cxt & (i32)(cm_len - 1);
                    const u32 pn = cm[idx];
                    p = stretch((i32)(pn >> 17));
                    const i32 count = (i32)(pn & 0x3ffu);
                    const i32 err = t32767 - (i32)(pn >> 17);
                    const i32 upd = wmul(err, (i32)s_dt[count]) & -1024;
                    cm[idx] = (u32)wadd(wadd((i32)pn, upd), count < climit ? 1 : 0);
                } else if (TYPE == ZT_ICM || TYPE == ZT_ISSE) {  // predictor.v:555-563,615-631,701-709,776-791
                    if (c8 == 1 || (c8 & 0xf0u) == 16u) {        // find_ht (predictor.v:495-532)
                        const u32 cx = hctx + 16u * c8;
                        const u32 chk = (cx >> (ca + 2)) & 255u;
                        const u32 h0 = (cx * 16u) & (ht_len - 16u);
                        const u32x4 A = *reinterpret_cast<const u32x4 *>(ht + h0);
                        const u32x4 Bq = *reinterpret_cast<const u32x4 *>(ht + (h0 ^ 16u));
                        const u32x4 Cq = *reinterpret_cast<const u32x4 *>(ht + (h0 ^ 32u));
                        const bool ma_ = (A.x & 255u) == chk, mb_ = (Bq.x & 255u) == chk, mc_ = (Cq.x & 255u) == chk;
                        const u32 qa = (A.x >> 8) & 255u, qb = (Bq.x >> 8) & 255u, qc = (Cq.x >> 8) & 255u;
                        const bool va = qa <= qb && qa <= qc, vb = qb < qc;
                        const bool hit = ma_ || mb_ || mc_;
                        const bool ua = ma_ || (!hit && va);
                        const bool ub = !ua && (mb_ || (!hit && vb));
                        roff = ua ? h0 : (ub ? (h0 ^ 16u) : (h0 ^ 32u));
                        r0 = hit ? (ua ? A.x : (ub ? Bq.x : Cq.x)) : chk;
                        r1 = hit ? (ua ? A.y : (ub ? Bq.y : Cq.y)) : 0u;
                        r2 = hit ? (ua ? A.z : (ub ? Bq.z : Cq.z)) : 0u;
                        r3 = hit ? (ua ? A.w : (ub ? Bq.w : Cq.w)) : 0u;
                    }
                    const u32 slotn = hmap4 & 15u;
                    const u32 dsel = (slotn & 8u) ? ((slotn & 4u) ? r3 : r2) : ((slotn & 4u) ? r1 : r0);
                    const u32 sh = (slotn & 3u) * 8u;
                    const u32 st = (dsel >> sh) & 255u;
                    if (TYPE == ZT_ICM) {
                        const u32 v = cm[st];
                        p = stretch((i32)(v >> 8));
                        cm[st] = (u32)wadd((i32)v, (t32767 - (i32)(v >> 8)) >> 2);
                    } else {
                        const uint2 w = *reinterpret_cast<const uint2 *>(cm + st * 2);
                        const i32 wt0 = (i32)w.x, wt1 = (i32)w.y;
                        const i32 pj = in0.get(kb);
                        p = clamp2k(wadd(wmul(wt0, pj), wmul(wt1, 64)) >> 16);
                        const i32 err = t32767 - squash(p);
                        const i32 n0 = clamp512k(wadd(wt0, wadd(wmul(err, pj), 1 << 12) >> 13));
                        const i32 n1 = clamp512k(wadd(wt1, (err + 16) >> 5));
                        *reinterpret_cast<uint2 *>(cm + st * 2) = make_uint2((u32)n0, (u32)n1);
                    }
                    const u32 nsv = s_ns[st * 4 + (u32)y];       // statetable.v:75-84
                    const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
                    r0 = (slotn < 4u) ? ins : r0;
                    r1 = (slotn >= 4u && slotn < 8u) ? ins : r1;
                    r2 = (slotn >= 8u && slotn < 12u) ? ins : r2;
                    r3 = (slotn >= 12u) ? ins : r3;
                    if ((kb & 3) == 3) *reinterpret_cast<u32x4 *>(ht + roff) = u32x4{r0, r1, r2, r3};   // the nibble's row goes back
                } else if (TYPE == ZT_MATCH) {                   // predictor.v:564-574,710-741
                    const i32 mask = (i32)(ht_len - 1);
                    const i32 idx = mlimit & mask;
                    const u32 cur = ht[idx];
                    if (ma == 0) p = 0;
                    else {
                        mc = (i32)(((u32)ht[wsub(mlimit, mb) & mask] >> (7u - mcxt)) & 1u);
                        p = stretch((s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767);
                    }
                    if (mc != y) ma = 0;
                    ht[idx] = (u8)((cur << 1) | (u32)y);
                    mcxt++;
                    if (mcxt >= 8) {
                        mcxt = 0;
                        mlimit = wadd(mlimit, 1) & mask;
                        const i32 cmi = (i32)hctx & (i32)(cm_len - 1);
                        if (ma == 0) {
                            mb = wsub(mlimit, (i32)cm[cmi]);
                            if ((mb & mask) != 0) {
                                while (ma < 255) {
                                    const i32 i1 = wsub(wsub(mlimit, ma), 1) & mask;
                                    const i32 i2 = wsub(wsub(wsub(mlimit, ma), mb), 1) & mask;
                                    if (ht[i1] != ht[i2]) break;
                                    ma++;
                                }
                            }
                        } else if (ma < 255) ma++;
                        cm[cmi] = (u32)mlimit;
                    }
                } else if (TYPE == ZT_AVG) {                     // predictor.v:575-585
                    p = wadd(wmul(in0.get(kb), cc), wmul(in1.get(kb), 256 - cc)) >> 8;
                } else if (TYPE == ZT_MIX2) {                    // predictor.v:586-599,744-762
                    const u32 cxt = (hctx + (c8 & (u32)cmask)) & (u32)(cc - 1);
                    const i32 w = (i32)a16[cxt];
                    const i32 pj = in0.get(kb), pk = in1.get(kb);
                    p = clamp2k(wadd(wmul(w, pj), wmul(65536 - w, pk)) >> 16);
                    const i32 err = wmul(t32767 - squash(p), crate) >> 5;
                    i32 wn = wadd(w, wadd(wmul(err, wsub(pj, pk)), 1 << 12) >> 13);
                    wn = min(max(wn, 0), 65535);
                    a16[cxt] = (u16)wn;
                } else if (TYPE == ZT_MIX) {                     // predictor.v:600-614,763-775
                    const u32 cxt = (u32)(wadd((i32)hctx, (i32)c8 & cmask) & (cc - 1));
                    u32 *const wrow = cm + (size_t)wmul((i32)cxt, climit);
                    i32 wv[8], pin[8];
#pragma unroll
                    for (int l = 0; l < 8; l++) wv[l] = l < climit ? (i32)wrow[l] : 0;
                    i32 sum = 0;
#pragma unroll
                    for (int l = 0; l < 8; l++) { pin[l] = inm[l].get(kb); sum = wadd(sum, wmul(wv[l] >> 8, pin[l])); }
                    p = clamp2k(sum >> 8);
                    const i32 err = wmul(t32767 - squash(p), crate) >> 4;
#pragma unroll
                    for (int l = 0; l < 8; l++)
                        if (l < climit) wrow[l] = (u32)clamp512k(wadd(wv[l], wadd(wmul(err, pin[l]), 1 << 12) >> 13));
                } else if (TYPE == ZT_SSE) {                     // predictor.v:632-660,792-805
                    const u32 cxt = (hctx + c8) * 32u;
                    i32 pq = wadd(in0.get(kb), 992);
                    pq = min(max(pq, 0), 1983);
                    const i32 wt = pq & 63;
                    pq >>= 6;
                    const i32 idx = wadd((i32)cxt, pq), idx2 = wadd(idx, 1);
                    const bool ok = idx >= 0 && idx2 < (i32)cm_len;
                    u32 e0 = 0, e1 = 0;
                    if (ok) { e0 = cm[idx]; e1 = cm[idx2]; }
                    p = ok ? stretch(wadd(wmul((i32)(e0 >> 10), 64 - wt), wmul((i32)(e1 >> 10), wt)) >> 13) : 0;
                    const i32 iu = (i32)((u32)idx + (u32)(wt >> 5)) & (i32)(cm_len - 1);
                    u32 v;
                    if (ok && iu == idx) v = e0;
                    else if (ok && iu == idx2) v = e1;
                    else v = cm[iu];
                    const i32 err = t32767 - (i32)(v >> 17);
                    const i32 count = (i32)v & 1023;
                    if (count < climit) v = (u32)wadd(wadd((i32)v, wadd(wmul(err, climit - count), 1 << 12) >> 13), 1);
                    cm[iu] = v;
                }
                out.set(kb, p);
                // bit context (predictor.v:807-823)
                c8 = (c8 << 1) | (u32)y;
                if (c8 >= 256u) { }
                else if (c8 >= 16u && c8 < 32u) hmap4 = ((hmap4 & 0xfu) << 5) | ((u32)y << 4) | 1u;
                else hmap4 = (hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u + (u32)y) & 0xfu);
            }
            *ring_at(ci, bi) = make_uint4(out.a, out.b, out.c, out.d);
            {   // ZPAQL.run(byte) of the shipped hash chain: H[ci] for the next byte
                u32 a = ch;
                for (u32 k = 0; k < hash_steps; k++) a = (a + prev + 512u) * 773u;
                hctx = hash_steps ? a : 0u;
                prev = ch;
            }
        }
        lds_barrier();
    }
}

// the coder wave (encoder.v:48-139): byte it - n, predictions from the last component's ring; returns the bytes written
__device__ __forceinline__ u32 coder_stage(const Stage &S)
{
    const DBatch &B = *S.B;
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const int n = S.n;
    const GCfg &G = *S.cfg;
    const uint4 *const link = reinterpret_cast<const uint4 *>(lds + L_LINK);
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    u32 low = 1, high = 0xFFFFFFFFu, opos = 0;
    u8 *const dst = S.dst;
    const u32 cap = S.cap;
    auto put = [&](const u32 b) { if (opos < cap) dst[opos] = (u8)b; opos++; };
    auto shift_out = [&]() {
        while ((high ^ low) < 0x1000000u) {
            put(high >> 24);
            low <<= 8; high = (high << 8) | 255u; low = low ? low : 1u;
        }
    };
    auto byte_at = [&](const u32 bi) -> u32 {
        const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
        const u32 v = (bi < S.total) ? (u32)S.src[pos] : 0u;
        return (pp && bi == 0) ? 0u : v;
    };
    u32 ch_next = S.active ? byte_at(0) : 0u;
    for (u32 it = 0; it < S.iters; it++) {
        const u32 bi = it - (u32)n;
        if (S.active && bi < S.total) {
            const u32 ch = ch_next;
            ch_next = byte_at(bi + 1u);
            const uint4 x = link[((u32)G.roff[n - 1] + (bi & (u32)G.rmask[n - 1])) * BPW + S.lane];
            const P8 pr = {x.x, x.y, x.z, x.w};
            low += 1;                                            // EOF flag: encode(0, 0) (encoder.v:108)
            shift_out();
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const bool y = ((ch >> (7 - kb)) & 1u) != 0;
                const u32 sq = s_squash[min(max(pr.get(kb) + 2047, 0), 4093)];
                const u32 p16 = sq * 2u + 1u;                    // encoder.v:60
                const u32 mid = low + mul_shr16(high - low, p16);
                high = y ? mid : high;
                low = y ? low : mid + 1;
                shift_out();
            }
        }
        lds_barrier();
    }
    if (S.active) {
        high = low;                                              // compress(-1) + flush (encoder.v:101-105,130-139)
        shift_out();
        for (int sft = 24; sft >= 0; sft -= 8) put(high >> sft);
    }
    return opos;
}

__global__ void __launch_bounds__(1024) k_gpipe(const DBatch B, const GCfg cfg)
{
    extern __shared__ __align__(16) u8 lds[];
    const DModel &M = *B.model;
    const int tid = threadIdx.x, nthr = blockDim.x;
    {
        u32 *st = reinterpret_cast<u32 *>(lds + L_STRETCH);
        for (int i = tid; i < 2048 + 128; i += nthr) st[i] = B.stretch_c[i];
        u16 *sq = reinterpret_cast<u16 *>(lds + L_SQUASH);
        for (int i = tid; i < 4096; i += nthr) sq[i] = (u16)B.squash[i];
        for (int i = tid; i < 1024; i += nthr) lds[L_NS + i] = B.ns[i];
        u32 *dt = reinterpret_cast<u32 *>(lds + L_DT);
        for (int i = tid; i < 1024; i += nthr) dt[i] = B.dt[i];
        int16_t *d2 = reinterpret_cast<int16_t *>(lds + L_DT2K);
        for (int i = tid; i < 256; i += nthr) d2[i] = B.dt2k[i];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int n = cfg.n;
    const int wg_slot0 = blockIdx.x * BPW;
    const int nslots = B.nslots;
    const int slot_id = wg_slot0 + lane;
    const bool lane_on = slot_id < nslots;
    u8 *const slot = B.slots + (u64)(lane_on ? slot_id : wg_slot0) * M.slot_bytes;
    u32 *const misc = reinterpret_cast<u32 *>(lds + L_LINK + (size_t)cfg.ring * BPW * 16);
    const int wg_slots = min(BPW, nslots - wg_slot0);

    for (int base = wg_slot0; base < B.nblocks; base += nslots) {       // rounds: slot s codes blocks s, s + nslots, ...
        const int blk = base + lane;
        const bool active = lane_on && blk < B.nblocks;
        const int nact = min(wg_slots, B.nblocks - base);
        {   // ---- Predictor.init + ZPAQL.clear for the round's blocks (predictor.v:325-470): zero, then the non-zero fills
            const u64 n16 = M.zero_bytes / 16;
            const uint4 zero = make_uint4(0, 0, 0, 0);
            for (int b = 0; b < nact; b++) {
                uint4 *z4 = reinterpret_cast<uint4 *>(B.slots + (u64)(wg_slot0 + b) * M.slot_bytes);
                for (u64 i = tid; i < n16; i += nthr) z4[i] = zero;
            }
            __syncthreads();
            for (int b = 0; b < nact; b++) {
                u8 *sl = B.slots + (u64)(wg_slot0 + b) * M.slot_bytes;
                for (int c = 0; c < n; c++) {
                    const DComp &cc = M.comp[c];
                    if (cc.cm_len && cc.cm_fill != ZF_ZERO) {
                        u32 *t = reinterpret_cast<u32 *>(sl + cc.cm_off);
                        if (cc.cm_fill == ZF_CONST) { for (u32 i = tid; i < cc.cm_len; i += nthr) t[i] = cc.cm_fill_val; }
                        else { const u32 *img = B.img + cc.cm_fill_val; for (u32 i = tid; i < cc.cm_len; i += nthr) t[i] = img[i % cc.cm_pat_len]; }
                    }
                    if (cc.a16_len && cc.a16_fill) {
                        u16 *t = reinterpret_cast<u16 *>(sl + cc.a16_off);
                        for (u32 i = tid; i < cc.a16_len; i += nthr) t[i] = (u16)cc.a16_fill;
                    }
                }
            }
            if (tid == 0) *misc = 0u;
        }
        __threadfence();
        __syncthreads();

        Stage S;
        S.B = &B; S.M = &M; S.lds = lds; S.ci = wave; S.lane = lane; S.n = n; S.cfg = &cfg; S.active = active;
        S.slot = slot;
        S.src = active ? B.in + B.in_off[blk] : B.in;
        S.nin = active ? (u32)(B.in_off[blk + 1] - B.in_off[blk]) : 0u;
        S.dst = active ? B.out + B.out_off[blk] : B.out;
        S.cap = active ? (u32)(B.out_off[blk + 1] - B.out_off[blk]) : 0u;
        S.total = active ? S.nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u) : 0u;
        if (wave == 0) atomicMax(misc, S.total);
        __syncthreads();
        S.iters = *misc + (u32)n;
        S.hashes = cfg.hashes;
        if (wave < n) {
            switch (M.comp[wave].type) {                         // uniform per wave
            case ZT_CONST: comp_stage<ZT_CONST>(S); break;
            case ZT_CM: comp_stage<ZT_CM>(S); break;
            case ZT_ICM: comp_stage<ZT_ICM>(S); break;
            case ZT_MATCH: comp_stage<ZT_MATCH>(S); break;
            case ZT_AVG: comp_stage<ZT_AVG>(S); break;
            case ZT_MIX2: comp_stage<ZT_MIX2>(S); break;
            case ZT_MIX: comp_stage<ZT_MIX>(S); break;
            case ZT_ISSE: comp_stage<ZT_ISSE>(S); break;
            default: comp_stage<ZT_SSE>(S); break;
            }
        } else {
            const u32 opos = coder_stage(S);
            if (active) {
                B.out_len[blk] = opos;
                B.status[blk] = opos > S.cap ? ZPQ_E_OVERFLOW : ZPQ_OK;
            }
        }
        __syncthreads();
    }
}

}  // namespace zpqg

// ------------------------------------------------------------------ host side
// The pipeline takes a model when every component is one of the nine types, names only EARLIER components as inputs, the
// HCOMP program is the shipped hash chain with a hash per component, and the rings fit the LDS.
static bool gpipe_cfg(const DModel *M, zpqg::GCfg *cfg, size_t *lds_bytes)
{
    const int n = M->n;
    if (n < 1 || n > 15) return false;
    const int hashes = zpq_vm_hashchain(M);
    if (hashes <= 0) return false;
    int far[16];                                                 // distance from a component to its farthest consumer
    for (int i = 0; i < n; i++) far[i] = 0;
    far[n - 1] = 1;                                              // the coder
    for (int i = 0; i < n; i++) {
        const DComp &c = M->comp[i];
        auto need = [&](int j) { if (j < 0 || j >= i) return false; if (i - j > far[j]) far[j] = i - j; return true; };
        switch (c.type) {
        case ZT_CONST: case ZT_CM: case ZT_ICM: break;
        case ZT_MATCH: if (c.cm_len < 1 || c.ht_len < 1) return false; break;
        case ZT_AVG: if (!need(c.a) || !need(c.b)) return false; break;
        case ZT_MIX2: if (!need(c.j) || !need(c.k)) return false; break;
        case ZT_MIX:
            if (c.limit < 1 || c.limit > 8 || c.b < 0 || c.b + c.limit > i) return false;   // every input an earlier component
            for (int l = 0; l < c.limit; l++) need(c.b + l);
            break;
        case ZT_ISSE: case ZT_SSE: if (!need(c.b)) return false; break;
        default: return false;
        }
    }
    memset(cfg, 0, sizeof *cfg);
    int total = 0;
    for (int i = 0; i < n; i++) {
        int depth = 2;
        while (depth <= far[i]) depth *= 2;
        cfg->roff[i] = (uint16_t)total; cfg->rmask[i] = (uint16_t)(depth - 1);
        total += depth;
    }
    cfg->n = n; cfg->ring = total; cfg->hashes = hashes; cfg->types = 0;
    *lds_bytes = (size_t)zpqg::L_LINK + (size_t)total * zpqg::BPW * 16 + 16;
    return *lds_bytes <= 160 * 1024;
}

// ZPQ_ENC_GPIPE=0 keeps general models on zpq_lanes.hip (tests compare the two)
extern "C" int zpq_gpipe_applies(const DModel *M)
{
    const char *ev = getenv("ZPQ_ENC_GPIPE");
    if (ev && atoi(ev) == 0) return 0;
    zpqg::GCfg cfg;
    size_t lds = 0;
    return gpipe_cfg(M, &cfg, &lds) ? 1 : 0;
}

extern "C" int zpq_gpipe_blocks_per_cu(const DModel *M)
{
    zpqg::GCfg cfg;
    size_t lds = 0;
    if (!gpipe_cfg(M, &cfg, &lds)) return 0;
    const int wgs = (int)((160 * 1024) / lds);                 // workgroups per CU by LDS; waves: n + 1 of 32 per CU
    const int by_waves = 32 / (cfg.n + 1);
    const int w = wgs < by_waves ? wgs : by_waves;
    return (w < 1 ? 1 : w) * zpqg::BPW;
}

extern "C" int zpq_launch_gpipe(const DBatch *B, const DModel *hostM, int nslots, hipStream_t stream)
{
    zpqg::GCfg cfg;
    size_t lds = 0;
    if (!gpipe_cfg(hostM, &cfg, &lds)) return ZPQ_E_INTERNAL;
    const int nwg = (nslots + zpqg::BPW - 1) / zpqg::BPW;
    (void)hipFuncSetAttribute((const void *)zpqg::k_gpipe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(zpqg::k_gpipe, dim3(nwg), dim3(64 * (cfg.n + 1)), lds, stream, *B, cfg);
    return ZPQ_OK;
}
