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
// exactly as zpq_lanes.hip lays it out (zpq_model.cpp); a lane touches only its own block's tables, so the order of the
// reference's reads and writes is the lane's own program order.  An encoder knows all eight bits of a byte when the byte
// starts, so a stage sends the byte's table loads out together and runs the eight predict + update steps on registers
// (comp_stage<TYPE, BATCH = true>; BATCH = false keeps one access per bit, in bit order -- ZPQ_GPIPE_BATCH=0, tests compare).
// The HCOMP program must be the shipped hash chain (contexts in registers); coded bytes are identical to zpq_lanes.hip's,
// zpq_generic.hip's and the CPU oracle's.
//
// Measured on C4b (all nine types, 16 384 x 64 KiB blocks, MI355X): 1.40-1.46 s against k_rows<encode>'s 2.83 s.  The stage
// that sets the pace is the MIX over seven inputs; FETCH_SIZE + WRITE_SIZE = 2.15 + 1.60 TB per launch = 31 + 23 random
// 64-byte lines per input byte (profiles/r03_gpipe_*.txt): the model's tables do not fit any cache at this residency, and
// the kernel runs at ~41 G random lines/s.
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
// vector accesses that are only word-aligned (a MIX row, an SSE pair)
typedef unsigned int u32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x3a __attribute__((ext_vector_type(3), aligned(4)));
typedef unsigned int u32x2a __attribute__((ext_vector_type(2), aligned(4)));

constexpr int BPW = 64;                          // blocks per workgroup: lane = block
constexpr int L_STRETCH = 0;                     // u32[2048+128]
constexpr int L_SQUASH = (2048 + 128) * 4;       // u16[4096]
constexpr int L_NS = L_SQUASH + 4096 * 2;        // u8[1024]
constexpr int L_DT = L_NS + 1024;                // u32[1024]
constexpr int L_DT2K = L_DT + 4096;              // i16[256]
constexpr int L_BYTES = L_DT2K + 512;            // u8 bytes[16][BPW]: the input bytes, published by wave 0 for the waves behind it
constexpr int L_LINK = L_BYTES + 16 * BPW;       // uint4 link[ring][BPW], then u32 misc

struct GCfg {
    int32_t n;
    int32_t ring;                // bytes of predictions kept, all components together (units of BPW uint4)
    uint16_t roff[16];           // component j's ring starts here ...
    uint16_t rmask[16];          // ... and keeps rmask[j] + 1 bytes: a power of two > the distance to its farthest consumer
    int32_t hashes;              // links of the HCOMP hash chain: H[i], i >= hashes, stays 0
    int32_t bpw;                 // decoder: blocks (lanes in use) per workgroup: 64, or fewer so that a small batch still fills every workgroup slot
    int32_t nlevels;             // decoder: levels of the prediction chain (level 0 = no inputs, level k = 1 + the deepest input)
    uint8_t level[16];           // decoder: component i predicts at this level
    uint8_t dslot[16];           // decoder: an SSE component's row buffer in LDS
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
    u64 busy;                    // (-DZPG_PROF) cycles between barriers
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
template <int TYPE, bool BATCH>
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
    u32 mcxt = 0, mpred = 0;
    u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0;               // ICM / ISSE: the nibble's bit-history row
    u32 roff = 0;
    (void)ma; (void)mb; (void)mc; (void)mlimit; (void)mcxt; (void)mpred; (void)r0; (void)r1; (void)r2; (void)r3; (void)roff;

    // Input bytes: wave 0 reads them from HBM -- one aligned dword per four bytes, the next one requested a byte early -- and
    // leaves each in an LDS ring for the waves behind it (wave i reads byte it - i, written i iterations ago).
    u8 *const s_bytes = lds + L_BYTES;
    auto src_addr = [&](const u32 bi) -> uintptr_t { return reinterpret_cast<uintptr_t>(S.src) + (pp ? (bi ? bi - 1u : 0u) : bi); };
    // BATCH: an encoder knows all eight bits of a byte, so every table address of the byte is known when the byte starts (a
    // bit-history row's states once the row is in).  The byte's loads go out together, the eight predict + update steps then
    // run on registers -- a value that an earlier bit of the byte already rewrote is taken from that bit, not from the load --
    // and the stores follow in bit order.  One or two memory round trips per byte instead of eight to ten (measured: a
    // dependent load -> store step costs ~4250 cycles in this kernel).  CONST and AVG touch no memory; a bit-history table
    // of fewer than 8192 bytes may hold both nibbles' rows in one line and keeps the bit-serial code.
    const bool batched = BATCH && TYPE != ZT_CONST && TYPE != ZT_AVG && !((TYPE == ZT_ICM || TYPE == ZT_ISSE) && ht_len < 8192u);
    u32 w_cur = 0;
    if (ci == 0 && S.active && S.nin) w_cur = *reinterpret_cast<const u32 *>(src_addr(0) & ~(uintptr_t)3);

    for (u32 it = 0; it < S.iters; it++) {
        const u32 bi = it - (u32)ci;
#ifdef ZPG_PROF
        const u64 t_in = __builtin_readcyclecounter();
#endif
        if (S.active && bi < total) {
            u32 ch;
            if (ci == 0) {
                const uintptr_t a = src_addr(bi), a1 = src_addr(bi + 1u);
                ch = (pp && bi == 0) ? 0u : (w_cur >> (8u * (u32)(a & 3))) & 255u;
                if (bi + 1u < total && (a1 & ~(uintptr_t)3) != (a & ~(uintptr_t)3)) w_cur = *reinterpret_cast<const u32 *>(a1 & ~(uintptr_t)3);
                s_bytes[(bi & 15u) * BPW + S.lane] = (u8)ch;
            } else ch = s_bytes[(bi & 15u) * BPW + S.lane];
            // inputs: the predictions of this byte by the components this one names (all earlier ones)
            P8 in0 = {0, 0, 0, 0}, in1 = {0, 0, 0, 0};
            P8 inm[8];
            if (TYPE == ZT_AVG) { in0 = ring_get(ca, bi); in1 = ring_get(cb, bi); }
            else if (TYPE == ZT_MIX2) { in0 = ring_get(cj, bi); in1 = ring_get(ck, bi); }
            else if (TYPE == ZT_ISSE || TYPE == ZT_SSE) in0 = ring_get(cb, bi);
            else if (TYPE == ZT_MIX && !batched) {
#pragma unroll
                for (int l = 0; l < 8; l++) inm[l] = l < climit ? ring_get(cb + l, bi) : P8{0, 0, 0, 0};
            }
            (void)in0; (void)in1; (void)inm;
            P8 out = {0, 0, 0, 0};
            if (batched) {
                // the byte's bit contexts (predictor.v:807-823), all eight up front
                u32 c8a[8], hma[8];
                i32 ya[8];
                {
                    u32 c8 = 1, hm = 1;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const u32 y = (ch >> (7 - k)) & 1u;
                        c8a[k] = c8; hma[k] = hm; ya[k] = (i32)y;
                        c8 = (c8 << 1) | y;
                        if (k == 3) hm = ((hm & 0xfu) << 5) | (y << 4) | 1u;
                        else hm = (hm & 0x1f0u) | (((hm & 0xfu) * 2u + y) & 0xfu);
                    }
                }
                if (TYPE == ZT_CM) {                             // predictor.v:549-554,681-700
                    u32 idx[8], v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { idx[k] = (hctx ^ hma[k]) & (cm_len - 1u); v[k] = cm[idx[k]]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        u32 pn = v[k];
#pragma unroll
                        for (int j = 0; j < k; j++) pn = idx[j] == idx[k] ? v[j] : pn;
                        out.set(k, stretch((i32)(pn >> 17)));
                        const i32 count = (i32)(pn & 0x3ffu);
                        const i32 err = (ya[k] ? 32767 : 0) - (i32)(pn >> 17);
                        const i32 upd = wmul(err, (i32)s_dt[count]) & -1024;
                        v[k] = (u32)wadd(wadd((i32)pn, upd), count < climit ? 1 : 0);
                        cm[idx[k]] = v[k];
                    }
                } else if (TYPE == ZT_ICM || TYPE == ZT_ISSE) {  // predictor.v:495-532,555-563,615-631,701-709,776-791
                    u32 R[2][4], ro[2];
                    {
                        u32x4 A[2], Bq[2], Cq[2];
                        u32 chk[2], h0[2];
#pragma unroll
                        for (int nb = 0; nb < 2; nb++) {
                            const u32 cx = hctx + 16u * c8a[4 * nb];
                            chk[nb] = (cx >> (ca + 2)) & 255u;
                            h0[nb] = (cx * 16u) & (ht_len - 16u);
                            A[nb] = *reinterpret_cast<const u32x4 *>(ht + h0[nb]);
                            Bq[nb] = *reinterpret_cast<const u32x4 *>(ht + (h0[nb] ^ 16u));
                            Cq[nb] = *reinterpret_cast<const u32x4 *>(ht + (h0[nb] ^ 32u));
                        }
#pragma unroll
                        for (int nb = 0; nb < 2; nb++) {
                            const bool ma_ = (A[nb].x & 255u) == chk[nb], mb_ = (Bq[nb].x & 255u) == chk[nb], mc_ = (Cq[nb].x & 255u) == chk[nb];
                            const u32 qa = (A[nb].x >> 8) & 255u, qb = (Bq[nb].x >> 8) & 255u, qc = (Cq[nb].x >> 8) & 255u;
                            const bool va = qa <= qb && qa <= qc, vb = qb < qc;
                            const bool hit = ma_ || mb_ || mc_;
                            const bool ua = ma_ || (!hit && va);
                            const bool ub = !ua && (mb_ || (!hit && vb));
                            ro[nb] = ua ? h0[nb] : (ub ? (h0[nb] ^ 16u) : (h0[nb] ^ 32u));
                            R[nb][0] = hit ? (ua ? A[nb].x : (ub ? Bq[nb].x : Cq[nb].x)) : chk[nb];
                            R[nb][1] = hit ? (ua ? A[nb].y : (ub ? Bq[nb].y : Cq[nb].y)) : 0u;
                            R[nb][2] = hit ? (ua ? A[nb].z : (ub ? Bq[nb].z : Cq[nb].z)) : 0u;
                            R[nb][3] = hit ? (ua ? A[nb].w : (ub ? Bq[nb].w : Cq[nb].w)) : 0u;
                        }
                    }
                    // the eight states: a bit reads a slot of its nibble's row that no earlier bit of the nibble writes
                    u32 st[8], w0[8], w1[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const u32 slotn = hma[k] & 15u;
                        const u32 *r = R[k >> 2];
                        const u32 dsel = (slotn & 8u) ? ((slotn & 4u) ? r[3] : r[2]) : ((slotn & 4u) ? r[1] : r[0]);
                        st[k] = (dsel >> ((slotn & 3u) * 8u)) & 255u;
                        if (TYPE == ZT_ICM) { w0[k] = cm[st[k]]; w1[k] = 0; }
                        else { const uint2 w = *reinterpret_cast<const uint2 *>(cm + st[k] * 2); w0[k] = w.x; w1[k] = w.y; }
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        u32 a0 = w0[k], a1 = w1[k];
#pragma unroll
                        for (int j = 0; j < k; j++) { const bool same = st[j] == st[k]; a0 = same ? w0[j] : a0; a1 = same ? w1[j] : a1; }
                        const i32 t32767 = ya[k] ? 32767 : 0;
                        if (TYPE == ZT_ICM) {
                            out.set(k, stretch((i32)(a0 >> 8)));
                            w0[k] = (u32)wadd((i32)a0, (t32767 - (i32)(a0 >> 8)) >> 2);
                            cm[st[k]] = w0[k];
                        } else {
                            const i32 pj = in0.get(k);
                            const i32 p = clamp2k(wadd(wmul((i32)a0, pj), wmul((i32)a1, 64)) >> 16);
                            out.set(k, p);
                            const i32 err = t32767 - squash(p);
                            w0[k] = (u32)clamp512k(wadd((i32)a0, wadd(wmul(err, pj), 1 << 12) >> 13));
                            w1[k] = (u32)clamp512k(wadd((i32)a1, (err + 16) >> 5));
                            *reinterpret_cast<uint2 *>(cm + st[k] * 2) = make_uint2(w0[k], w1[k]);
                        }
                        const u32 slotn = hma[k] & 15u, sh = (slotn & 3u) * 8u;
                        const u32 nsv = s_ns[st[k] * 4 + (u32)ya[k]];   // statetable.v:75-84
                        u32 *r = R[k >> 2];
#pragma unroll
                        for (int q = 0; q < 4; q++) r[q] = (slotn >> 2) == (u32)q ? ((r[q] & ~(255u << sh)) | (nsv << sh)) : r[q];
                        if ((k & 3) == 3) *reinterpret_cast<u32x4 *>(ht + ro[k >> 2]) = u32x4{r[0], r[1], r[2], r[3]};
                    }
                } else if (TYPE == ZT_MATCH) {                   // predictor.v:564-574,710-741 (see the bit-serial code below)
                    const i32 mask = (i32)(ht_len - 1);
                    const i32 cmi = (i32)hctx & (i32)(cm_len - 1);
                    const u32 cand = cm[cmi];                    // read at the end of the byte by the reference; nothing writes it in between
                    if (ma != 0) mpred = ht[wsub(mlimit, mb) & mask];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        i32 p = 0;
                        if (ma != 0) {
                            mc = (i32)((mpred >> (7 - k)) & 1u);
                            p = stretch((s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767);
                        }
                        out.set(k, p);
                        if (mc != ya[k]) ma = 0;
                    }
                    ht[mlimit & mask] = (u8)ch;
                    mlimit = wadd(mlimit, 1) & mask;
                    if (ma == 0) {
                        mb = wsub(mlimit, (i32)cand);
                        if ((mb & mask) != 0) {
                            while (ma < 255) {                   // four candidate bytes per round trip
                                u32 x[4], z[4];
#pragma unroll
                                for (int t = 0; t < 4; t++) {
                                    x[t] = ht[wsub(wsub(mlimit, ma), 1 + t) & mask];
                                    z[t] = ht[wsub(wsub(wsub(mlimit, ma), mb), 1 + t) & mask];
                                }
                                int eq = 0;
#pragma unroll
                                for (int t = 3; t >= 0; t--) eq = x[t] == z[t] ? eq + 1 : 0;
                                // eq = equal pairs counted from t = 0 up to the first mismatch
                                ma = min(ma + eq, 255);
                                if (eq < 4) break;
                            }
                        }
                    } else if (ma < 255) ma++;
                    cm[cmi] = (u32)mlimit;
                } else if (TYPE == ZT_MIX2) {                    // predictor.v:586-599,744-762
                    u32 idx[8], v[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { idx[k] = (hctx + (c8a[k] & (u32)cmask)) & (u32)(cc - 1); v[k] = a16[idx[k]]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        u32 wv_ = v[k];
#pragma unroll
                        for (int j = 0; j < k; j++) wv_ = idx[j] == idx[k] ? v[j] : wv_;
                        const i32 w = (i32)wv_;
                        const i32 pj = in0.get(k), pk = in1.get(k);
                        const i32 p = clamp2k(wadd(wmul(w, pj), wmul(65536 - w, pk)) >> 16);
                        out.set(k, p);
                        const i32 err = wmul((ya[k] ? 32767 : 0) - squash(p), crate) >> 5;
                        i32 wn = wadd(w, wadd(wmul(err, wsub(pj, pk)), 1 << 12) >> 13);
                        wn = min(max(wn, 0), 65535);
                        v[k] = (u32)wn;
                        a16[idx[k]] = (u16)wn;
                    }
                } else if (TYPE == ZT_MIX) {                     // predictor.v:600-614,763-775; a nibble's four weight rows at a time
                    const u8 *inb[8];
#pragma unroll
                    for (int l = 0; l < 8; l++) inb[l] = reinterpret_cast<const u8 *>(ring_at(l < climit ? cb + l : cb, bi));
#pragma unroll
                    for (int nb = 0; nb < 2; nb++) {
                        u32 cx[4];
                        i32 wv[4][8];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            cx[q] = (u32)(wadd((i32)hctx, (i32)c8a[4 * nb + q] & cmask) & (cc - 1));
                            const u32 *wrow = cm + (size_t)wmul((i32)cx[q], climit);
#pragma unroll
                            for (int l = 0; l < 8; l++) wv[q][l] = 0;
                            {   // eight words whatever the row's length (no branch around a load; see comp_dec's load_row)
                                const u32x4a lo = *reinterpret_cast<const u32x4a *>(wrow), hi = *reinterpret_cast<const u32x4a *>(wrow + 4);
                                wv[q][0] = (i32)lo.x; wv[q][1] = (i32)lo.y; wv[q][2] = (i32)lo.z; wv[q][3] = (i32)lo.w;
                                wv[q][4] = (i32)hi.x; wv[q][5] = (i32)hi.y; wv[q][6] = (i32)hi.z; wv[q][7] = (i32)hi.w;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int k = 4 * nb + q;
#pragma unroll
                            for (int j = 0; j < q; j++) {
                                const bool same = cx[j] == cx[q];
#pragma unroll
                                for (int l = 0; l < 8; l++) wv[q][l] = same ? wv[j][l] : wv[q][l];
                            }
                            i32 pin[8];
                            i32 sum = 0;
#pragma unroll
                            for (int l = 0; l < 8; l++) {
                                pin[l] = l < climit ? (i32)*reinterpret_cast<const int16_t *>(inb[l] + 2 * k) : 0;
                                sum = wadd(sum, wmul(wv[q][l] >> 8, pin[l]));
                            }
                            const i32 p = clamp2k(sum >> 8);
                            out.set(k, p);
                            const i32 err = wmul((ya[k] ? 32767 : 0) - squash(p), crate) >> 4;
#pragma unroll
                            for (int l = 0; l < 8; l++) wv[q][l] = clamp512k(wadd(wv[q][l], wadd(wmul(err, pin[l]), 1 << 12) >> 13));
                            u32 *wrow = cm + (size_t)wmul((i32)cx[q], climit);
                            auto stw = [&](const int at, const int cnt) {
                                if (cnt >= 4) *reinterpret_cast<u32x4a *>(wrow + at) = u32x4a{(u32)wv[q][at], (u32)wv[q][at + 1], (u32)wv[q][at + 2], (u32)wv[q][at + 3]};
                                else if (cnt == 3) *reinterpret_cast<u32x3a *>(wrow + at) = u32x3a{(u32)wv[q][at], (u32)wv[q][at + 1], (u32)wv[q][at + 2]};
                                else if (cnt == 2) *reinterpret_cast<u32x2a *>(wrow + at) = u32x2a{(u32)wv[q][at], (u32)wv[q][at + 1]};
                                else if (cnt == 1) wrow[at] = (u32)wv[q][at];
                            };
                            stw(0, climit);
                            if (climit > 4) stw(4, climit - 4);
                        }
                    }
                } else if (TYPE == ZT_SSE) {                     // predictor.v:632-660,792-805
                    i32 idx[8], wt[8];
                    u32 iu[8], e0[8], e1[8], nv[8];
                    bool ok[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const u32 cxt = (hctx + c8a[k]) * 32u;
                        i32 pq = wadd(in0.get(k), 992);
                        pq = min(max(pq, 0), 1983);
                        wt[k] = pq & 63;
                        pq >>= 6;
                        idx[k] = wadd((i32)cxt, pq);
                        ok[k] = idx[k] >= 0 && wadd(idx[k], 1) < (i32)cm_len;
                        iu[k] = ((u32)idx[k] + (u32)(wt[k] >> 5)) & (cm_len - 1u);
                        e0[k] = e1[k] = 0;
                        if (ok[k]) { const u32x2a e = *reinterpret_cast<const u32x2a *>(cm + idx[k]); e0[k] = e.x; e1[k] = e.y; }
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        u32 a0 = e0[k], a1 = e1[k];
#pragma unroll
                        for (int j = 0; j < k; j++) {            // entries an earlier bit of the byte rewrote
                            a0 = iu[j] == (u32)idx[k] ? nv[j] : a0;
                            a1 = iu[j] == (u32)idx[k] + 1u ? nv[j] : a1;
                        }
                        out.set(k, ok[k] ? stretch(wadd(wmul((i32)(a0 >> 10), 64 - wt[k]), wmul((i32)(a1 >> 10), wt[k])) >> 13) : 0);
                        u32 v;
                        if (ok[k] && iu[k] == (u32)idx[k]) v = a0;
                        else if (ok[k] && iu[k] == (u32)idx[k] + 1u) v = a1;
                        else v = cm[iu[k]];                      // (outside the pair: issued after the earlier bits' stores)
                        const i32 err = (ya[k] ? 32767 : 0) - (i32)(v >> 17);
                        const i32 count = (i32)v & 1023;
                        if (count < climit) v = (u32)wadd(wadd((i32)v, wadd(wmul(err, climit - count), 1 << 12) >> 13), 1);
                        nv[k] = v;
                        cm[iu[k]] = v;
                    }
                }
            } else {
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
                    // The reference shifts the bit into ht[limit] and, while a match is alive, re-reads the predicted byte
                    // ht[limit - b] every bit.  Neither byte changes under the other's feet (limit - b == limit needs
                    // b & mask == 0, and then no match is ever alive), so: the predicted byte is read once per byte, the coded
                    // byte is written once, whole (eight shifts of a u8 leave exactly the byte).
                    const i32 mask = (i32)(ht_len - 1);
                    if (kb == 0 && ma != 0) mpred = ht[wsub(mlimit, mb) & mask];
                    if (ma == 0) p = 0;
                    else {
                        mc = (i32)((mpred >> (7u - mcxt)) & 1u);
                        p = stretch((s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767);
                    }
                    if (mc != y) ma = 0;
                    mcxt++;
                    if (mcxt >= 8) {
                        mcxt = 0;
                        ht[mlimit & mask] = (u8)ch;
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
                    // the row of m <= 8 weights: one or two vector accesses (word-aligned) instead of m
                    i32 wv[8], pin[8];
#pragma unroll
                    for (int l = 0; l < 8; l++) wv[l] = 0;
                    auto ld = [&](const int at, const int cnt) {
                        if (cnt >= 4) { const u32x4a v = *reinterpret_cast<const u32x4a *>(wrow + at); wv[at] = (i32)v.x; wv[at + 1] = (i32)v.y; wv[at + 2] = (i32)v.z; wv[at + 3] = (i32)v.w; }
                        else if (cnt == 3) { const u32x3a v = *reinterpret_cast<const u32x3a *>(wrow + at); wv[at] = (i32)v.x; wv[at + 1] = (i32)v.y; wv[at + 2] = (i32)v.z; }
                        else if (cnt == 2) { const u32x2a v = *reinterpret_cast<const u32x2a *>(wrow + at); wv[at] = (i32)v.x; wv[at + 1] = (i32)v.y; }
                        else if (cnt == 1) wv[at] = (i32)wrow[at];
                    };
                    ld(0, climit);
                    if (climit > 4) ld(4, climit - 4);
                    i32 sum = 0;
#pragma unroll
                    for (int l = 0; l < 8; l++) { pin[l] = inm[l].get(kb); sum = wadd(sum, wmul(wv[l] >> 8, pin[l])); }
                    p = clamp2k(sum >> 8);
                    const i32 err = wmul(t32767 - squash(p), crate) >> 4;
#pragma unroll
                    for (int l = 0; l < 8; l++) wv[l] = clamp512k(wadd(wv[l], wadd(wmul(err, pin[l]), 1 << 12) >> 13));
                    auto st = [&](const int at, const int cnt) {
                        if (cnt >= 4) *reinterpret_cast<u32x4a *>(wrow + at) = u32x4a{(u32)wv[at], (u32)wv[at + 1], (u32)wv[at + 2], (u32)wv[at + 3]};
                        else if (cnt == 3) *reinterpret_cast<u32x3a *>(wrow + at) = u32x3a{(u32)wv[at], (u32)wv[at + 1], (u32)wv[at + 2]};
                        else if (cnt == 2) *reinterpret_cast<u32x2a *>(wrow + at) = u32x2a{(u32)wv[at], (u32)wv[at + 1]};
                        else if (cnt == 1) wrow[at] = (u32)wv[at];
                    };
                    st(0, climit);
                    if (climit > 4) st(4, climit - 4);
                } else if (TYPE == ZT_SSE) {                     // predictor.v:632-660,792-805
                    const u32 cxt = (hctx + c8) * 32u;
                    i32 pq = wadd(in0.get(kb), 992);
                    pq = min(max(pq, 0), 1983);
                    const i32 wt = pq & 63;
                    pq >>= 6;
                    const i32 idx = wadd((i32)cxt, pq), idx2 = wadd(idx, 1);
                    const bool ok = idx >= 0 && idx2 < (i32)cm_len;
                    u32 e0 = 0, e1 = 0;
                    if (ok) { const u32x2a e = *reinterpret_cast<const u32x2a *>(cm + idx); e0 = e.x; e1 = e.y; }
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
            }
            *ring_at(ci, bi) = make_uint4(out.a, out.b, out.c, out.d);
            {   // ZPAQL.run(byte) of the shipped hash chain: H[ci] for the next byte
                u32 a = ch;
                for (u32 k = 0; k < hash_steps; k++) a = (a + prev + 512u) * 773u;
                hctx = hash_steps ? a : 0u;
                prev = ch;
            }
        }
#ifdef ZPG_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const_cast<Stage &>(S).busy += __builtin_readcyclecounter() - t_in;
#endif
        lds_barrier();
    }
}

// the coder wave (encoder.v:48-139): byte it - n, predictions from the last component's ring; returns the bytes written
__device__ __forceinline__ u32 coder_stage(const Stage &S)
{
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const int n = S.n;
    const GCfg &G = *S.cfg;
    const uint4 *const link = reinterpret_cast<const uint4 *>(lds + L_LINK);
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
    const u8 *const s_bytes = lds + L_BYTES;
    for (u32 it = 0; it < S.iters; it++) {
        const u32 bi = it - (u32)n;
        if (S.active && bi < S.total) {
            const u32 ch = s_bytes[(bi & 15u) * BPW + S.lane];
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

// Predictor.init + ZPAQL.clear for the nact blocks of a workgroup's round (predictor.v:325-470): zero, then the non-zero fills
__device__ __forceinline__ void init_slots(const DBatch &B, const DModel &M, const int n, const int wg_slot0, const int nact, const int tid, const int nthr)
{
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
}

template <bool BATCH>
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
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform, and known to the compiler as such: component constants in SGPRs, scalar branches)
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
        init_slots(B, M, n, wg_slot0, nact, tid, nthr);
        if (tid == 0) *misc = 0u;
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
        S.busy = 0;
#ifdef ZPG_PROF
        const u64 t_all = __builtin_readcyclecounter();
#endif
        if (wave < n) {
            switch (M.comp[wave].type) {                         // uniform per wave
            case ZT_CONST: comp_stage<ZT_CONST, BATCH>(S); break;
            case ZT_CM: comp_stage<ZT_CM, BATCH>(S); break;
            case ZT_ICM: comp_stage<ZT_ICM, BATCH>(S); break;
            case ZT_MATCH: comp_stage<ZT_MATCH, BATCH>(S); break;
            case ZT_AVG: comp_stage<ZT_AVG, BATCH>(S); break;
            case ZT_MIX2: comp_stage<ZT_MIX2, BATCH>(S); break;
            case ZT_MIX: comp_stage<ZT_MIX, BATCH>(S); break;
            case ZT_ISSE: comp_stage<ZT_ISSE, BATCH>(S); break;
            default: comp_stage<ZT_SSE, BATCH>(S); break;
            }
        } else {
            const u32 opos = coder_stage(S);
            if (active) {
                B.out_len[blk] = opos;
                B.status[blk] = opos > S.cap ? ZPQ_E_OVERFLOW : ZPQ_OK;
            }
        }
#ifdef ZPG_PROF
        if (blockIdx.x == 0 && lane == 0 && wave < n)
            printf("wave %d type %d: busy %llu of %llu cycles, %u bytes\n", wave, M.comp[wave].type, (unsigned long long)S.busy,
                   (unsigned long long)(__builtin_readcyclecounter() - t_all), S.iters);
#endif
        __syncthreads();
    }
}


// ================================================================================================================
// The DECODER of the same models: wave i = component i of the workgroup's 64 blocks (lane = block), the last wave = the
// arithmetic decoder -- but bit-synchronous: a decoder cannot run ahead of the bit.  Per bit: every component wave sends
// its table loads out at once (addresses depend on contexts only; an SSE takes its whole 32-entry row, the entry is picked
// when its input is known), the predictions are then made level by level -- level 0 = components without inputs, level
// k = 1 + the deepest input -- and handed on through LDS with one s_barrier per level; the decoder wave reads the last
// component's prediction, decodes the bit and hands it back; every wave trains its component and moves to the next bit's
// contexts.  Against zpq_lanes.hip (lane = component, four blocks per wave) a wave issues only its own type's code, for
// 64 blocks, and the bit costs one trip to the tables plus the chain of levels.
#ifndef ZPG_HASH_SPEC
#define ZPG_HASH_SPEC 0                              // ICM / ISSE asking for the two states the next bit can meet: two random lines of the
                                                     // state table where one is needed -- measured 2081 ms against 2038 without (same box); 1: timing builds
#endif
// Which component types ask for the next bit's entries under both values of the bit -- decided by A/B on one box (C4b, 16 384
// blocks, decode ms; default 2079-2083): the kernel runs at ~0.85 of the chip's random-line rate, so a request pays only where it
// brings no new line (CM, MIX2: the neighbour word: 2090 / 2094 without) or takes a whole trip off the end of the prediction
// chain (SSE, the last level: 2330 without); the MIX's two rows (2001 WITHOUT) and the two states of an ICM / ISSE (2038
// without) cost more in lines than they hide.
#ifndef ZPG_CM_SPEC
#define ZPG_CM_SPEC 1
#endif
#ifndef ZPG_MIX2_SPEC
#define ZPG_MIX2_SPEC 1
#endif
#ifndef ZPG_MIX_SPEC
#define ZPG_MIX_SPEC 0
#endif
#ifndef ZPG_SSE_SPEC
#define ZPG_SSE_SPEC 1                               // an SSE asks for the next bit's two candidate rows (timing builds: 0)
#endif
constexpr int D_P = L_DT2K + 512;                // i32 p[16][BPW]: this bit's predictions
constexpr int D_Y = D_P + 16 * BPW * 4;          // u32 y[BPW]
constexpr int D_ALIVE = D_Y + BPW * 4;           // u32 alive[BPW]: the block has not met its EOF flag yet
constexpr int D_ANY = D_ALIVE + BPW * 4;         // u32: some block of the workgroup is alive
constexpr int D_SSE = D_ANY + 16;                // per SSE component: its 32-entry row per lane (entries 0..31: the input picks 0..30 and the next), 36 words apart
constexpr int D_SSE_BYTES = 36 * 4 * BPW;

struct DStage {
    const DBatch *B;
    const DModel *M;
    u8 *lds;
    const GCfg *cfg;
    int ci, lane, n;
    bool active;
    u8 *slot;
    const u8 *src;
    u32 nin;
    u8 *dst;
    u32 cap;
    int blk;
    u64 prof[4];                 // (-DZPG_PROF) cycles: loads in, levels, waiting for the bit, training
};

template <int TYPE>
__device__ __forceinline__ void comp_dec(const DStage &S)
{
    const DModel &M = *S.M;
    u8 *const lds = S.lds;
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + L_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const u8 *s_ns = lds + L_NS;
    const u32 *s_dt = reinterpret_cast<const u32 *>(lds + L_DT);
    const int16_t *s_dt2k = reinterpret_cast<const int16_t *>(lds + L_DT2K);
    auto squash = [&](i32 d) -> i32 { return s_squash[min(max(wadd(d, 2047), 0), 4093)]; };
    auto stretch = [&](i32 pr) -> i32 {
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
    i32 *const s_p = reinterpret_cast<i32 *>(lds + D_P);
    const u32 *const s_y = reinterpret_cast<const u32 *>(lds + D_Y);
    const u32 *const s_alive = reinterpret_cast<const u32 *>(lds + D_ALIVE);
    const u32 *const s_any = reinterpret_cast<const u32 *>(lds + D_ANY);
    u32 *const s_row = reinterpret_cast<u32 *>(lds + D_SSE + (int)G.dslot[ci] * D_SSE_BYTES) + S.lane * 36;   // (SSE only)
    auto pin_of = [&](const int comp) -> i32 { return s_p[comp * BPW + S.lane]; };
    // a MIX row of climit <= 8 weights, read as eight words whatever its length: no branch around a load (the compiler waits for
    // a load at the join behind it), words past the row are multiplied by inputs that are 0 and never stored; the slot pool
    // ends in 256 spare bytes for the last row of the last table
    auto load_row = [&](const u32 *wrow, i32 (&w)[8]) {
        const u32x4a lo = *reinterpret_cast<const u32x4a *>(wrow), hi = *reinterpret_cast<const u32x4a *>(wrow + 4);
        w[0] = (i32)lo.x; w[1] = (i32)lo.y; w[2] = (i32)lo.z; w[3] = (i32)lo.w;
        w[4] = (i32)hi.x; w[5] = (i32)hi.y; w[6] = (i32)hi.z; w[7] = (i32)hi.w;
    };
    (void)load_row;
    const int mylevel = G.level[ci], nlev = G.nlevels;
    const u32 hash_steps = (u32)ci < (u32)G.hashes ? (u32)ci + 1u : 0u;

    u32 prev = 0, hctx = 0;
    i32 ma = (TYPE == ZT_MATCH) ? ca : 0, mb = (TYPE == ZT_MATCH) ? cb : 0, mc = 0, mlimit = 0;   // quirk Q17
    u32 mpred = 0, mcand = 0;
    u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0, roff = 0;
    (void)ma; (void)mb; (void)mc; (void)mlimit; (void)mpred; (void)mcand; (void)r0; (void)r1; (void)r2; (void)r3; (void)roff;
    // Inside a nibble the NEXT bit's contexts are known now but for this bit: its table entries are asked for under both values
    // (CM, MIX2, MIX: neighbouring entries / rows; ICM, ISSE: the two states the nibble's row holds for them) while this bit
    // is still being predicted, and picked when the bit is known.  An entry that this bit's training then rewrites is taken from
    // the training (tr_*), not from the load that went out before it.
    constexpr bool SPEC = (TYPE == ZT_CM && ZPG_CM_SPEC) || ((TYPE == ZT_ICM || TYPE == ZT_ISSE) && ZPG_HASH_SPEC) || (TYPE == ZT_MIX2 && ZPG_MIX2_SPEC) ||
                          (TYPE == ZT_MIX && ZPG_MIX_SPEC) || (TYPE == ZT_SSE && ZPG_SSE_SPEC);
    // the entry arrives in nx0 / nx1 / nxw whether asked for early or not, and is taken from there -- or from the last training
    constexpr bool FWD = TYPE == ZT_CM || TYPE == ZT_ICM || TYPE == ZT_ISSE || TYPE == ZT_MIX2 || TYPE == ZT_MIX;   // (an SSE trains its CURRENT row; the next bit's row is patched in LDS if ever hit)
    bool spec = false;
    u32 sa0 = 0, sa1 = 0, sb0 = 0, sb1 = 0, tr_a = 0xFFFFFFFFu, tr0 = 0, tr1 = 0, yprev = 0;
    u32 nx0 = 0, nx1 = 0;                                        // ... picked as soon as the bit is known, BEFORE this bit's stores go out:
    i32 swa[8], swb[8], trw[8], nxw[8];                          // a wait placed behind the stores would wait for their acknowledgement too
    (void)nx0; (void)nx1; (void)nxw;
    u32x4 ra[8], rb[8];                                          // SSE: the next bit's two candidate rows
    bool sok0 = false, sok1 = false, nrow_ok = false;
    u32 nrow = 0;
    (void)ra; (void)rb; (void)sok0; (void)sok1; (void)nrow_ok; (void)nrow;
    (void)spec; (void)sa0; (void)sa1; (void)sb0; (void)sb1; (void)tr_a; (void)tr0; (void)tr1; (void)yprev; (void)swa; (void)swb; (void)trw;

    for (;;) {
        lds_barrier();                                           // the decoder wave has looked at the EOF flag
        const bool alive = S.active && s_alive[S.lane] != 0;
        if (*s_any == 0) break;
        u32 c8 = 1, hmap4 = 1;
#pragma unroll 1
        for (int kb = 0; kb < 8; kb++) {
#ifdef ZPG_PROF
            const u64 t0 = __builtin_readcyclecounter();
#endif
            // ---- this bit's table entries, requested as soon as the contexts are known
            u32 v0 = 0, v1 = 0, st = 0, idx = 0;
            i32 wv[8], pinv[8];
            bool row_ok = false;
            (void)v0; (void)v1; (void)st; (void)idx; (void)wv; (void)pinv; (void)row_ok;
            if (alive) {
                if (TYPE == ZT_CM) {
                    idx = (hctx ^ hmap4) & (cm_len - 1u);
                    if (!spec) nx0 = cm[idx];
                } else if (TYPE == ZT_ICM || TYPE == ZT_ISSE) {
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
                    st = (dsel >> ((slotn & 3u) * 8u)) & 255u;
                    idx = st;
                    if (!spec) {
                        if (TYPE == ZT_ICM) nx0 = cm[st];
                        else { const uint2 w = *reinterpret_cast<const uint2 *>(cm + st * 2); nx0 = w.x; nx1 = w.y; }
                    }
                } else if (TYPE == ZT_MATCH) {
                    if (kb == 0) {                               // the predicted byte, and the candidate position the byte's END will want
                        mpred = ht[wsub(mlimit, mb) & (i32)(ht_len - 1)];
                        mcand = cm[(i32)hctx & (i32)(cm_len - 1)];
                    }
                } else if (TYPE == ZT_MIX2) {
                    idx = (hctx + (c8 & (u32)cmask)) & (u32)(cc - 1);
                    if (!spec) nx0 = a16[idx];
                } else if (TYPE == ZT_MIX) {
                    idx = (u32)(wadd((i32)hctx, (i32)c8 & cmask) & (cc - 1));
                    if (!spec) load_row(cm + (size_t)wmul((i32)idx, climit), nxw);
                } else if (TYPE == ZT_SSE) {
                    idx = (hctx + c8) * 32u;                     // the row of this bit context; entry = row + f(input)
                    if (spec) row_ok = nrow_ok;                  // (the row was put into LDS when the last bit became known)
                    else {
                        row_ok = (i32)idx >= 0 && idx + 32u <= cm_len;
                        if (row_ok) {
#pragma unroll
                            for (int q = 0; q < 8; q++) *reinterpret_cast<u32x4 *>(s_row + 4 * q) = *reinterpret_cast<const u32x4 *>(cm + idx + 4 * q);
                        }
                    }
                }
            }
            // ---- the next bit's entries under both values of this bit
            // (a bit-history row changes with the nibble; everything else only needs the byte to go on: hctx stays)
            const bool spec_next = SPEC && alive && kb != 7 && !((TYPE == ZT_ICM || TYPE == ZT_ISSE) && kb == 3);
            if (spec_next) {
                const u32 hm0 = kb == 3 ? (((hmap4 & 0xfu) << 5) | 1u) : ((hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u) & 0xfu));
                const u32 hm1 = kb == 3 ? (((hmap4 & 0xfu) << 5) | 17u) : ((hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u + 1u) & 0xfu));
                const u32 c80 = c8 << 1, c81 = (c8 << 1) | 1u;
                (void)hm0; (void)hm1; (void)c80; (void)c81;
                if (TYPE == ZT_CM) {
                    sa0 = cm[(hctx ^ hm0) & (cm_len - 1u)];
                    sb0 = cm[(hctx ^ hm1) & (cm_len - 1u)];
                } else if (TYPE == ZT_ICM || TYPE == ZT_ISSE) {
                    // (a 64-bit shift, not a four-way select: two of those in a row the compiler turns into a stack array)
                    const u64 rlo = (u64)r0 | ((u64)r1 << 32), rhi = (u64)r2 | ((u64)r3 << 32);
                    auto state_at = [&](const u32 slotn) -> u32 { return (u32)(((slotn & 8u) ? rhi : rlo) >> ((slotn & 7u) * 8u)) & 255u; };
                    const u32 st0 = state_at(hm0 & 15u), st1 = state_at(hm1 & 15u);
                    if (TYPE == ZT_ICM) { sa0 = cm[st0]; sb0 = cm[st1]; }
                    else {
                        const uint2 wa = *reinterpret_cast<const uint2 *>(cm + st0 * 2), wb = *reinterpret_cast<const uint2 *>(cm + st1 * 2);
                        sa0 = wa.x; sa1 = wa.y; sb0 = wb.x; sb1 = wb.y;
                    }
                } else if (TYPE == ZT_MIX2) {
                    sa0 = a16[(hctx + (c80 & (u32)cmask)) & (u32)(cc - 1)];
                    sb0 = a16[(hctx + (c81 & (u32)cmask)) & (u32)(cc - 1)];
                } else if (TYPE == ZT_MIX) {
                    load_row(cm + (size_t)wmul(wadd((i32)hctx, (i32)c80 & cmask) & (cc - 1), climit), swa);
                    load_row(cm + (size_t)wmul(wadd((i32)hctx, (i32)c81 & cmask) & (cc - 1), climit), swb);
                } else if (TYPE == ZT_SSE) {
                    const u32 i0 = (hctx + c80) * 32u, i1 = (hctx + c81) * 32u;
                    sok0 = (i32)i0 >= 0 && i0 + 32u <= cm_len;
                    sok1 = (i32)i1 >= 0 && i1 + 32u <= cm_len;
                    const u32 j0 = sok0 ? i0 : 0u, j1 = sok1 ? i1 : 0u;     // (no branch around a load: a row outside the table reads row 0, unused)
#pragma unroll
                    for (int q = 0; q < 8; q++) ra[q] = *reinterpret_cast<const u32x4 *>(cm + j0 + 4 * q);
#pragma unroll
                    for (int q = 0; q < 8; q++) rb[q] = *reinterpret_cast<const u32x4 *>(cm + j1 + 4 * q);
                }
            }
#ifdef ZPG_PROF
            const u64 t1 = __builtin_readcyclecounter();         // (loads issued, not waited for)
#endif
            // ---- predictions, level by level
            i32 p = 0, sse_i = 0, sse_wt = 0;
            bool sse_ok = false;
            (void)sse_i; (void)sse_wt; (void)sse_ok;
            for (int lv = 0; lv < nlev; lv++) {
                if (lv == mylevel && alive) {
                    if (FWD) {                                   // the entry as it stands now: what came in, or what the last bit's training left there
                        const bool fwd = idx == tr_a;
                        v0 = fwd ? tr0 : nx0; v1 = fwd ? tr1 : nx1;
                        if (TYPE == ZT_MIX) {
#pragma unroll
                            for (int l = 0; l < 8; l++) wv[l] = fwd ? trw[l] : nxw[l];
                        }
                    }
                    if (TYPE == ZT_CONST) p = (ca - 128) * 16;
                    else if (TYPE == ZT_CM) p = stretch((i32)(v0 >> 17));
                    else if (TYPE == ZT_ICM) p = stretch((i32)(v0 >> 8));
                    else if (TYPE == ZT_MATCH) {
                        if (ma == 0) p = 0;
                        else {
                            mc = (i32)((mpred >> (7 - kb)) & 1u);
                            p = stretch((s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767);
                        }
                    } else if (TYPE == ZT_AVG) p = wadd(wmul(pin_of(ca), cc), wmul(pin_of(cb), 256 - cc)) >> 8;
                    else if (TYPE == ZT_MIX2) {
                        pinv[0] = pin_of(cj); pinv[1] = pin_of(ck);
                        p = clamp2k(wadd(wmul((i32)v0, pinv[0]), wmul(65536 - (i32)v0, pinv[1])) >> 16);
                    } else if (TYPE == ZT_ISSE) {
                        pinv[0] = pin_of(cb);
                        p = clamp2k(wadd(wmul((i32)v0, pinv[0]), wmul((i32)v1, 64)) >> 16);
                    } else if (TYPE == ZT_MIX) {
                        i32 sum = 0;
#pragma unroll
                        for (int l = 0; l < 8; l++) { pinv[l] = l < climit ? pin_of(cb + l) : 0; sum = wadd(sum, wmul(wv[l] >> 8, pinv[l])); }
                        p = clamp2k(sum >> 8);
                    } else if (TYPE == ZT_SSE) {
                        i32 pq = wadd(pin_of(cb), 992);
                        pq = min(max(pq, 0), 1983);
                        sse_wt = pq & 63;
                        pq >>= 6;
                        sse_i = wadd((i32)idx, pq);
                        sse_ok = sse_i >= 0 && wadd(sse_i, 1) < (i32)cm_len;
                        if (sse_ok) {                            // (then the row is in: idx >= 0 and idx + 32 <= cm_len)
                            v0 = s_row[pq]; v1 = s_row[pq + 1];
                            p = stretch(wadd(wmul((i32)(v0 >> 10), 64 - sse_wt), wmul((i32)(v1 >> 10), sse_wt)) >> 13);
                        } else p = 0;
                    }
                    s_p[ci * BPW + S.lane] = p;
                }
                lds_barrier();
            }
#ifdef ZPG_PROF
            const u64 t2 = __builtin_readcyclecounter();
#endif
            lds_barrier();                                       // the decoder wave has the bit
            const i32 y = (i32)(s_y[S.lane] & 1u);
#ifdef ZPG_PROF
            const u64 t3 = __builtin_readcyclecounter();
#endif
            if (spec_next) {
                nx0 = y ? sb0 : sa0; nx1 = y ? sb1 : sa1;
                if (TYPE == ZT_MIX) {
#pragma unroll
                    for (int l = 0; l < 8; l++) nxw[l] = y ? swb[l] : swa[l];
                }
                if (TYPE == ZT_SSE) {                            // (this bit's entries are in v0 / v1 already: the LDS row is free)
                    nrow_ok = y ? sok1 : sok0;
                    nrow = (hctx + ((c8 << 1) | (u32)y)) * 32u;
                    if (nrow_ok) {
#pragma unroll
                        for (int q = 0; q < 8; q++) *reinterpret_cast<u32x4 *>(s_row + 4 * q) = y ? rb[q] : ra[q];
                    }
                }
            }
            // ---- train (predictor.v:670-805)
            if (alive) {
                const i32 t32767 = y ? 32767 : 0;
                if (TYPE == ZT_CM) {
                    const i32 count = (i32)(v0 & 0x3ffu);
                    const i32 err = t32767 - (i32)(v0 >> 17);
                    const i32 upd = wmul(err, (i32)s_dt[count]) & -1024;
                    tr_a = idx; tr0 = (u32)wadd(wadd((i32)v0, upd), count < climit ? 1 : 0);
                    cm[idx] = tr0;
                } else if (TYPE == ZT_ICM || TYPE == ZT_ISSE) {
                    tr_a = st;
                    if (TYPE == ZT_ICM) { tr0 = (u32)wadd((i32)v0, (t32767 - (i32)(v0 >> 8)) >> 2); cm[st] = tr0; }
                    else {
                        const i32 err = t32767 - squash(p);
                        tr0 = (u32)clamp512k(wadd((i32)v0, wadd(wmul(err, pinv[0]), 1 << 12) >> 13));
                        tr1 = (u32)clamp512k(wadd((i32)v1, (err + 16) >> 5));
                        *reinterpret_cast<uint2 *>(cm + st * 2) = make_uint2(tr0, tr1);
                    }
                    const u32 slotn = hmap4 & 15u, sh = (slotn & 3u) * 8u;
                    const u32 nsv = s_ns[st * 4 + (u32)y];
                    const u32 dsel = (slotn & 8u) ? ((slotn & 4u) ? r3 : r2) : ((slotn & 4u) ? r1 : r0);
                    const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
                    r0 = (slotn < 4u) ? ins : r0;
                    r1 = (slotn >= 4u && slotn < 8u) ? ins : r1;
                    r2 = (slotn >= 8u && slotn < 12u) ? ins : r2;
                    r3 = (slotn >= 12u) ? ins : r3;
                    if ((kb & 3) == 3) *reinterpret_cast<u32x4 *>(ht + roff) = u32x4{r0, r1, r2, r3};
                } else if (TYPE == ZT_MATCH) {
                    if (mc != y) ma = 0;
                } else if (TYPE == ZT_MIX2) {
                    const i32 err = wmul(t32767 - squash(p), crate) >> 5;
                    i32 wn = wadd((i32)v0, wadd(wmul(err, wsub(pinv[0], pinv[1])), 1 << 12) >> 13);
                    wn = min(max(wn, 0), 65535);
                    tr_a = idx; tr0 = (u32)wn;
                    a16[idx] = (u16)wn;
                } else if (TYPE == ZT_MIX) {
                    const i32 err = wmul(t32767 - squash(p), crate) >> 4;
#pragma unroll
                    for (int l = 0; l < 8; l++) { wv[l] = clamp512k(wadd(wv[l], wadd(wmul(err, pinv[l]), 1 << 12) >> 13)); trw[l] = wv[l]; }
                    tr_a = idx;
                    u32 *wrow = cm + (size_t)wmul((i32)idx, climit);
                    auto stw = [&](const int at, const int cnt) {
                        if (cnt >= 4) *reinterpret_cast<u32x4a *>(wrow + at) = u32x4a{(u32)wv[at], (u32)wv[at + 1], (u32)wv[at + 2], (u32)wv[at + 3]};
                        else if (cnt == 3) *reinterpret_cast<u32x3a *>(wrow + at) = u32x3a{(u32)wv[at], (u32)wv[at + 1], (u32)wv[at + 2]};
                        else if (cnt == 2) *reinterpret_cast<u32x2a *>(wrow + at) = u32x2a{(u32)wv[at], (u32)wv[at + 1]};
                        else if (cnt == 1) wrow[at] = (u32)wv[at];
                    };
                    stw(0, climit);
                    if (climit > 4) stw(4, climit - 4);
                } else if (TYPE == ZT_SSE) {
                    const u32 iu = ((u32)sse_i + (u32)(sse_wt >> 5)) & (cm_len - 1u);
                    u32 v;
                    if (sse_ok && iu == (u32)sse_i) v = v0;
                    else if (sse_ok && iu == (u32)sse_i + 1u) v = v1;
                    else v = cm[iu];
                    const i32 err = t32767 - (i32)(v >> 17);
                    const i32 count = (i32)v & 1023;
                    if (count < climit) v = (u32)wadd(wadd((i32)v, wadd(wmul(err, climit - count), 1 << 12) >> 13), 1);
                    cm[iu] = v;
                    if (spec_next && nrow_ok && iu - nrow < 32u) s_row[iu - nrow] = v;   // (an index masked back into the table may land in the next row)
                }
            }
            c8 = (c8 << 1) | (u32)y;
            if (c8 >= 256u) { }
            else if (c8 >= 16u && c8 < 32u) hmap4 = ((hmap4 & 0xfu) << 5) | ((u32)y << 4) | 1u;
            else hmap4 = (hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u + (u32)y) & 0xfu);
            spec = spec_next;
            yprev = (u32)y;
#ifdef ZPG_PROF
            {
                const u64 t4 = __builtin_readcyclecounter();
                u64 *pr = const_cast<DStage &>(S).prof;
                pr[0] += t1 - t0; pr[1] += t2 - t1; pr[2] += t3 - t2; pr[3] += t4 - t3;
            }
#endif
        }
        const u32 byte = c8 - 256u;
        if (TYPE == ZT_MATCH && alive) {                         // the byte boundary of predictor.v:722-741
            const i32 mask = (i32)(ht_len - 1);
            ht[mlimit & mask] = (u8)byte;
            mlimit = wadd(mlimit, 1) & mask;
            const i32 cmi = (i32)hctx & (i32)(cm_len - 1);
            if (ma == 0) {
                mb = wsub(mlimit, (i32)mcand);
                if ((mb & mask) != 0) {
                    while (ma < 255) {
                        u32 x[4], z[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            x[t] = ht[wsub(wsub(mlimit, ma), 1 + t) & mask];
                            z[t] = ht[wsub(wsub(wsub(mlimit, ma), mb), 1 + t) & mask];
                        }
                        int eq = 0;
#pragma unroll
                        for (int t = 3; t >= 0; t--) eq = x[t] == z[t] ? eq + 1 : 0;
                        ma = min(ma + eq, 255);
                        if (eq < 4) break;
                    }
                }
            } else if (ma < 255) ma++;
            cm[cmi] = (u32)mlimit;
        }
        {
            u32 a = byte;
            for (u32 k = 0; k < hash_steps; k++) a = (a + prev + 512u) * 773u;
            hctx = hash_steps ? a : 0u;
            prev = byte;
        }
    }
}

// the decoder wave (decoder.v): EOF flag, eight bits per byte from the last component's predictions, output bytes
__device__ __forceinline__ void coder_dec(const DStage &S)
{
    const DBatch &B = *S.B;
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const GCfg &G = *S.cfg;
    const i32 *const s_p = reinterpret_cast<const i32 *>(lds + D_P);
    u32 *const s_y = reinterpret_cast<u32 *>(lds + D_Y);
    u32 *const s_alive = reinterpret_cast<u32 *>(lds + D_ALIVE);
    u32 *const s_any = reinterpret_cast<u32 *>(lds + D_ANY);
    const int n = S.n, nlev = G.nlevels;
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u8 *const src = S.src;
    const u32 nin = S.nin;
    // The coded stream: an eight-byte window over the read position (aligned dwords) and the dword behind it, asked for one bit
    // step before it can be needed.  No load sits inside the divergent byte loop of the decoder (the compiler would wait for
    // it at the loop's join): a step -- the first four bytes, an EOF flag, a bit -- takes at most four bytes out of the
    // window, refill() then moves the window on by a dword if it can and asks for the next one, on every lane alike.
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(src), base = s0 & ~(uintptr_t)3;
    auto dword_at = [&](const uintptr_t a) -> u32 {              // a: aligned; 0 when the dword holds no byte of the stream
        const bool in = S.active && a < s0 + nin && a + 4 > s0;
        const u32 v = *reinterpret_cast<const u32 *>(in ? a : reinterpret_cast<uintptr_t>(B.squash));
        return in ? v : 0u;
    };
    u64 win = (u64)dword_at(base) | ((u64)dword_at(base + 4) << 32);
    u32 wpos = (u32)(s0 & 3), wn = dword_at(base + 8), ipos = 0;
    uintptr_t naddr = base + 12;
    auto next_byte = [&]() -> u32 {
        const u32 c = ipos < nin ? (u32)(win >> (8u * wpos)) & 255u : 0u;
        const u32 adv = ipos < nin ? 1u : 0u;
        ipos += adv; wpos += adv;
        return c;
    };
    auto refill = [&]() {
        const bool need = wpos >= 4u;
        win = need ? ((win >> 32) | ((u64)wn << 32)) : win;
        wpos = need ? wpos - 4u : wpos;
        wn = dword_at(need ? naddr : naddr - 4);                 // (not needed: the same dword again)
        naddr += need ? 4 : 0;
    };
    u32 low = 1, high = 0xFFFFFFFFu, code = 0, opos = 0, first = 0xFFFFFFFFu;
    bool got_first = false, alive = S.active;
    if (alive) for (int k = 0; k < 4; k++) code = (code << 8) | next_byte();
    refill();
    auto shift_in = [&]() {
        while ((high ^ low) < 0x1000000u) {
            low <<= 8; high = (high << 8) | 255u; low = low ? low : 1u;
            code = (code << 8) | next_byte();
        }
    };
    for (;;) {
        if (alive) {                                             // the EOF flag: decode(0) (decoder.v)
            if (code <= low) { alive = false; high = low; }
            else low += 1;
            shift_in();
        }
        refill();
        s_alive[S.lane] = alive ? 1u : 0u;
        const bool any = __any(alive ? 1 : 0) != 0;
        if (S.lane == 0) *s_any = any ? 1u : 0u;
        lds_barrier();
        if (!any) break;
        u32 c8 = 1;
#pragma unroll 1
        for (int kb = 0; kb < 8; kb++) {
            for (int lv = 0; lv < nlev; lv++) lds_barrier();
            u32 y = 0;
            if (alive) {
                const u32 sq = s_squash[min(max(s_p[(n - 1) * BPW + S.lane] + 2047, 0), 4093)];
                const u32 p16 = sq * 2u + 1u;
                const u32 mid = low + mul_shr16(high - low, p16);
                y = code <= mid ? 1u : 0u;
                high = y ? mid : high;
                low = y ? low : mid + 1;
                shift_in();
            }
            s_y[S.lane] = y;
            refill();
            c8 = (c8 << 1) | y;
            lds_barrier();
        }
        if (alive) {
            const u32 byte = c8 - 256u;
            if (pp && !got_first) { first = byte; got_first = true; }
            else {
                if (opos < S.cap) S.dst[opos] = (u8)byte;
                opos++;
                if (opos > S.cap) alive = false;
            }
        }
    }
    if (S.active) {
        B.out_len[S.blk] = opos;
        B.status[S.blk] = opos > S.cap ? ZPQ_E_OVERFLOW : ZPQ_OK;
        if (B.consumed) B.consumed[S.blk] = ipos;
        if (B.final_code) B.final_code[S.blk] = code;
        if (B.first_byte) B.first_byte[S.blk] = first;
    }
}

__global__ void __launch_bounds__(1024) k_gdec(const DBatch B, const GCfg cfg)
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
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform, and known to the compiler as such: component constants in SGPRs, scalar branches)
    const int n = cfg.n;
    const int wg_slot0 = blockIdx.x * cfg.bpw;
    const int nslots = B.nslots;
    const int slot_id = wg_slot0 + lane;
    const bool lane_on = lane < cfg.bpw && slot_id < nslots;
    u8 *const slot = B.slots + (u64)(lane_on ? slot_id : wg_slot0) * M.slot_bytes;
    const int wg_slots = min(cfg.bpw, nslots - wg_slot0);

    for (int base = wg_slot0; base < B.nblocks; base += nslots) {
        const int blk = base + lane;
        const bool active = lane_on && blk < B.nblocks;
        const int nact = min(wg_slots, B.nblocks - base);
        init_slots(B, M, n, wg_slot0, nact, tid, nthr);
        __threadfence();
        __syncthreads();
        DStage S;
        S.B = &B; S.M = &M; S.lds = lds; S.cfg = &cfg; S.ci = wave; S.lane = lane; S.n = n; S.active = active; S.slot = slot; S.blk = blk;
        S.src = active ? B.in + B.in_off[blk] : B.in;
        S.nin = active ? (u32)(B.in_off[blk + 1] - B.in_off[blk]) : 0u;
        S.dst = active ? B.out + B.out_off[blk] : B.out;
        S.cap = active ? (u32)(B.out_off[blk + 1] - B.out_off[blk]) : 0u;
        S.prof[0] = S.prof[1] = S.prof[2] = S.prof[3] = 0;
        if (wave < n) {
            switch (M.comp[wave].type) {                         // uniform per wave
            case ZT_CONST: comp_dec<ZT_CONST>(S); break;
            case ZT_CM: comp_dec<ZT_CM>(S); break;
            case ZT_ICM: comp_dec<ZT_ICM>(S); break;
            case ZT_MATCH: comp_dec<ZT_MATCH>(S); break;
            case ZT_AVG: comp_dec<ZT_AVG>(S); break;
            case ZT_MIX2: comp_dec<ZT_MIX2>(S); break;
            case ZT_MIX: comp_dec<ZT_MIX>(S); break;
            case ZT_ISSE: comp_dec<ZT_ISSE>(S); break;
            default: comp_dec<ZT_SSE>(S); break;
            }
        } else coder_dec(S);
#ifdef ZPG_PROF
        if (blockIdx.x == 0 && lane == 0 && wave < n)
            printf("gdec wave %d type %d level %d: loads %llu, levels %llu, bit %llu, train %llu cycles\n", wave, M.comp[wave].type, (int)cfg.level[wave],
                   (unsigned long long)S.prof[0], (unsigned long long)S.prof[1], (unsigned long long)S.prof[2], (unsigned long long)S.prof[3]);
#endif
        __syncthreads();
    }
}

}  // namespace zpqg

// ------------------------------------------------------------------ host side
// The pipeline takes a model when every component is one of the nine types, names only EARLIER components as inputs, the
// HCOMP program is the shipped hash chain with a hash per component, and the rings fit the LDS.
static bool gpipe_cfg(const DModel *M, zpqg::GCfg *cfg, size_t *lds_bytes, size_t *dec_lds = nullptr)
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
        case ZT_MATCH: if (c.cm_len < 1 || c.ht_len < 2) return false; break;   // (one-byte buffer: the predicted byte IS the coded byte)
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
    cfg->n = n; cfg->ring = total; cfg->hashes = hashes;
    int nsse = 0, nlev = 1;
    for (int i = 0; i < n; i++) {                                // (inputs are earlier components: one pass)
        const DComp &c = M->comp[i];
        int lv = 0;
        auto dep = [&](int j) { if (cfg->level[j] + 1 > lv) lv = cfg->level[j] + 1; };
        switch (c.type) {
        case ZT_AVG: dep(c.a); dep(c.b); break;
        case ZT_MIX2: dep(c.j); dep(c.k); break;
        case ZT_MIX: for (int l = 0; l < c.limit; l++) dep(c.b + l); break;
        case ZT_ISSE: dep(c.b); break;
        case ZT_SSE: dep(c.b); cfg->dslot[i] = (uint8_t)nsse++; break;
        default: break;
        }
        cfg->level[i] = (uint8_t)lv;
        if (lv + 1 > nlev) nlev = lv + 1;
    }
    cfg->nlevels = nlev;
    if (dec_lds) *dec_lds = (size_t)zpqg::D_SSE + (size_t)nsse * zpqg::D_SSE_BYTES;
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

// Workgroups of `threads` threads and `lds` bytes that one CU keeps resident: what the runtime says for the compiled kernel
// (registers included), never more than the hand calculation `bound`; the hand calculation alone where the query fails.
static int occupancy_wgs(const void *fn, int threads, size_t lds, int bound)
{
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); return bound; }
    return nb < bound ? nb : bound;
}

extern "C" int zpq_gpipe_blocks_per_cu(const DModel *M)
{
    zpqg::GCfg cfg;
    size_t lds = 0;
    if (!gpipe_cfg(M, &cfg, &lds)) return 0;
    const int wgs = (int)((160 * 1024) / lds);                 // workgroups per CU by LDS; waves: n + 1 of the 16 that 128 VGPRs allow
    const int by_waves = 16 / (cfg.n + 1);
    int w = wgs < by_waves ? wgs : by_waves;
    if (w < 1) w = 1;
    const int threads = 64 * (cfg.n + 1);
    const int a = occupancy_wgs((const void *)zpqg::k_gpipe<true>, threads, lds, w), b = occupancy_wgs((const void *)zpqg::k_gpipe<false>, threads, lds, w);
    return (a < b ? a : b) * zpqg::BPW;
}

extern "C" int zpq_launch_gpipe(const DBatch *B, const DModel *hostM, int nslots, hipStream_t stream)
{
    zpqg::GCfg cfg;
    size_t lds = 0;
    if (!gpipe_cfg(hostM, &cfg, &lds)) return ZPQ_E_INTERNAL;
    const int nwg = (nslots + zpqg::BPW - 1) / zpqg::BPW;
    const char *ev = getenv("ZPQ_GPIPE_BATCH");                   // "0": the bit-serial stages (tests compare the two)
    if (ev && atoi(ev) == 0) {
        (void)hipFuncSetAttribute((const void *)zpqg::k_gpipe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(zpqg::k_gpipe<false>, dim3(nwg), dim3(64 * (cfg.n + 1)), lds, stream, *B, cfg);
    } else {
        (void)hipFuncSetAttribute((const void *)zpqg::k_gpipe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(zpqg::k_gpipe<true>, dim3(nwg), dim3(64 * (cfg.n + 1)), lds, stream, *B, cfg);
    }
    return ZPQ_OK;
}

// ---- the decoder (k_gdec).  ZPQ_DEC_GPIPE=0 keeps general models' decoding on zpq_lanes.hip.
extern "C" int zpq_gdec_applies(const DModel *M)
{
    const char *ev = getenv("ZPQ_DEC_GPIPE");
    if (ev && atoi(ev) == 0) return 0;
    zpqg::GCfg cfg;
    size_t lds = 0, dlds = 0;
    return gpipe_cfg(M, &cfg, &lds, &dlds) && dlds <= 160 * 1024 ? 1 : 0;
}

static int gdec_wgs_per_cu(const zpqg::GCfg &cfg, size_t dlds)
{
    const int wgs = (int)((160 * 1024) / dlds);
    const int by_waves = 16 / (cfg.n + 1);                     // 98 VGPRs: four waves per SIMD
    const int w = wgs < by_waves ? wgs : by_waves;
    return w < 1 ? 1 : w;
}

extern "C" int zpq_gdec_blocks_per_cu(const DModel *M)
{
    zpqg::GCfg cfg;
    size_t lds = 0, dlds = 0;
    if (!gpipe_cfg(M, &cfg, &lds, &dlds)) return 0;
    return occupancy_wgs((const void *)zpqg::k_gdec, 64 * (cfg.n + 1), dlds, gdec_wgs_per_cu(cfg, dlds)) * zpqg::BPW;
}

extern "C" int zpq_launch_gdec(const DBatch *B, const DModel *hostM, int nslots, hipStream_t stream)
{
    zpqg::GCfg cfg;
    size_t lds = 0, dlds = 0;
    if (!gpipe_cfg(hostM, &cfg, &lds, &dlds)) return ZPQ_E_INTERNAL;
    // 64 blocks per workgroup.  (Spreading a batch that leaves workgroup slots empty over more workgroups of 32 or 16 lanes
    // was measured: C4b, 16 384 blocks: 2368 ms at 64 lanes, 3056 at 32, 4927 at 16 -- a bit costs a wave the same whatever
    // its lanes; ZPQ_GDEC_BPW keeps the knob.)
    const char *ev = getenv("ZPQ_GDEC_BPW");
    int bpw = zpqg::BPW;
    if (ev && (atoi(ev) == 16 || atoi(ev) == 32 || atoi(ev) == 64)) bpw = atoi(ev);
    cfg.bpw = bpw;
    const int nwg = (nslots + bpw - 1) / bpw;
    (void)hipFuncSetAttribute((const void *)zpqg::k_gdec, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(zpqg::k_gdec, dim3(nwg), dim3(64 * (cfg.n + 1)), dlds, stream, *B, cfg);
    return ZPQ_OK;
}
