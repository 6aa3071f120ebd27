// zpq_lanes.hip -- the all-component-types kernel in wave-parallel form: one ZPAQ block per
// wave, LANE i OWNS COMPONENT i (n <= 64).  Any model the reference accepts: CONST, CM, ICM,
// MATCH, AVG, MIX2, MIX, ISSE, SSE in any order, any ZPAQL program.
//
// Why: a component's table address depends only on contexts (h[i], c8, hmap4), never on
// another component's prediction, so the ~35 dependent HBM loads per bit that the lane-0
// interpreter (zpq_generic.hip) walks one after another are issued here by all lanes at once:
//   A. every lane fetches its own component's state (CM slot, ICM/ISSE row + table entry,
//      MATCH history byte, MIX2 weight), and all lanes together fetch the MIX weight vector (one
//      weight per lane) and the SSE row (32 entries, one per lane) whose addresses also depend
//      only on contexts -- one memory latency for the whole model;
//   B. predictions resolve in dependency order, handed between lanes in registers: components without
//      p-inputs (CONST/CM/ICM/MATCH) first, all together; then AVG/MIX2/ISSE/SSE one by one on
//      their own lane, and MIX as a dot product over ALL lanes (lane l takes weight l) reduced
//      with wavefront shuffles;
//   C. the coder runs on the lane of the last component; the decoded bit comes back through
//      v_readlane (one block per wave => it is wave-uniform);
//   D. every lane trains its own component (MIX again across lanes);
//   E. c8/hmap4 advance; bit-history rows are written back per nibble, ZPAQL runs per byte on
//      lane 0 (shared interpreter, zpq_vm.h).
// Tables squash / compact stretch / ns / dt / dt2k live in LDS, shared by the 4 waves (blocks)
// of a workgroup; component state stays in the block's HBM slot (same layout as
// zpq_generic.hip, which remains the fallback for n > 64 and multi-segment blocks).
//
// Reference behaviour reproduced (file:line under the reference's zpaq/):
//   predict predictor.v:536-668  update predictor.v:672-824  find_ht predictor.v:495-532
//   Encoder encoder.v:48-139     Decoder decoder.v:29-145     ZPAQL zpaql.v:167-954
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_vm.h"
#include "zpq_host.h"

namespace zpql {

typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using zpqvm::Vm;
using zpqvm::vm_run;

constexpr int WAVES = 4;                         // blocks per workgroup
constexpr int L_STRETCH = 0;                     // u32[2048+128]
constexpr int L_SQUASH = (2048 + 128) * 4;       // u16[4096]
constexpr int L_NS = L_SQUASH + 4096 * 2;        // u8[1024]
constexpr int L_DT = L_NS + 1024;                // u32[1024]
constexpr int L_DT2K = L_DT + 4096;              // i16[256]
constexpr int L_HDR = L_DT2K + 512;              // u8[ZPQ_MAX_HDR]
constexpr int L_WAVE = L_HDR + ZPQ_MAX_HDR;      // per-wave scratch follows
constexpr int W_ROW = 0;                         // u8 row[64][16]: the bit-history row each lane works in
constexpr int W_H = 1024;                        // u32 H[<= 256]: the ZPAQL H array when it fits (else it stays in the slot)
constexpr int W_H_WORDS = 256;
constexpr int W_BYTES = 1024 + 4 * W_H_WORDS;
constexpr int LDS_TOTAL = L_WAVE + WAVES * W_BYTES;

// components whose prediction needs other components' predictions, in index order
struct LCfg {
    int32_t n;
    uint64_t depmask;            // bit i: component i consumes other predictions (AVG/MIX2/MIX/ISSE/SSE)
    uint64_t mixmask;            // bit i: component i is a MIX
    int32_t mix_ci[2];           // the first two MIX components: weights fetched with everything else (-1 = none)
    int32_t sse_ci[2];           // the first two SSE components: whole 32-entry row fetched across lanes
    int32_t has_isse, has_mix2;
    int32_t vm_hashes;           // > 0: the program is the shipped hash chain (zpq_vm_hashchain), evaluated in registers
};

// -DZPQ_LANES_PROF: per-phase s_memtime totals of block 0, printed at the end (diagnostic builds only)
#ifdef ZPQ_LANES_PROF
#define LPROF(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const u64 t_ = __builtin_readcyclecounter(); \
                      prof[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define LPROF(k) do { } while (0)
#endif

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
__device__ __forceinline__ i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
// sum over the 64 lanes: four DPP row shifts leave each row's total in its lane 15, then four v_readlane
__device__ __forceinline__ i32 wave_sum(i32 x)
{
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));   // row_shr:1
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));   // row_shr:2
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));   // row_shr:4
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));   // row_shr:8
    return wadd(wadd(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31)),
                wadd(__builtin_amdgcn_readlane(x, 47), __builtin_amdgcn_readlane(x, 63)));
}
__device__ __forceinline__ i32 clamp2k(i32 x) { return min(max(x, -2048), 2047); }
__device__ __forceinline__ i32 clamp512k(i32 x) { return min(max(x, -262144), 262143); }

// VMH: the program is the shipped hash chain (cfg.vm_hashes > 0), evaluated in registers -- such launches do not carry
// the ZPAQL interpreter at all (fewer registers, no spills); any other program runs through it on lane 0.
template <bool DEC, bool VMH>
__global__ void __launch_bounds__(64 * WAVES, 4) k_lanes(const DBatch B, const LCfg cfg)
{
    extern __shared__ __align__(16) u8 lds[];
    const DModel &M = *B.model;
    const int tid = threadIdx.x;
    {
        u32 *st = reinterpret_cast<u32 *>(lds + L_STRETCH);
        for (int i = tid; i < 2048 + 128; i += 64 * WAVES) st[i] = B.stretch_c[i];
        u16 *sq = reinterpret_cast<u16 *>(lds + L_SQUASH);
        for (int i = tid; i < 4096; i += 64 * WAVES) sq[i] = (u16)B.squash[i];
        for (int i = tid; i < 1024; i += 64 * WAVES) lds[L_NS + i] = B.ns[i];
        u32 *dt = reinterpret_cast<u32 *>(lds + L_DT);
        for (int i = tid; i < 1024; i += 64 * WAVES) dt[i] = B.dt[i];
        int16_t *d2 = reinterpret_cast<int16_t *>(lds + L_DT2K);
        for (int i = tid; i < 256; i += 64 * WAVES) d2[i] = B.dt2k[i];
        for (int i = tid; i < M.hdr_len && i < ZPQ_MAX_HDR; i += 64 * WAVES) lds[L_HDR + i] = M.header[i];
    }
    __syncthreads();
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + L_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const u8 *s_ns = lds + L_NS;
    const u32 *s_dt = reinterpret_cast<const u32 *>(lds + L_DT);
    const int16_t *s_dt2k = reinterpret_cast<const int16_t *>(lds + L_DT2K);

    const int lane = tid & 63, wave = tid >> 6;
    u32 *hl = reinterpret_cast<u32 *>(lds + L_WAVE + wave * W_BYTES + W_H);
    u8 *myrow = lds + L_WAVE + wave * W_BYTES + W_ROW + lane * 16;

    const int n = cfg.n;
    const int last = n - 1;
    const bool act = lane < n;
    const int slot_id = blockIdx.x * WAVES + wave;
    const int nslots = B.nslots;
    u8 *slot = B.slots + (u64)slot_id * M.slot_bytes;

    // this lane's component (predictor.v:239-265 as Predictor.init leaves it)
    const DComp &C = M.comp[act ? lane : 0];
    const int type = act ? C.type : 0;
    const i32 ca = C.a, cb = C.b, cc = C.c, climit = C.limit, cj = C.j, ck = C.k, crate = C.rate, cmask = C.mask;
    const u32 cm_len = C.cm_len, ht_len = C.ht_len;
    u32 *cm = reinterpret_cast<u32 *>(slot + C.cm_off);
    u8 *ht = slot + C.ht_off;
    u16 *a16 = reinterpret_cast<u16 *>(slot + C.a16_off);
    const bool hashed = type == ZT_ICM || type == ZT_ISSE;
    const u32 cm_lo = (u32)(uintptr_t)cm, cm_hi = (u32)((u64)(uintptr_t)cm >> 32);
    auto tab_of = [&](int ci) -> u32 * {                          // component ci's u32 table, wave-uniform
        const u64 lo = (u32)__builtin_amdgcn_readlane((i32)cm_lo, ci), hi = (u32)__builtin_amdgcn_readlane((i32)cm_hi, ci);
        return reinterpret_cast<u32 *>((uintptr_t)(lo | (hi << 32)));
    };

    auto squash = [&](i32 d) -> i32 { return s_squash[min(max(wadd(d, 2047), 0), 4093)]; };      // predictor.v:193-202
    auto stretch = [&](i32 pr) -> i32 {                                                           // predictor.v:205-214
        const u32 q = (u32)min(max(pr, 1), 32767);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
    };

    for (int blk = slot_id; slot_id < nslots && blk < B.nblocks; blk += nslots) {
        // ---- Predictor.init + ZPAQL.clear (predictor.v:325-470, zpaql.v:54-95): the wave zeroes
        //      its slot with 16-B stores, then fills the non-zero tables
        {
            uint4 *z4 = reinterpret_cast<uint4 *>(slot);
            const u64 n16 = M.zero_bytes / 16;
            const uint4 zero = make_uint4(0, 0, 0, 0);
            for (u64 i = lane; i < n16; i += 64) z4[i] = zero;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (i32 ci = 0; ci < n; ci++) {
                const DComp &c = M.comp[ci];
                if (c.cm_len && c.cm_fill != ZF_ZERO) {
                    u32 *t = reinterpret_cast<u32 *>(slot + c.cm_off);
                    if (c.cm_fill == ZF_CONST) { for (u32 i = lane; i < c.cm_len; i += 64) t[i] = c.cm_fill_val; }
                    else { const u32 *img = B.img + c.cm_fill_val; for (u32 i = lane; i < c.cm_len; i += 64) t[i] = img[i % c.cm_pat_len]; }
                }
                if (c.a16_len && c.a16_fill) {
                    u16 *t = reinterpret_cast<u16 *>(slot + c.a16_off);
                    for (u32 i = lane; i < c.a16_len; i += 64) t[i] = (u16)c.a16_fill;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const bool h_in_lds = M.hlen <= (u32)W_H_WORDS;
        if (h_in_lds) for (u32 i = lane; i < M.hlen; i += 64) hl[i] = 0;
        for (int k = 0; k < 4; k++) reinterpret_cast<u32 *>(myrow)[k] = 0;

        const u8 *src = B.in + B.in_off[blk];
        const u32 nin = (u32)(B.in_off[blk + 1] - B.in_off[blk]);
        u8 *dst = B.out + B.out_off[blk];
        const u32 cap = (u32)(B.out_off[blk + 1] - B.out_off[blk]);
        i32 status = ZPQ_OK;

        Vm z;
        z.a = z.b = z.c = z.d = 0; z.f = 0; z.pc = 0; z.out = nullptr;
        z.m = slot + M.m_off; z.mlen = M.mlen;
        z.h = h_in_lds ? hl : reinterpret_cast<u32 *>(slot + M.h_off); z.hlen = M.hlen;
        z.r = reinterpret_cast<u32 *>(slot + M.r_off);
        z.hdr = lds + L_HDR; z.hdr_len = M.hdr_len; z.hbegin = M.hbegin; z.hend = M.hend;

        // per-lane component state
        u32 hctx = 0, cxt = 0, v0 = 0, v1 = 0, st = 0;
        i32 pown = 0;                                         // Predictor.p starts at 0 (predictor.v:326)
        u8 *raddr = ht;
        bool row_live = false;
        // MATCH: a = len, b = offset, c = predicted bit, cxt = bit position, limit = buffer position.
        // Quirk: init leaves sizebits/bufbits in a/b (predictor.v:372-373,566-572).
        i32 ma = (type == ZT_MATCH) ? ca : 0, mb = (type == ZT_MATCH) ? cb : 0, mc = 0, mlimit = 0;
        u32 mcxt = 0, mcur = 0;                               // mcur mirrors ht[limit], the byte being shifted in
        // SSE keeps both table entries it interpolated between
        u32 sse_idx = 0, sse_i0 = 0, sse_v0 = 0, sse_v1 = 0;
        bool sse_ok = false;

        u32 c8 = 1, hmap4 = 1, vm_prev = 0;
        u32 low = 1, high = 0xFFFFFFFFu, code = 0, opos = 0, ipos = 0, first = 0xFFFFFFFFu;
        bool got_first = false;
        if (DEC && lane == last)
            for (int k = 0; k < 4; k++) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
        const u32 total = DEC ? 0xFFFFFFFFu : nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u);

#ifdef ZPQ_LANES_PROF
        u64 prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u64 tprev = __builtin_readcyclecounter();
#endif
        for (u32 bi = 0; bi < total; bi++) {
            u32 ch = 0;
            if (!DEC) {
                if (B.flags & ZPQ_FLAG_PP) ch = (bi == 0) ? 0u : src[bi - 1];
                else ch = src[bi];
            }
            // ---- EOF flag (encoder.v:108 / decoder.v:128)
            i32 eof = 0;
            if (lane == last) {
                if (!DEC) low += 1;
                else { if (code <= low) { eof = 1; high = low; } else low += 1; }
                while ((high ^ low) < 0x1000000u) {
                    if (!DEC) { if (opos < cap) dst[opos] = (u8)(high >> 24); opos++; }
                    low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
                    if (DEC) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
                }
            }
            if (DEC) { eof = __builtin_amdgcn_readlane(eof, last); if (eof) break; }

            for (int bit = 7; bit >= 0; bit--) {
                const bool nib = (c8 == 1) || ((c8 & 0xf0u) == 16u);
                const u32 slotn = hmap4 & 15u;
                LPROF(0);
                // ================= A. every lane fetches its component's state =================
                if (type == ZT_CM) {
                    cxt = hctx ^ hmap4;
                    v0 = cm[(i32)cxt & (i32)(cm_len - 1)];
                } else if (hashed) {
                    if (nib) {                                   // find_ht (predictor.v:495-532)
                        const u32 cx = hctx + 16u * c8;
                        const u32 chk = (cx >> (ca + 2)) & 255u;
                        const u32 h0 = (cx * 16u) & (ht_len - 16u);
                        u8 *pa = ht + h0, *pb = ht + (h0 ^ 16u), *pc = ht + (h0 ^ 32u);
                        const u32x4 A = *reinterpret_cast<const u32x4 *>(pa);
                        const u32x4 Bq = *reinterpret_cast<const u32x4 *>(pb);
                        const u32x4 Cq = *reinterpret_cast<const u32x4 *>(pc);
                        const bool ma_ = (A.x & 255u) == chk, mb_ = (Bq.x & 255u) == chk, mc_ = (Cq.x & 255u) == chk;
                        const u32 qa = (A.x >> 8) & 255u, qb = (Bq.x >> 8) & 255u, qc = (Cq.x >> 8) & 255u;
                        const bool va = qa <= qb && qa <= qc, vb = qb < qc;
                        const bool hit = ma_ || mb_ || mc_;
                        const bool ua = ma_ || (!hit && va);
                        const bool ub = !ua && (mb_ || (!hit && vb));
                        raddr = ua ? pa : (ub ? pb : pc);
                        const u32x4 Rr = ua ? A : (ub ? Bq : Cq);
                        u32 *rw = reinterpret_cast<u32 *>(myrow);
                        rw[0] = hit ? Rr.x : chk; rw[1] = hit ? Rr.y : 0u; rw[2] = hit ? Rr.z : 0u; rw[3] = hit ? Rr.w : 0u;
                        row_live = true;
                    }
                    st = myrow[slotn];
                    if (type == ZT_ICM) v0 = cm[st];
                    else { const uint2 w = *reinterpret_cast<const uint2 *>(cm + st * 2); v0 = w.x; v1 = w.y; }
                } else if (type == ZT_MATCH) {
                    if (ma != 0) v0 = ht[wsub(mlimit, mb) & (i32)(ht_len - 1)];
                } else if (type == ZT_MIX2) {
                    cxt = (hctx + (c8 & (u32)cmask)) & (u32)(cc - 1);
                    v0 = a16[cxt];
                } else if (type == ZT_MIX) {
                    cxt = (u32)(wadd((i32)hctx, (i32)c8 & cmask) & (cc - 1));
                } else if (type == ZT_SSE) {
                    cxt = (hctx + c8) * 32u;
                }
                // MIX weights (owner lane (idx + l) mod 64 takes weight l) and SSE rows: addresses known now
                u32 pw0 = 0, pw1 = 0, sr0 = 0, sr1 = 0;
                u32 *pa0 = nullptr, *pa1 = nullptr;
                i32 pj0 = -1, pj1 = -1;
                if (cfg.mix_ci[0] >= 0) {
                    const int ci = cfg.mix_ci[0];
                    const i32 j = __builtin_amdgcn_readlane(cb, ci), m = __builtin_amdgcn_readlane(climit, ci);
                    const i32 idx = wmul((i32)__builtin_amdgcn_readlane((i32)cxt, ci), m);
                    const i32 l = (lane - idx) & 63;
                    if (l < m && (j + l) < n) { pj0 = j + l; pa0 = tab_of(ci) + (idx + l); pw0 = *pa0; }
                }
                if (cfg.mix_ci[1] >= 0) {
                    const int ci = cfg.mix_ci[1];
                    const i32 j = __builtin_amdgcn_readlane(cb, ci), m = __builtin_amdgcn_readlane(climit, ci);
                    const i32 idx = wmul((i32)__builtin_amdgcn_readlane((i32)cxt, ci), m);
                    const i32 l = (lane - idx) & 63;
                    if (l < m && (j + l) < n) { pj1 = j + l; pa1 = tab_of(ci) + (idx + l); pw1 = *pa1; }
                }
                if (cfg.sse_ci[0] >= 0) {
                    const int ci = cfg.sse_ci[0];
                    const i32 il = wadd((i32)__builtin_amdgcn_readlane((i32)cxt, ci), lane);
                    if (lane < 32 && il >= 0 && il < (i32)__builtin_amdgcn_readlane((i32)cm_len, ci)) sr0 = tab_of(ci)[il];
                }
                if (cfg.sse_ci[1] >= 0) {
                    const int ci = cfg.sse_ci[1];
                    const i32 il = wadd((i32)__builtin_amdgcn_readlane((i32)cxt, ci), lane);
                    if (lane < 32 && il >= 0 && il < (i32)__builtin_amdgcn_readlane((i32)cm_len, ci)) sr1 = tab_of(ci)[il];
                }
                LPROF(1);
                // ================= B. predictions in dependency order =================
                // Predictions travel between lanes in registers (v_readlane with the consumer's wave-uniform
                // input index; one bpermute for MIX's input vector) -- no LDS round trips in the chain.
                // An input index >= the consumer's own index names a prediction not made yet this bit: the
                // reference reads last bit's value there (predictor.v:536-668 walks i = 0..n-1 over one
                // persistent p[]), so every lane keeps its previous prediction in pprev.
                const i32 pprev = pown;
                {
                    // CONST / CM / ICM / MATCH / leftovers resolve together: one stretch() for all lanes instead of
                    // one per type branch (predictor.v:546-574)
                    const bool t_cm = type == ZT_CM, t_icm = type == ZT_ICM, t_match = type == ZT_MATCH;
                    const bool live_match = t_match && ma != 0;
                    mc = live_match ? (i32)((v0 >> (7u - mcxt)) & 1u) : mc;
                    const i32 mterm = (s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767;
                    const i32 sin = t_cm ? (i32)(v0 >> 17) : (t_icm ? (i32)(v0 >> 8) : mterm);
                    const i32 stv = stretch(sin);
                    const i32 val = (t_cm || t_icm || live_match) ? stv : (type == ZT_CONST ? (ca - 128) * 16 : 0);
                    const bool indep = type <= ZT_MATCH || type > ZT_SSE;          // NONE(0), CONST, CM, ICM, MATCH, unknown
                    pown = indep ? val : pown;
                }
                i32 pin0 = 0, pin1 = 0;
                for (u64 dm = cfg.depmask; dm != 0; dm &= dm - 1) {
                    const int ci = __builtin_ctzll(dm);
                    const int ty = __builtin_amdgcn_readlane(type, ci);
                    auto inp = [&](int x) -> i32 {                 // x wave-uniform, < n
                        return x < ci ? __builtin_amdgcn_readlane(pown, x) : __builtin_amdgcn_readlane(pprev, x);
                    };
                    if (ty == ZT_MIX) {
                        // p = clamp2k(sum_l (w[l] >> 8) * p[j+l] >> 8) (predictor.v:600-614)
                        const i32 merged = lane < ci ? pown : pprev;
                        i32 part;
                        if (ci == cfg.mix_ci[0]) {
                            const i32 t = __shfl(merged, pj0 & 63);
                            pin0 = pj0 >= 0 ? t : 0;
                            part = wmul((i32)pw0 >> 8, pin0);
                        } else if (ci == cfg.mix_ci[1]) {
                            const i32 t = __shfl(merged, pj1 & 63);
                            pin1 = pj1 >= 0 ? t : 0;
                            part = wmul((i32)pw1 >> 8, pin1);
                        } else {                                   // third and later MIX: weights fetched here
                            const i32 j = __builtin_amdgcn_readlane(cb, ci), m = __builtin_amdgcn_readlane(climit, ci);
                            const i32 idx = wmul((i32)__builtin_amdgcn_readlane((i32)cxt, ci), m);
                            const i32 l = (lane - idx) & 63;
                            const bool mine = l < m && (j + l) < n;
                            const i32 t = __shfl(merged, (j + l) & 63);
                            part = mine ? wmul((i32)tab_of(ci)[idx + l] >> 8, t) : 0;
                        }
                        const i32 sum = wave_sum(part);
                        if (lane == ci) pown = clamp2k(sum >> 8);
                    } else if (ty == ZT_AVG) {                     // predictor.v:590-594
                        const i32 xa = __builtin_amdgcn_readlane(ca, ci), xb = __builtin_amdgcn_readlane(cb, ci);
                        const bool ok = xa < n && xb < n;
                        const i32 ia = ok ? inp(xa) : 0, ib = ok ? inp(xb) : 0;
                        if (lane == ci) pown = ok ? (wadd(wmul(ia, cc), wmul(ib, 256 - cc)) >> 8) : 0;
                    } else if (ty == ZT_MIX2) {                    // predictor.v:595-599
                        const i32 xj = __builtin_amdgcn_readlane(cj, ci), xk = __builtin_amdgcn_readlane(ck, ci);
                        const bool ok = xj < n && xk < n;
                        const i32 ij = ok ? inp(xj) : 0, ik = ok ? inp(xk) : 0;
                        if (lane == ci) { const i32 w = (i32)v0; pown = ok ? clamp2k(wadd(wmul(w, ij), wmul(65536 - w, ik)) >> 16) : 0; }
                    } else if (ty == ZT_ISSE) {                    // predictor.v:615-631
                        const i32 xb = __builtin_amdgcn_readlane(cb, ci);
                        const bool ok = xb < n;
                        const i32 ib = ok ? inp(xb) : 0;
                        if (lane == ci) {
                            const i32 w0 = (i32)v0, w1 = (i32)v1;
                            pown = ok ? clamp2k(wadd(wmul(w0, ib), wmul(w1, 64)) >> 16) : clamp2k(w1 >> 10);
                        }
                    } else {                                       // SSE (predictor.v:632-659)
                        const i32 xb = __builtin_amdgcn_readlane(cb, ci);
                        i32 pq = 992;
                        if (xb < n) pq = wadd(inp(xb), 992);
                        pq = min(max(pq, 0), 1983);
                        const i32 wt = pq & 63;
                        pq >>= 6;
                        const bool pre0 = ci == cfg.sse_ci[0], pre1 = ci == cfg.sse_ci[1];
                        // prefetched row: entry e sits in lane e
                        const u32 srow = pre0 ? sr0 : sr1;
                        const u32 e0 = (u32)__builtin_amdgcn_readlane((i32)srow, pq), e1 = (u32)__builtin_amdgcn_readlane((i32)srow, pq + 1);
                        if (lane == ci) {
                            const i32 idx = wadd((i32)cxt, pq), idx2 = wadd(idx, 1);
                            sse_ok = idx >= 0 && idx2 < (i32)cm_len;
                            if (sse_ok) {
                                sse_i0 = (u32)idx;
                                if (pre0 || pre1) { sse_v0 = e0; sse_v1 = e1; }
                                else { sse_v0 = cm[idx]; sse_v1 = cm[idx2]; }
                                pown = stretch(wadd(wmul((i32)(sse_v0 >> 10), 64 - wt), wmul((i32)(sse_v1 >> 10), wt)) >> 13);
                            } else pown = 0;
                            sse_idx = (u32)idx + (u32)(wt >> 5);
                        }
                    }
                }
                LPROF(2);
                // ================= C. code the bit on the last component's lane =================
                const i32 sqown = squash(pown);
                i32 y = DEC ? 0 : (i32)((ch >> bit) & 1u);
                if (lane == last) {
                    const u32 p16 = (n == 0 ? 16384u : (u32)sqown) * 2u + 1u;
                    const u32 mid = low + (u32)(((u64)(high - low) * p16) >> 16);
                    if (DEC) y = code <= mid ? 1 : 0;
                    if (y) high = mid; else low = mid + 1;
                    while ((high ^ low) < 0x1000000u) {
                        if (!DEC) { if (opos < cap) dst[opos] = (u8)(high >> 24); opos++; }
                        low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
                        if (DEC) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
                    }
                }
                if (DEC) y = __builtin_amdgcn_readlane(y, last);
                LPROF(3);
                // ================= D. every lane trains its component =================
                const i32 t32767 = y ? 32767 : 0;
                if (hashed) myrow[slotn] = s_ns[st * 4 + y];     // next bit-history state (predictor.v:704,790)
                // training reads the finished predictions of this bit (predictor.v:672-824 runs after predict)
                const i32 fin_b = cfg.has_isse ? __shfl(pown, cb & 63) : 0;
                const i32 fin_j = cfg.has_mix2 ? __shfl(pown, cj & 63) : 0;
                const i32 fin_k = cfg.has_mix2 ? __shfl(pown, ck & 63) : 0;
                if (type == ZT_CM) {                               // predictor.v:681-700
                    const i32 idx = (i32)cxt & (i32)(cm_len - 1);
                    const i32 count = (i32)(v0 & 0x3ffu);
                    const i32 err = t32767 - (i32)(v0 >> 17);
                    const i32 upd = wmul(err, (i32)s_dt[count]) & -1024;
                    cm[idx] = (u32)wadd(wadd((i32)v0, upd), count < climit ? 1 : 0);
                } else if (type == ZT_ICM) {                       // predictor.v:701-709 (row state: below, with ISSE)
                    cm[st] = (u32)wadd((i32)v0, (t32767 - (i32)(v0 >> 8)) >> 2);
                } else if (type == ZT_ISSE) {                      // predictor.v:776-791
                    const i32 err = t32767 - sqown;
                    if (cb < n) {
                        const i32 w0 = clamp512k(wadd((i32)v0, wadd(wmul(err, fin_b), 1 << 12) >> 13));
                        const i32 w1 = clamp512k(wadd((i32)v1, (err + 16) >> 5));
                        *reinterpret_cast<uint2 *>(cm + st * 2) = make_uint2((u32)w0, (u32)w1);
                    }
                } else if (type == ZT_MATCH) {                     // predictor.v:710-741
                    const i32 mask = (i32)(ht_len - 1);
                    if (mc != y) ma = 0;
                    const i32 idx = mlimit & mask;
                    mcur = ((mcur << 1) | (u32)y) & 255u;
                    ht[idx] = (u8)mcur;
                    mcxt++;
                    if (mcxt >= 8) {
                        mcxt = 0;
                        mlimit = wadd(mlimit, 1) & mask;
                        mcur = ht[mlimit];
                        const i32 ci = (i32)hctx & (i32)(cm_len - 1);
                        if (ma == 0) {
                            mb = wsub(mlimit, (i32)cm[ci]);
                            if ((mb & mask) != 0) {
                                while (ma < 255) {
                                    const i32 i1 = wsub(wsub(mlimit, ma), 1) & mask;
                                    const i32 i2 = wsub(wsub(wsub(mlimit, ma), mb), 1) & mask;
                                    if (ht[i1] != ht[i2]) break;
                                    ma++;
                                }
                            }
                        } else if (ma < 255) ma++;
                        cm[ci] = (u32)mlimit;
                    }
                } else if (type == ZT_MIX2) {                      // predictor.v:744-762
                    const i32 err = wmul(t32767 - sqown, crate) >> 5;
                    if (cj < n && ck < n) {
                        i32 w = wadd((i32)v0, wadd(wmul(err, wsub(fin_j, fin_k)), 1 << 12) >> 13);
                        w = min(max(w, 0), 65535);
                        a16[cxt] = (u16)w;
                    }
                } else if (type == ZT_SSE) {                       // predictor.v:792-802
                    const i32 idx = (i32)sse_idx & (i32)(cm_len - 1);
                    // the trained entry is one of the two just interpolated whenever the index was in range
                    u32 v;
                    if (sse_ok && (u32)idx == sse_i0) v = sse_v0;
                    else if (sse_ok && (u32)idx == sse_i0 + 1u) v = sse_v1;
                    else v = cm[idx];
                    const i32 err = t32767 - (i32)(v >> 17);
                    const i32 count = (i32)v & 1023;
                    if (count < climit) v = (u32)wadd(wadd((i32)v, wadd(wmul(err, climit - count), 1 << 12) >> 13), 1);
                    cm[idx] = v;
                }
                for (u64 dm = cfg.mixmask; dm != 0; dm &= dm - 1) {   // MIX: the owner lane trains its weight (predictor.v:763-775)
                    const int ci = __builtin_ctzll(dm);
                    const i32 err = __builtin_amdgcn_readlane(wmul(t32767 - sqown, crate) >> 4, ci);
                    if (ci == cfg.mix_ci[0]) {
                        const i32 fin = __shfl(pown, pj0 & 63);
                        if (pj0 >= 0) *pa0 = (u32)clamp512k(wadd((i32)pw0, wadd(wmul(err, fin), 1 << 12) >> 13));
                    } else if (ci == cfg.mix_ci[1]) {
                        const i32 fin = __shfl(pown, pj1 & 63);
                        if (pj1 >= 0) *pa1 = (u32)clamp512k(wadd((i32)pw1, wadd(wmul(err, fin), 1 << 12) >> 13));
                    } else {
                        const i32 jj = __builtin_amdgcn_readlane(cb, ci), m = __builtin_amdgcn_readlane(climit, ci);
                        const i32 idx = wmul((i32)__builtin_amdgcn_readlane((i32)cxt, ci), m);
                        const i32 l = (lane - idx) & 63;                // same owner lane as in predict
                        const i32 fin = __shfl(pown, (jj + l) & 63);
                        u32 *wm = tab_of(ci);
                        if (l < m && (jj + l) < n) wm[idx + l] = (u32)clamp512k(wadd((i32)wm[idx + l], wadd(wmul(err, fin), 1 << 12) >> 13));
                    }
                }
                LPROF(4);
                // ================= E. bit context (predictor.v:807-823) =================
                c8 = (c8 << 1) | (u32)y;
                const bool nib_end = (bit & 3) == 0;
                if (nib_end && hashed && row_live) *reinterpret_cast<u32x4 *>(raddr) = *reinterpret_cast<const u32x4 *>(myrow);
                if (c8 >= 256) {
                    /* byte boundary handled below */
                } else if (c8 >= 16 && c8 < 32) hmap4 = ((hmap4 & 0xfu) << 5) | ((u32)y << 4) | 1u;
                else hmap4 = (hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u + (u32)y) & 0xfu);
            }
            LPROF(5);
            const u32 byte = c8 - 256;
            // ---- ZPAQL.run(byte); h[i] = z.h[i] (predictor.v:809-816)
            if (VMH) {
                // H[k] = hash^(k+1)(byte, previous byte), hash: a = (a + *b + 512) * 773 (zpaql.v HASH);
                // every lane walks the chain and keeps its own link
                u32 a = byte, hv = 0;
                for (int k = 0; k < cfg.vm_hashes; k++) { a = (a + vm_prev + 512u) * 773u; hv = (k == lane) ? a : hv; }
                vm_prev = byte;
                if (act) hctx = hv;
            } else {
                if (lane == 0) { if (!vm_run(z, byte)) status = ZPQ_E_VMSTEPS; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (act && (u32)lane < M.hlen) hctx = z.h[lane];
            }
            hmap4 = 1; c8 = 1;
            LPROF(6);

            if (DEC) {
                if ((B.flags & ZPQ_FLAG_PP) && !got_first) { first = byte; got_first = true; }
                else {
                    if (lane == last && opos < cap) dst[opos] = (u8)byte;
                    opos++;
                    if (opos > cap) break;
                }
            }
        }
        if (!DEC && lane == last) {                               // compress(-1) + flush (encoder.v:101-105,130-139)
            high = low;
            while ((high ^ low) < 0x1000000u) {
                if (opos < cap) dst[opos] = (u8)(high >> 24);
                opos++;
                low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
            }
            for (int sft = 24; sft >= 0; sft -= 8) { if (opos < cap) dst[opos] = (u8)(high >> sft); opos++; }
        }
#ifdef ZPQ_LANES_PROF
        if (blk == 0 && lane == 0)
            printf("LPROF %s eof/other=%llu A=%llu B=%llu C=%llu D=%llu E=%llu vm=%llu\n", DEC ? "dec" : "enc",
                   (unsigned long long)prof[0], (unsigned long long)prof[1], (unsigned long long)prof[2], (unsigned long long)prof[3],
                   (unsigned long long)prof[4], (unsigned long long)prof[5], (unsigned long long)prof[6]);
#endif
        const i32 st0 = __builtin_amdgcn_readlane(status, 0);
        if (lane == last) {
            i32 stt = st0;
            if (opos > cap && stt == ZPQ_OK) stt = ZPQ_E_OVERFLOW;
            B.out_len[blk] = opos;
            B.status[blk] = stt;
            if (DEC) {
                if (B.consumed) B.consumed[blk] = ipos;
                if (B.final_code) B.final_code[blk] = code;
                if (B.first_byte) B.first_byte[blk] = first;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}


// =====================================================================================================================
// k_rows: the same kernel with FOUR blocks per wave.  A model with at most 16 components leaves 48 of a wave's 64 lanes
// idle in k_lanes, and the per-bit instruction stream (every component type's code, one after the other) costs the same
// whether one block or four ride on it.  Here a block is a ROW of 16 lanes (a DPP row), lane li of the row owns
// component li; everything that k_lanes hands between lanes with v_readlane (wave-uniform) travels inside the row by
// ds_bpermute / DPP instead.  Model constants (types, input indices, rates) are the same for every block of a batch, so
// the dispatch over component types stays wave-uniform.  Requirements: n <= 16 and the shipped hash-chain program
// (evaluated in registers); anything else runs on k_lanes.
// =====================================================================================================================
constexpr int RL = 16;                           // lanes per block
constexpr int RPW = 64 / RL;                     // blocks per wave

__device__ __forceinline__ i32 row_sum_all(i32 x, int rb)
{
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));   // row_shr:1
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));   // row_shr:2
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));   // row_shr:4
    x = wadd(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));   // row_shr:8
    return __builtin_amdgcn_ds_bpermute((rb + 15) << 2, x);                  // the row's total, to every lane of the row
}

// VMH as in k_lanes: true = the shipped hash chain, evaluated in registers; false = any program, through the shared
// interpreter on lane 0 of every ROW (four interpreters per wave walking the same program on their own blocks' bytes).
template <bool DEC, bool VMH = true>
__global__ void __launch_bounds__(64 * WAVES, VMH ? 4 : 2) k_rows(const DBatch B, const LCfg cfg)
{
    extern __shared__ __align__(16) u8 lds[];
    const DModel &M = *B.model;
    const int tid = threadIdx.x;
    {
        u32 *st = reinterpret_cast<u32 *>(lds + L_STRETCH);
        for (int i = tid; i < 2048 + 128; i += 64 * WAVES) st[i] = B.stretch_c[i];
        u16 *sq = reinterpret_cast<u16 *>(lds + L_SQUASH);
        for (int i = tid; i < 4096; i += 64 * WAVES) sq[i] = (u16)B.squash[i];
        for (int i = tid; i < 1024; i += 64 * WAVES) lds[L_NS + i] = B.ns[i];
        u32 *dt = reinterpret_cast<u32 *>(lds + L_DT);
        for (int i = tid; i < 1024; i += 64 * WAVES) dt[i] = B.dt[i];
        int16_t *d2 = reinterpret_cast<int16_t *>(lds + L_DT2K);
        for (int i = tid; i < 256; i += 64 * WAVES) d2[i] = B.dt2k[i];
        if (!VMH) for (int i = tid; i < M.hdr_len && i < ZPQ_MAX_HDR; i += 64 * WAVES) lds[L_HDR + i] = M.header[i];
    }
    __syncthreads();
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + L_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const u8 *s_ns = lds + L_NS;
    const u32 *s_dt = reinterpret_cast<const u32 *>(lds + L_DT);
    const int16_t *s_dt2k = reinterpret_cast<const int16_t *>(lds + L_DT2K);

    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & (RL - 1), rb = lane & ~(RL - 1), row = lane / RL;
    u8 *myrow = lds + L_WAVE + wave * W_BYTES + W_ROW + lane * 16;

    const int n = cfg.n;
    const int last = n - 1;
    const bool act = li < n;
    const int slot_id = (blockIdx.x * WAVES + wave) * RPW + row;
    const int nslots = B.nslots;
    u8 *slot = B.slots + (u64)slot_id * M.slot_bytes;

    const DComp &C = M.comp[act ? li : 0];
    const int type = act ? C.type : 0;
    const i32 ca = C.a, cb = C.b, cc = C.c, climit = C.limit, cj = C.j, ck = C.k, crate = C.rate, cmask = C.mask;
    const u32 cm_len = C.cm_len, ht_len = C.ht_len;
    u32 *cm = reinterpret_cast<u32 *>(slot + C.cm_off);
    u8 *ht = slot + C.ht_off;
    u16 *a16 = reinterpret_cast<u16 *>(slot + C.a16_off);
    const bool hashed = type == ZT_ICM || type == ZT_ISSE;
    // Model constants of component ci: lane ci of ROW 0 holds them in registers (every row carries the same model), so
    // v_readlane hands them out as scalars -- no scalar memory load of M.comp[ci] per bit.
    const u32 cmo_lo = (u32)C.cm_off, cmo_hi = (u32)(C.cm_off >> 32);
    auto cst = [&](i32 reg, int ci) -> i32 { return __builtin_amdgcn_readlane(reg, ci); };
    auto tab_of = [&](int ci) -> u32 * {                          // this row's block, component ci
        const u64 off = (u64)(u32)cst((i32)cmo_lo, ci) | ((u64)(u32)cst((i32)cmo_hi, ci) << 32);
        return reinterpret_cast<u32 *>(slot + off);
    };
    auto rowget = [&](i32 v, int x) -> i32 { return __builtin_amdgcn_ds_bpermute((rb + (x & (RL - 1))) << 2, v); };  // lane x of my row

    auto squash = [&](i32 d) -> i32 { return s_squash[min(max(wadd(d, 2047), 0), 4093)]; };
    auto stretch = [&](i32 pr) -> i32 {
        const u32 q = (u32)min(max(pr, 1), 32767);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
    };

    for (int blk = slot_id; slot_id < nslots && blk < B.nblocks; blk += nslots) {
        {   // Predictor.init + ZPAQL.clear: the row zeroes its slot, then fills the non-zero tables
            uint4 *z4 = reinterpret_cast<uint4 *>(slot);
            const u64 n16 = M.zero_bytes / 16;
            const uint4 zero = make_uint4(0, 0, 0, 0);
            for (u64 i = li; i < n16; i += RL) z4[i] = zero;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (i32 ci = 0; ci < n; ci++) {
                const DComp &c = M.comp[ci];
                if (c.cm_len && c.cm_fill != ZF_ZERO) {
                    u32 *t = reinterpret_cast<u32 *>(slot + c.cm_off);
                    if (c.cm_fill == ZF_CONST) { for (u32 i = li; i < c.cm_len; i += RL) t[i] = c.cm_fill_val; }
                    else { const u32 *img = B.img + c.cm_fill_val; for (u32 i = li; i < c.cm_len; i += RL) t[i] = img[i % c.cm_pat_len]; }
                }
                if (c.a16_len && c.a16_fill) {
                    u16 *t = reinterpret_cast<u16 *>(slot + c.a16_off);
                    for (u32 i = li; i < c.a16_len; i += RL) t[i] = (u16)c.a16_fill;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        for (int k = 0; k < 4; k++) reinterpret_cast<u32 *>(myrow)[k] = 0;

        const u8 *src = B.in + B.in_off[blk];
        const u32 nin = (u32)(B.in_off[blk + 1] - B.in_off[blk]);
        u8 *dst = B.out + B.out_off[blk];
        const u32 cap = (u32)(B.out_off[blk + 1] - B.out_off[blk]);

        Vm z;                                                      // (!VMH) the block's ZPAQL machine: M, H, R in its slot
        z.a = z.b = z.c = z.d = 0; z.f = 0; z.pc = 0; z.out = nullptr;
        z.m = slot + M.m_off; z.mlen = M.mlen;
        z.h = reinterpret_cast<u32 *>(slot + M.h_off); z.hlen = M.hlen;
        z.r = reinterpret_cast<u32 *>(slot + M.r_off);
        z.hdr = lds + L_HDR; z.hdr_len = M.hdr_len; z.hbegin = M.hbegin; z.hend = M.hend;
        i32 vm_status = ZPQ_OK;

        u32 hctx = 0, cxt = 0, v0 = 0, v1 = 0, st = 0;
        i32 pown = 0;
        u8 *raddr = ht;
        bool row_live = false;
        i32 ma = (type == ZT_MATCH) ? ca : 0, mb = (type == ZT_MATCH) ? cb : 0, mc = 0, mlimit = 0;   // quirk Q17
        u32 mcxt = 0, mcur = 0;
        u32 sse_idx = 0, sse_i0 = 0, sse_v0 = 0, sse_v1 = 0;
        bool sse_ok = false;

        u32 c8 = 1, hmap4 = 1, vm_prev = 0;
        u32 low = 1, high = 0xFFFFFFFFu, code = 0, opos = 0, ipos = 0, first = 0xFFFFFFFFu;
        bool got_first = false;
        if (DEC && li == last)
            for (int k = 0; k < 4; k++) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
        const u32 total = DEC ? 0xFFFFFFFFu : nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u);

        for (u32 bi = 0; bi < total; bi++) {
            u32 ch = 0;
            if (!DEC) {
                if (B.flags & ZPQ_FLAG_PP) ch = (bi == 0) ? 0u : src[bi - 1];
                else ch = src[bi];
            }
            i32 eof = 0;
            if (li == last) {
                if (!DEC) low += 1;
                else { if (code <= low) { eof = 1; high = low; } else low += 1; }
                while ((high ^ low) < 0x1000000u) {
                    if (!DEC) { if (opos < cap) dst[opos] = (u8)(high >> 24); opos++; }
                    low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
                    if (DEC) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
                }
            }
            if (DEC) { eof = rowget(eof, last); if (eof) break; }

            for (int bit = 7; bit >= 0; bit--) {
                const bool nib = (c8 == 1) || ((c8 & 0xf0u) == 16u);
                const u32 slotn = hmap4 & 15u;
                // ================= A. every lane fetches its component's state =================
                if (type == ZT_CM) {
                    cxt = hctx ^ hmap4;
                    v0 = cm[(i32)cxt & (i32)(cm_len - 1)];
                } else if (hashed) {
                    if (nib) {                                   // find_ht (predictor.v:495-532)
                        const u32 cx = hctx + 16u * c8;
                        const u32 chk = (cx >> (ca + 2)) & 255u;
                        const u32 h0 = (cx * 16u) & (ht_len - 16u);
                        u8 *pa = ht + h0, *pb = ht + (h0 ^ 16u), *pc = ht + (h0 ^ 32u);
                        const u32x4 A = *reinterpret_cast<const u32x4 *>(pa);
                        const u32x4 Bq = *reinterpret_cast<const u32x4 *>(pb);
                        const u32x4 Cq = *reinterpret_cast<const u32x4 *>(pc);
                        const bool ma_ = (A.x & 255u) == chk, mb_ = (Bq.x & 255u) == chk, mc_ = (Cq.x & 255u) == chk;
                        const u32 qa = (A.x >> 8) & 255u, qb = (Bq.x >> 8) & 255u, qc = (Cq.x >> 8) & 255u;
                        const bool va = qa <= qb && qa <= qc, vb = qb < qc;
                        const bool hit = ma_ || mb_ || mc_;
                        const bool ua = ma_ || (!hit && va);
                        const bool ub = !ua && (mb_ || (!hit && vb));
                        raddr = ua ? pa : (ub ? pb : pc);
                        u32 *rw = reinterpret_cast<u32 *>(myrow);
                        rw[0] = hit ? (ua ? A.x : (ub ? Bq.x : Cq.x)) : chk;
                        rw[1] = hit ? (ua ? A.y : (ub ? Bq.y : Cq.y)) : 0u;
                        rw[2] = hit ? (ua ? A.z : (ub ? Bq.z : Cq.z)) : 0u;
                        rw[3] = hit ? (ua ? A.w : (ub ? Bq.w : Cq.w)) : 0u;
                        row_live = true;
                    }
                    st = myrow[slotn];
                    if (type == ZT_ICM) v0 = cm[st];
                    else { const uint2 w = *reinterpret_cast<const uint2 *>(cm + st * 2); v0 = w.x; v1 = w.y; }
                } else if (type == ZT_MATCH) {
                    if (ma != 0) v0 = ht[wsub(mlimit, mb) & (i32)(ht_len - 1)];
                } else if (type == ZT_MIX2) {
                    cxt = (hctx + (c8 & (u32)cmask)) & (u32)(cc - 1);
                    v0 = a16[cxt];
                } else if (type == ZT_MIX) {
                    cxt = (u32)(wadd((i32)hctx, (i32)c8 & cmask) & (cc - 1));
                } else if (type == ZT_SSE) {
                    cxt = (hctx + c8) * 32u;
                }
                // MIX weights (owner lane (idx + l) mod 16 of the row takes weight l) and SSE rows (two entries per lane)
                u32 pw0 = 0, pw1 = 0, srA0 = 0, srB0 = 0, srA1 = 0, srB1 = 0;
                u32 *pa0 = nullptr, *pa1 = nullptr;
                i32 pj0 = -1, pj1 = -1;
                if (cfg.mix_ci[0] >= 0) {
                    const int ci = cfg.mix_ci[0];
                    const i32 j = cst(cb, ci), m = cst(climit, ci);
                    const i32 idx = wmul(rowget((i32)cxt, ci), m);
                    const i32 l = (li - idx) & (RL - 1);
                    if (l < m && (j + l) < n) { pj0 = j + l; pa0 = tab_of(ci) + (idx + l); pw0 = *pa0; }
                }
                if (cfg.mix_ci[1] >= 0) {
                    const int ci = cfg.mix_ci[1];
                    const i32 j = cst(cb, ci), m = cst(climit, ci);
                    const i32 idx = wmul(rowget((i32)cxt, ci), m);
                    const i32 l = (li - idx) & (RL - 1);
                    if (l < m && (j + l) < n) { pj1 = j + l; pa1 = tab_of(ci) + (idx + l); pw1 = *pa1; }
                }
                if (cfg.sse_ci[0] >= 0) {
                    const int ci = cfg.sse_ci[0];
                    const i32 base = rowget((i32)cxt, ci), len = cst((i32)cm_len, ci);
                    const i32 ia = wadd(base, li), ib = wadd(base, li + RL);
                    if (ia >= 0 && ia < len) srA0 = tab_of(ci)[ia];
                    if (ib >= 0 && ib < len) srB0 = tab_of(ci)[ib];
                }
                if (cfg.sse_ci[1] >= 0) {
                    const int ci = cfg.sse_ci[1];
                    const i32 base = rowget((i32)cxt, ci), len = cst((i32)cm_len, ci);
                    const i32 ia = wadd(base, li), ib = wadd(base, li + RL);
                    if (ia >= 0 && ia < len) srA1 = tab_of(ci)[ia];
                    if (ib >= 0 && ib < len) srB1 = tab_of(ci)[ib];
                }
                // ================= B. predictions in dependency order =================
                const i32 pprev = pown;
                {
                    const bool t_cm = type == ZT_CM, t_icm = type == ZT_ICM, t_match = type == ZT_MATCH;
                    const bool live_match = t_match && ma != 0;
                    mc = live_match ? (i32)((v0 >> (7u - mcxt)) & 1u) : mc;
                    const i32 mterm = (s_dt2k[ma & 255] * (mc * -2 + 1)) & 32767;
                    const i32 sin = t_cm ? (i32)(v0 >> 17) : (t_icm ? (i32)(v0 >> 8) : mterm);
                    const i32 stv = stretch(sin);
                    const i32 val = (t_cm || t_icm || live_match) ? stv : (type == ZT_CONST ? (ca - 128) * 16 : 0);
                    const bool indep = type <= ZT_MATCH || type > ZT_SSE;
                    pown = indep ? val : pown;
                }
                i32 pin0 = 0, pin1 = 0;
                for (u64 dm = cfg.depmask; dm != 0; dm &= dm - 1) {
                    const int ci = __builtin_ctzll(dm);
                    const int ty = cst(type, ci);
                    auto inp = [&](int x) -> i32 { return x < ci ? rowget(pown, x) : rowget(pprev, x); };   // x uniform, < n
                    if (ty == ZT_MIX) {
                        const i32 merged = li < ci ? pown : pprev;
                        i32 part;
                        if (ci == cfg.mix_ci[0]) {
                            const i32 t = rowget(merged, pj0);
                            pin0 = pj0 >= 0 ? t : 0;
                            part = wmul((i32)pw0 >> 8, pin0);
                        } else if (ci == cfg.mix_ci[1]) {
                            const i32 t = rowget(merged, pj1);
                            pin1 = pj1 >= 0 ? t : 0;
                            part = wmul((i32)pw1 >> 8, pin1);
                        } else {
                            const i32 j = cst(cb, ci), m = cst(climit, ci);
                            const i32 idx = wmul(rowget((i32)cxt, ci), m);
                            const i32 l = (li - idx) & (RL - 1);
                            const bool mine = l < m && (j + l) < n;
                            const i32 t = rowget(merged, j + l);
                            part = mine ? wmul((i32)tab_of(ci)[idx + l] >> 8, t) : 0;
                        }
                        const i32 sum = row_sum_all(part, rb);
                        if (li == ci) pown = clamp2k(sum >> 8);
                    } else if (ty == ZT_AVG) {
                        const i32 xa = cst(ca, ci), xb = cst(cb, ci);
                        const bool ok = xa < n && xb < n;
                        const i32 ia = ok ? inp(xa) : 0, ib = ok ? inp(xb) : 0;
                        if (li == ci) pown = ok ? (wadd(wmul(ia, cc), wmul(ib, 256 - cc)) >> 8) : 0;
                    } else if (ty == ZT_MIX2) {
                        const i32 xj = cst(cj, ci), xk = cst(ck, ci);
                        const bool ok = xj < n && xk < n;
                        const i32 ij = ok ? inp(xj) : 0, ik = ok ? inp(xk) : 0;
                        if (li == ci) { const i32 w = (i32)v0; pown = ok ? clamp2k(wadd(wmul(w, ij), wmul(65536 - w, ik)) >> 16) : 0; }
                    } else if (ty == ZT_ISSE) {
                        const i32 xb = cst(cb, ci);
                        const bool ok = xb < n;
                        const i32 ib = ok ? inp(xb) : 0;
                        if (li == ci) {
                            const i32 w0 = (i32)v0, w1 = (i32)v1;
                            pown = ok ? clamp2k(wadd(wmul(w0, ib), wmul(w1, 64)) >> 16) : clamp2k(w1 >> 10);
                        }
                    } else {                                       // SSE
                        const i32 xb = cst(cb, ci);
                        i32 pq = 992;
                        if (xb < n) pq = wadd(inp(xb), 992);
                        pq = min(max(pq, 0), 1983);
                        const i32 wt = pq & 63;
                        pq >>= 6;
                        const bool pre0 = ci == cfg.sse_ci[0], pre1 = ci == cfg.sse_ci[1];
                        // prefetched row: entry e sits in lane e & 15 of the row, register e >> 4
                        const u32 sA = pre0 ? srA0 : srA1, sB = pre0 ? srB0 : srB1;
                        const u32 e0a = (u32)rowget((i32)sA, pq), e0b = (u32)rowget((i32)sB, pq);
                        const u32 e1a = (u32)rowget((i32)sA, pq + 1), e1b = (u32)rowget((i32)sB, pq + 1);
                        const u32 e0 = (pq & RL) ? e0b : e0a, e1 = ((pq + 1) & RL) ? e1b : e1a;
                        if (li == ci) {
                            const i32 idx = wadd((i32)cxt, pq), idx2 = wadd(idx, 1);
                            sse_ok = idx >= 0 && idx2 < (i32)cm_len;
                            if (sse_ok) {
                                sse_i0 = (u32)idx;
                                if (pre0 || pre1) { sse_v0 = e0; sse_v1 = e1; }
                                else { sse_v0 = cm[idx]; sse_v1 = cm[idx2]; }
                                pown = stretch(wadd(wmul((i32)(sse_v0 >> 10), 64 - wt), wmul((i32)(sse_v1 >> 10), wt)) >> 13);
                            } else pown = 0;
                            sse_idx = (u32)idx + (u32)(wt >> 5);
                        }
                    }
                }
                // ================= C. code the bit on the last component's lane =================
                const i32 sqown = squash(pown);
                i32 y = DEC ? 0 : (i32)((ch >> bit) & 1u);
                if (li == last) {
                    const u32 p16 = (u32)sqown * 2u + 1u;
                    const u32 mid = low + (u32)(((u64)(high - low) * p16) >> 16);
                    if (DEC) y = code <= mid ? 1 : 0;
                    if (y) high = mid; else low = mid + 1;
                    while ((high ^ low) < 0x1000000u) {
                        if (!DEC) { if (opos < cap) dst[opos] = (u8)(high >> 24); opos++; }
                        low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
                        if (DEC) { u32 c = 0; if (ipos < nin) c = src[ipos++]; code = (code << 8) | c; }
                    }
                }
                if (DEC) y = rowget(y, last);
                // ================= D. every lane trains its component =================
                const i32 t32767 = y ? 32767 : 0;
                if (hashed) myrow[slotn] = s_ns[st * 4 + y];
                const i32 fin_b = cfg.has_isse ? rowget(pown, cb) : 0;
                const i32 fin_j = cfg.has_mix2 ? rowget(pown, cj) : 0;
                const i32 fin_k = cfg.has_mix2 ? rowget(pown, ck) : 0;
                if (type == ZT_CM) {
                    const i32 idx = (i32)cxt & (i32)(cm_len - 1);
                    const i32 count = (i32)(v0 & 0x3ffu);
                    const i32 err = t32767 - (i32)(v0 >> 17);
                    const i32 upd = wmul(err, (i32)s_dt[count]) & -1024;
                    cm[idx] = (u32)wadd(wadd((i32)v0, upd), count < climit ? 1 : 0);
                } else if (type == ZT_ICM) {
                    cm[st] = (u32)wadd((i32)v0, (t32767 - (i32)(v0 >> 8)) >> 2);
                } else if (type == ZT_ISSE) {
                    const i32 err = t32767 - sqown;
                    if (cb < n) {
                        const i32 w0 = clamp512k(wadd((i32)v0, wadd(wmul(err, fin_b), 1 << 12) >> 13));
                        const i32 w1 = clamp512k(wadd((i32)v1, (err + 16) >> 5));
                        *reinterpret_cast<uint2 *>(cm + st * 2) = make_uint2((u32)w0, (u32)w1);
                    }
                } else if (type == ZT_MATCH) {
                    const i32 mask = (i32)(ht_len - 1);
                    if (mc != y) ma = 0;
                    const i32 idx = mlimit & mask;
                    mcur = ((mcur << 1) | (u32)y) & 255u;
                    ht[idx] = (u8)mcur;
                    mcxt++;
                    if (mcxt >= 8) {
                        mcxt = 0;
                        mlimit = wadd(mlimit, 1) & mask;
                        mcur = ht[mlimit];
                        const i32 ci = (i32)hctx & (i32)(cm_len - 1);
                        if (ma == 0) {
                            mb = wsub(mlimit, (i32)cm[ci]);
                            if ((mb & mask) != 0) {
                                while (ma < 255) {
                                    const i32 i1 = wsub(wsub(mlimit, ma), 1) & mask;
                                    const i32 i2 = wsub(wsub(wsub(mlimit, ma), mb), 1) & mask;
                                    if (ht[i1] != ht[i2]) break;
                                    ma++;
                                }
                            }
                        } else if (ma < 255) ma++;
                        cm[ci] = (u32)mlimit;
                    }
                } else if (type == ZT_MIX2) {
                    const i32 err = wmul(t32767 - sqown, crate) >> 5;
                    if (cj < n && ck < n) {
                        i32 w = wadd((i32)v0, wadd(wmul(err, wsub(fin_j, fin_k)), 1 << 12) >> 13);
                        w = min(max(w, 0), 65535);
                        a16[cxt] = (u16)w;
                    }
                } else if (type == ZT_SSE) {
                    const i32 idx = (i32)sse_idx & (i32)(cm_len - 1);
                    u32 v;
                    if (sse_ok && (u32)idx == sse_i0) v = sse_v0;
                    else if (sse_ok && (u32)idx == sse_i0 + 1u) v = sse_v1;
                    else v = cm[idx];
                    const i32 err = t32767 - (i32)(v >> 17);
                    const i32 count = (i32)v & 1023;
                    if (count < climit) v = (u32)wadd(wadd((i32)v, wadd(wmul(err, climit - count), 1 << 12) >> 13), 1);
                    cm[idx] = v;
                }
                for (u64 dm = cfg.mixmask; dm != 0; dm &= dm - 1) {   // MIX: the owner lane trains its weight
                    const int ci = __builtin_ctzll(dm);
                    const i32 err = rowget(wmul(t32767 - sqown, crate) >> 4, ci);
                    if (ci == cfg.mix_ci[0]) {
                        const i32 fin = rowget(pown, pj0);
                        if (pj0 >= 0) *pa0 = (u32)clamp512k(wadd((i32)pw0, wadd(wmul(err, fin), 1 << 12) >> 13));
                    } else if (ci == cfg.mix_ci[1]) {
                        const i32 fin = rowget(pown, pj1);
                        if (pj1 >= 0) *pa1 = (u32)clamp512k(wadd((i32)pw1, wadd(wmul(err, fin), 1 << 12) >> 13));
                    } else {
                        const i32 jj = cst(cb, ci), m = cst(climit, ci);
                        const i32 idx = wmul(rowget((i32)cxt, ci), m);
                        const i32 l = (li - idx) & (RL - 1);
                        const i32 fin = rowget(pown, jj + l);
                        u32 *wm = tab_of(ci);
                        if (l < m && (jj + l) < n) wm[idx + l] = (u32)clamp512k(wadd((i32)wm[idx + l], wadd(wmul(err, fin), 1 << 12) >> 13));
                    }
                }
                // ================= E. bit context =================
                c8 = (c8 << 1) | (u32)y;
                const bool nib_end = (bit & 3) == 0;
                if (nib_end && hashed && row_live) *reinterpret_cast<u32x4 *>(raddr) = *reinterpret_cast<const u32x4 *>(myrow);
                if (c8 >= 256) {
                } else if (c8 >= 16 && c8 < 32) hmap4 = ((hmap4 & 0xfu) << 5) | ((u32)y << 4) | 1u;
                else hmap4 = (hmap4 & 0x1f0u) | (((hmap4 & 0xfu) * 2u + (u32)y) & 0xfu);
            }
            const u32 byte = c8 - 256;
            if (VMH) {  // H[k] = hash^(k+1)(byte, previous byte): every lane walks the chain and keeps its own link
                u32 a = byte, hv = 0;
                for (int k = 0; k < cfg.vm_hashes; k++) { a = (a + vm_prev + 512u) * 773u; hv = (k == li) ? a : hv; }
                vm_prev = byte;
                if (act) hctx = hv;
            } else {    // ZPAQL.run(byte); h[i] = z.h[i] (predictor.v:809-816), one interpreter per row
                if (li == 0) { if (!vm_run(z, byte)) vm_status = ZPQ_E_VMSTEPS; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (act && (u32)li < M.hlen) hctx = z.h[li];
            }
            hmap4 = 1; c8 = 1;

            if (DEC) {
                if ((B.flags & ZPQ_FLAG_PP) && !got_first) { first = byte; got_first = true; }
                else {
                    if (li == last && opos < cap) dst[opos] = (u8)byte;
                    opos++;
                    if (opos > cap) break;
                }
            }
        }
        if (!DEC && li == last) {
            high = low;
            while ((high ^ low) < 0x1000000u) {
                if (opos < cap) dst[opos] = (u8)(high >> 24);
                opos++;
                low <<= 8; high = (high << 8) | 255u; if (low == 0) low = 1;
            }
            for (int sft = 24; sft >= 0; sft -= 8) { if (opos < cap) dst[opos] = (u8)(high >> sft); opos++; }
        }
        const i32 vm_st = VMH ? (i32)ZPQ_OK : rowget(vm_status, 0);   // (the interpreter's step cap: lane 0 of the row)
        if (li == last) {
            i32 stt = vm_st;
            if (opos > cap && stt == ZPQ_OK) stt = ZPQ_E_OVERFLOW;
            B.out_len[blk] = opos;
            B.status[blk] = stt;
            if (DEC) {
                if (B.consumed) B.consumed[blk] = ipos;
                if (B.final_code) B.final_code[blk] = code;
                if (B.first_byte) B.first_byte[blk] = first;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace zpql

// ------------------------------------------------------------------ host side
static bool lanes_cfg(const DModel *M, zpql::LCfg *cfg)
{
    if (M->n < 1 || M->n > 64) return false;
    memset(cfg, 0, sizeof *cfg);
    cfg->n = M->n;
    cfg->mix_ci[0] = cfg->mix_ci[1] = cfg->sse_ci[0] = cfg->sse_ci[1] = -1;
    int nmix = 0, nsse = 0;
    for (int i = 0; i < M->n; i++) {
        const int t = M->comp[i].type;
        if (t == ZT_AVG || t == ZT_MIX2 || t == ZT_MIX || t == ZT_ISSE || t == ZT_SSE) cfg->depmask |= 1ull << i;
        if (t == ZT_MIX) { cfg->mixmask |= 1ull << i; if (nmix < 2) cfg->mix_ci[nmix++] = i; }
        if (t == ZT_SSE && nsse < 2) cfg->sse_ci[nsse++] = i;
        if (t == ZT_ISSE) cfg->has_isse = 1;
        if (t == ZT_MIX2) cfg->has_mix2 = 1;
    }
    cfg->vm_hashes = zpq_vm_hashchain(M);
    return true;
}

extern "C" int zpq_lanes_supported(const DModel *M)
{
    zpql::LCfg cfg;
    return lanes_cfg(M, &cfg) ? 1 : 0;
}

// four blocks per wave (k_rows) when the model fits a 16-lane row; its program is evaluated in registers when it is the
// shipped hash chain and runs through the interpreter on the row's first lane otherwise (round 3)
static bool rows_ok(const zpql::LCfg &cfg)
{
    const char *ev = getenv("ZPQ_LANES_ROWS");                    // tuning / test knob: "0" keeps one block per wave
    if (ev && ev[0] == '0') return false;                         // ("h": only hash-chain programs, as in round 2)
    if (ev && ev[0] == 'h' && cfg.vm_hashes <= 0) return false;
    return cfg.n <= zpql::RL;
}

extern "C" int zpq_lanes_blocks_per_cu(const DModel *M)
{
    zpql::LCfg cfg;
    const bool rows = lanes_cfg(M, &cfg) && rows_ok(cfg);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rows ? (cfg.vm_hashes > 0 ? (const void *)zpql::k_rows<false, true> : (const void *)zpql::k_rows<false, false>) : (const void *)zpql::k_lanes<false, false>,
                                                     64 * zpql::WAVES, zpql::LDS_TOTAL) != hipSuccess || nb < 1)
        nb = 2;
    return nb * zpql::WAVES * (rows ? zpql::RPW : 1);
}

extern "C" const char *zpq_lanes_kernel_name(const DModel *M, int decode)
{
    zpql::LCfg cfg;
    const bool rows = lanes_cfg(M, &cfg) && rows_ok(cfg);
    return rows ? (decode ? "k_rows<decode>" : "k_rows<encode>") : (decode ? "k_lanes<decode>" : "k_lanes<encode>");
}

extern "C" int zpq_launch_lanes(const DBatch *B, const DModel *hostM, int decode, int nslots, hipStream_t stream)
{
    zpql::LCfg cfg;
    if (!lanes_cfg(hostM, &cfg)) return ZPQ_E_INTERNAL;
    if (rows_ok(cfg)) {
        const int per_wg = zpql::WAVES * zpql::RPW;
        const dim3 g((nslots + per_wg - 1) / per_wg), t(64 * zpql::WAVES);
        if (cfg.vm_hashes > 0) {
            if (decode) hipLaunchKernelGGL((zpql::k_rows<true, true>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
            else hipLaunchKernelGGL((zpql::k_rows<false, true>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
        } else {
            if (decode) hipLaunchKernelGGL((zpql::k_rows<true, false>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
            else hipLaunchKernelGGL((zpql::k_rows<false, false>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
        }
        return ZPQ_OK;
    }
    const int grid = (nslots + zpql::WAVES - 1) / zpql::WAVES;
    const dim3 g(grid), t(64 * zpql::WAVES);
    if (cfg.vm_hashes > 0) {
        if (decode) hipLaunchKernelGGL((zpql::k_lanes<true, true>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
        else hipLaunchKernelGGL((zpql::k_lanes<false, true>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
    } else {
        if (decode) hipLaunchKernelGGL((zpql::k_lanes<true, false>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
        else hipLaunchKernelGGL((zpql::k_lanes<false, false>), g, t, zpql::LDS_TOTAL, stream, *B, cfg);
    }
    return ZPQ_OK;
}
