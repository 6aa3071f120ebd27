// zpq_dpipe.hip -- the DECODER of the chain models (ICM + ISSEs: levels 1-3, levels.v:53-181) as cooperating WAVES.
//
// Decoding feeds every decoded bit back into every component (decoder.v:122-145 -> predictor.v:536-824), so unlike the
// encoder (zpq_pipe.hip) no stage can run bytes ahead.  zpq_chain.hip therefore keeps a block inside one wave (lane =
// component), and all lanes issue the union of every role's instructions: ~185 per coded bit and wave of 8 blocks.
//
// Here the roles are separated by WAVE and meet once per coded bit:
//   * wave c < NCH owns component c of every block of the workgroup.  A block is a PAIR of neighbouring lanes: lane
//     2b assumes that the bit being decoded is 0, lane 2b+1 that it is 1.  While the coder works on bit t, each lane
//     does, for ITS outcome of bit t, everything that follows the bit: trains the component's table entry, looks the
//     next bit-history state up (at a nibble boundary: resolves the next nibble's row, requested one bit earlier for
//     its outcome), fetches that state's entry (forwarding the trained one when the state repeats) -- and publishes
//     the entry bit t+1 will be predicted from: stretch(p) for the ICM, the weight pair for an ISSE.
//   * the last wave is the arithmetic decoder (decoder.v:73-145), lane = block.  For bit t it picks, per component,
//     the published entry of the lane that assumed the right bit t-1, runs the chain p0 -> p1 -> ... (one multiply-add,
//     shift and clamp per link: predictor.v:615-631), squash, decodes the bit and publishes it.
//   * a component wave needs its own input and output prediction to train (predictor.v:776-791); it recomputes the
//     links below it from the same published entries instead of waiting for another hand-off.
//   * ONE workgroup barrier per bit (LDS-only wait in front of it, as in zpq_pipe.hip); the entries and the bits are
//     double-buffered by bit parity, so a writer never meets a reader of the previous bit.
// When the bit is known the lane that assumed it commits (table entry, row), its neighbour takes its registers over
// with one DPP move per register (quad_perm [1,0,3,2]).
// Decoded bytes are identical to zpq_chain.hip's, zpq_generic.hip's and the CPU oracle's.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_chain_cfg.h"

namespace zpqd {

using namespace zpqc;

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 clamp2k(i32 x) { return min(max(x, -2048), 2047); }
__device__ __forceinline__ i32 clamp512k(i32 x) { return min(max(x, -262144), 262143); }
__device__ __forceinline__ uint32_t mul_shr16(uint32_t range, uint32_t p16)   // see zpq_chain.hip
{
    return (uint32_t)__umul24(range >> 16, p16) + ((uint32_t)__umul24(range & 0xFFFFu, p16) >> 16);
}
// LDS traffic of this wave done, then the workgroup barrier (no wait for global memory: row requests stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef ZPD_PROF
// Timing build (tools/variant.sh): per wave of workgroup 0, cycles spent working and waiting per bit position.
// prof[wave][kb][0] = cycles between leaving a barrier and reaching the next, [1] = cycles inside the barrier.
__device__ unsigned long long zpd_prof[16][8][2];
struct Prof {
    unsigned long long t_rel, acc[8][2];
    bool on;
    __device__ void init(bool o) { on = o; for (int k = 0; k < 8; k++) { acc[k][0] = 0; acc[k][1] = 0; } t_rel = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void barrier(const int kb)
    {
        const unsigned long long t0 = __builtin_readcyclecounter();
        lds_barrier();
        const unsigned long long t1 = __builtin_readcyclecounter();
        acc[kb][0] += t0 - t_rel; acc[kb][1] += t1 - t0; t_rel = t1;
    }
    __device__ void flush(const int wave)
    {
        if (on && (threadIdx.x & 63) == 0) for (int k = 0; k < 8; k++) { zpd_prof[wave][k][0] += acc[k][0]; zpd_prof[wave][k][1] += acc[k][1]; }
    }
};
#define ZPD_BARRIER(kb_) prof.barrier(kb_)
#else
#define ZPD_BARRIER(kb_) lds_barrier()
#endif
// the same register of the block's other hypothesis lane (lane ^ 1)
__device__ __forceinline__ u32 xchg(const u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((i32)v, (i32)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, false);
}

// LDS behind the per-block state
struct DLds {
    int32_t cand_off;    // uint4 cand[NCH][2][blocks_per_wg]: the entry the NEXT bit is predicted from, {for bit = 0 | for bit = 1}
    int32_t ymail_off;   // u32 ymail[2][blocks_per_wg]: the decoded bit (bit 0), "block has ended" (bit 1)
    int32_t stat_off;    // i32 stat[NCH][blocks_per_wg]
    int32_t misc_off;
};

struct StageArgs {
    const DBatch *B;
    const Cfg *cfg;
    u8 *lds;
    int ci;              // component index (component waves)
    int lane, bpw;
    bool active;         // this lane has a block this round
    u8 *slot, *my;
    const u8 *src;
    u32 nin;
    u8 *dst;
    u32 cap;
    u32 blk;
    DLds L;
};

struct Req {             // a request for the three candidate rows of one nibble context, in flight
    u32x4 A, B, C;
    u32 po, chk;
};
struct Row {             // a nibble's bit-history row (byte 0 = check) and its tbase offset
    u32 x, y, z, w, off;
};

// ------------------------------------------------------------------ a component wave
template <int NCH, bool IS_ICM>
__device__ __forceinline__ void dcomp_loop(const StageArgs &S)
{
    const DBatch &B = *S.B;
    const Cfg &cfg = *S.cfg;
    const DModel &M = *B.model;
    u8 *const lds = S.lds;
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + LDS_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + LDS_SQUASH);
    const u8 *s_ns = lds + LDS_NS;
    auto stretch_of = [&](u32 cm) -> i32 {                              // see zpq_chain.hip
        u32 q = cm >> 8;
        q = min(max(q, 1u), 32767u);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
    };
    const int ci = S.ci;
    const u32 h = (u32)S.lane & 1u;                    // the outcome this lane assumes for the bit being decoded
    const int bl = S.lane >> 1;                        // its block inside the workgroup
    const DComp &C = M.comp[ci];
    const u32 ht_mask = C.ht_len - 16u;
    const bool swz = C.ht_len >= 512u;                 // (tables smaller than 512 bytes have no bit 8: tests)
    u8 *const tbase = S.slot + C.ht_off;
    const int sizebits = C.a + 2;
    u32 *const t32 = reinterpret_cast<u32 *>(S.my + cfg.lds_off32[ci]);
    u8 *const t8 = S.my + cfg.lds_off8[ci];
    const int bpw = S.bpw;
    const uint4 *const cand_in4 = reinterpret_cast<const uint4 *>(lds + S.L.cand_off) + (bl < bpw ? bl : 0);
    uint2 *const cand_out = reinterpret_cast<uint2 *>(lds + S.L.cand_off) + ((size_t)ci * 2 * bpw + (bl < bpw ? bl : 0)) * 2 + h;
    const u32 *const ymail = reinterpret_cast<const u32 *>(lds + S.L.ymail_off) + (bl < bpw ? bl : 0);

#ifdef ZPD_PROF
    Prof prof; prof.init(blockIdx.x == 0);
#endif
    i32 status = ZPQ_OK;
    const bool on = S.active;                          // this lane has a block this round: the others run along (their loads are
                                                       // harmless) but must not write -- their addresses alias block 0's
    bool alive = S.active;
    u32 prev = 0, m4 = 0, b4 = 0, hctx = 0;
    u32 slotn = 1, c8 = 1, yprev = 0;

#ifdef ZPD_DEBUG_NO_ROWS   // timing experiment only (wrong output): no hash-row traffic
#define ZPD_LOAD_ROWS(q_, po_) do { q_.A = u32x4{(po_), 0, 0, 0}; q_.B = q_.A; q_.C = q_.A; } while (0)
#else
#define ZPD_LOAD_ROWS(q_, po_)                                                          \
    do {                                                                                \
        q_.A = *reinterpret_cast<const u32x4 *>(tbase + (po_));                         \
        q_.B = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 16u));                 \
        q_.C = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 32u));                 \
    } while (0)
#endif
    // request the three candidate rows of context (hc, c8v) -- h0, h0 ^ 16, h0 ^ 32 of one 64-byte line (predictor.v:495-532)
    auto request = [&](const u32 hc, const u32 c8v) -> Req {
        Req q;
        const u32 cx = hc + 16u * c8v;
        q.chk = (cx >> sizebits) & 255u;
        const u32 h0 = (cx * 16u) & ht_mask;
        // Inside this kernel a table's 64-byte lines live at their address with bits 6 and 8 exchanged (a bijection; the
        // rows h0 ^ 16, h0 ^ 32 stay inside the line).  The two outcomes of a byte's FOURTH bit lead to contexts 16 apart
        // (predictor.v:558-560), i.e. lines 256 bytes apart: exchanged, the lane pair asks for the two halves of one
        // aligned 128-byte block -- one request to the memory system instead of two (tools/micro/rowlat.hip: -20 % latency).
        q.po = swz ? ((h0 & ~0x140u) | ((h0 >> 2) & 0x40u) | ((h0 << 2) & 0x100u)) : h0;
        ZPD_LOAD_ROWS(q, q.po);
        return q;
    };
    // Resolve hit / victim among the three candidates with selects (find_ht).  L1 = the row of the nibble that is just
    // ending, as it will be if this lane's assumed bit is the decoded one: it is still in registers, not in memory.
    // No store happens here: what this lane resolves may be the wrong outcome's.
    auto consume = [&](const Req &q, const Row &L1) -> Row {
        const u32 pa = q.po, pb = q.po ^ 16u, pc = q.po ^ 32u;
        const bool a1 = pa == L1.off, b1 = pb == L1.off, c1 = pc == L1.off;
        auto fwd = [](const bool f1, const Row &R1, const u32x4 N) -> u32x4 {
            return u32x4{f1 ? R1.x : N.x, f1 ? R1.y : N.y, f1 ? R1.z : N.z, f1 ? R1.w : N.w};
        };
        const u32x4 A = fwd(a1, L1, q.A), Bq = fwd(b1, L1, q.B), Cq = fwd(c1, L1, q.C);
        const u32 chk = q.chk;
        const bool ma = (A.x & 255u) == chk, mb = (Bq.x & 255u) == chk, mc = (Cq.x & 255u) == chk;
        const u32 qa = (A.x >> 8) & 255u, qb = (Bq.x >> 8) & 255u, qc = (Cq.x >> 8) & 255u;
        const bool va = qa <= qb && qa <= qc, vb = qb < qc;             // victim order (predictor.v:513-531)
        const bool hit = ma || mb || mc;
        const bool ua = ma || (!hit && va);
        const bool ub = !ua && (mb || (!hit && vb));
        Row R;
        R.off = ua ? pa : (ub ? pb : pc);
        const u32 Rx = ua ? A.x : (ub ? Bq.x : Cq.x), Ry = ua ? A.y : (ub ? Bq.y : Cq.y);
        const u32 Rz = ua ? A.z : (ub ? Bq.z : Cq.z), Rw = ua ? A.w : (ub ? Bq.w : Cq.w);
        R.x = hit ? Rx : chk; R.y = hit ? Ry : 0u; R.z = hit ? Rz : 0u; R.w = hit ? Rw : 0u;
        return R;
    };
    // ZPAQL.run(byte) + h[] copy (predictor.v:809-816) for the two shipped program shapes -> this component's context;
    // vm_hash does not touch the VM's state (the byte may not be the one decoded), vm_commit does once it is certain
    auto vm_hash = [&](const u32 byte) -> u32 {
        u32 hv = 0;
        if (NCH != 2) {
            // b=c c-- *c=a d=0 (hash *d=a d++)* hash *d=a halt: H[k] = hash^(k+1) of (byte, prev)
            u32 a = byte;
            for (int k = 0; k <= ci; k++) a = (a + prev + 512u) * 773u;
            hv = a;
        } else {
            // level 1: *b=a a=0 d=0 hash b-- hash *d=a d++ b-- hash b-- hash *d=a halt, M = 4 bytes
            const u32 mm = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8));
            u32 bb = b4;
            u32 a = 0;
            a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            const u32 h0v = a; bb--;
            a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            hv = (ci == 0) ? h0v : a;
        }
        return hv;
    };
    auto vm_commit = [&](const u32 byte) {
        if (NCH != 2) prev = byte;
        else { m4 = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8)); b4 -= 3u; }
    };

    Row row = {0, 0, 0, 0, 0xFFFFFFFFu};               // the current nibble's row (both lanes of a block hold it)
    Row nrow = {0, 0, 0, 0, 0};                        // the next nibble's, as this lane resolved it for ITS outcome
    Req req;
    {
        const u32x4 z4 = {0, 0, 0, 0};
        req.A = z4; req.B = z4; req.C = z4; req.po = 0; req.chk = 0;
    }
    u32 cur_s = 0, cur_v = 0;                          // the state the bit being decoded is predicted from, its packed entry
    i32 cur_b = 0;
    u32 hn_spec = 0;                                   // the next byte's context hash under this lane's outcome of the last bit
    auto icm_st = [](u32 v, i32 b) -> i32 { return (i32)((u32)b << 9) | (i32)(v >> 23); };
    // what the next bit is predicted from, as the consumers want it: stretch(p) (ICM) / w0, w1 << 6 (ISSE)
    auto publish = [&](const int par, const u32 v, const i32 b) {
        uint2 o;
        if (IS_ICM) { o.x = (u32)icm_st(v, b); o.y = 0u; }
        else { o.x = (u32)(((i32)(v << 12)) >> 12); o.y = (u32)((i32)(((u32)b << 12) | (v >> 20)) << 6); }
        if (on) cand_out[(size_t)par * bpw * 2] = o;
    };

    // The published entries of the components below this one, for the bit about to be predicted: both outcomes' halves,
    // read right behind the barrier together with the decoded bit (a read whose address depends on the bit would be a
    // second LDS round trip in series).
    constexpr int NIN = NCH - 1;
    uint4 cin[NIN > 0 ? NIN : 1];
    auto load_cands = [&](const int par) {
#pragma unroll
        for (int j = 0; j < NIN; j++)
            if (j < ci) cin[j] = cand_in4[(size_t)(j * 2 + par) * bpw];
    };

    // One coded bit.  K = bit inside the nibble (selects at compile time which dword of the row holds the slot:
    // slot 1 | 2..3 | 4..7 | 8..15, predictor.v:817-823), NB = nibble of the byte.
    // Program order = the order the waits come in: state and entry reads that depend on nothing newer than the last
    // commit go out first, the chain and the squash read follow, everything is consumed behind them.
    auto cycle = [&](auto kc, auto nbc) {
        constexpr int K = decltype(kc)::value;
        constexpr int NB = decltype(nbc)::value;
        constexpr int kb = NB * 4 + K;                 // position in coding order 0..7
        constexpr int par = kb & 1;
        const u32 s = cur_s;
        // ---- (1) reads that need only the committed state
        const u32 ns01 = *reinterpret_cast<const u16 *>(s_ns + s * 4);       // next state for y=0 | y=1 << 8
        u32 sN = 0, rNv = 0;
        i32 rNb = 0;
        if (K < 3) {
            u32 pair;
            if (K == 0) pair = row.x >> 16;
            else if (K == 1) pair = row.y >> ((slotn & 1u) * 16u);
            else pair = ((slotn & 2u) ? row.w : row.z) >> ((slotn & 1u) * 16u);
            sN = h ? ((pair >> 8) & 255u) : (pair & 255u);
            rNv = t32[sN];
            rNb = (i32)(int8_t)t8[sN];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- (2) the prediction of this bit, up to this component
        i32 pin = 0, sq = 0;
        const i32 w0 = ((i32)(cur_v << 12)) >> 12;                           // ISSE: sext20
        const i32 w1 = (i32)(((u32)cur_b << 12) | (cur_v >> 20));
        if (!IS_ICM) {
            pin = (i32)(yprev ? cin[0].z : cin[0].x);
#pragma unroll
            for (int j = 1; j < NIN; j++) {
                if (j < ci) {
                    const i32 a = (i32)(yprev ? cin[j].z : cin[j].x), b = (i32)(yprev ? cin[j].w : cin[j].y);
                    pin = clamp2k((__mul24(a, pin) + b) >> 16);
                }
            }
            const i32 p = clamp2k((__mul24(w0, pin) + (w1 << 6)) >> 16);     // predictor.v:615-631
            sq = s_squash[p + 2048];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- (3) this lane's outcome of the bit: h
        u32 nv;
        i32 nb;
        if (IS_ICM) {
            // cm += (y*32767 - (cm >> 8)) >> 2 (predictor.v:701-709); the entry carries stretch(cm >> 8)
            const u32 cmv = cur_v & 0x7FFFFFu;
            const u32 cmn = (u32)wadd((i32)cmv, ((h ? 32767 : 0) - (i32)(cmv >> 8)) >> 2);
            const i32 st_new = stretch_of(cmn);
            nv = cmn | (((u32)st_new & 0x1FFu) << 23);
            nb = st_new >> 9;
        } else {
            const i32 err = (h ? 32767 : 0) - sq;                            // predictor.v:776-791
            const i32 nw0 = clamp512k(w0 + ((__mul24(err, pin) + (1 << 12)) >> 13));
            const i32 nw1 = clamp512k(w1 + ((err + 16) >> 5));
            nv = ((u32)nw0 & 0xFFFFFu) | ((u32)nw1 << 20);
            nb = nw1 >> 12;
        }
        if (K == 3) {
            // the nibble ends with this bit: resolve the next nibble's row for this lane's outcome (requested when the
            // third bit became known).  The row that is ending counts as updated with this lane's outcome.
            Row rh = row;
            const u32 nsv = h ? (ns01 >> 8) : (ns01 & 255u);
            const u32 sh = (slotn & 3u) * 8u;
            const u32 dsel = (slotn & 4u) ? rh.w : rh.z;
            const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
            rh.w = (slotn & 4u) ? ins : rh.w; rh.z = (slotn & 4u) ? rh.z : ins;
            nrow = consume(req, rh);
            sN = (nrow.x >> 8) & 255u;
            rNv = t32[sN];
            rNb = (i32)(int8_t)t8[sN];
        }
        const bool same = sN == s;
        const u32 nxt_v = same ? nv : rNv;
        const u32 nxt_bs = ((u32)(same ? nb : rNb) & 255u) | (sN << 8);
        publish(par, nxt_v, (i32)(int8_t)(nxt_bs & 255u));
        // what the neighbour lane holds for ITS outcome: fetched before the bit is known, picked after
        const u32 xv = xchg(nxt_v), xb = xchg(nxt_bs);
        u32 x0 = 0, x1 = 0, x2 = 0, x3 = 0, xo = 0, xh = 0;
        if (K == 3) {
            x0 = xchg(nrow.x); x1 = xchg(nrow.y); x2 = xchg(nrow.z); x3 = xchg(nrow.w); xo = xchg(nrow.off);
            if (NB == 1) xh = xchg(hn_spec);
        }
        ZPD_BARRIER(kb);
        // ---- (4) the bit, and the entries the next bit is predicted from
        const u32 ym = ymail[(size_t)par * bpw];
        load_cands(par);
        __builtin_amdgcn_sched_barrier(0);
        const u32 y = ym & 1u;
        alive = alive && (ym & 2u) == 0u;
        const bool mine = y == h;
        if (mine && on) { t32[s] = nv; t8[s] = (u8)nb; }    // one lane trains the block's table entry
        {
            cur_v = mine ? nxt_v : xv;
            const u32 bs = mine ? nxt_bs : xb;
            cur_b = (i32)(int8_t)(bs & 255u);
            cur_s = bs >> 8;
        }
        // next bit-history state into the row (statetable.v:75-84)
        {
            const u32 nsv = y ? (ns01 >> 8) : (ns01 & 255u);
            const u32 sh = (slotn & 3u) * 8u;
            const u32 dsel = (K <= 1) ? row.x : (K == 2 ? row.y : ((slotn & 4u) ? row.w : row.z));
            const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
            if (K <= 1) row.x = ins;
            else if (K == 2) row.y = ins;
            else { row.w = (slotn & 4u) ? ins : row.w; row.z = (slotn & 4u) ? row.z : ins; }
        }
        c8 = (c8 << 1) | y;
        if (K == 3) {
            // the finished row goes back to its table (one lane of the pair stores; the requests have all been consumed)
#ifndef ZPD_DEBUG_NO_ROWS
            if (alive && h == 0u && row.off != 0xFFFFFFFFu) *reinterpret_cast<u32x4 *>(tbase + row.off) = u32x4{row.x, row.y, row.z, row.w};
#endif
            row.x = mine ? nrow.x : x0; row.y = mine ? nrow.y : x1; row.z = mine ? nrow.z : x2; row.w = mine ? nrow.w : x3;
            row.off = mine ? nrow.off : xo;
            slotn = 1;
            if (NB == 1) {
                hctx = mine ? hn_spec : xh;
                vm_commit(c8 - 256u);
                c8 = 1;
            }
        } else {
            slotn = slotn * 2u + y;
        }
        if (K == 2) {
            // three bits of the nibble are known: ask for the next nibble's rows under this lane's outcome of the fourth
            const u32 c8n = (c8 << 1) | h;
            if (NB == 0) req = request(hctx, c8n);
            else { hn_spec = vm_hash(c8n - 256u); req = request(hn_spec, 1u); }
        }
        yprev = y;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ---- first nibble of the first byte: h = 0, c8 = 1; both lanes of a block publish the same entry
    {
        req = request(0u, 1u);
        const Row none = {0, 0, 0, 0, 0xFFFFFFFFu};
        row = consume(req, none);
        cur_s = (row.x >> 8) & 255u;
        cur_v = t32[cur_s];
        cur_b = (i32)(int8_t)t8[cur_s];
        publish(1, cur_v, cur_b);
        lds_barrier();
        load_cands(1);
    }
    for (;;) {
        cycle(I0{}, I0{});
        cycle(I1{}, I0{});
        cycle(I2{}, I0{});
        cycle(I3{}, I0{});
        cycle(I0{}, I1{});
        cycle(I1{}, I1{});
        cycle(I2{}, I1{});
        cycle(I3{}, I1{});
        if (__ballot(alive) == 0ull) break;            // (every wave sees the same "ended" bits at the same bit)
    }
    // (the last nibble's row is not written back: the slot is re-initialised for the next block)
    if ((S.lane & 1) == 0 && bl < bpw) reinterpret_cast<i32 *>(lds + S.L.stat_off)[ci * bpw + bl] = status;
#ifdef ZPD_PROF
    prof.flush(ci);
#endif
#undef ZPD_LOAD_ROWS
}

// ------------------------------------------------------------------ the decoder wave (decoder.v:73-145)
struct DecState {
    u32 low, high, code, ipos, opos;
    u32 first;
    i32 status;
};

template <int NCH>
__device__ __forceinline__ void dcoder_loop(const StageArgs &S, DecState &X)
{
    const DBatch &B = *S.B;
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + LDS_SQUASH);
    const int bpw = S.bpw;
    const int bl = S.lane < bpw ? S.lane : 0;
    const uint2 *const cand_in = reinterpret_cast<const uint2 *>(lds + S.L.cand_off) + bl * 2;
    u32 *const ymail = reinterpret_cast<u32 *>(lds + S.L.ymail_off) + bl;
    const u8 *const src = S.src;
    const u32 nin = S.nin;
    u8 *const dst = S.dst;
    const u32 cap = S.cap;
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;

    // The coded input: a FOUR-dword window, double-buffered per byte of output (zpq_chain.hip): at the top of every
    // byte iteration the window requested one iteration ago is adopted and the next one requested, unconditionally.
    const u32 mis = (u32)(reinterpret_cast<uintptr_t>(src) & 3u);
    const u32 *src4 = reinterpret_cast<const u32 *>(src - mis);
    const u32 ndw = (nin + mis + 3u) >> 2;
    const u32 *const enc4 = ndw ? src4 : reinterpret_cast<const u32 *>(B.in_off);   // nin == 0: any readable dword
    const u32 dlast = ndw ? ndw - 1u : 0u;
    u32 dW0 = 0, dW1 = 0, dW2 = 0, dW3 = 0, wd = 0;
    u32 nW0 = 0, nW1 = 0, nW2 = 0, nW3 = 0, nwd = 0;
    auto dec_request = [&](const u32 pos) {
        nwd = (pos + mis) >> 2;
        nW0 = enc4[min(nwd, dlast)];
        nW1 = enc4[min(nwd + 1u, dlast)];
        nW2 = enc4[min(nwd + 2u, dlast)];
        nW3 = enc4[min(nwd + 3u, dlast)];
    };
    auto dec_adopt = [&]() { dW0 = nW0; dW1 = nW1; dW2 = nW2; dW3 = nW3; wd = nwd; };
    auto in_byte = [&](u32 pos) -> u32 {               // src[pos]; 0 past the end
        const u32 vp = pos + mis;
        const u32 bo = vp - 4u * wd;
        const u64 lo = (u64)dW0 | ((u64)dW1 << 32), hi = (u64)dW2 | ((u64)dW3 << 32);
        u32 c = (u32)(((bo & 8u) ? hi : lo) >> ((bo & 7u) * 8u)) & 255u;
        if (bo > 15u) {
            u32 t = enc4[min(vp >> 2, dlast)];
            asm volatile("; coded input beyond the window: waited for here, not at the join" : "+v"(t));
            c = (t >> ((vp & 3u) * 8u)) & 255u;
        }
        return pos < nin ? c : 0u;
    };
    auto renorm = [&]() {
        while ((X.high ^ X.low) < 0x1000000u) {
            X.low <<= 8; X.high = (X.high << 8) | 255u; X.low = X.low ? X.low : 1u;
            const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin);
            X.code = (X.code << 8) | c;
        }
    };

#ifdef ZPD_PROF
    Prof prof; prof.init(blockIdx.x == 0);
#endif
    bool dead = !S.active;                             // the block has ended (EOF, output full) or the lane has none
    bool got_first = false;
    bool pend_store = false;
    u32 pend_pos = 0, pend_val = 0;
    dec_request(0u);
    dec_adopt();
    dec_request(0u);                                   // (adopted by the first byte iteration)
    if (S.active) {
        for (int k = 0; k < 4; k++) { const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin); X.code = (X.code << 8) | c; }
    }
    u32 yprev = 0;
    lds_barrier();                                     // the components have published the first bit's entries
    for (;;) {
        // (unconditional: a load inside an exec-masked branch is waited for at the branch join, at once)
        dec_adopt();
        // The byte decoded in the last iteration is stored HERE, behind the wait for the window that was requested a whole
        // byte ago (vmcnt retires in order: a store issued at the end of the last iteration would be waited for as well),
        // and unconditionally -- lanes with nothing to store aim at their block's state slot (the ZPAQL registers no chain
        // kernel uses): a store under a branch makes the compiler wait for everything at the next use of a loaded register.
        {
            u8 *const sp = pend_store ? dst + pend_pos : S.slot;
            *sp = (u8)pend_val;
            pend_store = false;
        }
        dec_request(X.ipos);
        // ---- EOF flag: decode(0) (decoder.v:128): p = 0, mid = low
        if (!dead) {
            if (X.code <= X.low) { dead = true; X.high = X.low; } else { X.low = X.low + 1; }
            renorm();
        }
        u32 c8 = 1;
        bool dead_pub = dead;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            const int par = kb & 1, ppar = par ^ 1;
            // the chain p0 -> p1 -> ... from the entries the components published for this bit
            const uint2 *const cin = cand_in + yprev;
            i32 p = (i32)cin[(size_t)(0 * 2 + ppar) * bpw * 2].x;
#pragma unroll
            for (int j = 1; j < NCH; j++) {
                const uint2 cj = cin[(size_t)(j * 2 + ppar) * bpw * 2];
                p = clamp2k((__mul24((i32)cj.x, p) + (i32)cj.y) >> 16);
            }
            const u32 sq = s_squash[p + 2048];
            u32 y = 0;
            if (!dead) {
                const u32 p16 = sq * 2u + 1u;                               // decoder.v:86
                const u32 mid = X.low + mul_shr16(X.high - X.low, p16);
                y = X.code <= mid ? 1u : 0u;
                X.high = y ? mid : X.high;
                X.low = y ? X.low : mid + 1;
                renorm();
            }
            c8 = (c8 << 1) | y;
            if (S.lane < bpw) ymail[(size_t)par * bpw] = y | (dead ? 2u : 0u);
            dead_pub = dead;
            yprev = y;
            ZPD_BARRIER(kb);
        }
        if (!dead) {
            const u32 byte = c8 - 256u;
            if (pp && !got_first) { X.first = byte; got_first = true; }
            else {
                pend_store = X.opos < cap; pend_pos = X.opos; pend_val = byte;
                X.opos++;
                if (X.opos > cap) dead = true;
            }
        }
        if (__ballot(!dead_pub) == 0ull) break;
    }
    if (pend_store) dst[pend_pos] = (u8)pend_val;
#ifdef ZPD_PROF
    prof.flush(NCH);
#endif
}

// NCH = chain length (ICM + ISSEs); waves: NCH + the decoder
template <int NCH>
__global__ void __launch_bounds__(64 * (NCH + 1)) k_dpipe(const DBatch B, const Cfg cfg, const DLds L)
{
    extern __shared__ __align__(16) u8 lds[];
    const DModel &M = *B.model;
    const int tid = threadIdx.x, nthr = blockDim.x;
    {
        u32 *st = reinterpret_cast<u32 *>(lds + LDS_STRETCH);
        for (int i = tid; i < 2048 + 128; i += nthr) st[i] = B.stretch_c[i];
        u16 *sq = reinterpret_cast<u16 *>(lds + LDS_SQUASH);
        for (int i = tid; i < 4096; i += nthr) sq[i] = (u16)B.squash[min(max(i - 1, 0), 4093)];   // entry p + 2048 = squash(p): no clamp in the bit loop (|p| <= 2048)
        u8 *ns = lds + LDS_NS;
        for (int i = tid; i < 1024; i += nthr) ns[i] = B.ns[i];
    }
    __syncthreads();
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + LDS_STRETCH);
    const int lane = tid & 63, wave = tid >> 6;
    const int bpw = cfg.blocks_per_wg;
    const int wg_slot0 = blockIdx.x * bpw;
    const int nslots = B.nslots;
    const bool is_coder = wave == NCH;
    const int bl = is_coder ? lane : (lane >> 1);      // this lane's block inside the workgroup
    const int slot_id = wg_slot0 + bl;
    const bool lane_on = bl < bpw && slot_id < nslots;
    u8 *const slot = B.slots + (u64)(lane_on ? slot_id : wg_slot0) * M.slot_bytes;
    u8 *const my = lds + LDS_STATE + (lane_on ? bl : 0) * cfg.lds_per_block;
    const int wg_slots = min(bpw, nslots - wg_slot0);

    for (int base = wg_slot0; base < B.nblocks; base += nslots) {       // rounds: slot s decodes blocks s, s + nslots, ...
        const int blk = base + bl;
        const bool active = lane_on && blk < B.nblocks;
        const int nact = min(wg_slots, B.nblocks - base);
        // ---- Predictor.init + ZPAQL.clear for the round's blocks (predictor.v:325-470, zpaql.v:54-95)
        {
            const u64 n16 = M.zero_bytes / 16;
            const uint4 zero = make_uint4(0, 0, 0, 0);
            for (int b = 0; b < nact; b++) {
                uint4 *z4 = reinterpret_cast<uint4 *>(B.slots + (u64)(wg_slot0 + b) * M.slot_bytes);
                for (u64 i = tid; i < n16; i += nthr) z4[i] = zero;
            }
            for (int idx = tid; idx < nact * 256; idx += nthr) {
                const int b = idx >> 8, i = idx & 255;
                u8 *blk_lds = lds + LDS_STATE + b * cfg.lds_per_block;
                {
                    const u32 cmi = B.img[i];                           // cminit(i) (statetable.v:108-116), < 2^23
                    u32 q = cmi >> 8;
                    q = min(max(q, 1u), 32767u);
                    const u32 wv = s_stretch[q >> 4];
                    const u32 ei = q < 64u ? q : (q - 32704u + 64u);
                    const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
                    const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
                    const i32 sti = (q < 64u || q >= 32704u) ? endv : midv;
                    reinterpret_cast<u32 *>(blk_lds + cfg.lds_off32[0])[i] = cmi | (((u32)sti & 0x1FFu) << 23);
                    (blk_lds + cfg.lds_off8[0])[i] = (u8)(sti >> 9);
                }
                const u32 a0 = B.img[256 + 2 * i], a1 = B.img[257 + 2 * i];
                for (int c = 1; c < NCH; c++) {
                    reinterpret_cast<u32 *>(blk_lds + cfg.lds_off32[c])[i] = (a0 & 0xFFFFFu) | (a1 << 20);
                    (blk_lds + cfg.lds_off8[c])[i] = (u8)((i32)a1 >> 12);
                }
            }
        }
        __syncthreads();

        StageArgs S;
        S.B = &B; S.cfg = &cfg; S.lds = lds; S.ci = wave; S.lane = lane; S.bpw = bpw; S.active = active;
        S.slot = slot; S.my = my; S.L = L; S.blk = (u32)blk;
        S.src = active ? B.in + B.in_off[blk] : B.in;
        S.nin = active ? (u32)(B.in_off[blk + 1] - B.in_off[blk]) : 0u;
        S.dst = active ? B.out + B.out_off[blk] : B.out;
        S.cap = active ? (u32)(B.out_off[blk + 1] - B.out_off[blk]) : 0u;

        DecState X;
        X.low = 1; X.high = 0xFFFFFFFFu; X.code = 0; X.ipos = 0; X.opos = 0; X.first = 0xFFFFFFFFu; X.status = ZPQ_OK;
        if (wave == 0) dcomp_loop<NCH, true>(S);
        else if (wave < NCH) dcomp_loop<NCH, false>(S);
        else dcoder_loop<NCH>(S, X);
        __syncthreads();
        if (is_coder && active) {
            const i32 *stat = reinterpret_cast<const i32 *>(lds + L.stat_off);
            i32 st = ZPQ_OK;
            for (int c = 0; c < NCH; c++) { const i32 sc = stat[c * bpw + lane]; st = st ? st : sc; }
            if (X.opos > S.cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
            B.out_len[blk] = X.opos;
            B.status[blk] = st;
            if (B.consumed) B.consumed[blk] = X.ipos;
            if (B.final_code) B.final_code[blk] = X.code;
            if (B.first_byte) B.first_byte[blk] = X.first;
        }
        __syncthreads();
    }
}

}  // namespace zpqd

// ------------------------------------------------------------------ host side
using zpqc::Cfg;

static bool dpipe_layout(const Cfg &cfg, int bpw, zpqd::DLds *L, size_t *lds_bytes)
{
    const int nch = cfg.nch_spec;
    size_t off = (size_t)zpqc::LDS_STATE + (size_t)bpw * cfg.lds_per_block;
    off = (off + 15) & ~(size_t)15;
    L->cand_off = (int32_t)off; off += (size_t)nch * 2 * bpw * 16;
    L->ymail_off = (int32_t)off; off += (size_t)2 * bpw * 4;
    L->stat_off = (int32_t)off; off += (size_t)nch * bpw * 4;
    L->misc_off = (int32_t)off; off += 16;
    *lds_bytes = off;
    return off <= 160 * 1024;
}

// The wave-split decoder exists for the dense chains of levels 1-3 and is OPT-IN (ZPQ_DEC_PIPE=1): measured on MI355X it
// is slower than the lane-per-component decoder (level 2 x 8192: 300 vs 269 ms; EXPERIMENTS.md 4.5 has the per-wave cycle
// breakdown) -- both are bound by the two HBM round trips per byte that nothing can be overlapped with once the LDS is
// full of blocks, and the faster bit step leaves more of them exposed.  It stays as a third, independently written
// device implementation that the tests compare the others with.  A block is a lane PAIR of a component wave: at most 32
// per workgroup; and a wave must not live on a handful of lanes (zpq_pipe.hip): fewer than 12 resident blocks stay
// with zpq_chain.hip.
extern "C" int zpq_dpipe_applies(const DModel *M, int blocks_per_wg, int nslots)
{
    if (nslots < 12) return 0;
    const char *ev = getenv("ZPQ_DEC_PIPE");
    if (!ev || atoi(ev) == 0) return 0;
    Cfg cfg;
    if (!zpq_chain_build_cfg(M, &cfg)) return 0;
    if (cfg.has_mix2 || cfg.sparse) return 0;
    if (!(cfg.nch_spec == 2 || cfg.nch_spec == 3 || cfg.nch_spec == 5)) return 0;
    if (blocks_per_wg < 1 || blocks_per_wg > 32 || blocks_per_wg > cfg.blocks_per_wg) return 0;
    zpqd::DLds L;
    size_t lds = 0;
    return dpipe_layout(cfg, blocks_per_wg, &L, &lds) ? 1 : 0;
}

#ifdef ZPD_PROF
extern "C" int zpq_debug_dpipe_prof(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(zpqd::zpd_prof), sizeof(unsigned long long) * 16 * 8 * 2) != hipSuccess) return ZPQ_E_INTERNAL;
    if (reset) { static unsigned long long z[16 * 8 * 2]; if (hipMemcpyToSymbol(HIP_SYMBOL(zpqd::zpd_prof), z, sizeof z) != hipSuccess) return ZPQ_E_INTERNAL; }
    return ZPQ_OK;
}
#endif

extern "C" int zpq_launch_dpipe(const DBatch *B, const DModel *hostM, int nwg, int blocks_per_wg, hipStream_t stream)
{
    Cfg cfg;
    if (!zpq_chain_build_cfg(hostM, &cfg)) return ZPQ_E_INTERNAL;
    if (!zpq_dpipe_applies(hostM, blocks_per_wg, B->nslots)) return ZPQ_E_INTERNAL;
    if (B->prog_counter) return ZPQ_E_INTERNAL;              // (early download: the lane-per-component decoder)
    {   // regroup the plan's slots evenly, at least 16 per workgroup where the batch has them (zpq_pipe.hip)
        const int nslots = B->nslots;
        int cap = cfg.blocks_per_wg < 32 ? cfg.blocks_per_wg : 32;
        int per = blocks_per_wg < 16 ? (cap < 16 ? cap : 16) : (blocks_per_wg < cap ? blocks_per_wg : cap);
        { const char *ev = getenv("ZPQ_DPIPE_PER"); if (ev && atoi(ev) >= 9 && atoi(ev) <= cap) per = atoi(ev); }
        nwg = (nslots + per - 1) / per;
        blocks_per_wg = (nslots + nwg - 1) / nwg;
    }
    cfg.blocks_per_wg = blocks_per_wg;
    zpqd::DLds L;
    size_t lds = 0;
    if (!dpipe_layout(cfg, blocks_per_wg, &L, &lds)) return ZPQ_E_INTERNAL;
    { const char *ev = getenv("ZPQ_DPIPE_SHARE"); if (!(ev && atoi(ev) == 1) && lds < 81 * 1024) lds = 81 * 1024; }   // one workgroup per CU unless asked
#define ZPD_LAUNCH(N)                                                                                                \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void *)zpqd::k_dpipe<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((zpqd::k_dpipe<N>), dim3(nwg), dim3(64 * ((N) + 1)), lds, stream, *B, cfg, L);            \
    } while (0)
    switch (cfg.nch_spec) {
    case 2: ZPD_LAUNCH(2); break;
    case 3: ZPD_LAUNCH(3); break;
    case 5: ZPD_LAUNCH(5); break;
    default: return ZPQ_E_INTERNAL;
    }
#undef ZPD_LAUNCH
    return ZPQ_OK;
}
