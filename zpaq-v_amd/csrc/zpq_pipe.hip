// zpq_pipe.hip -- the ENCODER of the chain models (ICM + ISSEs [+ MIX2]: levels 1-5, levels.v:53-375) as a pipeline of WAVES.
//
// Every context, coded bit and bit-history state of the encoder is a function of the input alone; only predictions
// flow down the chain ICM -> ISSE -> ... [-> MIX2] -> coder, and every component trains on its OWN prediction
// (predictor.v:701-709,776-791).  zpq_chain.hip maps a block to a group of lanes (lane = component) and, when
// encoding, lets lane c run c bytes behind the ICM -- but all lanes of a wave still issue the union of the ICM's,
// the ISSE's and the coder's instructions, and a wave that is alone on its SIMD issues one instruction every five
// cycles: ~190 instructions = ~800 cycles per coded bit.
//
// Here the roles are separated by WAVE, and a lane is a block:
//   * wave c < NCH owns component c of every block of the workgroup (its bit-history rows in HBM, its counters /
//     weights in LDS -- the layout of zpq_chain_cfg.h), a MIX2 (levels 4-5) has the next wave, the last wave is the
//     arithmetic coder (encoder.v:48-139);
//   * in iteration `it` wave c works on byte it - c of every block; it takes its predecessor's eight predictions
//     of that byte from LDS (16 bytes per block and link, double-buffered) and leaves its own there; the last
//     stage leaves squash(p) and the bit, which is what the coder needs;
//   * one s_barrier per byte keeps the waves in step (with an LDS-only wait in front of it: a full fence would
//     also wait for the row prefetch every byte).
// Each wave then issues only its own role's instructions (~50-135 per bit) on the CU's four SIMDs side by side.
// Coded bytes are identical to zpq_chain.hip's, zpq_generic.hip's and the CPU oracle's.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_chain_cfg.h"

namespace zpqp {

using namespace zpqc;

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
__device__ __forceinline__ i32 clamp2k(i32 x) { return min(max(x, -2048), 2047); }
__device__ __forceinline__ i32 clamp512k(i32 x) { return min(max(x, -262144), 262143); }
__device__ __forceinline__ uint32_t mul_shr16(uint32_t range, uint32_t p16)   // see zpq_chain.hip
{
    return (uint32_t)__umul24(range >> 16, p16) + ((uint32_t)__umul24(range & 0xFFFFu, p16) >> 16);
}
// LDS traffic of this wave done, then the workgroup barrier (no wait for global memory: the row prefetch stays in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Striped upload (host_pipeline, DBatch::gate_flag): the rest of the input may still be crossing PCIe.  Wait for the
// host's signal (an acquire at system scope, so that nothing read afterwards is stale); bounded, so that a host that
// died cannot hang the GPU.  false = gave up.
__device__ __forceinline__ bool gate_wait(const uint32_t *flag)
{
    u32 tries = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __builtin_amdgcn_s_sleep(64);
        if (++tries > (1u << 22)) return false;
    }
    return true;
}

// LDS behind the per-block state: the links between the stages and the per-stage status words
struct PipeLds {
    int32_t link_off;    // uint4 link[NCH][2][blocks_per_wg]
    int32_t stat_off;    // int32 stat[NCH][blocks_per_wg]
    int32_t misc_off;    // u32: longest block of the round
    int32_t st_off;      // k_pipe2: uint2 states[NCH][2][blocks_per_wg] (a byte's eight bit-history states per component)
};

struct StageArgs {
    const DBatch *B;
    const Cfg *cfg;
    u8 *lds;
    int ci;              // stage = component index (the coder: NCH)
    int delay;           // comp_loop: in iteration `it` the stage works on byte it - delay (k_pipe: = ci)
    int lane, bpw;
    bool active;         // this lane has a block this round
    u8 *slot, *my;
    const u8 *src;
    u32 nin, total, iters;
    u32 split;           // HIO: iterations before the gate
    u8 *dst;
    u32 cap;
    u32 blk;
    PipeLds L;
};

// The input bytes of a block, one per iteration: a register window of two dwords plus the one requested behind them,
// sliding by register moves, with an unconditional look-ahead load per byte (zpq_chain.hip, enc_byte).  The byte after
// the current one is always inside the two dwords that have arrived.
struct InWin {
    const u32 *enc4;
    u32 enc_last, mis, nin;
    u32 win0, win1, win2, wdw;
    __device__ __forceinline__ void open(const u8 *src, const u32 n, const uint64_t *fallback)
    {
        nin = n;
        mis = (u32)(reinterpret_cast<uintptr_t>(src) & 3u);
        const u32 *src4 = reinterpret_cast<const u32 *>(src - mis);
        const u32 ndw = (nin + mis + 3u) >> 2;
        enc4 = ndw ? src4 : reinterpret_cast<const u32 *>(fallback);    // nin == 0: any readable dword
        enc_last = ndw ? ndw - 1u : 0u;
        win0 = enc4[0];
        win1 = enc4[min(1u, enc_last)];
        win2 = enc4[min(2u, enc_last)];
        wdw = 0;
    }
    __device__ __forceinline__ u32 byte(const u32 pos)                  // pos advances by at most one per call
    {
        const u32 vp = pos + mis;
        const bool slide = (vp >> 2) != wdw;
        win0 = slide ? win1 : win0;
        win1 = slide ? win2 : win1;
        wdw = slide ? wdw + 1u : wdw;
        win2 = enc4[min(wdw + 2u, enc_last)];
        const u32 c = (win0 >> ((vp & 3u) * 8u)) & 255u;
        return pos < nin ? c : 0u;
    }
    __device__ __forceinline__ u32 peek(const u32 pos) const            // pos = the last byte()'s position, or one more
    {
        const u32 vp = pos + mis;
        const u32 d = (vp >> 2) != wdw ? win1 : win0;
        const u32 c = (d >> ((vp & 3u) * 8u)) & 255u;
        return pos < nin ? c : 0u;
    }
};

// a request for the three candidate rows of one nibble context, in flight
struct Req {
    u32x4 A, B, C, tags;
    u32 po, chk, key, si, off;
    u32 tw;              // dense table that is not cleared (zpq_touch_layout): the word holding the line's four "touched" bits
};
// a nibble's bit-history row (byte 0 = check) and its tbase offset
struct Row {
    u32 x, y, z, w, off;
};

// ------------------------------------------------------------------ a component stage
// IS_LAST: the component the coder follows (its link carries squash(p) and the bit).  FWD_PIN: the last ISSE of a chain
// that a MIX2 follows: it hands its INPUT on as well (the MIX2 mixes p[n-3] and p[n-2], levels.v:199-218,290-375).
template <int NCH, bool SP, bool HIO, bool IS_ICM, bool IS_LAST, bool FWD_PIN>
__device__ __forceinline__ void comp_loop(const StageArgs &S)
{
    const DBatch &B = *S.B;
    const Cfg &cfg = *S.cfg;
    const DModel &M = *B.model;
    u8 *const lds = S.lds;
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + LDS_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + LDS_SQUASH);
    const u8 *s_ns = lds + LDS_NS;
    auto stretch_of = [&](u32 cm) -> i32 {                              // see zpq_chain.hip
#ifdef ZPQ_STRETCH_ENDS
        u32 q = cm >> 8;
        q = min(max(q, 1u), 32767u);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
#else
        const u32 wv = s_stretch[(cm >> 12) & 2047u];                   // (an ICM counter's index stays below 32767)
        return ((i32)wv >> 16) + (i32)__popc(__builtin_amdgcn_ubfe(wv, 1u, (cm >> 8) & 15u));
#endif
    };
    const int ci = S.ci;
    const DComp &C = M.comp[ci];
    const u32 ht_mask = C.ht_len - 16u;
    const u32 sp_cap = SP ? C.sp_cap : 0u;
    const u32 sp_groups = sp_cap >> 2;
    u8 *const slot = S.slot;
    u32 *const sp_tags = reinterpret_cast<u32 *>(slot + C.sp_tag_off);
    u8 *const tbase = sp_cap ? slot + C.sp_line_off : slot + C.ht_off;
    // A dense table may come WITHOUT having been cleared (zpq_touch_layout: 12 MiB per level-2 block otherwise): one
    // "touched" bit per 16-byte row says whether the row has been written in this block; an untouched row reads as zeros.
    // Built, parity-green and MEASURED: level 2 encode 135-145 ms against 123-126 ms with the 15 ms of clearing (level 1 133
    // against 136, level 3 dense 167 against 167.5) -- the extra load per request, the twelve selects and the word store cost
    // what the clearing costs.  Compiled only with -DZPP_TOUCH (tools/variant.sh).
#ifdef ZPP_TOUCH
    constexpr bool TOUCHC = !SP;
#else
    constexpr bool TOUCHC = false;
#endif
    const bool touch = TOUCHC && C.tb_off != 0;
    u32 *const tb32 = reinterpret_cast<u32 *>(slot + (touch ? C.tb_off : 0));
    u32 tc_word = 0xFFFFFFFFu, tc_bit = 0;             // the last bit set: a request that was in flight then holds a stale word
    const int sizebits = C.a + 2;
    u32 *const t32 = reinterpret_cast<u32 *>(S.my + cfg.lds_off32[ci]);
    u8 *const t8 = S.my + cfg.lds_off8[ci];
    const uint4 *const link_in = reinterpret_cast<const uint4 *>(lds + S.L.link_off) + (size_t)(ci > 0 ? ci - 1 : 0) * 2 * S.bpw + S.lane;
    uint4 *const link_out = reinterpret_cast<uint4 *>(lds + S.L.link_off) + (size_t)ci * 2 * S.bpw + S.lane;
    uint4 *const link_pin = reinterpret_cast<uint4 *>(lds + S.L.link_off) + (size_t)(NCH + 1) * 2 * S.bpw + S.lane;   // (FWD_PIN)
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u32 total = S.total;
    // Dense tables: a nibble's rows are requested a whole BYTE (two nibbles) before they are used -- every context of
    // the encoder is known from the input -- so the HBM round trip (~2000 cycles under this load) hides behind eight
    // bit steps instead of four.  Two nibbles finish between request and use; what they wrote is forwarded from
    // registers (FWD2).  The line store keeps the one-nibble distance: its tags and claims would need forwarding too.
#ifdef ZPP_NO_FWD2
    constexpr bool FWD2 = false;
#else
    constexpr bool FWD2 = !SP;
#endif

    i32 status = ZPQ_OK;
    InWin W;
    if (S.active) W.open(S.src, S.nin, B.in_off);
    u32 prev = 0, m4 = 0, b4 = 0, hctx = 0;
    u32 slotn = 1;                                     // hmap4 & 15
    u32 ch = 0;
    bool sp_full = false;
    u32 sp_claims = 0;
    const u32 sp_limit = sp_cap - (sp_cap >> 4);

#ifdef ZPP_DEBUG_NO_ROWS   // timing experiment only (wrong output): no hash-row traffic
#define ZPP_LOAD_ROWS(q_, po_) do { q_.A = u32x4{(po_), 0, 0, 0}; q_.B = q_.A; q_.C = q_.A; } while (0)
#else
#define ZPP_LOAD_ROWS(q_, po_)                                                          \
    do {                                                                                \
        q_.A = *reinterpret_cast<const u32x4 *>(tbase + (po_));                         \
        q_.B = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 16u));                 \
        q_.C = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 32u));                 \
    } while (0)
#endif
    // request the three candidate rows of context (hc, c8v) -- h0, h0 ^ 16, h0 ^ 32 of one 64-byte line
    // (predictor.v:495-532); compact line store as in zpq_chain.hip
    auto request = [&](const u32 hc, const u32 c8v) -> Req {
        Req q;
        const u32 cx = hc + 16u * c8v;
        q.chk = (cx >> sizebits) & 255u;
        const u32 h0 = (cx * 16u) & ht_mask;
        u32 pox = h0;
        q.key = 0; q.si = 0; q.off = 0; q.tags = u32x4{0, 0, 0, 0};
        q.tw = 0xFFFFFFFFu;
        if (TOUCHC) q.tw = tb32[touch ? (h0 >> 9) : 0u];
        if (SP && sp_cap) {
            q.key = (h0 >> 6) + 1u;
            q.si = __umulhi(q.key * 0x9E3779B1u, sp_cap);
            q.off = h0 & 48u;
            q.tags = *reinterpret_cast<const u32x4 *>(sp_tags + (q.si & ~3u));
            pox = (q.si << 6) + q.off;
        }
        q.po = pox;
        ZPP_LOAD_ROWS(q, pox);
        return q;
    };
    // Consume a request: resolve hit / victim among the three candidates with selects (find_ht), taking rows that
    // were finished after the request went out from registers -- L1 = the nibble that just ended (not yet stored),
    // L2 = the one before it (stored after the request was issued) -- THEN store L1 (vmcnt retires in order: a store
    // issued before the wait would be waited for as well).
    auto consume = [&](Req q, const bool have1, const Row L1, const bool have2, const Row L2) -> Row {
        bool claim = false;
        u32 claim_si = 0;
        if (SP && sp_cap) {
            const u32 o = q.si & 3u;
            auto probe_group = [&](const u32x4 T) -> u32 {
                const u32 mm = (min(T.x ^ q.key, T.x) == 0u ? 1u : 0u) | (min(T.y ^ q.key, T.y) == 0u ? 2u : 0u) |
                               (min(T.z ^ q.key, T.z) == 0u ? 4u : 0u) | (min(T.w ^ q.key, T.w) == 0u ? 8u : 0u);
                return ((mm * 17u) >> o) & 15u;
            };
            u32x4 T = q.tags;
            u32 g = q.si >> 2;
            u32 r = probe_group(T);
            if (r == 0u && !sp_full) {
                for (u32 tries = 1; tries < sp_groups; tries++) {
                    g = (g + 1u == sp_groups) ? 0u : g + 1u;
                    T = *reinterpret_cast<const u32x4 *>(sp_tags + 4u * g);
                    r = probe_group(T);
                    if (r) break;
                }
            }
            const u32 idx = (o + (u32)__builtin_ctz(r | 16u)) & 3u;
            const u32 t = (idx & 2u) ? ((idx & 1u) ? T.w : T.z) : ((idx & 1u) ? T.y : T.x);
            const u32 si = 4u * g + idx;
            if (r == 0u) { status = ZPQ_E_TOOBIG; sp_full = true; }
            else if (t == 0u && ++sp_claims > sp_limit) { status = ZPQ_E_TOOBIG; sp_full = true; }
            else if (t == 0u) {
                const u32x4 z4 = {0, 0, 0, 0};
                claim = true;
                claim_si = si;
                q.A = z4; q.B = z4; q.C = z4;
                q.po = (si << 6) + q.off;
            } else if (si != q.si) {
                q.po = (si << 6) + q.off;
                ZPP_LOAD_ROWS(q, q.po);
            }
        }
        u32 tw = 0xFFFFFFFFu;
        if (TOUCHC) {
            // (the bit of the nibble that was resolved while this request was in flight is not in its word yet)
            tw = touch ? (q.tw | ((q.po >> 9) == tc_word ? tc_bit : 0u)) : 0xFFFFFFFFu;
            const u32 rb = (q.po >> 4) & 31u;
            const bool ta = ((tw >> rb) & 1u) != 0, tb = ((tw >> (rb ^ 1u)) & 1u) != 0, tc = ((tw >> (rb ^ 2u)) & 1u) != 0;
            q.A = u32x4{ta ? q.A.x : 0u, ta ? q.A.y : 0u, ta ? q.A.z : 0u, ta ? q.A.w : 0u};
            q.B = u32x4{tb ? q.B.x : 0u, tb ? q.B.y : 0u, tb ? q.B.z : 0u, tb ? q.B.w : 0u};
            q.C = u32x4{tc ? q.C.x : 0u, tc ? q.C.y : 0u, tc ? q.C.z : 0u, tc ? q.C.w : 0u};
        }
        const u32 pa = q.po, pb = q.po ^ 16u, pc = q.po ^ 32u;
        const bool a1 = have1 && pa == L1.off, b1 = have1 && pb == L1.off, c1 = have1 && pc == L1.off;
        const bool a2 = FWD2 && have2 && pa == L2.off, b2 = FWD2 && have2 && pb == L2.off, c2 = FWD2 && have2 && pc == L2.off;
        // Forwarding decides on the candidates' FIRST dwords only (check byte and priority live there); the other three dwords
        // are forwarded for the chosen candidate alone: 20 selects per nibble instead of 32 (round 4).
        auto fwd1 = [](const bool f1, const u32 r1, const bool f2, const u32 r2, const u32 n) -> u32 { return f1 ? r1 : (f2 ? r2 : n); };
        const u32 Ax = fwd1(a1, L1.x, a2, L2.x, q.A.x), Bx = fwd1(b1, L1.x, b2, L2.x, q.B.x), Cx = fwd1(c1, L1.x, c2, L2.x, q.C.x);
        const u32 chk = q.chk;
        const bool ma = (Ax & 255u) == chk, mb = (Bx & 255u) == chk, mc = (Cx & 255u) == chk;
        const u32 qa = (Ax >> 8) & 255u, qb = (Bx >> 8) & 255u, qc = (Cx >> 8) & 255u;
        const bool va = qa <= qb && qa <= qc, vb = qb < qc;             // victim order (predictor.v:513-531)
        const bool hit = ma || mb || mc;
        const bool ua = ma || (!hit && va);
        const bool ub = !ua && (mb || (!hit && vb));
        Row R;
        R.off = ua ? pa : (ub ? pb : pc);
        const bool g1 = ua ? a1 : (ub ? b1 : c1), g2 = ua ? a2 : (ub ? b2 : c2);     // the chosen candidate is a row still in registers
        const u32 Rx = ua ? Ax : (ub ? Bx : Cx);
        const u32 Ry = fwd1(g1, L1.y, g2, L2.y, ua ? q.A.y : (ub ? q.B.y : q.C.y));
        const u32 Rz = fwd1(g1, L1.z, g2, L2.z, ua ? q.A.z : (ub ? q.B.z : q.C.z));
        const u32 Rw = fwd1(g1, L1.w, g2, L2.w, ua ? q.A.w : (ub ? q.B.w : q.C.w));
        R.x = hit ? Rx : chk; R.y = hit ? Ry : 0u; R.z = hit ? Rz : 0u; R.w = hit ? Rw : 0u;
        u32 poff2 = L1.off, cpo = q.po & ~63u;
        asm volatile("; order: row store after the prefetched rows are consumed" : "+v"(poff2), "+v"(cpo) : "v"(R.x), "v"(R.w));
        if (SP && claim) {
            const u32x4 z4 = {0, 0, 0, 0};
            u8 *line = tbase + cpo;
            sp_tags[claim_si] = q.key;
            *reinterpret_cast<u32x4 *>(line) = z4;
            *reinterpret_cast<u32x4 *>(line + 16) = z4;
            *reinterpret_cast<u32x4 *>(line + 32) = z4;
            *reinterpret_cast<u32x4 *>(line + 48) = z4;
        }
#ifndef ZPP_DEBUG_NO_ROWS
        if (have1) *reinterpret_cast<u32x4 *>(tbase + poff2) = u32x4{L1.x, L1.y, L1.z, L1.w};
        if (TOUCHC) {
            // the resolved row counts as written from now on (it is stored when the next nibble's rows are consumed; until
            // then every request that meets it takes it from the registers)
            const u32 nbit = 1u << ((R.off >> 4) & 31u);
            if (touch && (tw & nbit) == 0u) tb32[R.off >> 9] = tw | nbit;
            tc_word = R.off >> 9; tc_bit = nbit;
        }
#endif
        return R;
    };
    // ZPAQL.run(byte) + h[] copy (predictor.v:809-816) for the two shipped program shapes -> this component's context
    auto run_vm = [&](const u32 byte) -> u32 {
        u32 hv = 0;
        if (NCH != 2) {
            // b=c c-- *c=a d=0 (hash *d=a d++)* hash *d=a halt: H[k] = hash^(k+1) of (byte, prev)
            u32 a = byte;
            for (int k = 0; k <= ci; k++) a = (a + prev + 512u) * 773u;
            hv = a;
            prev = byte;
        } else {
            // level 1: *b=a a=0 d=0 hash b-- hash *d=a d++ b-- hash b-- hash *d=a halt, M = 4 bytes
            m4 = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8));
            u32 bb = b4;
            u32 a = 0;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            const u32 h0v = a; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            hv = (ci == 0) ? h0v : a;
            b4 -= 3u;
        }
        return hv;
    };

    Row rowA = {0, 0, 0, 0, 0}, rowB = {0, 0, 0, 0, 0};    // the byte's first / second nibble
    u32 cur_s = 0, cur_v = 0;                          // state byte, packed entry (the previous bit's update forwarded)
    i32 cur_b = 0;
    auto icm_st = [](u32 v, i32 b) -> i32 { return (i32)((u32)b << 9) | (i32)(v >> 23); };
    auto nibble_begin = [&](const Row &R) {
        cur_s = (R.x >> 8) & 255u;
        cur_v = t32[cur_s];
        cur_b = (i32)(int8_t)t8[cur_s];
    };
    u32 pi0 = 0, pi1 = 0, pi2 = 0, pi3 = 0;            // the predecessor's eight predictions of this byte (i16 each)
    u32 po0 = 0, po1 = 0, po2 = 0, po3 = 0;            // this component's

    // One bit (zpq_chain.hip's pipelined encode step, one role): next bit's entry fetched before this bit's update
    // is stored, the update forwarded in registers when the state repeats.
    auto bitstep = [&](auto kc, auto nbc) {
        constexpr int K = decltype(kc)::value;
        constexpr int NB = decltype(nbc)::value;
        constexpr int bit = (NB ? 3 : 7) - K;
        constexpr int KB = 7 - bit;
        Row &R = NB ? rowB : rowA;
        const u32 s = cur_s;
        const u32 yk = (ch >> bit) & 1u;
        u32 sA = 0, rAv = 0;
        i32 rAb = 0;
        if (K < 3) {
            u32 pair;
            if (K == 0) pair = R.x >> 16;
            else if (K == 1) pair = R.y >> ((slotn & 1u) * 16u);
            else pair = ((slotn & 2u) ? R.w : R.z) >> ((slotn & 1u) * 16u);
            sA = yk ? ((pair >> 8) & 255u) : (pair & 255u);
            rAv = t32[sA];
            rAb = (i32)(int8_t)t8[sA];
        }
        const u32 ns01 = *reinterpret_cast<const u16 *>(s_ns + s * 4);
        u32 nv, outv;
        i32 nb;
        if (IS_ICM) {
            // p = stretch(cm >> 8), carried with the entry; cm += (y*32767 - (cm >> 8)) >> 2 (predictor.v:555-563,701-709)
            const u32 cmv = cur_v & 0x7FFFFFu;
            outv = (u32)icm_st(cur_v, cur_b);
            const u32 cmn = (u32)wadd((i32)cmv, ((yk ? 32767 : 0) - (i32)(cmv >> 8)) >> 2);
            const i32 st_new = stretch_of(cmn);
            nv = cmn | (((u32)st_new & 0x1FFu) << 23);
            nb = st_new >> 9;
        } else {
            // ISSE (predictor.v:615-631,776-791)
            const i32 w0 = ((i32)(cur_v << 12)) >> 12;
            const i32 w1 = (i32)(((u32)cur_b << 12) | (cur_v >> 20));
            const u32 pw = (KB >> 1) == 0 ? pi0 : ((KB >> 1) == 1 ? pi1 : ((KB >> 1) == 2 ? pi2 : pi3));
            const i32 pin = (i32)(int16_t)(pw >> ((KB & 1) * 16));
            const i32 p = clamp2k((__mul24(w0, pin) + (w1 << 6)) >> 16);
            const i32 sq = s_squash[p + 2048];
            const i32 err = (yk ? 32767 : 0) - sq;
            const i32 nw0 = clamp512k(w0 + ((__mul24(err, pin) + (1 << 12)) >> 13));
            const i32 nw1 = clamp512k(w1 + ((err + 16) >> 5));
            nv = ((u32)nw0 & 0xFFFFFu) | ((u32)nw1 << 20);
            nb = nw1 >> 12;
            // the last component hands the coder what it needs: squash(p) (15 bits) and, in bit 15, the bit to code
            outv = IS_LAST ? (u32)sq | (yk << 15) : (u32)p;
        }
        {
            const u32 ov = (outv & 0xFFFFu) << ((KB & 1) * 16);
            if ((KB >> 1) == 0) po0 = (KB & 1) ? (po0 | ov) : ov;
            else if ((KB >> 1) == 1) po1 = (KB & 1) ? (po1 | ov) : ov;
            else if ((KB >> 1) == 2) po2 = (KB & 1) ? (po2 | ov) : ov;
            else po3 = (KB & 1) ? (po3 | ov) : ov;
        }
        t32[s] = nv;
        t8[s] = (u8)nb;
        if (K < 3) {
            const bool same = sA == s;
            cur_v = same ? nv : rAv;
            cur_b = same ? nb : rAb;
            cur_s = sA;
        }
        // next bit-history state into the row (statetable.v:75-84)
        const u32 nsv = yk ? (ns01 >> 8) : (ns01 & 255u);
        const u32 sh = (slotn & 3u) * 8u;
        const u32 dsel = (K <= 1) ? R.x : (K == 2 ? R.y : ((slotn & 4u) ? R.w : R.z));
        const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
        if (K <= 1) R.x = ins;
        else if (K == 2) R.y = ins;
        else { R.w = (slotn & 4u) ? ins : R.w; R.z = (slotn & 4u) ? R.z : ins; }
        slotn = (K == 3) ? 1u : (slotn * 2u + yk);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ch(bi): the PP byte 0 first when the flag is set, then the input
    Req reqX, reqY;                                    // in flight: a first-nibble context, a second-nibble context
    {
        const u32x4 z4 = {0, 0, 0, 0};
        reqX.A = z4; reqX.B = z4; reqX.C = z4; reqX.tags = z4; reqX.po = 0; reqX.chk = 0; reqX.key = 0; reqX.si = 0; reqX.off = 0;
        reqY = reqX;
    }
    if (S.active && total) {
        reqX = request(0u, 1u);                        // first nibble of the first byte: h = 0, c8 = 1
        if (FWD2) {
            const u32 ch0 = pp ? 0u : W.peek(0u);
            reqY = request(0u, 16u | (ch0 >> 4));
        }
    }
    // HIO: the byte loop runs in two phases with the wait for the host between them (the wait inside the loop would
    // cost every iteration; zpq_chain.hip measured 7 %).  No stage has then read beyond S.split + 16 bytes of a block.
    u32 it = 0;
    for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
    const u32 it_end = (HIO && phase == 0) ? min(S.iters, S.split) : S.iters;
    for (; it < it_end; it++) {
        const u32 bi = it - (u32)S.delay;
#ifdef ZPP_DEBUG_NO_COMP   // timing experiment only
        if (false) {
#else
        if (S.active && bi < total) {
#endif
            const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
            const u32 cb = W.byte(pos);
            ch = (pp && bi == 0) ? 0u : cb;
            const u32 chn = W.peek(pp ? bi : bi + 1u);            // the next byte (0 past the end)
            if (!IS_ICM) {
                const uint4 v = link_in[((it - 1u) & 1u) * S.bpw];
                pi0 = v.x; pi1 = v.y; pi2 = v.z; pi3 = v.w;
            }
            slotn = 1;
            rowA = consume(reqX, bi != 0, rowB, bi != 0, rowA);
            const u32 hnext = run_vm(ch);
            if (FWD2) reqX = request(hnext, 1u);                  // first nibble of the next byte
            else reqY = request(hctx, 16u | (ch >> 4));           // second nibble of this byte
            nibble_begin(rowA);
            bitstep(I0{}, I0{});
            bitstep(I1{}, I0{});
            bitstep(I2{}, I0{});
            bitstep(I3{}, I0{});
            rowB = consume(reqY, true, rowA, bi != 0, rowB);
            if (FWD2) reqY = request(hnext, 16u | (chn >> 4));    // second nibble of the next byte
            else reqX = request(hnext, 1u);
            nibble_begin(rowB);
            bitstep(I0{}, I1{});
            bitstep(I1{}, I1{});
            bitstep(I2{}, I1{});
            bitstep(I3{}, I1{});
            hctx = hnext;
            link_out[(it & 1u) * S.bpw] = make_uint4(po0, po1, po2, po3);
            if (FWD_PIN) link_pin[(it & 1u) * S.bpw] = make_uint4(pi0, pi1, pi2, pi3);
        }
        lds_barrier();
    }
    if (HIO && phase == 0 && B.gate_flag && !gate_wait(B.gate_flag)) status = ZPQ_E_INTERNAL;
    }   // phase
    // (the last nibbles' rows are not written back: the slot is re-initialised for the next block)
    if (S.lane < S.bpw) reinterpret_cast<i32 *>(lds + S.L.stat_off)[ci * S.bpw + S.lane] = status;
#undef ZPP_LOAD_ROWS
}

// ------------------------------------------------------------------ a component's stage SPLIT in two (levels 1-2)
// An ISSE stage is the pipeline's period (~100 instructions per bit against the ICM's 88 and the coder's 26 + loops), and
// half of it never touches a weight: finding the nibble's row (find_ht, forwarding, requests), walking the bit-history
// states through it, writing the next state back.  All of that is a function of the INPUT alone.  So the stage is cut:
//   * hist_loop: rows and states.  Per byte it leaves the eight states the byte's bits are predicted from (8 bytes per
//     block) in LDS;
//   * pred_loop: weights.  One iteration later it takes those states and its predecessor's predictions, predicts, trains,
//     hands its predictions on.
// Waves are placed so that the two history waves share one SIMD (no dependent LDS round trip in them: they interleave
// well) and the coder shares one with a prediction wave: k_pipe2 below.
template <int NCH, bool HIO>
__device__ __forceinline__ void hist_loop(const StageArgs &S, const int delay)
{
    const DBatch &B = *S.B;
    const DModel &M = *B.model;
    u8 *const lds = S.lds;
    const u8 *s_ns = lds + LDS_NS;
    const int ci = S.ci;
    const DComp &C = M.comp[ci];
    const u32 ht_mask = C.ht_len - 16u;
    u8 *const tbase = S.slot + C.ht_off;
    const int sizebits = C.a + 2;
    uint2 *const st_out = reinterpret_cast<uint2 *>(lds + S.L.st_off) + (size_t)ci * 2 * S.bpw + S.lane;
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u32 total = S.total;
    InWin W;
    if (S.active) W.open(S.src, S.nin, B.in_off);
    u32 prev = 0, m4 = 0, b4 = 0;
    u32 slotn = 1;
    u32 ch = 0;
#ifdef ZPP_DEBUG_NO_ROWS   // timing experiment only (wrong output): no hash-row traffic
#define ZPH_LOAD_ROWS(q_, po_) do { q_.A = u32x4{(po_), 0, 0, 0}; q_.B = q_.A; q_.C = q_.A; } while (0)
#else
#define ZPH_LOAD_ROWS(q_, po_)                                                          \
    do {                                                                                \
        q_.A = *reinterpret_cast<const u32x4 *>(tbase + (po_));                         \
        q_.B = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 16u));                 \
        q_.C = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 32u));                 \
    } while (0)
#endif
    auto request = [&](const u32 hc, const u32 c8v) -> Req {
        Req q;
        const u32 cx = hc + 16u * c8v;
        q.chk = (cx >> sizebits) & 255u;
        q.po = (cx * 16u) & ht_mask;
        q.key = 0; q.si = 0; q.off = 0; q.tags = u32x4{0, 0, 0, 0}; q.tw = 0;
        ZPH_LOAD_ROWS(q, q.po);
        return q;
    };
    // find_ht with the rows of the two nibbles finished since the request went out taken from registers (comp_loop, FWD2)
    auto consume = [&](Req q, const bool have1, const Row L1, const bool have2, const Row L2) -> Row {
        const u32 pa = q.po, pb = q.po ^ 16u, pc = q.po ^ 32u;
        const bool a1 = have1 && pa == L1.off, b1 = have1 && pb == L1.off, c1 = have1 && pc == L1.off;
        const bool a2 = have2 && pa == L2.off, b2 = have2 && pb == L2.off, c2 = have2 && pc == L2.off;
        // Forwarding decides on the candidates' FIRST dwords only (check byte and priority live there); the other three dwords
        // are forwarded for the chosen candidate alone: 20 selects per nibble instead of 32 (round 4).
        auto fwd1 = [](const bool f1, const u32 r1, const bool f2, const u32 r2, const u32 n) -> u32 { return f1 ? r1 : (f2 ? r2 : n); };
        const u32 Ax = fwd1(a1, L1.x, a2, L2.x, q.A.x), Bx = fwd1(b1, L1.x, b2, L2.x, q.B.x), Cx = fwd1(c1, L1.x, c2, L2.x, q.C.x);
        const u32 chk = q.chk;
        const bool ma = (Ax & 255u) == chk, mb = (Bx & 255u) == chk, mc = (Cx & 255u) == chk;
        const u32 qa = (Ax >> 8) & 255u, qb = (Bx >> 8) & 255u, qc = (Cx >> 8) & 255u;
        const bool va = qa <= qb && qa <= qc, vb = qb < qc;             // victim order (predictor.v:513-531)
        const bool hit = ma || mb || mc;
        const bool ua = ma || (!hit && va);
        const bool ub = !ua && (mb || (!hit && vb));
        Row R;
        R.off = ua ? pa : (ub ? pb : pc);
        const bool g1 = ua ? a1 : (ub ? b1 : c1), g2 = ua ? a2 : (ub ? b2 : c2);     // the chosen candidate is a row still in registers
        const u32 Rx = ua ? Ax : (ub ? Bx : Cx);
        const u32 Ry = fwd1(g1, L1.y, g2, L2.y, ua ? q.A.y : (ub ? q.B.y : q.C.y));
        const u32 Rz = fwd1(g1, L1.z, g2, L2.z, ua ? q.A.z : (ub ? q.B.z : q.C.z));
        const u32 Rw = fwd1(g1, L1.w, g2, L2.w, ua ? q.A.w : (ub ? q.B.w : q.C.w));
        R.x = hit ? Rx : chk; R.y = hit ? Ry : 0u; R.z = hit ? Rz : 0u; R.w = hit ? Rw : 0u;
        u32 poff2 = L1.off;
        asm volatile("; order: row store after the prefetched rows are consumed" : "+v"(poff2) : "v"(R.x), "v"(R.w));
#ifndef ZPP_DEBUG_NO_ROWS
        if (have1) *reinterpret_cast<u32x4 *>(tbase + poff2) = u32x4{L1.x, L1.y, L1.z, L1.w};
#endif
        return R;
    };
    auto run_vm = [&](const u32 byte) -> u32 {         // comp_loop
        u32 hv = 0;
        if (NCH != 2) {
            u32 a = byte;
            for (int k = 0; k <= ci; k++) a = (a + prev + 512u) * 773u;
            hv = a;
            prev = byte;
        } else {
            m4 = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8));
            u32 bb = b4;
            u32 a = 0;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            const u32 h0v = a; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
            a = (a + ((m4 >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
            hv = (ci == 0) ? h0v : a;
            b4 -= 3u;
        }
        return hv;
    };
    Row rowA = {0, 0, 0, 0, 0}, rowB = {0, 0, 0, 0, 0};
    u32 cur_s = 0;
    u32 so0 = 0, so1 = 0;                              // the byte's eight states, in coding order
    auto bitstep = [&](auto kc, auto nbc) {
        constexpr int K = decltype(kc)::value;
        constexpr int NB = decltype(nbc)::value;
        constexpr int bit = (NB ? 3 : 7) - K;
        constexpr int KB = 7 - bit;
        Row &R = NB ? rowB : rowA;
        const u32 s = cur_s;
        const u32 yk = (ch >> bit) & 1u;
        {
            const u32 sv = s << ((KB & 3) * 8);
            if (KB < 4) so0 = (KB & 3) ? (so0 | sv) : sv;
            else so1 = (KB & 3) ? (so1 | sv) : sv;
        }
        if (K < 3) {
            u32 pair;
            if (K == 0) pair = R.x >> 16;
            else if (K == 1) pair = R.y >> ((slotn & 1u) * 16u);
            else pair = ((slotn & 2u) ? R.w : R.z) >> ((slotn & 1u) * 16u);
            cur_s = yk ? ((pair >> 8) & 255u) : (pair & 255u);
        }
        const u32 ns01 = *reinterpret_cast<const u16 *>(s_ns + s * 4);
        const u32 nsv = yk ? (ns01 >> 8) : (ns01 & 255u);
        const u32 sh = (slotn & 3u) * 8u;
        const u32 dsel = (K <= 1) ? R.x : (K == 2 ? R.y : ((slotn & 4u) ? R.w : R.z));
        const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
        if (K <= 1) R.x = ins;
        else if (K == 2) R.y = ins;
        else { R.w = (slotn & 4u) ? ins : R.w; R.z = (slotn & 4u) ? R.z : ins; }
        slotn = (K == 3) ? 1u : (slotn * 2u + yk);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    Req reqX, reqY;
    {
        const u32x4 z4 = {0, 0, 0, 0};
        reqX.A = z4; reqX.B = z4; reqX.C = z4; reqX.tags = z4; reqX.po = 0; reqX.chk = 0; reqX.key = 0; reqX.si = 0; reqX.off = 0; reqX.tw = 0;
        reqY = reqX;
    }
    if (S.active && total) {
        reqX = request(0u, 1u);
        const u32 ch0 = pp ? 0u : W.peek(0u);
        reqY = request(0u, 16u | (ch0 >> 4));
    }
    i32 status = ZPQ_OK;
    u32 it = 0;
    for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
    const u32 it_end = (HIO && phase == 0) ? min(S.iters, S.split) : S.iters;
    for (; it < it_end; it++) {
        const u32 bi = it - (u32)delay;
        if (S.active && bi < total) {
            const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
            const u32 cb = W.byte(pos);
            ch = (pp && bi == 0) ? 0u : cb;
            const u32 chn = W.peek(pp ? bi : bi + 1u);
            slotn = 1;
            rowA = consume(reqX, bi != 0, rowB, bi != 0, rowA);
            const u32 hnext = run_vm(ch);
            reqX = request(hnext, 1u);                            // first nibble of the next byte
            cur_s = (rowA.x >> 8) & 255u;
            bitstep(I0{}, I0{});
            bitstep(I1{}, I0{});
            bitstep(I2{}, I0{});
            bitstep(I3{}, I0{});
            rowB = consume(reqY, true, rowA, bi != 0, rowB);
            reqY = request(hnext, 16u | (chn >> 4));              // second nibble of the next byte
            cur_s = (rowB.x >> 8) & 255u;
            bitstep(I0{}, I1{});
            bitstep(I1{}, I1{});
            bitstep(I2{}, I1{});
            bitstep(I3{}, I1{});
            st_out[(it & 1u) * S.bpw] = make_uint2(so0, so1);
        }
        lds_barrier();
    }
    if (HIO && phase == 0 && B.gate_flag && !gate_wait(B.gate_flag)) status = ZPQ_E_INTERNAL;
    }   // phase
    if (S.lane < S.bpw) reinterpret_cast<i32 *>(lds + S.L.stat_off)[ci * S.bpw + S.lane] = status;
#undef ZPH_LOAD_ROWS
}

// the counter / weights half of a stage (see hist_loop): states from its history wave, an ISSE's inputs from its predecessor
// PAIR (round 4): the wave's two halves are the weights of two ISSEs (lane = block in both halves, S.ci per lane): whether the
// stage is the chain's last one is then a per-lane fact
template <int NCH, bool HIO, bool IS_LAST, bool IS_ICM, bool PAIR = false>
__device__ __forceinline__ void pred_loop(const StageArgs &S, const int delay)
{
    const DBatch &B = *S.B;
    const Cfg &cfg = *S.cfg;
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + LDS_SQUASH);
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + LDS_STRETCH);
    auto stretch_of = [&](u32 cm) -> i32 {                              // see zpq_chain.hip
#ifdef ZPQ_STRETCH_ENDS
        u32 q = cm >> 8;
        q = min(max(q, 1u), 32767u);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
#else
        const u32 wv = s_stretch[(cm >> 12) & 2047u];                   // (an ICM counter's index stays below 32767)
        return ((i32)wv >> 16) + (i32)__popc(__builtin_amdgcn_ubfe(wv, 1u, (cm >> 8) & 15u));
#endif
    };
    const int ci = S.ci;
    const bool is_last = PAIR ? ci == NCH - 1 : IS_LAST;
    u32 *const t32 = reinterpret_cast<u32 *>(S.my + (PAIR ? (ci == 2 ? cfg.lds_off32[2] : cfg.lds_off32[1]) : cfg.lds_off32[ci]));
    u8 *const t8 = S.my + (PAIR ? (ci == 2 ? cfg.lds_off8[2] : cfg.lds_off8[1]) : cfg.lds_off8[ci]);
    const uint2 *const st_in = reinterpret_cast<const uint2 *>(lds + S.L.st_off) + (size_t)ci * 2 * S.bpw + S.lane;
    const uint4 *const link_in = reinterpret_cast<const uint4 *>(lds + S.L.link_off) + (size_t)(ci > 0 ? ci - 1 : 0) * 2 * S.bpw + S.lane;
    uint4 *const link_out = reinterpret_cast<uint4 *>(lds + S.L.link_off) + (size_t)ci * 2 * S.bpw + S.lane;
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u32 total = S.total;
    InWin W;
    if (S.active) W.open(S.src, S.nin, B.in_off);
    u32 it = 0;
    for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
    const u32 it_end = (HIO && phase == 0) ? min(S.iters, S.split) : S.iters;
    for (; it < it_end; it++) {
        const u32 bi = it - (u32)delay;
        if (S.active && bi < total) {
            const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
            const u32 cb = W.byte(pos);
            const u32 ch = (pp && bi == 0) ? 0u : cb;
            const uint2 sv = st_in[((it - 1u) & 1u) * S.bpw];
            uint4 v = make_uint4(0, 0, 0, 0);
            if (!IS_ICM) v = link_in[((it - 1u) & 1u) * S.bpw];
            u32 po0 = 0, po1 = 0, po2 = 0, po3 = 0;
            // the byte's first entry; afterwards the next bit's entry is fetched before this bit's update is stored and
            // the update is forwarded in registers when the state repeats (comp_loop's bit step, the weights half)
            u32 s = sv.x & 255u;
            u32 cur_v = t32[s];
            i32 cur_b = (i32)(int8_t)t8[s];
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const u32 yk = (ch >> (7 - kb)) & 1u;
                u32 sA = 0, rAv = 0;
                i32 rAb = 0;
                if (kb < 7) {
                    const u32 w = (kb + 1) < 4 ? sv.x : sv.y;
                    sA = (w >> (((kb + 1) & 3) * 8)) & 255u;
                    rAv = t32[sA];
                    rAb = (i32)(int8_t)t8[sA];
                }
                u32 nv, outv;
                i32 nb;
                if (IS_ICM) {
                    // p = stretch(cm >> 8), carried with the entry; cm += (y*32767 - (cm >> 8)) >> 2 (predictor.v:555-563,701-709)
                    const u32 cmv = cur_v & 0x7FFFFFu;
                    outv = (u32)((i32)((u32)cur_b << 9) | (i32)(cur_v >> 23));
                    const u32 cmn = (u32)wadd((i32)cmv, ((yk ? 32767 : 0) - (i32)(cmv >> 8)) >> 2);
                    const i32 st_new = stretch_of(cmn);
                    nv = cmn | (((u32)st_new & 0x1FFu) << 23);
                    nb = st_new >> 9;
                } else {
                    const i32 w0 = ((i32)(cur_v << 12)) >> 12;
                    const i32 w1 = (i32)(((u32)cur_b << 12) | (cur_v >> 20));
                    const u32 pw = (kb >> 1) == 0 ? v.x : ((kb >> 1) == 1 ? v.y : ((kb >> 1) == 2 ? v.z : v.w));
                    const i32 pin = (i32)(int16_t)(pw >> ((kb & 1) * 16));
                    const i32 p = clamp2k((__mul24(w0, pin) + (w1 << 6)) >> 16);   // predictor.v:615-631
                    const i32 sq = s_squash[p + 2048];
                    const i32 err = (yk ? 32767 : 0) - sq;                         // predictor.v:776-791
                    const i32 nw0 = clamp512k(w0 + ((__mul24(err, pin) + (1 << 12)) >> 13));
                    const i32 nw1 = clamp512k(w1 + ((err + 16) >> 5));
                    nv = ((u32)nw0 & 0xFFFFFu) | ((u32)nw1 << 20);
                    nb = nw1 >> 12;
                    outv = is_last ? (u32)sq | (yk << 15) : (u32)p;
                }
                {
                    const u32 ov = (outv & 0xFFFFu) << ((kb & 1) * 16);
                    if ((kb >> 1) == 0) po0 = (kb & 1) ? (po0 | ov) : ov;
                    else if ((kb >> 1) == 1) po1 = (kb & 1) ? (po1 | ov) : ov;
                    else if ((kb >> 1) == 2) po2 = (kb & 1) ? (po2 | ov) : ov;
                    else po3 = (kb & 1) ? (po3 | ov) : ov;
                }
                t32[s] = nv;
                t8[s] = (u8)nb;
                if (kb < 7) {
                    const bool same = sA == s;
                    cur_v = same ? nv : rAv;
                    cur_b = same ? nb : rAb;
                    s = sA;
                }
            }
            link_out[(it & 1u) * S.bpw] = make_uint4(po0, po1, po2, po3);
        }
        lds_barrier();
    }
    if (HIO && phase == 0 && B.gate_flag) (void)gate_wait(B.gate_flag);
    }   // phase
}

// ------------------------------------------------------------------ the MIX2 stage (levels 4-5; predictor.v:587-592,776-791)
// Component NCH mixes p[NCH-2] and p[NCH-1] with a 16-bit weight selected by (h + c8) -- with mask 255 the eight weights
// of a byte are eight DISTINCT entries, known when the byte begins: they are loaded there (after the previous byte's
// trained weights are stored: an entry may recur), trained in registers, stored when the next byte begins.  The wait
// for the loads is this wave's own; its SIMD is shared with component waves that keep issuing.
template <int NCH, bool HIO>
__device__ __forceinline__ void mix_loop(const StageArgs &S)
{
    const DBatch &B = *S.B;
    const DModel &M = *B.model;
    u8 *const lds = S.lds;
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + LDS_SQUASH);
    const DComp &C = M.comp[NCH];
    const i32 mix_rate = C.rate;
    const u32 cmask = (u32)(C.c - 1);
    u16 *const a16 = reinterpret_cast<u16 *>(S.slot + C.a16_off);
    const uint4 *const link_k = reinterpret_cast<const uint4 *>(lds + S.L.link_off) + (size_t)(NCH - 1) * 2 * S.bpw + S.lane;
    const uint4 *const link_j = reinterpret_cast<const uint4 *>(lds + S.L.link_off) + (size_t)(NCH + 1) * 2 * S.bpw + S.lane;
    uint4 *const link_out = reinterpret_cast<uint4 *>(lds + S.L.link_off) + (size_t)NCH * 2 * S.bpw + S.lane;
    const bool pp = (B.flags & ZPQ_FLAG_PP) != 0;
    const u32 total = S.total;
    InWin W;
    if (S.active) W.open(S.src, S.nin, B.in_off);
    u32 prev = 0, hctx = 0, mh_prev = 0, mch_prev = 0;
    u32 mw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 it = 0;
    for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
    const u32 it_end = (HIO && phase == 0) ? min(S.iters, S.split) : S.iters;
    for (; it < it_end; it++) {
        const u32 bi = it - (u32)NCH;
        if (S.active && bi < total) {
            const u32 pos = pp ? (bi ? bi - 1u : 0u) : bi;
            const u32 cb = W.byte(pos);
            const u32 ch = (pp && bi == 0) ? 0u : cb;
            const uint4 vk = link_k[((it - 1u) & 1u) * S.bpw], vj = link_j[((it - 1u) & 1u) * S.bpw];
            if (bi != 0) {
#pragma unroll
                for (int t = 0; t < 8; t++) a16[(mh_prev + ((1u << t) | (mch_prev >> (8 - t)))) & cmask] = (u16)mw[t];
            }
#pragma unroll
            for (int t = 0; t < 8; t++) mw[t] = a16[(hctx + ((1u << t) | (ch >> (8 - t)))) & cmask];
            mh_prev = hctx;
            mch_prev = ch;
            u32 hnext;
            {   // H[NCH] = hash^(NCH+1)(byte, previous byte)
                u32 a = ch;
                for (int k = 0; k <= NCH; k++) a = (a + prev + 512u) * 773u;
                hnext = a;
                prev = ch;
            }
            u32 po0 = 0, po1 = 0, po2 = 0, po3 = 0;
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const u32 yk = (ch >> (7 - kb)) & 1u;
                const u32 wk = (kb >> 1) == 0 ? vk.x : ((kb >> 1) == 1 ? vk.y : ((kb >> 1) == 2 ? vk.z : vk.w));
                const u32 wj = (kb >> 1) == 0 ? vj.x : ((kb >> 1) == 1 ? vj.y : ((kb >> 1) == 2 ? vj.z : vj.w));
                const i32 pk = (i32)(int16_t)(wk >> ((kb & 1) * 16)), pj = (i32)(int16_t)(wj >> ((kb & 1) * 16));
                const i32 w = (i32)mw[kb];
                const i32 p = clamp2k(wadd(wmul(w, pj), wmul(65536 - w, pk)) >> 16);
                const i32 sq = s_squash[p + 2048];
                const i32 err = (yk ? 32767 : 0) - sq;
                const i32 em = wmul(err, mix_rate) >> 5;
                i32 wn = wadd(w, wadd(wmul(em, pj - pk), 1 << 12) >> 13);
                wn = min(max(wn, 0), 65535);
                mw[kb] = (u32)wn;
                const u32 ov = (((u32)sq | (yk << 15)) & 0xFFFFu) << ((kb & 1) * 16);
                if ((kb >> 1) == 0) po0 = (kb & 1) ? (po0 | ov) : ov;
                else if ((kb >> 1) == 1) po1 = (kb & 1) ? (po1 | ov) : ov;
                else if ((kb >> 1) == 2) po2 = (kb & 1) ? (po2 | ov) : ov;
                else po3 = (kb & 1) ? (po3 | ov) : ov;
            }
            hctx = hnext;
            link_out[(it & 1u) * S.bpw] = make_uint4(po0, po1, po2, po3);
        }
        lds_barrier();
    }
    if (HIO && phase == 0 && B.gate_flag) (void)gate_wait(B.gate_flag);
    }   // phase
    // (the last byte's weights are not written back: the slot is re-initialised for the next block)
    if (S.lane < S.bpw) reinterpret_cast<i32 *>(lds + S.L.stat_off)[NCH * S.bpw + S.lane] = ZPQ_OK;
}

// ------------------------------------------------------------------ the coder stage (encoder.v:48-139)
// Its input is the last component's link: per bit squash(p) and, in bit 15, the bit to code.  Coded bytes
// go straight to the block's slab (this wave has no loads in flight that a store could hold up).
struct Coder {
    u32 low, high, opos, cap;
    u8 *dst;
    __device__ __forceinline__ void put(const u32 b)                    // Writer.put (encoder.v:76-83)
    {
        if (opos < cap) dst[opos] = (u8)b;
        opos++;
    }
    __device__ __forceinline__ void shift_out()
    {
        while ((high ^ low) < 0x1000000u) {
            put(high >> 24);
            low <<= 8; high = (high << 8) | 255u; low = low ? low : 1u;
        }
    }
};

// NST = stages in front of the coder (= its distance in bytes from the ICM); their last one's link is its input
template <int NST, bool HIO>
__device__ __forceinline__ void coder_loop_d(const StageArgs &S, Coder &X, const int delay)
{
    u8 *const lds = S.lds;
    const uint4 *const link_in = reinterpret_cast<const uint4 *>(lds + S.L.link_off) + (size_t)(NST - 1) * 2 * S.bpw + S.lane;
    const u32 total = S.total;
    u32 it = 0;
    for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
    const u32 it_end = (HIO && phase == 0) ? min(S.iters, S.split) : S.iters;
    for (; it < it_end; it++) {
        const u32 bi = it - (u32)delay;
#ifdef ZPP_DEBUG_NO_CODER   // timing experiment only
        if (false) {
#else
        if (S.active && bi < total) {
#endif
            const uint4 v = link_in[((it - 1u) & 1u) * S.bpw];
            // EOF flag: encode(0, 0) (encoder.v:108): mid = low, low = mid + 1
            X.low += 1;
            X.shift_out();
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const u32 pw = (kb >> 1) == 0 ? v.x : ((kb >> 1) == 1 ? v.y : ((kb >> 1) == 2 ? v.z : v.w));
                const u32 hv = pw >> ((kb & 1) * 16);
                const u32 p16 = (hv & 0x7FFFu) * 2u + 1u;               // encoder.v:60
                const bool y = (hv & 0x8000u) != 0;
#ifdef ZPP_DEBUG_DUMP   // debug aid: the link values instead of the coded stream
                X.put(hv & 255u); X.put((hv >> 8) & 255u);
                continue;
#endif
                const u32 mid = X.low + mul_shr16(X.high - X.low, p16);
                X.high = y ? mid : X.high;
                X.low = y ? X.low : mid + 1;
                X.shift_out();
            }
        }
        lds_barrier();
    }
    }   // phase (the coder reads no input: nothing to wait for)
}
template <int NST, bool HIO>
__device__ __forceinline__ void coder_loop(const StageArgs &S, Coder &X) { coder_loop_d<NST, HIO>(S, X, NST); }

// NCH = chain length (ICM + ISSEs), MIXT = a MIX2 follows it (levels 4-5); waves: NCH [+ 1] + the coder
template <int NCH, bool MIXT, bool SP, bool HIO>
__global__ void __launch_bounds__(64 * (NCH + (MIXT ? 2 : 1))) k_pipe(const DBatch B, const Cfg cfg, const PipeLds L)
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
    const int slot_id = wg_slot0 + lane;
    const bool lane_on = lane < bpw && slot_id < nslots;
    u8 *const slot = B.slots + (u64)(lane_on ? slot_id : wg_slot0) * M.slot_bytes;
    u8 *const my = lds + LDS_STATE + (lane_on ? lane : 0) * cfg.lds_per_block;
    u32 *const misc = reinterpret_cast<u32 *>(lds + L.misc_off);
    const int wg_slots = min(bpw, nslots - wg_slot0);                  // slots this workgroup owns (> 0)

    for (int base = wg_slot0; base < B.nblocks; base += nslots) {       // rounds: slot s codes blocks s, s + nslots, ...
        const int blk = base + lane;
        const bool active = lane_on && blk < B.nblocks;
        const int nact = min(wg_slots, B.nblocks - base);               // active slots are the first nact of the workgroup
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
            if (MIXT) {                                                 // a16[] = 32768 (predictor.v:396)
                __syncthreads();                                        // (behind the zero-fill above, which other waves did)
                const DComp &cm2 = M.comp[NCH];
                const u32 words = (cm2.a16_len + 1) / 2;
                for (int b = 0; b < nact; b++) {
                    u32 *w = reinterpret_cast<u32 *>(B.slots + (u64)(wg_slot0 + b) * M.slot_bytes + cm2.a16_off);
                    for (u32 i = tid; i < words; i += nthr) w[i] = 0x80008000u;
                }
            }
            if (tid == 0) *misc = 0u;
        }
        __syncthreads();

        StageArgs S;
        S.B = &B; S.cfg = &cfg; S.lds = lds; S.ci = wave; S.delay = wave; S.lane = lane; S.bpw = bpw; S.active = active;
        S.slot = slot; S.my = my; S.L = L; S.blk = (u32)blk;
        S.src = active ? B.in + B.in_off[blk] : B.in;
        S.nin = active ? (u32)(B.in_off[blk + 1] - B.in_off[blk]) : 0u;
        S.dst = active ? B.out + B.out_off[blk] : B.out;
        S.cap = active ? (u32)(B.out_off[blk + 1] - B.out_off[blk]) : 0u;
        S.total = active ? S.nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u) : 0u;
        if (wave == 0) atomicMax(misc, S.total);
        __syncthreads();
        constexpr int NST = NCH + (MIXT ? 1 : 0);                       // stages in front of the coder
        S.iters = *misc + (u32)NST;
        // (the ICM is the stage furthest ahead: in iteration `it` it is at input byte <= it and holds <= 12 bytes more)
        S.split = (HIO && B.gate_flag) ? (B.gate_pos > 64u ? B.gate_pos - 64u : 0u) : S.iters;

        Coder X;
        X.low = 1; X.high = 0xFFFFFFFFu; X.opos = 0; X.cap = S.cap; X.dst = S.dst;
        if (wave == 0) comp_loop<NCH, SP, HIO, true, false, false>(S);
        else if (wave < NCH - 1) comp_loop<NCH, SP, HIO, false, false, false>(S);
        else if (wave == NCH - 1) comp_loop<NCH, SP, HIO, false, !MIXT, MIXT>(S);
        else if (MIXT && wave == NCH) mix_loop<NCH, HIO>(S);
        else coder_loop<NST, HIO>(S, X);
        __syncthreads();
        if (wave == NST && active) {
            // ---- segment end: compress(-1) + flush (encoder.v:101-105,130-139)
            X.high = X.low;                                           // encode(1, 0): mid = low, high = mid
            X.shift_out();
            for (int sft = 24; sft >= 0; sft -= 8) X.put(X.high >> sft);
            const i32 *stat = reinterpret_cast<const i32 *>(lds + L.stat_off);
            i32 st = ZPQ_OK;
            for (int c = 0; c < NST; c++) { const i32 sc = stat[c * bpw + lane]; st = st ? st : sc; }
            if (X.opos > X.cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
            B.out_len[blk] = X.opos;
            B.status[blk] = st;
        }
        __syncthreads();
    }
}

// The encoder of the dense short chains (levels 1-2) with every stage split into a history wave and a counter / weights
// wave (hist_loop / pred_loop): 2 * NCH + 1 waves.  Wave -> role, chosen for the SIMD a wave lands on (wave w runs on SIMD
// w & 3) so that no SIMD issues more than ~92 instructions per bit (one wave per stage: the ISSE's 100, the ICM's 100):
//   NCH = 3 (level 2): 0 hist(0) | 1 hist(2) | 2 pred(1) | 3 pred(2) | 4 hist(1) [SIMD 0] | 5 pred(0) [SIMD 1] | 6 coder [SIMD 2]
//   NCH = 2 (level 1): 0 coder | 1 hist(0) | 2 hist(1) | 3 pred(1) | 4 pred(0) [SIMD 0, with the coder]
// In iteration `it` hist(c) works on byte it - c, pred(c) on byte it - c - 1, the coder on byte it - NCH - 1.
template <int NCH, bool HIO>
__global__ void __launch_bounds__(64 * (2 * NCH + 1)) k_pipe2(const DBatch B, const Cfg cfg, const PipeLds L)
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
    const int wave = tid >> 6;
    // role of this wave: 1 = history of component comp, 2 = counter / weights of component comp, 3 = coder
    // (cfg.split_enc: four bits per wave, role id 0 H0 | 1 P0 | 2 H1 | 3 P1 | 4 H2 | 5 P2 | 6 coder | 8 + c: component c whole |
    //  12 H1+H2 | 13 P1+P2: a PAIRED wave -- both ISSEs of a three-component chain run the same code, so lanes 0-31 are
    //  component 1 and lanes 32-63 component 2 of the workgroup's (at most 32) blocks: five roles instead of seven)
    const int rid = (int)(((u64)cfg.split_enc >> (4 * wave)) & 15u);
    const bool paired = NCH == 3 && (rid == 12 || rid == 13);
    const int lane = paired ? (tid & 31) : (tid & 63);
    const int bpw = cfg.blocks_per_wg;
    const int wg_slot0 = blockIdx.x * bpw;
    const int nslots = B.nslots;
    const int slot_id = wg_slot0 + lane;
    const bool lane_on = lane < bpw && slot_id < nslots;
    u8 *const slot = B.slots + (u64)(lane_on ? slot_id : wg_slot0) * M.slot_bytes;
    u8 *const my = lds + LDS_STATE + (lane_on ? lane : 0) * cfg.lds_per_block;
    u32 *const misc = reinterpret_cast<u32 *>(lds + L.misc_off);
    const int wg_slots = min(bpw, nslots - wg_slot0);
    const int role = rid == 6 ? 3 : (paired ? (rid == 12 ? 1 : 2) : (rid >= 8 ? 0 : ((rid & 1) ? 2 : 1)));
    const int comp = rid == 6 ? 0 : (paired ? 1 + ((tid >> 5) & 1) : (rid >= 8 ? rid - 8 : (rid >> 1)));
    // delays: a whole stage works one iteration behind its predecessor's output, a split one's weights wave too, its history
    // wave one iteration ahead of that; split_mask bit c = component c is split
    int dH[3] = {0, 0, 0}, dP[3] = {0, 0, 0}, dC = 0;
    {
        const int split_mask = (int)((cfg.split_enc >> 32) & 7u);
        int o = -1;
        for (int c = 0; c < NCH; c++) {
            if ((split_mask >> c) & 1) { dP[c] = o + 1 < 1 ? 1 : o + 1; dH[c] = dP[c] - 1; o = dP[c]; }
            else { dP[c] = o + 1; dH[c] = dP[c]; o = dP[c]; }
        }
        dC = o + 1;
    }

    for (int base = wg_slot0; base < B.nblocks; base += nslots) {
        const int blk = base + lane;
        const bool active = lane_on && blk < B.nblocks;
        const int nact = min(wg_slots, B.nblocks - base);
        {   // ---- Predictor.init + ZPAQL.clear for the round's blocks (k_pipe)
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
                    const u32 cmi = B.img[i];
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
            if (tid == 0) *misc = 0u;
        }
        __syncthreads();

        StageArgs S;
        S.B = &B; S.cfg = &cfg; S.lds = lds; S.ci = comp; S.lane = lane; S.bpw = bpw; S.active = active;
        S.slot = slot; S.my = my; S.L = L; S.blk = (u32)blk;
        S.src = active ? B.in + B.in_off[blk] : B.in;
        S.nin = active ? (u32)(B.in_off[blk + 1] - B.in_off[blk]) : 0u;
        S.dst = active ? B.out + B.out_off[blk] : B.out;
        S.cap = active ? (u32)(B.out_off[blk + 1] - B.out_off[blk]) : 0u;
        S.total = active ? S.nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u) : 0u;
        if (wave == 0) atomicMax(misc, S.total);
        __syncthreads();
        S.iters = *misc + (u32)dC;
        S.split = (HIO && B.gate_flag) ? (B.gate_pos > 64u ? B.gate_pos - 64u : 0u) : S.iters;
        S.delay = comp == 0 ? dP[0] : (comp == 1 ? dP[1] : dP[2]);

        Coder X;
        X.low = 1; X.high = 0xFFFFFFFFu; X.opos = 0; X.cap = S.cap; X.dst = S.dst;
        if (role == 0) {
            if (comp == 0) comp_loop<NCH, false, HIO, true, false, false>(S);
            else if (comp == NCH - 1) comp_loop<NCH, false, HIO, false, true, false>(S);
            else comp_loop<NCH, false, HIO, false, false, false>(S);
        }
        else if (role == 1) hist_loop<NCH, HIO>(S, comp == 0 ? dH[0] : (comp == 1 ? dH[1] : dH[2]));
        else if (role == 2) {
            if (paired) pred_loop<NCH, HIO, false, false, true>(S, comp == 2 ? dP[2] : dP[1]);
            else if (comp == 0) pred_loop<NCH, HIO, false, true>(S, dP[comp]);
            else if (comp == NCH - 1) pred_loop<NCH, HIO, true, false>(S, dP[comp]);
            else pred_loop<NCH, HIO, false, false>(S, dP[comp]);
        }
        else coder_loop_d<NCH, HIO>(S, X, dC);
        __syncthreads();
        if (role == 3 && active) {
            X.high = X.low;                                           // encode(1, 0): mid = low, high = mid
            X.shift_out();
            for (int sft = 24; sft >= 0; sft -= 8) X.put(X.high >> sft);
            const i32 *stat = reinterpret_cast<const i32 *>(lds + L.stat_off);
            i32 st = ZPQ_OK;
            for (int c = 0; c < NCH; c++) { const i32 sc = stat[c * bpw + lane]; st = st ? st : sc; }
            if (X.opos > X.cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
            B.out_len[blk] = X.opos;
            B.status[blk] = st;
        }
        __syncthreads();
    }
}

}  // namespace zpqp

// ------------------------------------------------------------------ host side
using zpqc::Cfg;

static bool pipe_layout(const Cfg &cfg, int bpw, zpqp::PipeLds *L, size_t *lds_bytes)
{
    const int nch = cfg.nch_spec;
    const int nlinks = nch + (cfg.has_mix2 ? 2 : 0);         // + the MIX2's own, + the last ISSE's input handed on to it
    size_t off = (size_t)zpqc::LDS_STATE + (size_t)bpw * cfg.lds_per_block;
    off = (off + 15) & ~(size_t)15;
    L->link_off = (int32_t)off; off += (size_t)nlinks * 2 * bpw * 16;
    L->stat_off = (int32_t)off; off += (size_t)(nch + 1) * bpw * 4;
    L->misc_off = (int32_t)off; off += 16;
    L->st_off = (int32_t)off;
    if (cfg.split_enc) off += (size_t)nch * 2 * bpw * 8;
    *lds_bytes = off;
    return off <= 160 * 1024;
}

// The wave-pipelined encoder exists for the dense and line-store chains of levels 1-5.  ZPQ_ENC_PIPE=0 keeps the
// lane-per-component encoder (tests compare the two).
// (a batch of fewer than 12 resident blocks stays with the lane-per-component encoder: see zpq_launch_pipe)
// 1 = this build's encoder reads dense hash rows through "touched" bitmaps (zpq_touch_layout)
extern "C" int zpq_pipe_touch(void)
{
#ifdef ZPP_TOUCH
    return 1;
#else
    return 0;
#endif
}

extern "C" int zpq_pipe_applies(const DModel *M, int blocks_per_wg, int nslots)
{
    if (nslots < 12) return 0;
    const char *ev = getenv("ZPQ_ENC_PIPE");
    if (ev && atoi(ev) == 0) return 0;
    Cfg cfg;
    if (!zpq_chain_build_cfg(M, &cfg)) return 0;
    if (cfg.has_mix2 ? !(cfg.nch_spec == 6 || cfg.nch_spec == 8) : !(cfg.nch_spec == 2 || cfg.nch_spec == 3 || cfg.nch_spec == 5)) return 0;
    if (blocks_per_wg < 1 || blocks_per_wg > 64 || blocks_per_wg > cfg.blocks_per_wg) return 0;
    zpqp::PipeLds L;
    size_t lds = 0;
    return pipe_layout(cfg, blocks_per_wg, &L, &lds) ? 1 : 0;
}

// A wave -> role order for k_pipe2 is usable when every component c < nch is there exactly once, either whole (8 + c) or as
// its history (2c) + weights (2c + 1) pair, and there is exactly one coder (6); anything else would index past the model's
// components or leave out_len / status unwritten.
static bool split_order_valid(const char *order, int nch)
{
    int whole[3] = {0, 0, 0}, hist[3] = {0, 0, 0}, pred[3] = {0, 0, 0}, coder = 0, nw = 0;
    for (const char *q = order; *q; q++, nw++) {
        const int id = *q == 'a' ? 10 : *q == 'c' ? 12 : *q == 'd' ? 13 : (*q >= '0' && *q <= '9') ? *q - '0' : -1;
        if (id < 0 || id == 7) return false;
        if (id == 6) { coder++; continue; }
        if (id >= 12) {                                      // both ISSEs of a three-component chain on one wave's two halves
            if (nch != 3) return false;
            if (id == 12) { hist[1]++; hist[2]++; } else { pred[1]++; pred[2]++; }
            continue;
        }
        const int c = id >= 8 ? id - 8 : id >> 1;
        if (c >= nch || c > 2) return false;
        if (id >= 8) whole[c]++; else if (id & 1) pred[c]++; else hist[c]++;
    }
    if (coder != 1 || nw > 2 * nch + 1) return false;
    for (int c = 0; c < nch; c++)
        if (!((whole[c] == 1 && !hist[c] && !pred[c]) || (!whole[c] && hist[c] == 1 && pred[c] == 1))) return false;
    return true;
}

// (internal, for the CPU tests: is `order` a complete wave order for a chain of nch components?)
extern "C" int zpq_pipe_split_order_valid(const char *order, int nch) { return order && nch >= 1 && nch <= 3 && split_order_valid(order, nch) ? 1 : 0; }

extern "C" int zpq_launch_pipe(const DBatch *B, const DModel *hostM, int nwg, int blocks_per_wg, hipStream_t stream, const char **name_out)
{
    if (name_out) *name_out = "k_pipe<encode>";
    Cfg cfg;
    if (!zpq_chain_build_cfg(hostM, &cfg)) return ZPQ_E_INTERNAL;
    if (!zpq_pipe_applies(hostM, blocks_per_wg, B->nslots)) return ZPQ_E_INTERNAL;
    // Slot s is lane s % B of workgroup s / B for ANY B, so this kernel regroups the plan's slots: at least 16 per
    // workgroup where the batch has them, and evenly, so that no workgroup is left with a handful.  A wave that LIVES
    // on <= 8 active lanes issues VALU code at a third of its speed on this GPU (tools/micro/lanes.hip; one workgroup
    // alone: 8 blocks 250 ms, 12 blocks 112 ms, 32 blocks 108 ms; same instruction count, ten times the
    // SQ_WAIT_INST_ANY cycles), and here a lane is a block.
    {
        const int nslots = B->nslots;
        const int cap = cfg.blocks_per_wg;
        int per = blocks_per_wg < 16 ? (cap < 16 ? cap : 16) : blocks_per_wg;
        { const char *ev = getenv("ZPQ_PIPE_PER"); if (ev && atoi(ev) >= 9 && atoi(ev) <= cap) per = atoi(ev); }   // experiments
        nwg = (nslots + per - 1) / per;
        blocks_per_wg = (nslots + nwg - 1) / nwg;
    }
    const bool hio = B->gate_flag != nullptr;
    if (hio && cfg.sparse) return ZPQ_E_INTERNAL;            // (striped uploads: dense levels 1-2, see zpq_chain_has_hio)
    // Dense level 1: every stage split into a history wave and a counter / weights wave (k_pipe2): 136 -> 101 ms for 4096
    // blocks.  Level 2 has seven such roles for four SIMDs; every order and every mix of split and whole stages that was
    // tried ran at 122-137 ms against 123 for k_pipe (two waves sharing a SIMD issue slower together than the instruction
    // counts promise), so it stays on k_pipe.  ZPQ_ENC_SPLIT=0: never split; ZPQ_ENC_SPLIT=<role ids, wave 0 first>: that
    // order (0 H0 | 1 P0 | 2 H1 | 3 P1 | 4 H2 | 5 P2 | 6 coder | 8 W0 | 9 W1 | a W2 = whole stages | c H1+H2 | d P1+P2 = both
    // ISSEs of level 2 on the two halves of one wave), levels 1-2.
    {
        const char *ev = getenv("ZPQ_ENC_SPLIT");
        cfg.split_enc = 0;
        if (!cfg.sparse && !cfg.has_mix2 && (cfg.nch_spec == 2 || cfg.nch_spec == 3) && !(ev && atoi(ev) == 0 && strlen(ev) == 1)) {
            // (level 2, round 4: both ISSEs on the two halves of one wave, "60cd1" -- five roles as at level 1 -- takes the floor
            //  without row traffic from 113.7 to 84.8 ms, but with the rows it runs at the memory system's rate for random lines
            //  read and written back: 119.4 / 123.7-133 / 126.3 ms against 122.7 / 121-131 / 123.7 for k_pipe on three boxes.
            //  Opt-in: ZPQ_ENC_SPLIT=60cd1; EXPERIMENTS.md R4.10)
            const char *order = cfg.nch_spec == 2 ? "60231" : nullptr;
            if (ev && strlen(ev) >= 4) {
                const bool has_pair = strchr(ev, 'c') || strchr(ev, 'd');   // (a paired wave holds 2 x 32 blocks at most)
                if (split_order_valid(ev, cfg.nch_spec) && !(has_pair && blocks_per_wg > 32)) order = ev;
                else fprintf(stderr, "[zpaq_hip] ZPQ_ENC_SPLIT=%s is not a complete wave order for %d components: ignored\n", ev, cfg.nch_spec);
            }
            if (order) {
                uint64_t v = 0;
                int mask = 0, nw = 0;
                for (int w = 0; order[w]; w++, nw++) {
                    const int id = order[w] == 'a' ? 10 : order[w] == 'c' ? 12 : order[w] == 'd' ? 13 : order[w] - '0';
                    v |= (uint64_t)id << (4 * w);
                    if (id < 6) mask |= 1 << (id >> 1);
                    else if (id >= 12) mask |= 6;
                }
                cfg.split_enc = (int64_t)(v | ((uint64_t)mask << 32) | ((uint64_t)nw << 40) | (1ull << 48));
            }
        }
    }
    cfg.blocks_per_wg = blocks_per_wg;
    zpqp::PipeLds L;
    size_t lds = 0;
    if (!pipe_layout(cfg, blocks_per_wg, &L, &lds)) return ZPQ_E_INTERNAL;
    // A small batch has few blocks per workgroup and would fit several workgroups into one CU's LDS -- whose waves
    // would then share SIMDs while other CUs idle (measured: 1024 blocks 254 ms instead of 114).  Asking for more than
    // half of the LDS keeps it at one workgroup per CU, one wave per SIMD.
    { const char *ev = getenv("ZPQ_PIPE_SHARE"); if (!(ev && atoi(ev) == 1) && lds < 81 * 1024) lds = 81 * 1024; }
#define ZPP_LAUNCH(N, MX, SPv, HIOv)                                                                                 \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void *)zpqp::k_pipe<N, MX, SPv, HIOv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((zpqp::k_pipe<N, MX, SPv, HIOv>), dim3(nwg), dim3(64 * ((N) + ((MX) ? 2 : 1))), lds, stream, *B, cfg, L); \
    } while (0)
    if (hio && !(cfg.nch_spec == 2 || cfg.nch_spec == 3)) return ZPQ_E_INTERNAL;
#define ZPP_LAUNCH2(N, HIOv)                                                                                         \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void *)zpqp::k_pipe2<N, HIOv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((zpqp::k_pipe2<N, HIOv>), dim3(nwg), dim3(64 * (int)((cfg.split_enc >> 40) & 15)), lds, stream, *B, cfg, L);  \
    } while (0)
    if (cfg.split_enc) {
        if (name_out) *name_out = "k_pipe2<encode>";
        if (cfg.nch_spec == 2) { if (hio) ZPP_LAUNCH2(2, true); else ZPP_LAUNCH2(2, false); }
        else { if (hio) ZPP_LAUNCH2(3, true); else ZPP_LAUNCH2(3, false); }
        return ZPQ_OK;
    }
#undef ZPP_LAUNCH2
    switch (cfg.nch_spec) {
    case 2: if (cfg.sparse) ZPP_LAUNCH(2, false, true, false); else if (hio) ZPP_LAUNCH(2, false, false, true); else ZPP_LAUNCH(2, false, false, false); break;
    case 3: if (cfg.sparse) ZPP_LAUNCH(3, false, true, false); else if (hio) ZPP_LAUNCH(3, false, false, true); else ZPP_LAUNCH(3, false, false, false); break;
    case 5: if (cfg.sparse) ZPP_LAUNCH(5, false, true, false); else ZPP_LAUNCH(5, false, false, false); break;
    case 6: if (cfg.sparse) ZPP_LAUNCH(6, true, true, false); else ZPP_LAUNCH(6, true, false, false); break;       /* level 4 */
    case 8: if (cfg.sparse) ZPP_LAUNCH(8, true, true, false); else ZPP_LAUNCH(8, true, false, false); break;       /* level 5 */
    default: return ZPQ_E_INTERNAL;
    }
#undef ZPP_LAUNCH
    return ZPQ_OK;
}
