// zpq_chain.hip -- the LDS-resident kernel for "chain" models: component 0 is an
// ICM, every further component is an ISSE fed by its predecessor (j = i-1) and
// the last may be a MIX2.  That is every level the reference ships
// (levels.v:53-375: level 1 = ICM+ISSE ... level 5 = ICM+7xISSE+MIX2).
//
// Mapping (gfx950, wave64):
//   * one ZPAQ block = one GROUP of G lanes inside a DPP row (G = 8 for models with <= 8
//     components, else 16); a wave carries 64/G blocks, a workgroup up to 32 blocks (= one CU's
//     LDS).  Lane c of a group owns component c.
//   * per block in LDS: the ICM's 256 entries (23-bit cm + its 12-bit stretch = u32 + u8) and each
//     ISSE's 256 weight pairs packed to 20+20 bits (u32 + u8): 3.75 KiB per level-2 block, 32 blocks
//     per CU.  Shared per workgroup in LDS: squash (u16[4096]), the state table ns[1024] and an
//     8.5 KiB packing of the 64 KiB stretch table.
//   * per block in HBM (its state slot): the hash tables (64*2^sizebits bytes per component, or a
//     compact line store for levels 4-5), M/H for the ZPAQL VM, MIX2 weights.  A nibble's
//     bit-history row (16 B: check byte + 15 states) is fetched as h0 / h0^16 / h0^32 from ONE
//     64-byte line AHEAD of the nibble that uses it, kept in 4 VGPRs for the nibble's four bits
//     (the nibble is unrolled so the dword holding the slot is a compile-time choice), and written
//     back once (predictor.v:495-532,558-563,619-622,704,790).
//   * the prediction chain p0 -> p1 -> ... is handed lane-to-lane with DPP row_shr:1; the decoded
//     bit goes back down the row by DPP (one quad_perm for <= 4 components); MIX2 inputs by
//     ds_bpermute.
//   * the arithmetic coder (encoder.v:48-89 / decoder.v:73-118) runs on the lane that owns the
//     last component.
//   * encode: software-pipelined bit step (next bit's entry fetched before this bit's update,
//     forwarded on a state match); decode: the next entry is fetched the moment the bit is known.
// All arithmetic is integer and reproduces V's 32-bit wrap/arithmetic-shift
// semantics; results are bit-identical to zpq_generic.hip and the CPU oracle.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_vm.h"
#include "zpq_host.h"
#include "zpq_chain_cfg.h"

// Measured on MI355X: the pipelined step wins for encode, for decode the step with
// fewer instructions wins (both-candidate speculation was slower every time it was tried).
#ifndef ZPQ_CHAIN_G_DEFAULT
#define ZPQ_CHAIN_G_DEFAULT 8
#endif
#ifndef ZPQ_CHAIN_SPEC_ENC
#define ZPQ_CHAIN_SPEC_ENC 1
#endif
#ifndef ZPQ_CHAIN_SPEC_DEC
#define ZPQ_CHAIN_SPEC_DEC 0
#endif

namespace zpqc {

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
// The MIX2's products in the specialised decoders: a weight (<= 65536) or an error term (|err| < 2^15, rate < 2^8, |em| < 2^18) times a
// prediction or a difference of two (|p| <= 2048, |pj - pk| < 2^12): every operand fits 24 signed bits and no product leaves 32 -- the
// full-rate 24-bit multiply is exact where V's wrapping 32-bit multiply (a quarter-rate v_mul_lo_u32 / v_mad_u64_u32) is.
__device__ __forceinline__ i32 mul24x(i32 a, i32 b) { return __mul24(a, b); }
__device__ __forceinline__ i32 clamp2k(i32 x) { return min(max(x, -2048), 2047); }
// (u64(range) * p16) >> 16 for p16 < 2^16 (encoder.v:60-61, decoder.v:86-87) without the 64-bit multiply:
// range = hi * 2^16 + lo  =>  hi * p16 + ((lo * p16) >> 16), both products of 16-bit operands (two full-rate v_mul_u32_u24)
__device__ __forceinline__ uint32_t mul_shr16(uint32_t range, uint32_t p16)
{
    return (uint32_t)__umul24(range >> 16, p16) + ((uint32_t)__umul24(range & 0xFFFFu, p16) >> 16);   // (__umul24 returns int)
}
__device__ __forceinline__ i32 clamp512k(i32 x) { return min(max(x, -262144), 262143); }

// value of lane (li-1) of the same 16-lane row; lane 0 of a row gets `self`
__device__ __forceinline__ i32 row_shr1(i32 self)
{
    return __builtin_amdgcn_update_dpp(self, self, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
}
// broadcast from lane `src` (0..15) of the own row
__device__ __forceinline__ i32 row_bcast(i32 v, int src_lane_abs)
{
    return __builtin_amdgcn_ds_bpermute(src_lane_abs << 2, v);
}

__device__ __forceinline__ u32 row_byte(const uint4 &r, u32 slot)
{
    const u32 d = (slot & 8) ? ((slot & 4) ? r.w : r.z) : ((slot & 4) ? r.y : r.x);
    return (d >> ((slot & 3) * 8)) & 255u;
}
__device__ __forceinline__ void row_set(uint4 &r, u32 slot, u32 v)
{
    const u32 sh = (slot & 3) * 8;
    const u32 m = ~(255u << sh), b = v << sh;
    const u32 q = slot >> 2;
    r.x = (q == 0) ? ((r.x & m) | b) : r.x;
    r.y = (q == 1) ? ((r.y & m) | b) : r.y;
    r.z = (q == 2) ? ((r.z & m) | b) : r.z;
    r.w = (q == 3) ? ((r.w & m) | b) : r.w;
}

// Programs that are not one of the recognised shapes run through the shared interpreter
// (zpq_vm.h) on lane 0 of the group, with M/H/R in the block's HBM slot.
using zpqvm::Vm;
using zpqvm::vm_run;

// One coded bit for every lane of the group.  K = bit index inside the nibble
// (0..3) selects at compile time which dword(s) of the 16-byte row can hold the
// bit-history slot: slot 1 | 2..3 | 4..7 | 8..15 (predictor.v:817-823).
struct BitCtx {
    u32 r0, r1, r2, r3;      // the nibble's bit-history row (byte 0 = check)
    u32 slot;                // hmap4 & 15
    u32 c8;
    u32 low, high, code, opos, ipos;
};

// SPEC = software-pipelined bit step (next bit's table entries fetched early, update
// forwarded in registers); !SPEC = plain read-predict-update per bit (fewer instructions).
// NCH > 0: chain length (ICM + ISSEs) known at compile time, MIXT = a MIX2 follows it (levels
// 1-3: 2/3/5 without, level 4: 6 with, level 5: 8 with): straight-line chain and broadcast.
// NCH == 0: any chain model, runtime loops.
// GG = lanes per ZPAQ block (16, or 8 when the model has <= 8 components).
// SP = some hash table of the model lives in a compact line store (levels 1 and 3-5; level 2's tables are smaller
// than a store and stay dense).  Only SP kernels carry the store's code: its collision walk -- a divergent loop
// with global loads inside take_prefetched -- cost the dense level-2 encode 17 % by its mere presence.
// HIO = the launch takes part in a striped host transfer (host_pipeline): the encoder waits ONCE, between two phases of
// its byte loop, for the rest of its input; the decoder reports ONCE, there, that the first stripe of its output is
// stored.  Only these instantiations carry that code: the mere presence of the wait inside the byte loop cost the
// level-2 encoder 7 % (218 vs 204 ms), more than the overlap gains.
template <bool DEC, bool SPEC, int NCH, bool MIXT, int GG, bool SP, bool HIO = false>
__global__ void __launch_bounds__(64 * MAXW) k_chain(const DBatch B, const Cfg cfg)
{
    constexpr int G = GG;            // shadows zpqc::G inside the kernel
    constexpr int BPW = 64 / GG;
    extern __shared__ __align__(16) u8 lds[];
    const DModel &M = *B.model;
    const int tid = threadIdx.x, nthr = blockDim.x;

    // ---- shared read-only tables -> LDS
    // Level 1's decoders (two components, at most 32 blocks of 2.6 KiB per workgroup) have the LDS to spare for the WHOLE
    // stretch table, 32768 x i16: one lookup instead of the sixteen instructions that unpack the 8.5 KiB form.
#ifdef ZPQ_NO_DST
    constexpr bool DST = false;
#else
    constexpr bool DST = DEC && NCH == 2;
#endif
    constexpr int L_SQUASH = DST ? LDS_DST_BYTES : LDS_SQUASH;
    constexpr int L_NS = L_SQUASH + 4096 * 2, L_STATE = L_NS + 1024;
    static_assert(DST || L_STATE == LDS_STATE, "layout");
    {
        if constexpr (DST) {
            u32 *st = reinterpret_cast<u32 *>(lds);
            const u32 *src = reinterpret_cast<const u32 *>(B.stretch);
            for (int i = tid; i < 16384; i += nthr) st[i] = src[i];
            __syncthreads();
            if (tid == 0) reinterpret_cast<int16_t *>(lds)[0] = B.stretch[1];              // stretch(0) = stretch(1) (predictor.v:205-214)
        } else {
            u32 *st = reinterpret_cast<u32 *>(lds + LDS_STRETCH);
            for (int i = tid; i < 2048 + 128; i += nthr) st[i] = B.stretch_c[i];
        }
        u16 *sq = reinterpret_cast<u16 *>(lds + L_SQUASH);
        for (int i = tid; i < 4096; i += nthr) sq[i] = (u16)B.squash[min(max(i - 1, 0), 4093)];   // entry p + 2048 = squash(p): no clamp in the bit loop (|p| <= 2048)
        u8 *ns = lds + L_NS;
        for (int i = tid; i < 1024; i += nthr) ns[i] = B.ns[i];
    }
    __syncthreads();
    const u32 *s_stretch = reinterpret_cast<const u32 *>(lds + LDS_STRETCH);
    const u16 *s_squash = reinterpret_cast<const u16 *>(lds + L_SQUASH);
    const u8 *s_ns = lds + L_NS;

    // stretch(cm >> 8) (predictor.v:205-214) from the 8.5 KiB LDS packing, or from the whole table (DST; a counter is below
    // 2^23, garbage on lanes that hold none is cut to the table); `cm` may be any u32
    auto stretch_of = [&](u32 cm) -> i32 {
        if constexpr (DST) return reinterpret_cast<const int16_t *>(lds)[(cm >> 8) & 32767u];
#ifdef ZPQ_STRETCH_ENDS
        u32 q = cm >> 8;
        q = min(max(q, 1u), 32767u);
        const u32 wv = s_stretch[q >> 4];
        const u32 ei = q < 64u ? q : (q - 32704u + 64u);
        const i32 endv = (i32)(int16_t)s_stretch[2048 + (ei & 127u)];
        const i32 midv = (i32)(int16_t)(wv >> 16) + __popc(wv & ((2u << (q & 15u)) - 1u) & 0xFFFEu);
        return (q < 64u || q >= 32704u) ? endv : midv;
#else
        // The table rises by at most 1 per entry everywhere but at its last one (stretch(32767) = 2047 after 375; checked
        // exhaustively where the packing is built, zpq_model.cpp) -- and no ICM counter gets there: cminit's largest is
        // 31987 << 8 (checked there too), and an update adds (32767 - q) >> 2 to cm = 256 q + r, which cannot carry q to 32767 (that needs
        // 255 + d / 4 >= 256 d for d = 32767 - q >= 1).  So: word q >> 4 = base + 15 step bits, eight instructions;
        // word 0's base is stretch(1), which is also the reference's stretch(0) (predictor.v:205-214).
        const u32 wv = s_stretch[(cm >> 12) & 2047u];
        return ((i32)wv >> 16) + (i32)__popc(__builtin_amdgcn_ubfe(wv, 1u, (cm >> 8) & 15u));
#endif
    };
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = lane / G, li = lane % G;
    const int row_base = lane & ~(G - 1);
    const int bslot = wave * BPW + grp;
    const int slot_id = blockIdx.x * cfg.blocks_per_wg + bslot;
    const int nslots = B.nslots;                       // may be less than gridDim.x * blocks_per_wg
    u8 *slot = B.slots + (u64)slot_id * M.slot_bytes;
    u8 *my = lds + L_STATE + bslot * cfg.lds_per_block;

    const int n = NCH ? NCH + (MIXT ? 1 : 0) : cfg.n;
    const int last = n - 1;
    const int nisse_end = NCH ? NCH : cfg.nisse_end;
    const bool has_mix2 = NCH ? MIXT : (cfg.has_mix2 != 0);
    // Decode of the short chains (levels 1-2, dense tables): TWO HYPOTHESES per block.  The block's eight lanes are two
    // copies of the chain -- lanes 0..3 assume the bit being decoded is 0, lanes 4..7 assume it is 1 -- and each copy
    // does the whole update for ITS outcome (next bit-history state and its table entry, trained weights / counter
    // with its stretch, the next nibble's row request) while the coder is still working on the bit.  When the bit is
    // known the copy that guessed right commits its table entry and the other takes over its registers (two DPP
    // moves per register).  What used to follow the bit in series -- two LDS round trips and ~40 instructions, at
    // the end of a nibble an HBM round trip -- now runs beside the chain -> squash -> coder path.  Both copies run
    // the coder (same input, same state), so the bit needs no trip between them.
#ifdef ZPQ_NO_HYP
    constexpr bool HYP = false, HYP16 = false;
#else
    // Round 4: the chain of five (level 3) as well, at SIXTEEN lanes per block -- ten of them the two copies of its five
    // components -- and with the line store: 16 blocks per CU are then four waves, one per SIMD, where the eight-lane
    // decoder left two SIMDs without a wave.  With the store only the copy that guessed the nibble's
    // last bit right probes, claims and reloads (take_prefetched); the other takes over its result.
    // ... and the chain of six with its MIX2 (level 4): fourteen lanes, the MIX2's two copies on lanes 12 and 13.
    constexpr bool HYP16 = DEC && !SPEC && GG == 16 && ((!MIXT && NCH == 5) || (MIXT && NCH == 6));
    constexpr bool HYP = (DEC && !SPEC && !SP && !MIXT && (NCH == 2 || NCH == 3) && GG == 8) || HYP16;
#endif
    // The other decoders (longer chains, MIX2, line store) have no lanes to spare for a second copy.  They request the
    // next nibble's rows for BOTH outcomes of the nibble's last bit as soon as its third bit is known -- a whole bit step
    // before they are needed -- and pick one when they arrive (twice the row reads, no exposed HBM round trip).
    // (Dense tables without a MIX2 only: with the line store the second probe doubles tag loads and selects, and the
    //  MIX2 levels have eight tables to fetch twice -- measured slower there: level 5 505 vs 466 ms.)
    constexpr bool TWO = DEC && !SPEC && !HYP && NCH > 0 && !SP && !MIXT;
    // Line-store decoders (levels 3-5 at scale): the two outcomes of a byte's FOURTH bit lead to neighbouring contexts, and the
    // store places neighbouring lines in one group of four slots (see ZPQ_PREFETCH) -- so for the mid-byte boundary, and only
    // there, both outcomes are requested when the third bit is known: one tag group and two neighbouring lines three times out
    // of four, a whole bit step early.  (The byte boundary's two outcomes are unrelated lines: asking for both was measured
    // slower in round 2, level 5 505 vs 466 ms.)
    // Built in round 3, parity-green (65 GPU tests), MEASURED SLOWER at the shapes bench.py ships: level 3 x 4096 393.5 against
    // 382.6 ms, level 4 x 4096 511.6 against 491.1, level 5 x 3072 503.0 against 477.4 -- the second set of tag and row loads,
    // the seventeen selects and 30-60 more registers cost more than the hidden half round trip, and four neighbouring lines
    // filling one home group push unrelated lines into the walk.  Compiled only with -DZPQ_TWOM (tools/variant.sh).
#ifdef ZPQ_TWOM
    constexpr bool TWOM = DEC && !SPEC && SP && NCH > 0;
#else
    constexpr bool TWOM = false;
#endif
    // The store's PLACEMENT that goes with it -- neighbouring second-nibble contexts in one group of four slots, home slot =
    // index & 3 -- also serves the sixteen-lane two-hypothesis decoders (-DZPQ_HYP16_NBR): the two copies' mid-byte requests then
    // read ONE tag group and two neighbouring lines instead of two unrelated pairs.  The layout lives and dies with the launch.
#ifdef ZPQ_HYP16_NBR
    constexpr bool NBR = TWOM || (HYP16 && SP);
#else
    constexpr bool NBR = TWOM;
#endif
    // (Requesting only the LIKELIER outcome early -- as soon as the last bit's probability is known, asking again after
    //  a wrong guess -- was measured as well: level 3 375 -> 472 ms, level 5 467 -> 557 ms.  Every speculative row read
    //  these decoders add costs more in memory latency under load than it hides; they request after the bit is known.)
    // (the two copies of a component sit on NEIGHBOURING lanes, 2c and 2c + 1: what one copy takes over from the other is
    //  then one DPP quad_perm move per register instead of two bank-masked row shifts)
    const int hyp = HYP ? (li & 1) : 0;                  // the outcome this lane assumes
    const int lc = HYP ? (li >> 1) : li;                 // the component this lane works for
    auto comp_lane = [](const int c) -> int { return HYP ? 2 * c : c; };   // lane (inside the block) of component c, copy 0
    const int ctype = (lc < n) ? M.comp[lc].type : 0;
    const bool hashed = ctype == ZT_ICM || ctype == ZT_ISSE;
    const bool is_icm = ctype == ZT_ICM, is_last = lc == last;
    const DComp &C = M.comp[lc < n ? lc : 0];
    // lanes without a hash table (idle, MIX2) run the same row loads against the first 64
    // bytes of the slot: no exec-masked branch around the loads (a branch join would make
    // the compiler wait for them at once and defeat the prefetch); only the store is masked
    const u32 ht_mask = hashed ? ((C.ht_len - 16u) & cfg.dbg_ht_and) : 0u;
    // compact line store: u32 tags[cap] (dense line index + 1, 0 = free) + 64-byte lines[cap] instead of the dense
    // table.  cap is a multiple of 4, about 1.12x the lines the largest block can touch (zpq_ctx_set_max_block_bytes).
    constexpr bool SPARSE = SP;
    const u32 sp_cap = (SPARSE && hashed) ? C.sp_cap : 0u;
    const u32 sp_groups = sp_cap >> 2;
    const u32 sp_qbits = (SPARSE && hashed && C.ht_len >= 256u) ? (u32)(31 - __clz((int)C.ht_len)) - 8u : 0u;   // log2(lines / 4)
    u32 *sp_tags = reinterpret_cast<u32 *>(slot + C.sp_tag_off);
    // every row access of this lane is tbase + a 32-bit byte offset: the dense table, or the store's line array
    // (offsets, not pointers, so that the h0 ^ 16 / ^ 32 neighbours stay provably global addresses)
    u8 *const tbase = hashed ? (sp_cap ? slot + C.sp_line_off : slot + C.ht_off) : slot;
    // "touched" bitmap of a dense table that is not cleared (two-hypothesis decoder; zpq_touch_layout): one bit per row.
    // Built, parity-green and MEASURED SLOWER (level 2 decode 283.6 ms against 261 ms with the 15 ms of clearing; the code
    // alone, switched off at run time, cost 12 ms): one more load per request in the in-order vmcnt queue, twelve selects,
    // one more register to exchange between the copies.  Compiled only with -DZPQ_TOUCH_DEC (tools/variant.sh).
#ifdef ZPQ_TOUCH_DEC
    constexpr bool TOUCHC = true;
#else
    constexpr bool TOUCHC = false;
#endif
    const bool touch = TOUCHC && hashed && C.tb_off != 0;
    u32 *const tb32 = reinterpret_cast<u32 *>(slot + (touch ? C.tb_off : 0));
    const int sizebits = C.a + 2;
    // Packed per-block state.  ISSE weights are 20-bit two's complement (clamp512k,
    // predictor.v:228-236): t32[s] = (w0 & 0xFFFFF) | (w1 << 20), t8[s] = w1 >> 12.  An ICM entry is
    // its 23-bit cm[s] PLUS stretch(cm[s] >> 8) (12 bits), refreshed when the entry is trained:
    // t32[s] = cm | (st & 0x1FF) << 23, t8[s] = st >> 9 -- the table lookup behind stretch then sits
    // in the update phase instead of in front of the prediction chain.  Lanes without a table
    // (idle, MIX2) use per-workgroup dummy tables, so the bit loop needs no role branches.
    u8 *dummy = lds + L_STATE + cfg.lds_dummy;
    u32 *t32 = reinterpret_cast<u32 *>(hashed ? my + cfg.lds_off32[lc] : dummy);
    u8 *t8 = hashed ? my + cfg.lds_off8[lc] : dummy + 1024;
    u16 *a16 = reinterpret_cast<u16 *>(slot + C.a16_off);
    const int mix_j = M.comp[last].j, mix_k = M.comp[last].k, mix_rate = M.comp[last].rate;
    const u32 mix_mask = (u32)M.comp[last].mask, mix_cmask = (u32)(M.comp[last].c - 1);

    for (int blk = slot_id; slot_id < nslots && blk < B.nblocks; blk += nslots) {
        // ---- Predictor.init + ZPAQL.clear for this block (predictor.v:325-470, zpaql.v:54-95)
        {
            uint4 *z4 = reinterpret_cast<uint4 *>(slot);
            const u64 n16 = M.zero_bytes / 16;
            const uint4 zero = make_uint4(0, 0, 0, 0);
            for (u64 i = li; i < n16; i += G) z4[i] = zero;
            for (int c = 0; c < n; c++) {
                const DComp &cc = M.comp[c];
                u32 *d32 = reinterpret_cast<u32 *>(my + cfg.lds_off32[c]);
                if (cc.type == ZT_ICM) {
                    u8 *d8 = my + cfg.lds_off8[c];
                    for (int i = li; i < 256; i += G) {
                        const u32 cmi = B.img[i];                          // cminit(i) (statetable.v:108-116), < 2^23
                        const i32 sti = stretch_of(cmi);
                        d32[i] = cmi | (((u32)sti & 0x1FFu) << 23);
                        d8[i] = (u8)(sti >> 9);
                    }
                }
                else if (cc.type == ZT_ISSE) {
                    u8 *d8 = my + cfg.lds_off8[c];
                    for (int i = li; i < 256; i += G) {
                        const u32 a0 = B.img[256 + 2 * i], a1 = B.img[257 + 2 * i];
                        d32[i] = (a0 & 0xFFFFFu) | (a1 << 20);
                        d8[i] = (u8)((i32)a1 >> 12);
                    }
                }
                else if (cc.type == ZT_MIX2) {
                    u32 *w = reinterpret_cast<u32 *>(slot + cc.a16_off);
                    const u32 words = (cc.a16_len + 1) / 2;
                    for (u32 i = li; i < words; i += G) w[i] = 0x80008000u;   // a16[] = 32768 (predictor.v:396)
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const u8 *src = B.in + B.in_off[blk];
        const u32 nin = (u32)(B.in_off[blk + 1] - B.in_off[blk]);
        u8 *dst = B.out + B.out_off[blk];
        const u32 cap = (u32)(B.out_off[blk + 1] - B.out_off[blk]);
        i32 status = ZPQ_OK;

        Vm z;
        z.a = z.b = z.c = z.d = 0; z.f = 0; z.pc = 0; z.out = nullptr;
        z.m = slot + M.m_off; z.mlen = M.mlen;
        z.h = reinterpret_cast<u32 *>(slot + M.h_off); z.hlen = M.hlen;
        z.r = reinterpret_cast<u32 *>(slot + M.r_off);
        z.hdr = M.header; z.hdr_len = M.hdr_len; z.hbegin = M.hbegin; z.hend = M.hend;
        u32 prev = 0, m4 = 0, b4 = 0, hctx = 0;

        // Input window: the next bytes of `src` live in two registers, refilled by aligned
        // dword loads issued a whole dword ahead of use (never a load on the bit path).
        // Virtual position vp = byte index + (src & 3); dword k covers vp 4k..4k+3.
        const u32 mis = (u32)(reinterpret_cast<uintptr_t>(src) & 3u);
        const u32 *src4 = reinterpret_cast<const u32 *>(src - mis);
        const u32 ndw = (nin + mis + 3u) >> 2;              // dwords that hold at least one valid byte
        u32 win0 = ndw > 0 ? src4[0] : 0u, win1 = ndw > 1 ? src4[1] : 0u;
        u32 wdw = 0;                                       // dword index held in win0
        // Encode walks the input one byte per iteration, so its window needs no decision: every iteration slides by
        // register moves and re-requests the dword behind the window UNCONDITIONALLY; the value is first looked at an
        // iteration later.  (A load inside the "crossed a dword" branch was waited for right at the branch join --
        // an exposed memory round trip whenever ANY lane of the wave crossed, i.e. almost every byte once lanes of
        // a block work on different bytes.)  Past the end the clamped address re-reads the last dword; the byte is
        // masked by pos < nin.
        const u32 *const enc4 = ndw ? src4 : reinterpret_cast<const u32 *>(B.in_off);   // nin == 0: any readable dword
        const u32 enc_last = ndw ? ndw - 1u : 0u;
        if (!DEC) win1 = enc4[min(1u, enc_last)];
        auto enc_byte = [&](u32 pos) -> u32 {              // src[pos], pos advancing by at most one dword per call
            const u32 vp = pos + mis;
            const bool slide = (vp >> 2) != wdw;
            win0 = slide ? win1 : win0;
            wdw = slide ? wdw + 1u : wdw;
            win1 = enc4[min(wdw + 1u, enc_last)];
            const u32 c = (win0 >> ((vp & 3u) * 8u)) & 255u;
            return pos < nin ? c : 0u;
        };
        // Decode consumes 0..4 coded bytes per bit, at positions only the coder knows.  Its window is FOUR dwords,
        // double-buffered per byte of output: at the top of every byte iteration the window requested one iteration
        // ago is adopted and the next one (from the coder's current position) is requested, unconditionally -- no load
        // and no wait inside the bit steps.  (The old two-dword window reloaded inside the renormalisation loop; the
        // compiler's wait for that load sat at the branch join and was a vmcnt(0): every pass through the loop also
        // waited for whatever else was in flight -- the row write-back, the next nibble's row request.)  A window
        // covers what two iterations consume unless the stream expands more than ~6x locally; then one dword is
        // fetched on the spot.
        const u32 dlast = ndw ? ndw - 1u : 0u;
        u32 dW0 = 0, dW1 = 0, dW2 = 0, dW3 = 0, wd = 0;        // dwords [wd, wd + 4) of the coded input
        u32 nW0 = 0, nW1 = 0, nW2 = 0, nW3 = 0, nwd = 0;       // requested, not yet looked at
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
            const u32 bo = vp - 4u * wd;                   // byte offset inside the 16-byte window
            // (two 64-bit halves and a 64-bit shift: a four-way select by index gets turned into a scratch array)
            const u64 lo = (u64)dW0 | ((u64)dW1 << 32), hi = (u64)dW2 | ((u64)dW3 << 32);
            u32 c = (u32)(((bo & 8u) ? hi : lo) >> ((bo & 7u) * 8u)) & 255u;
            if (bo > 15u) {
                u32 t = enc4[min(vp >> 2, dlast)];
                asm volatile("; coded input beyond the window: waited for here, not at the join" : "+v"(t));
                c = (t >> ((vp & 3u) * 8u)) & 255u;
            }
            return pos < nin ? c : 0u;
        };

        BitCtx X;
        X.low = 1; X.high = 0xFFFFFFFFu; X.code = 0; X.opos = 0; X.ipos = 0;
        X.r0 = X.r1 = X.r2 = X.r3 = 0; X.slot = 1; X.c8 = 1;
        u32 first = 0xFFFFFFFFu;
        bool got_first = false;
        if (DEC) {
            dec_request(0u);
            dec_adopt();
            dec_request(0u);                               // (adopted by the first byte iteration)
            for (int k = 0; k < 4; k++) { const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin); X.code = (X.code << 8) | c; }
        }
        const u32 total = DEC ? 0xFFFFFFFFu : nin + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u);
        u32 ch = 0;
        u32 roff = 0;                                      // tbase offset of the row in X.r0..r3

        auto stretch_lds = stretch_of;
        // the stretch an ICM entry carries (meaningless on other lanes, where it is masked out)
        auto icm_st = [](u32 v, i32 b) -> i32 { return (i32)((u32)b << 9) | (i32)(v >> 23); };
        // decoded bit from lane `last` to the lanes below it: log-step DPP row_shl, no LDS round trip
        const int bdist = last - lc;                       // > 0 on lanes that need the value
        auto bcast_down = [&](i32 v) -> i32 {
            if constexpr (HYP && NCH == 2) {
                // lanes c0h0 c0h1 c1h0 c1h1 share a quad: every lane takes its own copy's coder lane (both decode the same bit)
                return __builtin_amdgcn_update_dpp(v, v, 0xEE /*quad_perm:[2,3,2,3]*/, 0xf, 0xf, false);
            }
            if constexpr (HYP && NCH == 3) {
                // the coder lanes are 4 and 5 of the block's eight: spread them over their quad, then hand the quad down
                v = __builtin_amdgcn_update_dpp(v, v, 0x44 /*quad_perm:[0,1,0,1]*/, 0xf, 0xa, false);
                return __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0x5, false);
            }
            if constexpr (HYP && NCH == 5) {
                // the coder lanes are 8 and 9 of the block's sixteen (both hold the bit): over their quad, then to the two
                // quads below (the lanes above work for no component)
                v = __builtin_amdgcn_update_dpp(v, v, 0x44 /*quad_perm:[0,1,0,1]*/, 0xf, 0x4, false);
                v = __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0x2, false);
                return __builtin_amdgcn_update_dpp(v, v, 0x108 /*row_shl:8*/, 0xf, 0x1, false);
            }
            if constexpr (HYP && NCH == 6) {
                // MIX2 = component 6: the coder lanes are 12 and 13 of the row; over their quad (lanes 14, 15 follow the bits
                // too: they help fetch the MIX2's candidate weights), then to the three quads below
                v = __builtin_amdgcn_update_dpp(v, v, 0x44 /*quad_perm:[0,1,0,1]*/, 0xf, 0x8, false);
                const i32 a = __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0x4, false);
                const i32 b = __builtin_amdgcn_update_dpp(a, a, 0x108 /*row_shl:8*/, 0xf, 0x3, false);
                return b;
            }
            if constexpr (NCH > 0 && NCH + (MIXT ? 1 : 0) <= 4) {
                // all of the block's lanes are in one quad: one quad_perm broadcast of the coder lane
                constexpr int L = NCH + (MIXT ? 1 : 0) - 1;
                return __builtin_amdgcn_update_dpp(v, v, L * 0x55 /*quad_perm:[L,L,L,L]*/, 0xf, 0xf, false);
            }
            if constexpr (DEC && !HYP && NCH > 0 && GG == 8 && NCH + (MIXT ? 1 : 0) > 4) {
                // one copy, eight lanes per block, the coder on lane 4..7 of the block: over its quad, then to the quad below
                constexpr int P = (NCH + (MIXT ? 1 : 0) - 1) & 3;
                v = __builtin_amdgcn_update_dpp(v, v, P * 0x55 /*quad_perm:[P,P,P,P]*/, 0xf, 0xa, false);
                return __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0x5, false);
            }
            if constexpr (DEC && !HYP && NCH == 8 && MIXT && GG == 16) {
                // level 5: the coder (MIX2) on lane 8 of the row: over its quad, then to the two quads below
                v = __builtin_amdgcn_update_dpp(v, v, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0x4, false);
                v = __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0x2, false);
                return __builtin_amdgcn_update_dpp(v, v, 0x108 /*row_shl:8*/, 0xf, 0x1, false);
            }
            if (n > 1) { const i32 t = __builtin_amdgcn_update_dpp(v, v, 0x101 /*row_shl:1*/, 0xf, 0xf, false); v = (bdist == 1) ? t : v; }
            if (n > 2) { const i32 t = __builtin_amdgcn_update_dpp(v, v, 0x102 /*row_shl:2*/, 0xf, 0xf, false); v = (bdist >= 2 && bdist < 4) ? t : v; }
            if (n > 4) { const i32 t = __builtin_amdgcn_update_dpp(v, v, 0x104 /*row_shl:4*/, 0xf, 0xf, false); v = (bdist >= 4 && bdist < 8) ? t : v; }
            if (n > 8) { const i32 t = __builtin_amdgcn_update_dpp(v, v, 0x108 /*row_shl:8*/, 0xf, 0xf, false); v = (bdist >= 8) ? t : v; }
            return v;
        };

        // find_ht (predictor.v:495-532): the three candidate rows h0, h0^16, h0^32 share one
        // 64-byte line.  select_row resolves hit / victim with selects only.
        auto select_row = [&](const u32x4 A, const u32x4 Bq, const u32x4 Cq, const u32 pa, const u32 chk) {
            const u32 pb = pa ^ 16u, pc = pa ^ 32u;                             // rows share a 64-B aligned line
            const bool ma = (A.x & 255u) == chk, mb = (Bq.x & 255u) == chk, mc = (Cq.x & 255u) == chk;
            const u32 qa = (A.x >> 8) & 255u, qb = (Bq.x >> 8) & 255u, qc = (Cq.x >> 8) & 255u;
            const bool va = qa <= qb && qa <= qc, vb = qb < qc;            // victim order (predictor.v:513-531)
            const bool hit = ma || mb || mc;
            const bool ua = ma || (!hit && va);
            const bool ub = !ua && (mb || (!hit && vb));
            roff = ua ? pa : (ub ? pb : pc);
            const u32x4 Rr = ua ? A : (ub ? Bq : Cq);
            X.r0 = hit ? Rr.x : chk; X.r1 = hit ? Rr.y : 0u; X.r2 = hit ? Rr.z : 0u; X.r3 = hit ? Rr.w : 0u;
        };
        // Rows are requested AHEAD of the nibble that uses them.  Encode: every context is a
        // function of input bytes only, so the request goes out one whole nibble early and the
        // HBM latency hides behind four bit steps.  Decode: the request goes out inside the
        // previous nibble's last bit step, the moment that bit is decoded, and overlaps its
        // update work.  If a requested row is the one being updated right now, the registers
        // win (exact forwarding in take_prefetched).
        u32x4 nA = {0, 0, 0, 0}, nB = {0, 0, 0, 0}, nC = {0, 0, 0, 0};
        u32 n_po = 0;                                      // tbase offset of candidate row A of the probe in flight
        u32 n_chk = 0;
        u32 n_key = 0, n_si = 0, n_off = 0;                // compact store: the probe in flight
        u32x4 n_tags = {0, 0, 0, 0};
        bool sp_full = false;
        u32 sp_claims = 0;
        const u32 sp_limit = sp_cap - (sp_cap >> 4);       // a block that needs more than 15/16 of the store is refused
        // second request of the non-HYP decoders (see TWO below): the rows for the other outcome of the nibble's last bit
        u32x4 aA = {0, 0, 0, 0}, aB = {0, 0, 0, 0}, aC = {0, 0, 0, 0}, a_tags = {0, 0, 0, 0};
        u32 a_po = 0, a_chk = 0, a_key = 0, a_si = 0, a_off = 0;
        bool sel_alt = false;                              // the nibble that really follows is the one the alt request was for
        // (a macro, not a lambda taking references: locals handed on by reference through a second closure are not
        //  promoted to registers -- the whole probe state ended up in scratch, 5x slower)
#ifdef ZPQ_DEBUG_NO_ROWS   // timing experiment only (wrong output): no hash-row traffic at all
#define ZPQ_LOAD_ROWS(A_, B_, C_, po_) do { A_ = u32x4{(po_), 0, 0, 0}; B_ = A_; C_ = A_; } while (0)
#else
#define ZPQ_LOAD_ROWS(A_, B_, C_, po_)                                                  \
    do {                                                                                \
        A_ = *reinterpret_cast<const u32x4 *>(tbase + (po_));                           \
        B_ = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 16u));                   \
        C_ = *reinterpret_cast<const u32x4 *>(tbase + ((po_) ^ 32u));                   \
    } while (0)
#endif
        // Dense line index -> slot of the compact store.  Open addressing over GROUPS of four slots: the home
        // slot is mulhi(hash, cap) (capacity need not be a power of two); probing visits the home group's
        // slots cyclically from the home slot, then the following groups the same way.  One 16-byte load
        // brings the home group's four tags, and the HOME slot's rows are fetched with them: a line that
        // sits in its home slot, or a new line (it takes the first free slot it meets, and a new line is all
        // zero -- nothing to load), costs one memory round trip; only a line that was displaced when it
        // was claimed needs a second one for its rows, and only a full group is walked past.
#define ZPQ_PREFETCH(A_, B_, C_, tags_, po_, chk_, key_, si_, off_, tw_, hc_, c8v_)     \
    do {                                                                                \
        const u32 cx_ = (hc_) + 16u * (c8v_);                                           \
        chk_ = (cx_ >> sizebits) & 255u;                                                \
        const u32 h0_ = (cx_ * 16u) & ht_mask;                                          \
        u32 pox_ = SWZ ? swz_addr(h0_) : h0_;                                           \
        if (SPARSE && sp_cap) {                                                         \
            /* the line's index with the two lowest bits moved to the top (neighbouring second-nibble contexts = */ \
            /* neighbouring indices); four neighbouring indices share a home group, index & 3 is the home slot     */ \
            const u32 tl_ = NBR ? (((h0_ >> 6) & 3u) << sp_qbits) | (h0_ >> 8) : (h0_ >> 6); \
            key_ = tl_ + 1u;                                                            \
            si_ = NBR ? 4u * __umulhi((tl_ >> 2) * 0x9E3779B1u, sp_groups) + (tl_ & 3u) : __umulhi(key_ * 0x9E3779B1u, sp_cap); \
            off_ = h0_ & 48u;                                                           \
            tags_ = *reinterpret_cast<const u32x4 *>(sp_tags + (si_ & ~3u));            \
            pox_ = (si_ << 6) + off_;                                                   \
        }                                                                               \
        po_ = pox_;                                                                     \
        if (HYP && TOUCHC) tw_ = tb32[touch ? (pox_ >> 9) : 0u];   /* the line's four "touched" bits (32 rows per word) */ \
        ZPQ_LOAD_ROWS(A_, B_, C_, pox_);                                                \
    } while (0)
        // Two-hypothesis decode: inside this launch a table's 64-byte lines are TRANSPOSED -- line (h0 >> 6) & 3 of every
        // 256-byte group moves to quarter (h0 >> 6) & 3 of the table, the groups keep their order inside a quarter (a
        // bijection on lines; h0 ^ 16 and h0 ^ 32 stay inside the line; the layout lives and dies with the launch).  The
        // contexts of a byte's second nibble are 16 * c8 apart (predictor.v:558-560), their lines 256 bytes apart: transposed,
        // the lines of neighbouring c8 are NEIGHBOURS, so the two copies' requests for the two outcomes of the fourth bit fall
        // into one 128-byte block half of the time (one request to the memory system instead of two; tools/micro/rowlat.hip).
#ifdef ZPQ_NO_SWZ
        constexpr bool SWZ = false;
#else
        constexpr bool SWZ = HYP;
#endif
        const u32 swz_q = hashed && C.ht_len >= 1024u ? (u32)(31 - __clz((int)C.ht_len)) - 8u : 0u;   // log2(quarter) - 6; 0 = table too small
        auto swz_addr = [&](const u32 h0) -> u32 {
            const u32 t = ((h0 & 0xC0u) << swz_q) | ((h0 >> 2) & ((64u << swz_q) - 64u)) | (h0 & 63u);
            return swz_q ? t : h0;
        };
        auto load_rows = [&](const u32 po) { ZPQ_LOAD_ROWS(nA, nB, nC, po); };
        u32 n_tw = 0xFFFFFFFFu, a_tw = 0xFFFFFFFFu;        // "touched" words of the requests in flight
        auto prefetch_rows = [&](const u32 hc, const u32 c8v) { ZPQ_PREFETCH(nA, nB, nC, n_tags, n_po, n_chk, n_key, n_si, n_off, n_tw, hc, c8v); };
        auto prefetch_alt = [&](const u32 hc, const u32 c8v) { ZPQ_PREFETCH(aA, aB, aC, a_tags, a_po, a_chk, a_key, a_si, a_off, a_tw, hc, c8v); };
        // Consume the rows requested one nibble ago, THEN write the finished row back (so that
        // the wait for the loads does not also wait for a just-issued store), then the caller
        // requests the next nibble's rows.  The finished row is forwarded from registers if it
        // is one of the candidates.
        // two-hypothesis decode: the same register of the block's OTHER copy (lane ^ 1)
        // (mov_dpp, not update_dpp(old = v, ...): every lane has a source under this permutation, and with no `old` operand
        //  tied to the destination the compiler needs no copy of v in front of the v_mov_b32_dpp)
        auto xchg = [&](const u32 v) -> u32 {
#ifdef ZPQ_DPP_OLD
            return (u32)__builtin_amdgcn_update_dpp((i32)v, (i32)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, false);
#else
            return (u32)__builtin_amdgcn_mov_dpp((i32)v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true);
#endif
        };
        const bool wr_lane = hashed && (!HYP || hyp == 0);  // this lane writes finished rows back
        const bool mix_lane = ctype == ZT_MIX2;
        bool row_mine = true;                              // this copy requested the rows of the nibble that really follows
        // Two-hypothesis decode, a byte's SECOND nibble: its 16 possible contexts are hctx + 16 * (16..31) (predictor.v:558-560),
        // i.e. -- in the transposed layout above -- 16 neighbouring lines.  So the request need not wait for the first nibble's
        // third bit: when the SECOND bit is known each copy asks for the two lines of ITS outcome of the third bit (both outcomes
        // of the fourth: neighbouring lines, half of the time one 128-byte block), two whole bit steps before they are needed --
        // the HBM round trip (~2000 cycles under this load, tools/micro/rowlat.hip) then hides completely instead of by half.
        // Four lines per table and byte instead of two, but four NEIGHBOURING ones: asking two steps early for four lines 256
        // bytes apart was measured slower in round 2 (321 vs 267 ms).
        // Round 4 (EXPERIMENTS.md R4.11): on the dieted kernel the early request buys nothing any more -- level 2 x 8192, processes
        // balanced over a box's fast and slow places: 235.3-236.9 ms without against 235.7-239.4 with -- and it reads a third more
        // lines (4 instead of 2 per table for a byte's second nibble).  Off by default; -DZPQ_HYP4 (tools/variant.sh) brings it back.
#ifndef ZPQ_HYP4
        constexpr bool HYP4 = false;
#else
#ifdef ZPQ_HYP4_L1
        constexpr bool HYP4 = HYP && SWZ;
#else
        constexpr bool HYP4 = HYP && SWZ && NCH == 3;         // (level 1, one hashed table of 32 MiB beside a small one: measured 272 vs 269 ms)
#endif
#endif
        auto take_prefetched = [&](const bool have_prev, auto midc) {
            constexpr bool MID = decltype(midc)::value;        // the rows of a byte's second nibble
            bool claim = false;
            u32 claim_si = 0;
            if (TWO || ((HYP4 || TWOM) && MID)) {              // (selects, not a branch: both requests are waited for here anyway)
                auto sel4 = [](const bool c, const u32x4 a, const u32x4 b) -> u32x4 {
                    return u32x4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w};
                };
                nA = sel4(sel_alt, aA, nA); nB = sel4(sel_alt, aB, nB); nC = sel4(sel_alt, aC, nC);
                n_po = sel_alt ? a_po : n_po; n_chk = sel_alt ? a_chk : n_chk;
                if (HYP && TOUCHC) n_tw = sel_alt ? a_tw : n_tw;
                if (SPARSE) {
                    n_tags = sel4(sel_alt, a_tags, n_tags);
                    n_key = sel_alt ? a_key : n_key; n_si = sel_alt ? a_si : n_si; n_off = sel_alt ? a_off : n_off;
                }
                sel_alt = false;
            }
            // two hypotheses over a line store: the copy whose request was for the nibble that really follows resolves it (walk,
            // claim, reload); the other copy's request was for a context that never came -- it claims nothing (a claim there
            // would fill the store with lines no context owns) and takes the result over below
            const bool sp_act = !HYP || row_mine;
            if (SPARSE && sp_cap) {
                const u32 o = n_si & 3u;
                // slots of a group that end the probe -- the line's own tag, or a free slot -- as a 4-bit mask rotated
                // so that bit j stands for slot (o + j) & 3: the lowest set bit is the first such slot met
                auto probe_group = [&](const u32x4 T) -> u32 {
                    const u32 mm = (min(T.x ^ n_key, T.x) == 0u ? 1u : 0u) | (min(T.y ^ n_key, T.y) == 0u ? 2u : 0u) |
                                   (min(T.z ^ n_key, T.z) == 0u ? 4u : 0u) | (min(T.w ^ n_key, T.w) == 0u ? 8u : 0u);
                    return ((mm * 17u) >> o) & 15u;
                };
                u32x4 T = n_tags;
                u32 g = n_si >> 2;
                u32 r = probe_group(T);
                if (r == 0u && !sp_full && sp_act) {                           // home group full of other lines: walk on
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
                if (!sp_act) { }                                               // (wrong copy: nothing to resolve)
                else if (r == 0u) { status = ZPQ_E_TOOBIG; sp_full = true; }   // every slot taken: the block is larger than promised
                else if (t == 0u && ++sp_claims > sp_limit) { status = ZPQ_E_TOOBIG; sp_full = true; }   // (nearly) full: stop before probes get long
                else if (t == 0u) {
                    // claim: the line becomes this context's and starts out all zero, like an untouched dense line.
                    // Its tag and its zeros are STORED further down, behind the consumption of the loads: a store
                    // issued here would sit in front of the reload below in the in-order vmcnt queue.
                    const u32x4 z4 = {0, 0, 0, 0};
                    claim = true;
                    claim_si = si;
                    nA = z4; nB = z4; nC = z4;
                    n_po = (si << 6) + n_off;
                } else if (si != n_si) {                                       // displaced line: its rows were not the ones fetched
                    n_po = (si << 6) + n_off;
                    load_rows(n_po);
                }
            }
            u32 tw_line = 0xFFFFFFFFu;
            if constexpr (HYP && TOUCHC) {
                // a row of a table that is never cleared counts only once it has been written: untouched rows read as zeros
                tw_line = touch ? n_tw : 0xFFFFFFFFu;
                const u32 rb = (n_po >> 4) & 31u;
                const u32x4 z4 = {0, 0, 0, 0};
                const bool ta = ((tw_line >> rb) & 1u) != 0, tb = ((tw_line >> (rb ^ 1u)) & 1u) != 0, tc = ((tw_line >> (rb ^ 2u)) & 1u) != 0;
                nA = u32x4{ta ? nA.x : 0u, ta ? nA.y : 0u, ta ? nA.z : 0u, ta ? nA.w : 0u};
                nB = u32x4{tb ? nB.x : 0u, tb ? nB.y : 0u, tb ? nB.z : 0u, tb ? nB.w : 0u};
                nC = u32x4{tc ? nC.x : 0u, tc ? nC.y : 0u, tc ? nC.z : 0u, tc ? nC.w : 0u};
                (void)z4;
            }
            const u32x4 Rp = {X.r0, X.r1, X.r2, X.r3};
            const u32 poff = roff;
            const bool fa = have_prev && n_po == poff;
            const bool fb = have_prev && (n_po ^ 16u) == poff;
            const bool fc = have_prev && (n_po ^ 32u) == poff;
            const u32x4 A = fa ? Rp : nA, Bq = fb ? Rp : nB, Cq = fc ? Rp : nC;
            select_row(A, Bq, Cq, n_po, n_chk);
            if constexpr (HYP) {
                // each copy asked for the rows of ITS outcome; the one that was wrong takes the other's resolved row
                const u32 x0 = xchg(X.r0), x1 = xchg(X.r1), x2 = xchg(X.r2), x3 = xchg(X.r3), xo = xchg(roff);
                X.r0 = row_mine ? X.r0 : x0; X.r1 = row_mine ? X.r1 : x1; X.r2 = row_mine ? X.r2 : x2; X.r3 = row_mine ? X.r3 : x3;
                roff = row_mine ? roff : xo;
                if (TOUCHC) { const u32 xt = xchg(tw_line); tw_line = row_mine ? tw_line : xt; }
                if constexpr (SPARSE) {
                    // the store's bookkeeping lives on both copies and must agree: what the resolving copy counted and found
                    const u32 xc = xchg(sp_claims), xs = xchg((u32)status), xf = xchg(sp_full ? 1u : 0u);
                    sp_claims = row_mine ? sp_claims : xc;
                    status = row_mine ? status : (i32)xs;
                    sp_full = row_mine ? sp_full : (xf != 0u);
                }
            }
            // keep the stores BELOW the wait for the loads above (vmcnt is in-order: a store issued
            // first would be waited for as well)
            u32 poff2 = poff, cpo = n_po & ~63u;
            asm volatile("; order: row store after the prefetched rows are consumed" : "+v"(poff2), "+v"(cpo) : "v"(X.r0), "v"(X.r3));
            if (SPARSE && claim) {
                const u32x4 z4 = {0, 0, 0, 0};
                u8 *line = tbase + cpo;
                sp_tags[claim_si] = n_key;
                *reinterpret_cast<u32x4 *>(line) = z4;
                *reinterpret_cast<u32x4 *>(line + 16) = z4;
                *reinterpret_cast<u32x4 *>(line + 32) = z4;
                *reinterpret_cast<u32x4 *>(line + 48) = z4;
            }
#ifndef ZPQ_DEBUG_NO_ROWS
            // (one predicate, the lane's part of it computed once: `have_prev && hashed && ...` as written became a twenty-instruction
            //  maze of exec-mask moves around one store)
            if (have_prev & wr_lane) *reinterpret_cast<u32x4 *>(tbase + poff2) = Rp;   // (both copies hold the same row: copy 0 writes)
            if constexpr (HYP && TOUCHC) {
                // the row this nibble works on counts as written from now on (it is stored when the nibble ends; until then a
                // request that meets it takes it from the registers)
                const u32 nbit = 1u << ((roff >> 4) & 31u);
                if (touch && hyp == 0 && (tw_line & nbit) == 0u) tb32[roff >> 9] = tw_line | nbit;
            }
#endif
        };
        // ZPAQL.run(byte) + h[] copy (predictor.v:809-816) -> this lane's next context hash
        // vm_hash: the contexts a byte leads to, WITHOUT touching the VM's state (the two-hypothesis decoder asks for a
        // byte that may not be the one decoded); vm_commit: the state change once the byte is certain.
        u32 vm_last = 0;                                   // (set by vm_hash) the LAST component's context hash for that byte: the MIX2's
        auto vm_hash = [&](const u32 byte) -> u32 {
            u32 hv = 0;
            // the specialised kernels are only launched for their level's own program shape (chain of 2 = level 1's
            // program, longer chains = the hash chain): the interpreter and its state exist only in the runtime-loop kernel
            const int vm_kind = NCH == 0 ? cfg.vm_kind : (NCH == 2 ? (int)VM_LEVEL1 : (int)VM_HASHCHAIN);
            if (vm_kind == VM_HASHCHAIN) {
                // b=c c-- *c=a d=0 (hash *d=a d++)* hash *d=a halt: H[k] = hash^(k+1) of (byte, prev)
                u32 a = byte;
                for (int k = 0; k < n; k++) { a = (a + prev + 512u) * 773u; hv = (k == lc) ? a : hv; }
                vm_last = a;
            } else if (vm_kind == VM_LEVEL1) {
                // *b=a a=0 d=0 hash b-- hash *d=a d++ b-- hash b-- hash *d=a halt, M = 4 bytes
                const u32 mm = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8));
                u32 bb = b4;
                u32 a = 0;
                a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
                a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
                const u32 h0v = a; bb--;
                a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u; bb--;
                a = (a + ((mm >> ((bb & 3) * 8)) & 255u) + 512u) * 773u;
                hv = (lc == 0) ? h0v : ((lc == 1) ? a : 0u);
            }
            return hv;
        };
        auto vm_commit = [&](const u32 byte) {
            const int vm_kind = NCH == 0 ? cfg.vm_kind : (NCH == 2 ? (int)VM_LEVEL1 : (int)VM_HASHCHAIN);
            if (vm_kind == VM_HASHCHAIN) prev = byte;
            else if (vm_kind == VM_LEVEL1) { m4 = (m4 & ~(255u << ((b4 & 3) * 8))) | (byte << ((b4 & 3) * 8)); b4 -= 3u; }
        };
        // Encode: coded bytes are queued in a register pair and stored at ONE fixed point per input byte, right behind
        // the wait for the first nibble's rows.  A byte store issued inside the coder would be the youngest memory
        // operation when the next row wait comes, and vmcnt retires in order: the wait would sit out the store.
        u64 oq = 0;
        u32 oqn = 0;                                       // queued bytes (they end at X.opos)
        auto oq_flush = [&]() {
            const u32 base = X.opos - oqn;
            for (u32 k = 0; k < oqn; k++) {
                if (base + k < cap) dst[base + k] = (u8)(oq >> (8u * k));
            }
            oq = 0; oqn = 0;
        };
        auto put_byte = [&](const u32 b) {                 // Writer.put (encoder.v:76-83)
            oq |= (u64)(b & 255u) << (8u * oqn);
            oqn++;
            X.opos++;
            if (oqn == 8u) oq_flush();
        };
        // ZPAQL.run(byte) + h[] copy (predictor.v:809-816) -> this lane's next context hash
        auto run_vm = [&](const u32 byte) -> u32 {
            u32 hv = 0;
            const int vm_kind = NCH == 0 ? cfg.vm_kind : (NCH == 2 ? (int)VM_LEVEL1 : (int)VM_HASHCHAIN);
            if (vm_kind == VM_HASHCHAIN || vm_kind == VM_LEVEL1) {
                hv = vm_hash(byte);
                vm_commit(byte);
            } else {
                if (li == 0) { if (!vm_run(z, byte)) status = ZPQ_E_VMSTEPS; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                hv = (li < n && (u32)li < M.hlen) ? z.h[li] : 0u;
            }
            return hv;
        };
        // Software-pipelined bit step.  Carried between steps: the current state byte, its
        // table entry with the previous bit's update already forwarded, and (ICM) its stretch.
        u32 hnext_dec = 0;                                 // decode: contexts of the next byte, set in the last bit step
        u32 cur_s = 0, cur_v = 0;                          // state byte, packed entry (forwarded)
        i32 cur_b = 0, cur_pst = 0;                        // w1's top byte, ICM stretch

        auto nibble_begin = [&]() {                        // after find_row: slot 1 = byte 1 of the row
            cur_s = (X.r0 >> 8) & 255u;
            cur_v = t32[cur_s];
            cur_b = (i32)(int8_t)t8[cur_s];
            cur_pst = icm_st(cur_v, cur_b);
        };

        // MIX2 weights when ENCODING (levels 4-5): cxt = (h[i] + (c8 & mask)) & (size-1) (predictor.v:587-592) with
        // mask = 255 walks eight DISTINCT entries per byte -- c8 = 1, 1b7, 1b7b6, ... -- all known when the byte
        // begins.  They are loaded into registers there, BEFORE the row prefetch is issued, trained in registers and
        // written back when the next byte begins.  A global load inside the bit steps would be waited for with the
        // in-order vmcnt, i.e. together with the row prefetch issued before it, exposing that HBM fetch every nibble.
        // Lanes other than the MIX2's aim the same (unconditional, so the compiler can count them) accesses at one
        // scratch word per workgroup.
        // (the specialised MIX2 kernels are only launched when this holds; the runtime-loop kernel checks it)
        const bool mixreg = NCH ? MIXT : (has_mix2 && (mix_mask & 255u) == 255u && mix_cmask >= 255u);
        u32 mw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u32 mh_prev = 0, mch_prev = 0;
        u16 *const ax16 = (ctype == ZT_MIX2) ? a16
                                             : reinterpret_cast<u16 *>(B.slots + (u64)(blockIdx.x * cfg.blocks_per_wg) * M.slot_bytes + M.comp[0].cm_off);
        const u32 axmask = (ctype == ZT_MIX2) ? mix_cmask : 0u;
        auto mix_byte_begin = [&](const u32 byte, const bool have_prev) {
            if (have_prev) {
#pragma unroll
                for (int t = 0; t < 8; t++) ax16[(mh_prev + ((1u << t) | (mch_prev >> (8 - t)))) & axmask] = (u16)mw[t];
            }
#pragma unroll
            for (int t = 0; t < 8; t++) mw[t] = ax16[(hctx + ((1u << t) | (byte >> (8 - t)))) & axmask];
            mh_prev = hctx;
            mch_prev = byte;
        };

        // MIX2 weights when DECODING: the contexts of a nibble are not known ahead, but there are only 15 candidates
        // (c8 = prefix extended by 0..3 bits = the same slot numbering 1..15 as the bit-history row).  The group's
        // lanes fetch them together with the nibble's rows (lane l takes slots l and l+G..), park them in LDS, the MIX2
        // lane reads and trains its weight there, and the lanes write their candidates back when the nibble ends.
        u16 *const w16s = reinterpret_cast<u16 *>(my + cfg.lds_mixw);
        u16 *const a16m = reinterpret_cast<u16 *>(slot + M.comp[last].a16_off);
        constexpr int MQ = 16 / GG;
        u32 mwa[MQ], mwl[MQ], mwf[MQ];
        bool mwhit[MQ];
#pragma unroll
        for (int q = 0; q < MQ; q++) { mwa[q] = 0; mwl[q] = 0; mwf[q] = 0; mwhit[q] = false; }
        bool mix_live = false;                              // candidates of the current nibble are in w16s
        u32 mix_hm = 0, mix_prefix = 1;                     // what the current nibble's candidates were derived from
        // `prefix` = c8 at the start of the nibble (1, or 1hhhh); `hm` = the MIX2 component's context hash
        auto mixw_request = [&](const u32 hm, const u32 prefix) {
            const u32 plen_old = 32u - (u32)__clz((int)mix_prefix);      // bit length of the old prefix (1 or 5)
#pragma unroll
            for (int q = 0; q < MQ; q++) {
                const u32 sl = (u32)li + (u32)q * GG;        // this lane's slot (0 = none)
                if (sl >= 1u && sl <= 15u) {
                    if (mix_live) a16m[mwa[q]] = w16s[sl];   // the nibble that just ended (values are final)
                    const u32 L = 31u - (u32)__clz((int)sl);
                    const u32 c8c = (prefix << L) | (sl - (1u << L));
                    const u32 na = (hm + (c8c & mix_mask)) & mix_cmask;
                    // The index is a hash: this entry may have been a candidate of the nibble that just ended,
                    // under ANOTHER lane.  That lane's write-back and this lane's load are not ordered against
                    // each other, so such an entry is taken from LDS, where its trained value still sits:
                    // d = na - old hash must read (old prefix << Lo) | r with Lo = 0..3, r < 2^Lo.
                    const u32 d = (na - mix_hm) & mix_cmask;
                    const u32 dlen = 32u - (u32)__clz((int)(d | 1u));
                    const u32 Lo = dlen - plen_old;
                    const bool in_old = mix_live && d != 0u && d <= 255u && dlen >= plen_old && Lo <= 3u && (d >> Lo) == mix_prefix;
                    const u32 so = (1u << (Lo & 3u)) | (d & ((1u << (Lo & 3u)) - 1u));
                    mwhit[q] = in_old;
                    mwf[q] = w16s[in_old ? so : 0u];
                    mwa[q] = na;
                    mwl[q] = a16m[na];
                }
            }
            mix_live = true;
            mix_hm = hm;
            mix_prefix = prefix;
        };
        auto mixw_arrive = [&]() {                          // after take_prefetched: the loads have landed
#pragma unroll
            for (int q = 0; q < MQ; q++) {
                const u32 sl = (u32)li + (u32)q * GG;
                if (sl >= 1u && sl <= 15u) w16s[sl] = (u16)(mwhit[q] ? mwf[q] : mwl[q]);
            }
        };

        // Two-hypothesis decoder with a MIX2 (level 4, HYP16): the candidates are asked for PER HYPOTHESIS, a bit step early, like
        // the rows.  When three bits of a nibble are known each copy derives the next nibble's prefix under ITS outcome of the
        // fourth bit and its eight lanes fetch the fifteen candidates (lane pair q: slots q and q + 8).  When the bit is known
        // (mixs_arrive, behind take_prefetched) copy 0 writes the ended nibble's candidates back, the copy that was right
        // parks its loaded values in LDS -- an entry that was also a candidate of the nibble that just ended is taken from
        // LDS, where its trained value sits (its load went out before that nibble's last training and write-back) -- and both
        // copies adopt its addresses.  ZPQ_MIXW_SPEC=0 at build time: ask after the bit, as the one-copy decoders do.
#ifndef ZPQ_MIXW_SPEC
#define ZPQ_MIXW_SPEC 1
#endif
        constexpr bool MIXS = HYP && MIXT && (ZPQ_MIXW_SPEC != 0);
        u32 ms_cur[2] = {0, 0}, ms_new[2] = {0, 0}, ms_ld[2] = {0, 0};
        u32 ms_hm = 0, ms_prefix = 1;                      // what THIS copy's request in flight was derived from
        u32 hctx_mix = 0, hn_spec_mix = 0, hnext_mix = 0;  // the MIX2's context hash: this byte / next byte under this copy's outcome / next byte
        auto mixs_request = [&](const u32 hm, const u32 prefix) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32 sl = (u32)(li >> 1) + 8u * (u32)q;
                const u32 slv = max(sl, 1u);                 // (slot 0 does not exist: lane pair 0 re-reads slot 1, unused)
                const u32 L = 31u - (u32)__clz((int)slv);
                const u32 c8c = (prefix << L) | (slv - (1u << L));
                const u32 na = (hm + (c8c & mix_mask)) & mix_cmask;
                ms_new[q] = na;
                ms_ld[q] = a16m[na];
            }
            ms_hm = hm;
            ms_prefix = prefix;
        };
        auto mixs_arrive = [&]() {
            const u32 plen_old = 32u - (u32)__clz((int)mix_prefix);
            u32 fwd[2];
            bool hit[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32 sl = (u32)(li >> 1) + 8u * (u32)q;
                if (mix_live && hyp == 0 && sl >= 1u) a16m[ms_cur[q]] = w16s[sl];     // the nibble that ended: values are final
                const u32 d = (ms_new[q] - mix_hm) & mix_cmask;
                const u32 dlen = 32u - (u32)__clz((int)(d | 1u));
                const u32 Lo = dlen - plen_old;
                hit[q] = mix_live && d != 0u && d <= 255u && dlen >= plen_old && Lo <= 3u && (d >> Lo) == mix_prefix;
                const u32 so = (1u << (Lo & 3u)) | (d & ((1u << (Lo & 3u)) - 1u));
                fwd[q] = w16s[hit[q] ? so : 0u];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const u32 sl = (u32)(li >> 1) + 8u * (u32)q;
                if (row_mine && sl >= 1u) w16s[sl] = (u16)(hit[q] ? fwd[q] : ms_ld[q]);
                const u32 xa = xchg(ms_new[q]);
                ms_cur[q] = row_mine ? ms_new[q] : xa;
            }
            const u32 xh = xchg(ms_hm), xp = xchg(ms_prefix);
            mix_hm = row_mine ? ms_hm : xh;
            mix_prefix = row_mine ? ms_prefix : xp;
            mix_live = true;
        };

#ifdef ZPQ_NO_SKEW
        constexpr bool SKEW = false;
#else
        constexpr bool SKEW = !DEC && NCH > 0;              // byte-skewed chain pipeline (see the main loop)
#endif
        i32 pout[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // this lane's predictions for the byte it is coding
        i32 pin_cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // its predecessor's predictions for that byte
        i32 pj_cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // MIX2 lane: the predecessor's input (= p[j])

        u32 hn_main = 0, hn_alt = 0;                       // TWO: the next byte's context hash for last bit 0 / 1
        auto bitstep = [&](auto kc, auto nbc) {
            constexpr int K = decltype(kc)::value;
            constexpr int bit = (decltype(nbc)::value ? 3 : 7) - K;   // bit of the byte this step codes (7..0)
            constexpr int KB = 7 - bit;                               // its position in coding order (0..7)
            // (plain form: cur_* of bits 1..3 were fetched at the end of the previous bit step, see (4b))
            const u32 s = cur_s;
            const i32 yk = DEC ? 0 : (i32)((ch >> bit) & 1u);  // encode knows its bit up front
            // ---- (1) next bit's candidate states and their table entries, read BEFORE this
            //          bit's update is stored; slots 2*slot and 2*slot+1 are adjacent row bytes
            u32 sA = 0, sB = 0;
            u32 rAv = 0, rBv = 0;
            i32 rAb = 0, rBb = 0;
            if (SPEC && K < 3) {
                u32 pair;
                if (K == 0) pair = X.r0 >> 16;
                else if (K == 1) pair = X.r1 >> ((X.slot & 1u) * 16u);
                else pair = ((X.slot & 2u) ? X.r3 : X.r2) >> ((X.slot & 1u) * 16u);
                sA = pair & 255u;
                sB = (pair >> 8) & 255u;
                if (DEC) { rAv = t32[sA]; rBv = t32[sB]; rAb = (i32)(int8_t)t8[sA]; rBb = (i32)(int8_t)t8[sB]; }
                else { sA = yk ? sB : sA; rAv = t32[sA]; rAb = (i32)(int8_t)t8[sA]; }
            }
            const u32 ns01 = *reinterpret_cast<const u16 *>(s_ns + s * 4);   // next state for y=0 | y=1 << 8
            // ---- (2) predict: chain p0 -> p1 -> ... (predictor.v:555-563,615-631)
            const u32 cmv = cur_v & 0x7FFFFFu;                                   // ICM lanes
            const i32 w0 = ((i32)(cur_v << 12)) >> 12;                           // ISSE lanes: sext20
            const i32 w1 = (i32)(((u32)cur_b << 12) | (cur_v >> 20));
            i32 p = is_icm ? cur_pst : 0, pin = 0;
            // (decoders of the specialised chains, round 4: the hop without an `old` operand tied to its destination -- no copy in
            //  front of the v_mov_b32_dpp; lane 0 of a row reads 0, it is an ICM -- and the lane's INPUT taken by one more hop once the
            //  chain is through instead of a select per link.  See chain_of, where this was measured.)
            constexpr bool DIET = DEC && NCH > 0;
            auto chain_step = [&](const int i) {
                const i32 pv = DIET ? __builtin_amdgcn_mov_dpp(p, 0x111 /*row_shr:1*/, 0xf, 0xf, true) : row_shr1(p);
                const i32 pn = clamp2k((__mul24(w0, pv) + (w1 << 6)) >> 16);   // |w0|<2^18, |pv|<=2^11: exact in 32 bits
                const bool me = lc == i;
                if (!DIET) pin = me ? pv : pin;
                p = me ? pn : p;
            };
            if constexpr (SKEW) {
                // every ISSE lane at once, each on its own byte, fed by what its predecessor handed over
                pin = pin_cur[KB];
                const i32 pn = clamp2k((__mul24(w0, pin) + (w1 << 6)) >> 16);
                p = is_icm ? cur_pst : pn;
            } else if (NCH) {
#pragma unroll
                for (int i = 1; i < (NCH ? NCH : 1); i++) chain_step(i);
                if (DIET) pin = __builtin_amdgcn_mov_dpp(p, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
            } else {
                for (int i = 1; i < nisse_end; i++) chain_step(i);
            }
            i32 pj = 0, pk = 0, wmix = 0;
            u32 mcx = 0;
            if (has_mix2) {
                if constexpr (SKEW) { pj = pj_cur[KB]; pk = pin_cur[KB]; }
                else if constexpr (DIET) {
                    // the specialised chains' MIX2 mixes the two components right below it (k = n - 2, j = n - 3: checked on the
                    // host): two DPP hops instead of two trips through the LDS crossbar on the path to the coder
                    pk = pin;
                    pj = __builtin_amdgcn_mov_dpp(p, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
                } else {
                    pj = row_bcast(p, row_base + mix_j);
                    pk = row_bcast(p, row_base + mix_k);
                }
                if (!DEC && mixreg) {
                    // encode: the byte's eight weights sit in registers since the byte began (mix_byte_begin)
                    wmix = (i32)mw[7 - bit];
                    if (is_last) p = clamp2k(wadd(wmul(wmix, pj), wmul(65536 - wmix, pk)) >> 16);
                } else if (DEC && mixreg) {
                    wmix = (i32)w16s[X.slot];
                    const i32 pm = clamp2k(wadd(mul24x(wmix, pj), mul24x(65536 - wmix, pk)) >> 16);
                    p = is_last ? pm : p;                             // (a select on every lane, not a divergent region for one)
                } else if (is_last) {
                    mcx = (hctx + (X.c8 & mix_mask)) & mix_cmask;
                    wmix = a16[mcx];
                    p = clamp2k(wadd(wmul(wmix, pj), wmul(65536 - wmix, pk)) >> 16);
                }
            }
            if constexpr (SKEW) pout[KB] = p;
            const i32 sq = s_squash[p + 2048];      // squash(p[li]) (predictor.v:193-202,667)
            // ---- (3) off the critical path: the ICM's stretch for the next bit, for both
            //          outcomes, with this bit's cm update forwarded when the state repeats
            // cm += (y*32767 - (cm >> 8)) >> 2 (predictor.v:706-708).  Encode knows y and computes one outcome; the
            // speculative decoder wants both ahead of the coder; the plain decoder computes its one after the bit.
            auto cm_next = [&](const i32 yy) -> u32 { return (u32)wadd((i32)cmv, ((yy ? 32767 : 0) - (i32)(cmv >> 8)) >> 2); };
            const u32 cm0 = (DEC && SPEC) ? cm_next(0) : 0u, cm1 = (DEC && SPEC) ? cm_next(1) : 0u;
            u32 cmn = DEC ? 0u : cm_next(yk);
            // the trained entry's stretch travels with it; encode knows y already, so it is computed here,
            // off the path to the coder; decode computes it after the bit is known
            i32 stA = 0, stB = 0, st0 = 0, st1 = 0, st_new = 0;
            if (!DEC) st_new = stretch_lds(cmn);
            else if (SPEC) { st0 = stretch_lds(cm0); st1 = stretch_lds(cm1); }
            if (SPEC && K < 3) {
                if (DEC) {
                    stA = sA == s ? st0 : icm_st(rAv, rAb);
                    stB = sB == s ? st1 : icm_st(rBv, rBb);
                } else {
                    stA = sA == s ? st_new : icm_st(rAv, rAb);
                }
            }
            // ---- (4) code the bit on the lane that owns the final prediction
            i32 y = yk;
            if constexpr (DIET) {
                // (as in bitstep_hyp: the split on every lane -- lanes that are no coder hold a range that means nothing and is
                //  never renormalised -- so that only the loop is a divergent region)
                const u32 p16 = (u32)sq * 2u + 1u;
                const u32 mid = X.low + mul_shr16(X.high - X.low, p16);
                y = X.code <= mid ? 1 : 0;
                X.high = y ? mid : X.high;
                X.low = y ? X.low : mid + 1;
                while (is_last && (X.high ^ X.low) < 0x1000000u) {
                    X.low <<= 8; X.high = (X.high << 8) | 255u; X.low = X.low ? X.low : 1u;
                    const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin); X.code = (X.code << 8) | c;
                }
            } else if (is_last) {
                const u32 p16 = (u32)sq * 2u + 1u;
                const u32 mid = X.low + mul_shr16(X.high - X.low, p16);
                if (DEC) y = X.code <= mid ? 1 : 0;
                X.high = y ? mid : X.high;
                X.low = y ? X.low : mid + 1;
                while ((X.high ^ X.low) < 0x1000000u) {
                    if (!DEC) put_byte(X.high >> 24);
                    X.low <<= 8; X.high = (X.high << 8) | 255u; X.low = X.low ? X.low : 1u;
                    if (DEC) { const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin); X.code = (X.code << 8) | c; }
                }
            }
            if (DEC) y = bcast_down(y);
            if (DEC && K == 3 && !TWO) {
                // the nibble's last bit is known: request the next nibble's rows now, so that their
                // latency overlaps this bit's update work (contexts: predictor.v:558-560,809-816)
                const u32 c8n = (X.c8 << 1) | (u32)y;
                if (bit == 4) { if (TWOM) sel_alt = y != 0; else prefetch_rows(hctx, c8n); }
                else { hnext_dec = run_vm(c8n - 256u); prefetch_rows(hnext_dec, 1u); }
            }
            if (DEC && K == 3 && TWO) {
                sel_alt = y != 0;
                if (bit == 0) { hnext_dec = y ? hn_alt : hn_main; vm_commit(((X.c8 << 1) | (u32)y) - 256u); }
            }
            // ---- (4b) plain form: the bit is known, so is the next slot (2*slot + y).  Fetch the next bit's
            //           entry NOW, before this bit's update is computed and stored -- otherwise the read
            //           would queue behind the store and behind the update's own stretch lookup (two
            //           LDS round trips in series).  The update is forwarded in (6) when the state repeats.
            u32 sN = 0, rNv = 0;
            i32 rNb = 0;
            if (!SPEC && K < 3) {
                u32 pair;
                if (K == 0) pair = X.r0 >> 16;
                else if (K == 1) pair = X.r1 >> ((X.slot & 1u) * 16u);
                else pair = ((X.slot & 2u) ? X.r3 : X.r2) >> ((X.slot & 1u) * 16u);
                sN = y ? ((pair >> 8) & 255u) : (pair & 255u);
                rNv = t32[sN];
                rNb = (i32)(int8_t)t8[sN];
            }
            // ---- (5) update (predictor.v:701-709,776-791): one 8-byte LDS store per lane
            const i32 err = (y ? 32767 : 0) - sq;
            const i32 nw0 = clamp512k(w0 + ((__mul24(err, pin) + (1 << 12)) >> 13));  // |err|<2^15, |pin|<=2^11
            const i32 nw1 = clamp512k(w1 + ((err + 16) >> 5));
            if (DEC) {
                cmn = SPEC ? (y ? cm1 : cm0) : cm_next(y);
                st_new = SPEC ? (y ? st1 : st0) : stretch_lds(cmn);            // (both outcomes ahead of the coder: measured slower)
            }
            const u32 nv = is_icm ? (cmn | (((u32)st_new & 0x1FFu) << 23)) : (((u32)nw0 & 0xFFFFFu) | ((u32)nw1 << 20));
            const i32 nb = is_icm ? (st_new >> 9) : (nw1 >> 12);
            t32[s] = nv;
            t8[s] = (u8)nb;
            if constexpr (DIET && MIXT) {
                // (specialised decoders: the weight's training on every lane, the store under an index select -- slot 0 holds no
                //  candidate -- instead of a divergent region for the MIX2's one lane)
                const i32 em = mul24x(err, mix_rate) >> 5;
                i32 w = wadd(wmix, wadd(mul24x(em, pj - pk), 1 << 12) >> 13);
                w = min(max(w, 0), 65535);
                w16s[(ctype == ZT_MIX2) ? X.slot : 0u] = (u16)w;
            } else if (has_mix2 && ctype == ZT_MIX2) {
                const i32 em = wmul(err, mix_rate) >> 5;
                i32 w = wadd(wmix, wadd(wmul(em, pj - pk), 1 << 12) >> 13);
                w = min(max(w, 0), 65535);
                if (!DEC && mixreg) mw[7 - bit] = (u32)w;
                else if (DEC && mixreg) w16s[X.slot] = (u16)w;
                else a16[mcx] = (u16)w;
            }
            if (DEC && K == 3 && mixreg) {
                // this nibble's weights are final: write the candidates back and request the next nibble's
                // (after the update above; stores before loads, an entry may recur under another context hash)
                // (hash and prefix from the coder lane: idle lanes above it never see the decoded bits)
                const u32 c8n2 = (u32)row_bcast((i32)((X.c8 << 1) | (u32)y), row_base + last);
                if (bit == 4) mixw_request(row_bcast((i32)hctx, row_base + last), c8n2);
                else mixw_request(row_bcast((i32)hnext_dec, row_base + last), 1u);
            }
            // ---- (6) hand the next bit its state, forwarded entry and stretch
            if (SPEC && K < 3) {
                const u32 sn = DEC ? (y ? sB : sA) : sA;
                const u32 rv = DEC ? (y ? rBv : rAv) : rAv;
                const i32 rb = DEC ? (y ? rBb : rAb) : rAb;
                const bool same = sn == s;
                cur_v = same ? nv : rv;
                cur_b = same ? nb : rb;
                cur_pst = DEC ? (y ? stB : stA) : stA;
                cur_s = sn;
            }
            if (!SPEC && K < 3) {
                const bool same = sN == s;
                cur_v = same ? nv : rNv;
                cur_b = same ? nb : rNb;
                cur_pst = icm_st(cur_v, cur_b);
                cur_s = sN;
            }
            // next bit-history state into the row (statetable.v:75-84)
            const u32 nsv = y ? (ns01 >> 8) : (ns01 & 255u);
            const u32 sh = (X.slot & 3u) * 8u;
            const u32 dsel = (K <= 1) ? X.r0 : (K == 2 ? X.r1 : ((X.slot & 4u) ? X.r3 : X.r2));
            const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
            if (K <= 1) X.r0 = ins;
            else if (K == 2) X.r1 = ins;
            else { X.r3 = (X.slot & 4u) ? ins : X.r3; X.r2 = (X.slot & 4u) ? X.r2 : ins; }
            X.c8 = (X.c8 << 1) | (u32)y;
            X.slot = (K == 3) ? 1u : (X.slot * 2u + (u32)y);
            if (TWOM && bit == 5) {
                const u32 c8n = X.c8 << 1;                       // (mid-byte: both values of the fourth bit, see TWOM)
                prefetch_alt(hctx, c8n | 1u);
                prefetch_rows(hctx, c8n);
            }
            if (TWO && K == 2) {
                // three bits of the nibble are known: request the next nibble's rows for both values of the fourth
                const u32 c8n = X.c8 << 1;
                if (bit == 5) { prefetch_alt(hctx, c8n | 1u); prefetch_rows(hctx, c8n); }
                else {
                    hn_main = vm_hash(c8n - 256u);
                    hn_alt = vm_hash((c8n | 1u) - 256u);
                    prefetch_alt(hn_alt, 1u);
                    prefetch_rows(hn_main, 1u);
                }
            }
        };

        // Two-hypothesis decode step (see HYP above).  Program order = what may run beside what: the chain and the
        // squash request first; then everything this copy can do for ITS assumed bit yh without the squash value
        // (next state and its entry, the counter's next value and its stretch, the next nibble's rows); then the
        // squash-dependent training; then the coder; then commit / take-over.
        u32 hn_spec = 0;                                   // next byte's context hash under this copy's outcome of the last bit
        // The prediction chain of ONE bit from every lane's table entry (v, b): p0 -> p1 -> ... (-> MIX2), each copy among its
        // own lanes.  Returns this lane's prediction, its input, and (MIX2) what its training needs.
        struct Pred { i32 p, pin, pj, pk, wmix; };
        auto chain_of = [&](const u32 v, const i32 b, const u32 mslot) -> Pred {
            const i32 w0c = ((i32)(v << 12)) >> 12;
            const i32 w1c = (i32)(((u32)b << 12) | (v >> 20));
            Pred R;
            R.p = is_icm ? icm_st(v, b) : 0; R.pin = 0; R.pj = 0; R.pk = 0; R.wmix = 0;
#pragma unroll
            for (int i = 1; i < (NCH ? NCH : 1); i++) {
#ifdef ZPQ_DPP_OLD
                const i32 pv = __builtin_amdgcn_update_dpp(R.p, R.p, 0x112 /*row_shr:2: the same copy of the component below*/, 0xf, 0xf, false);
#else
                // (no `old` operand tied to the destination: no copy in front of the v_mov_b32_dpp.  Lanes 0 and 1 of a DPP row have
                //  no lane two below and read 0; they, and lanes 8 and 9 where a row carries two blocks of eight, are the ICM's
                //  copies, which use no input prediction)
                const i32 pv = __builtin_amdgcn_mov_dpp(R.p, 0x112 /*row_shr:2: the same copy of the component below*/, 0xf, 0xf, true);
#endif
                const i32 pn = clamp2k((__mul24(w0c, pv) + (w1c << 6)) >> 16);
                R.p = (lc == i) ? pn : R.p;
            }
            // every lane's INPUT is the finished prediction of the lane two below: one more hop once the chain is through (it is
            // needed by the training only, off the path to the coder) instead of a select per link
            R.pin = __builtin_amdgcn_mov_dpp(R.p, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
            if constexpr (MIXT) {
                // the MIX2 mixes p[NCH - 2] and p[NCH - 1] (checked on the host) of ITS copy: the lanes four and two below
                R.pk = R.pin;
                R.pj = __builtin_amdgcn_mov_dpp(R.p, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
                R.wmix = (i32)w16s[mslot];                    // the nibble's candidate weights live in LDS (mixw_request / mixs_arrive)
                const i32 pm = clamp2k(wadd(mul24x(R.wmix, R.pj), mul24x(65536 - R.wmix, R.pk)) >> 16);
                R.p = is_last ? pm : R.p;
            }
            return R;
        };
        // AHEAD (round 4): inside a nibble each copy also runs the NEXT bit's chain and squash lookup for its outcome of this
        // bit -- its next state's entry with its own training forwarded is all the chain needs -- BEFORE the coder resolves
        // this bit.  The copy that was right hands (input, squash, MIX2 operands) over with the entry, and the next step starts
        // at its coder: the chain's DPP hops and the squash's LDS round trip leave the bit-to-bit critical path (they were
        // ~200 cycles of the ~340 a bit step spends waiting).  The first bit of a nibble has a new row and predicts as before.
        // MEASURED SLOWER on every level (same box, alternating runs, profiles/r04_ahead_ab.txt): level 2 x 8192 272-274 against
        // 257.7-257.9 ms, level 1 282 against 261, level 3 360 against 337, level 4 484 against 455-459.  The bit step's one LDS
        // round trip does not go away -- the next state's entry and the ICM's stretch of its trained counter (needed when the state
        // repeats) must have arrived before the early chain can start, where they used to arrive behind the coder -- and the
        // hand-over grows by two to five registers.  Parity-green (the whole GPU suite ran on it).  Compiled only with -DZPQ_AHEAD.
#ifdef ZPQ_AHEAD
        constexpr bool AHEAD = HYP;
#else
        constexpr bool AHEAD = false;
#endif
        i32 a_pin = 0, a_sq = 0, a_pj = 0, a_pk = 0, a_wmix = 0;
        auto bitstep_hyp = [&](auto kc, auto nbc) {
            constexpr int K = decltype(kc)::value;
            constexpr int bit = (decltype(nbc)::value ? 3 : 7) - K;
            const u32 s = cur_s;
            const u32 ns01 = *reinterpret_cast<const u16 *>(s_ns + s * 4);
            const u32 cmv = cur_v & 0x7FFFFFu;
            const i32 w0 = ((i32)(cur_v << 12)) >> 12;
            const i32 w1 = (i32)(((u32)cur_b << 12) | (cur_v >> 20));
            i32 pin, pj, pk, wmix, sq;
            if constexpr (AHEAD && K > 0) {
                pin = a_pin; pj = a_pj; pk = a_pk; wmix = a_wmix; sq = a_sq;     // predicted during the previous bit
            } else {
                const Pred P = chain_of(cur_v, cur_b, X.slot);
                pin = P.pin; pj = P.pj; pk = P.pk; wmix = P.wmix;
                sq = s_squash[P.p + 2048];
            }
            // ---- this copy's outcome
            const i32 yh = hyp;
            u32 sN = 0, rNv = 0;
            i32 rNb = 0;
            if (K < 3) {
                u32 pair;
                if (K == 0) pair = X.r0 >> 16;
                else if (K == 1) pair = X.r1 >> ((X.slot & 1u) * 16u);
                else pair = ((X.slot & 2u) ? X.r3 : X.r2) >> ((X.slot & 1u) * 16u);
                sN = yh ? ((pair >> 8) & 255u) : (pair & 255u);
                rNv = t32[sN];
                rNb = (i32)(int8_t)t8[sN];
            }
            const u32 cmn = (u32)wadd((i32)cmv, ((yh ? 32767 : 0) - (i32)(cmv >> 8)) >> 2);
            const i32 st_new = stretch_of(cmn);
            const i32 err = (yh ? 32767 : 0) - sq;
            const i32 nw0 = clamp512k(w0 + ((__mul24(err, pin) + (1 << 12)) >> 13));
            const i32 nw1 = clamp512k(w1 + ((err + 16) >> 5));
            // (The compiler turns these selects into a divergent if / else around the two roles' code: five exec-mask instructions
            //  per bit.  Bitwise merges under a per-lane role mask remove them -- and were measured SLOWER: level 2 257.8-259.3
            //  against 253.6-255.3 ms, level 1 264.8 against 258.0, level 3 334.4 against 327.5: with the if / else each role's
            //  lanes skip the other role's LDS reads, which the merged form issues for every lane.)
            const u32 nv_icm = cmn | (((u32)st_new & 0x1FFu) << 23), nv_isse = ((u32)nw0 & 0xFFFFFu) | ((u32)nw1 << 20);
            const u32 nv = is_icm ? nv_icm : nv_isse;
            const i32 nb = is_icm ? (st_new >> 9) : (nw1 >> 12);
            u32 nxt_v = 0, nxt_bs = 0;
            if (K < 3) {
                const bool same = sN == s;
                nxt_v = same ? nv : rNv;
                nxt_bs = ((u32)(same ? nb : rNb) & 255u) | (sN << 8);
            }
            // ---- the next bit's chain under this copy's outcome (see AHEAD)
            Pred Q = {0, 0, 0, 0, 0};
            i32 sq2 = 0;
            if constexpr (AHEAD && K < 3) {
                Q = chain_of(nxt_v, (i32)(int8_t)(nxt_bs & 255u), X.slot * 2u + (u32)yh);
                sq2 = s_squash[Q.p + 2048];
            }
            // ---- the bit (both copies decode it: same code, same window, same bounds).  The split runs on EVERY lane -- lanes that
            // are not a coder hold a range of their own that means nothing and is never renormalised (no memory access outside the
            // loop) -- so that only the loop is a divergent region: one exec-mask save / branch / restore less per bit.
            i32 y;
            {
                const u32 p16 = (u32)sq * 2u + 1u;
                const u32 mid = X.low + mul_shr16(X.high - X.low, p16);
                y = X.code <= mid ? 1 : 0;
                X.high = y ? mid : X.high;
                X.low = y ? X.low : mid + 1;
                while (is_last && (X.high ^ X.low) < 0x1000000u) {
                    X.low <<= 8; X.high = (X.high << 8) | 255u; X.low = X.low ? X.low : 1u;
                    const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin); X.code = (X.code << 8) | c;
                }
            }
            y = bcast_down(y);
            // ---- commit / take over
            const bool mine = y == yh;
            {
                // one copy trains the (shared) table entry; the other stores into the workgroup's dummy table -- two address selects
                // instead of an exec-mask region around the stores (level 2 249.0 -> 246.3 ms, level 3 323.2 -> 319.2)
                u32 *const w32 = mine ? t32 : reinterpret_cast<u32 *>(dummy);
                u8 *const w8 = mine ? t8 : dummy + 1024;
                w32[s] = nv; w8[s] = (u8)nb;
            }
            if constexpr (MIXT) {
                // MIX2 weight (predictor.v:744-762), trained by the copy that was right
                const i32 em = mul24x(err, mix_rate) >> 5;
                i32 w = wadd(wmix, wadd(mul24x(em, pj - pk), 1 << 12) >> 13);
                w = min(max(w, 0), 65535);
                w16s[(mine & mix_lane) ? X.slot : 0u] = (u16)w;           // (slot 0 holds no candidate: everybody else's store lands there; level 4 433.6 -> 431.0 ms)
                if (K == 3 && !MIXS) {
                    // this nibble's weights are final: write the candidates back and request the next nibble's (all sixteen
                    // lanes of the row, after the bit: hash and prefix from the coder lane, as in bitstep)
                    const u32 c8n2 = (u32)row_bcast((i32)((X.c8 << 1) | (u32)y), row_base + comp_lane(last));
                    if (bit == 4) mixw_request((u32)row_bcast((i32)hctx, row_base + comp_lane(last)), c8n2);
                    else {
                        const u32 xh2 = xchg(hn_spec);
                        const u32 hn_true = mine ? hn_spec : xh2;
                        mixw_request((u32)row_bcast((i32)hn_true, row_base + comp_lane(last)), 1u);
                    }
                }
            }
            if (K < 3) {
                const u32 xv = xchg(nxt_v), xb = xchg(nxt_bs);
                cur_v = mine ? nxt_v : xv;
                const u32 bs = mine ? nxt_bs : xb;
                cur_b = (i32)(int8_t)(bs & 255u);
                cur_s = bs >> 8;
                cur_pst = icm_st(cur_v, cur_b);
                if constexpr (AHEAD) {
                    const i32 xq = (i32)xchg((u32)sq2), xi = (i32)xchg((u32)Q.pin);
                    a_sq = mine ? sq2 : xq;
                    a_pin = mine ? Q.pin : xi;
                    if constexpr (MIXT) {
                        const i32 xj = (i32)xchg((u32)Q.pj), xk = (i32)xchg((u32)Q.pk), xw = (i32)xchg((u32)Q.wmix);
                        a_pj = mine ? Q.pj : xj; a_pk = mine ? Q.pk : xk; a_wmix = mine ? Q.wmix : xw;
                    }
                }
            } else {
                if (HYP4 && bit == 4) sel_alt = y != 0;         // (the copy whose third bit was right holds both outcomes of this one)
                else row_mine = mine;
                if (bit == 0) {
                    const u32 xh = xchg(hn_spec);
                    hnext_dec = mine ? hn_spec : xh;
                    if (MIXS) { const u32 xm = xchg(hn_spec_mix); hnext_mix = mine ? hn_spec_mix : xm; }
                    vm_commit(((X.c8 << 1) | (u32)y) - 256u);
                }
            }
            // next bit-history state into the row (statetable.v:75-84), with the decoded bit
            const u32 nsv = y ? (ns01 >> 8) : (ns01 & 255u);
            const u32 sh = (X.slot & 3u) * 8u;
            const u32 dsel = (K <= 1) ? X.r0 : (K == 2 ? X.r1 : ((X.slot & 4u) ? X.r3 : X.r2));
            const u32 ins = (dsel & ~(255u << sh)) | (nsv << sh);
            if (K <= 1) X.r0 = ins;
            else if (K == 2) X.r1 = ins;
            else { X.r3 = (X.slot & 4u) ? ins : X.r3; X.r2 = (X.slot & 4u) ? X.r2 : ins; }
            X.c8 = (X.c8 << 1) | (u32)y;
            X.slot = (K == 3) ? 1u : (X.slot * 2u + (u32)y);
            if (HYP4 && bit == 6) {
                // two bits of the byte's first nibble are known: the second nibble's rows for this copy's outcome of the
                // third bit, both outcomes of the fourth (see HYP4 above)
                const u32 c8n = (X.c8 << 2) | ((u32)hyp << 1);
                prefetch_alt(hctx, c8n | 1u);
                prefetch_rows(hctx, c8n);
            }
            if (HYP4 && bit == 5) row_mine = mine;
            if (K == 2 && !(HYP4 && bit == 5)) {
                // Three bits of the nibble are known: this copy asks for the rows of the next nibble under ITS outcome of
                // the fourth.  The request has the whole last bit step (~1000 cycles) to travel before it is needed;
                // the HBM round trip measured here is ~1350 cycles.
                const u32 c8n = (X.c8 << 1) | (u32)hyp;
                if (bit == 5) { prefetch_rows(hctx, c8n); if (MIXS) mixs_request(hctx_mix, c8n); }
                else {
                    hn_spec = vm_hash(c8n - 256u);
                    prefetch_rows(hn_spec, 1u);
                    if (MIXS) { hn_spec_mix = vm_last; mixs_request(hn_spec_mix, 1u); }
                }
                // (Asking one bit earlier still -- both outcomes of the last bit under this copy's outcome of the third,
                //  four lines per nibble and table -- was measured: 320.7 vs 266.6 ms.  The memory system is loaded
                //  enough that doubling the row reads costs more latency than the earlier request hides.)
            }
        };
        auto step = [&](auto kc, auto nbc) {
            if constexpr (HYP) bitstep_hyp(kc, nbc); else bitstep(kc, nbc);
        };

        // nibble start: find_ht (predictor.v:495-532); three rows of one 64-byte line

        prefetch_rows(0u, 1u);                             // first nibble of the first byte: h = 0, c8 = 1
        if (DEC && mixreg) { if (MIXS) mixs_request(0u, 1u); else mixw_request(0u, 1u); }
        // Encode, specialised kernels: the prediction chain as a PIPELINE over bytes.  Every context, bit and
        // bit-history state of the encoder is a function of the input alone; only predictions flow down the chain.
        // So in iteration `it` lane c works on byte it - c: the ICM is one byte ahead of the first ISSE, which is one
        // byte ahead of the next, ...; a lane's eight predictions of a byte are handed to its successor by eight DPP
        // moves when the iteration ends.  Inside an iteration no lane waits for another: the per-bit dependent chain
        // ICM -> ISSE -> ... -> coder (one DPP + multiply + clamp per link, in series) becomes one link per bit.
        // The coder stays on the last component's lane and so sees bytes in order; it - c < 0 or >= total: lane idle.
        bool pend_store = false;                           // decode: the last decoded byte still has to be stored
        u32 pend_pos = 0, pend_val = 0;
        const u32 skew_delay = SKEW ? (u32)li : 0u;
        const u32 iters = SKEW ? total + (u32)last : total;
        // HIO: the byte loop runs in two phases with the host hand-shake between them (wave-uniform iteration counts: the
        // encoder's lanes are at most 15 bytes apart, every decoder iteration after the PP byte stores one byte)
        const u32 split = !HIO ? iters
                               : (DEC ? (B.prog_counter ? B.prog_pos + ((B.flags & ZPQ_FLAG_PP) ? 1u : 0u) : iters)
                                      : (B.gate_flag ? (B.gate_pos > 160u ? B.gate_pos - 160u : 0u) : iters));
        u32 it = 0;
        bool stop = false;
        for (int phase = 0; phase < (HIO ? 2 : 1); phase++) {
        const u32 it_end = (HIO && phase == 0) ? min(iters, split) : iters;
        for (; it < it_end; it++) {
            const u32 bi = it - skew_delay;
            // (lanes beyond the last component stay in step on dummy tables: masking them off made level 1 30 % slower --
            //  measured 286 vs 219 ms, cause not understood)
            if (!SKEW || bi < total) {
            if (DEC) { dec_adopt(); dec_request(X.ipos); }
            if (!DEC) {
                const u32 cpos = (B.flags & ZPQ_FLAG_PP) ? bi - 1u : bi;
                const u32 cb = enc_byte((B.flags & ZPQ_FLAG_PP) && bi == 0 ? 0u : cpos);
                ch = ((B.flags & ZPQ_FLAG_PP) && bi == 0) ? 0u : cb;
            }
            // ---- EOF flag: encode(0,0) / decode(0)  (encoder.v:108, decoder.v:128)
            if (!DEC) {
                if (is_last) {
                    X.low += 1;                               // p=0, y=0: mid = low, low = mid+1
                    while ((X.high ^ X.low) < 0x1000000u) {
                        put_byte(X.high >> 24);
                        X.low <<= 8; X.high = (X.high << 8) | 255u; if (X.low == 0) X.low = 1;
                    }
                }
            } else {
                i32 eof = 0;
                if constexpr (NCH > 0) {
                    // (as in the bit steps: the split on every lane, only the renormalisation loop is a divergent region)
                    eof = X.code <= X.low ? 1 : 0;                                                  // p=0: mid = low
                    X.high = eof ? X.low : X.high;
                    X.low = eof ? X.low : X.low + 1;
                    while (is_last && (X.high ^ X.low) < 0x1000000u) {
                        X.low <<= 8; X.high = (X.high << 8) | 255u; if (X.low == 0) X.low = 1;
                        const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin);
                        X.code = (X.code << 8) | c;
                    }
                } else if (is_last) {
                    if (X.code <= X.low) { eof = 1; X.high = X.low; } else { X.low = X.low + 1; }   // p=0: mid = low
                    while ((X.high ^ X.low) < 0x1000000u) {
                        X.low <<= 8; X.high = (X.high << 8) | 255u; if (X.low == 0) X.low = 1;
                        const u32 c = in_byte(X.ipos); X.ipos += (X.ipos < nin);
                        X.code = (X.code << 8) | c;
                    }
                }
                eof = row_bcast(eof, row_base + comp_lane(last));
                if (eof) { stop = true; break; }
            }

            X.c8 = 1; X.slot = 1;
            u32 hnext = 0;
            take_prefetched(bi != 0, std::false_type{});      // rows of this byte's first nibble
            if (DEC && pend_store) { dst[pend_pos] = (u8)pend_val; pend_store = false; }   // (see pend_store)
            if (!DEC && is_last) oq_flush();
            if (DEC && mixreg) { if (MIXS) mixs_arrive(); else mixw_arrive(); }
            if (!DEC && mixreg) mix_byte_begin(ch, bi != 0);  // (before the prefetch below, see mix_byte_begin)
            if (!DEC) {
                hnext = run_vm(ch);                           // contexts of the NEXT byte: known now
                prefetch_rows(hctx, 16u | (ch >> 4));         // second nibble of this byte, a nibble ahead
            }
            nibble_begin();
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});     // decode: requests the next rows inside
            take_prefetched(true, std::true_type{});
            if (DEC && mixreg) { if (MIXS) mixs_arrive(); else mixw_arrive(); }
            if (!DEC) prefetch_rows(hnext, 1u);               // first nibble of the next byte
            nibble_begin();
            step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            const u32 byte = X.c8 - 256;
            hctx = DEC ? hnext_dec : hnext;
            if (MIXS) hctx_mix = hnext_mix;

            if (DEC) {
                if ((B.flags & ZPQ_FLAG_PP) && !got_first) { first = byte; got_first = true; }
                else {
                    // The byte is STORED behind the next iteration's wait for its rows: a store issued here would be
                    // the youngest memory operation when that wait comes, and vmcnt retires in order.
                    pend_store = is_last && hyp == 0 && X.opos < cap;                 // (one copy writes)
                    pend_pos = X.opos;
                    pend_val = byte;
                    X.opos++;                                 // uniform across the group when decoding
                    if (X.opos > cap) { stop = true; break; }
                }
            }
            }   // active
            if constexpr (SKEW) {
                // hand this byte's predictions down the chain (lane c-1 -> lane c); a MIX2 lane also receives its
                // predecessor's INPUT, which is the other prediction it mixes (j = k - 1, checked on the host)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (MIXT) pj_cur[k] = row_shr1(pin_cur[k]);
                    pin_cur[k] = row_shr1(pout[k]);
                }
            }
        }
        if (HIO && phase == 0) {
            if (!DEC && B.gate_flag) {
                // striped upload: the rest of the input may still be crossing PCIe.  Wait for the host's signal (an acquire
                // at system scope, so that nothing read afterwards is stale); bounded, so that a host that died cannot hang the GPU
                u32 tries = 0;
                while (__hip_atomic_load(B.gate_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
                    __builtin_amdgcn_s_sleep(64);
                    if (++tries > (1u << 22)) { status = ZPQ_E_INTERNAL; break; }
                }
            }
            if (DEC && B.prog_counter) {
                // early download: this block's first stripe is decoded (or the block has ended).  Store the byte still
                // pending, push this lane's stores and the L2's dirty lines out to memory, then count the block in
                if (pend_store) { dst[pend_pos] = (u8)pend_val; pend_store = false; }
                if (is_last && hyp == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                    __hip_atomic_fetch_add(B.prog_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (stop) break;
        }   // phase

        if (DEC && pend_store) dst[pend_pos] = (u8)pend_val;
        // (the last nibble's row is not written back: the slot is re-initialised for the next block)
        // ---- segment end: compress(-1) + flush (encoder.v:101-105,130-139)
        if (!DEC && is_last) {
            X.high = X.low;                                   // encode(1, 0): mid = low, high = mid
            while ((X.high ^ X.low) < 0x1000000u) {
                put_byte(X.high >> 24);
                X.low <<= 8; X.high = (X.high << 8) | 255u; if (X.low == 0) X.low = 1;
            }
            for (int sft = 24; sft >= 0; sft -= 8) put_byte(X.high >> sft);
            oq_flush();
        }
        i32 st0 = row_bcast(status, row_base);                 // VM status lives on lane 0
        for (int c = 1; c < n; c++) { const i32 sc = row_bcast(status, row_base + comp_lane(c)); st0 = st0 ? st0 : sc; }   // line-store overflow: any hashed lane
        if (is_last && hyp == 0) {
            i32 st = st0;
            if (X.opos > cap && st == ZPQ_OK) st = ZPQ_E_OVERFLOW;
            B.out_len[blk] = X.opos;
            B.status[blk] = st;
            if (DEC) {
                if (B.consumed) B.consumed[blk] = X.ipos;
                if (B.final_code) B.final_code[blk] = X.code;
                if (B.first_byte) B.first_byte[blk] = first;
            }
        }
    }
}

}  // namespace zpqc

// ------------------------------------------------------------------ host side
using zpqc::Cfg;

static bool build_cfg(const DModel *M, Cfg *cfg)
{
    if (!M->fast_kind || M->n < 1 || M->n > zpqc::G) return false;
    memset(cfg, 0, sizeof *cfg);
    cfg->n = M->n;
    {
        const char *ev = getenv("ZPQ_CHAIN_G");          // tuning knob
        const int want = ev ? atoi(ev) : ZPQ_CHAIN_G_DEFAULT;
        cfg->g = (want == 8 && M->n <= 8) ? 8 : 16;
    }
    cfg->dbg_ht_and = 0xFFFFFFFFu;
#ifdef ZPQ_DEBUG_KNOBS
    {
        // timing experiments only (make EXTRA=-DZPQ_DEBUG_KNOBS): an AND-mask on hash-table offsets keeps the tables
        // cache-resident -- the output is WRONG, which is why a normal build cannot reach this
        const char *ev = getenv("ZPQ_DEBUG_HT_AND");
        if (ev) cfg->dbg_ht_and = (uint32_t)strtoul(ev, nullptr, 0);
    }
#endif
    // Per-block LDS layout.  Tables are indexed by the bit-history state, and lanes of
    // different blocks / components very often hold EQUAL states, so tables whose bases share
    // a bank (all of them, if laid out at 1 KiB multiples) collide on every access -- rocprof
    // showed SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.67.  Component c's tables therefore
    // start at bank c, and the per-block stride advances the bank by n, so the n * (blocks per
    // wave) lanes of a wave land on distinct banks for equal states (exact for n*bpwave <= 32).
    int off = 0, i = 0;
    auto place = [&](int bytes, int bank) {
        while (((off >> 2) & 31) != (bank & 31)) off += 4;
        const int o = off;
        off += bytes;
        return o;
    };
    if (M->comp[0].type != ZT_ICM) return false;
    for (int c = 0; c < zpqc::G; c++) cfg->lds_off8[c] = 0xFFFF;
    cfg->lds_off32[0] = (uint16_t)place(1024, 0);
    cfg->lds_off8[0] = (uint16_t)place(256, 0);                // the ICM's stretch high bits
    for (i = 1; i < M->n && M->comp[i].type == ZT_ISSE; i++) {
        if (M->comp[i].b != i - 1) return false;               // chain: ISSE i is fed by component i-1
        cfg->lds_off32[i] = (uint16_t)place(1024, i);
        cfg->lds_off8[i] = (uint16_t)place(256, i);
    }
    cfg->nisse_end = i;
    if (i < M->n) {
        if (i != M->n - 1 || M->comp[i].type != ZT_MIX2) return false;
        if (M->comp[i].j >= i || M->comp[i].k >= i) return false;
        cfg->has_mix2 = 1;
        cfg->lds_off32[i] = 0;
        cfg->lds_mixw = place(32, i);
    }
    {
        const int nch = cfg->nisse_end;
        bool spec = cfg->has_mix2 ? (nch == 6 || nch == 8) : (nch == 2 || nch == 3 || nch == 5);
        bool any_sparse = false;
        for (int c = 0; c < M->n; c++) any_sparse = any_sparse || M->comp[c].sp_cap != 0;
        cfg->sparse = any_sparse ? 1 : 0;
        cfg->nch_spec = spec ? nch : 0;
        if (spec) cfg->g = (M->n <= 8) ? 8 : 16;              // specialised kernels exist for one G each
    }
    while (((off >> 2) & 31) != (M->n & 31)) off += 4;         // next block starts n banks further
    cfg->lds_per_block = off;
    // recognise the shipped HCOMP programs (levels.v:73-87,126-141,...)
    cfg->vm_kind = zpqc::VM_GENERIC;
    const uint8_t *p = M->header + M->hbegin;
    const int plen = M->hend - M->hbegin;
    {
        if (zpq_vm_hashchain(M) == M->n) cfg->vm_kind = zpqc::VM_HASHCHAIN;
        static const uint8_t l1[] = {96, 4, 28, 59, 10, 59, 112, 25, 10, 59, 10, 59, 112, 56};
        if (plen == (int)sizeof l1 && memcmp(p, l1, sizeof l1) == 0 && M->mlen == 4 && M->hlen == 2 && M->n == 2)
            cfg->vm_kind = zpqc::VM_LEVEL1;
    }
    // a specialised kernel evaluates exactly one program shape in registers and, with a MIX2, keeps its weights out
    // of the bit loop (needs mask 255 and >= 256 weights); anything else -> runtime-loop kernel
    const bool mix_ok = !cfg->has_mix2 || ((M->comp[M->n - 1].mask & 255) == 255 && M->comp[M->n - 1].c >= 256 &&
                                           M->comp[M->n - 1].k == M->n - 2 && M->comp[M->n - 1].j == M->n - 3);
    if (cfg->nch_spec && (!mix_ok || cfg->vm_kind != (cfg->nch_spec == 2 ? zpqc::VM_LEVEL1 : zpqc::VM_HASHCHAIN))) {
        cfg->nch_spec = 0;
        const char *ev = getenv("ZPQ_CHAIN_G");
        const int want = ev ? atoi(ev) : ZPQ_CHAIN_G_DEFAULT;
        cfg->g = (want == 8 && M->n <= 8) ? 8 : 16;
    }
    const int bpwave2 = 64 / cfg->g;   // blocks one wave carries
    // (the level-1 decoders stage the whole stretch table: LDS_DST_BYTES instead of the 8.5 KiB packing)
    const int shared_tables = cfg->nch_spec == 2 ? zpqc::LDS_STATE - zpqc::LDS_SQUASH + zpqc::LDS_DST_BYTES : zpqc::LDS_STATE;
    const int avail = 160 * 1024 - shared_tables - 1280 /*dummy tables*/ - 256;
    int bpw = avail / cfg->lds_per_block;
    if (bpw > 32) bpw = 32;                                           // 8 waves of 4 blocks or 4 waves of 8
    bpw = bpw / bpwave2 * bpwave2;
    if (bpw < bpwave2) return false;
    cfg->lds_dummy = bpw * cfg->lds_per_block;
    cfg->blocks_per_wg = bpw;
    return true;
}

bool zpq_chain_build_cfg(const DModel *M, zpqc::Cfg *cfg) { return build_cfg(M, cfg); }

extern "C" int zpq_chain_blocks_per_wg(const DModel *M)
{
    Cfg cfg;
    return build_cfg(M, &cfg) ? cfg.blocks_per_wg : 0;
}

// Blocks per workgroup for a batch: as few as keeps every CU busy (each wave then has a SIMD
// to itself), never more than the LDS allows; a multiple of the blocks one wave carries.
static int plan_blocks_per_wg(const Cfg &cfg, int nblocks, int cus)
{
    const int bpwave = 64 / cfg.g;
    int want = (nblocks + cus - 1) / cus;
    want = (want + bpwave - 1) / bpwave * bpwave;
    if (want < bpwave) want = bpwave;
    {
        const char *ev = getenv("ZPQ_CHAIN_BPW");         // tuning knob: at least this many blocks per workgroup
        const int floor_ = ev ? atoi(ev) / bpwave * bpwave : 0;
        if (floor_ > want) want = floor_;
    }
    return want < cfg.blocks_per_wg ? want : cfg.blocks_per_wg;
}

extern "C" int zpq_chain_plan(const DModel *M, int nblocks, int cus, int *blocks_per_wg)
{
    Cfg cfg;
    if (!build_cfg(M, &cfg)) return 0;
    *blocks_per_wg = plan_blocks_per_wg(cfg, nblocks, cus);
    return 1;
}

extern "C" int zpq_chain_max_wgs(const DModel *M, int cus)
{
    (void)M;
    return cus;   // one workgroup per CU (LDS-bound)
}

// The two-hypothesis decoder (dense short chains, levels 1-2) reads hash rows through "touched" bitmaps when the slot
// layout carries them (zpq_touch_layout); the wave-split decoder (opt-in) does not.
extern "C" int zpq_chain_touch_decode(const DModel *M)
{
    Cfg cfg;
    const char *ev = getenv("ZPQ_DEC_PIPE");
    if (ev && atoi(ev) != 0) return 0;
#ifndef ZPQ_TOUCH_DEC
    return 0;                                                // (measured slower: see `touch` in k_chain)
#endif
    return build_cfg(M, &cfg) && !cfg.sparse && !cfg.has_mix2 && (cfg.nch_spec == 2 || cfg.nch_spec == 3) && cfg.g == 8 ? 1 : 0;
}

// striped host transfers (HIO kernels) exist for the dense short chains: levels 1 and 2
extern "C" int zpq_chain_has_hio(const DModel *M)
{
    Cfg cfg;
    return build_cfg(M, &cfg) && !cfg.sparse && (cfg.nch_spec == 2 || cfg.nch_spec == 3) ? 1 : 0;
}

// zpq_pipe.hip: the wave-pipelined encoder of the chains without a MIX2
extern "C" int zpq_pipe_applies(const DModel *M, int blocks_per_wg, int nslots);
extern "C" int zpq_launch_pipe(const DBatch *B, const DModel *hostM, int nwg, int blocks_per_wg, hipStream_t stream, const char **name_out);

// zpq_dpipe.hip: the wave-split decoder of the dense chains without a MIX2
extern "C" int zpq_dpipe_applies(const DModel *M, int blocks_per_wg, int nslots);
extern "C" int zpq_launch_dpipe(const DBatch *B, const DModel *hostM, int nwg, int blocks_per_wg, hipStream_t stream);

extern "C" int zpq_launch_chain(const DBatch *B, const DModel *hostM, int decode, int nwg, int blocks_per_wg,
                                hipStream_t stream, const char **name_out)
{
    Cfg cfg;
    if (!build_cfg(hostM, &cfg)) return ZPQ_E_INTERNAL;
    const bool hio = B->gate_flag != nullptr || B->prog_counter != nullptr;
    if (hio && !zpq_chain_has_hio(hostM)) return ZPQ_E_INTERNAL;
    // The DECODER of the chain of five (level 3) gives a block sixteen lanes, two copies of every component (HYP16 in k_chain);
    // ZPQ_DEC_HYP16=0 keeps the eight-lane decoder (tests and A/B runs compare the two).
    bool hyp16 = false;
    if (decode && ((cfg.nch_spec == 5 && !cfg.has_mix2) || (cfg.nch_spec == 6 && cfg.has_mix2)) && cfg.g == 8 && blocks_per_wg % 4 == 0) {
        const char *ev = getenv("ZPQ_DEC_HYP16");
        hyp16 = !(ev && atoi(ev) == 0);
    }
    if (hyp16) cfg.g = 16;
    if (blocks_per_wg < 64 / cfg.g || blocks_per_wg > cfg.blocks_per_wg || blocks_per_wg % (64 / cfg.g)) return ZPQ_E_INTERNAL;
    if (name_out) *name_out = decode ? "k_chain<decode>" : "k_chain<encode>";
    if (!decode && zpq_pipe_applies(hostM, blocks_per_wg, B->nslots)) {
        return zpq_launch_pipe(B, hostM, nwg, blocks_per_wg, stream, name_out);   // "k_pipe<encode>" or "k_pipe2<encode>" (split stages)
    }
    if (decode && !B->prog_counter && zpq_dpipe_applies(hostM, blocks_per_wg, B->nslots)) {
        if (name_out) *name_out = "k_dpipe<decode>";
        return zpq_launch_dpipe(B, hostM, nwg, blocks_per_wg, stream);
    }
    cfg.blocks_per_wg = blocks_per_wg;
    cfg.lds_dummy = blocks_per_wg * cfg.lds_per_block;
    const int threads = cfg.blocks_per_wg / (64 / cfg.g) * 64;
#ifdef ZPQ_NO_DST
    const bool dst = false;
#else
    const bool dst = decode && cfg.nch_spec == 2;
#endif
    const size_t lds = (size_t)zpqc::LDS_STATE + (dst ? (size_t)(zpqc::LDS_DST_BYTES - zpqc::LDS_SQUASH) : 0) +
                       (size_t)cfg.blocks_per_wg * cfg.lds_per_block + 1280;
    if (lds > 160 * 1024) return ZPQ_E_INTERNAL;
    // encode uses the pipelined bit step, decode the plain one (measured, see above)
#define ZPQ_LAUNCH(D, N, MX, GGv, SPv)                                                                   \
    do {                                                                                                 \
        constexpr bool S_ = (D) ? (ZPQ_CHAIN_SPEC_DEC != 0) : (ZPQ_CHAIN_SPEC_ENC != 0);                 \
        (void)hipFuncSetAttribute((const void *)zpqc::k_chain<D, S_, N, MX, GGv, SPv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((zpqc::k_chain<D, S_, N, MX, GGv, SPv>), dim3(nwg), dim3(threads), lds, stream, *B, cfg); \
    } while (0)
#define ZPQ_LAUNCH_HIO(D, N)                                                                             \
    do {                                                                                                 \
        constexpr bool S_ = (D) ? (ZPQ_CHAIN_SPEC_DEC != 0) : (ZPQ_CHAIN_SPEC_ENC != 0);                 \
        (void)hipFuncSetAttribute((const void *)zpqc::k_chain<D, S_, N, false, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((zpqc::k_chain<D, S_, N, false, 8, false, true>), dim3(nwg), dim3(threads), lds, stream, *B, cfg); \
    } while (0)
#define ZPQ_LAUNCH_SP(D, N, GGv)                                                                         \
    do {                                                                                                 \
        if (cfg.sparse) ZPQ_LAUNCH(D, N, false, GGv, true);                                              \
        else if (hio && ((N) == 2 || (N) == 3)) ZPQ_LAUNCH_HIO(D, ((N) == 2 ? 2 : 3));                   \
        else ZPQ_LAUNCH(D, N, false, GGv, false);                                                        \
    } while (0)
#define ZPQ_DISPATCH(D)                                                                                  \
    do {                                                                                                 \
        switch (cfg.nch_spec) {                                                                          \
        case 2: ZPQ_LAUNCH_SP(D, 2, 8); break;          /* level 1 */                                    \
        case 3: ZPQ_LAUNCH_SP(D, 3, 8); break;          /* level 2 */                                    \
        case 5: if ((D) && hyp16) ZPQ_LAUNCH_SP(D, 5, ((D) ? 16 : 8)); else ZPQ_LAUNCH_SP(D, 5, 8); break;   /* level 3 */ \
        case 6: if ((D) && hyp16) ZPQ_LAUNCH(D, 6, true, ((D) ? 16 : 8), true); else ZPQ_LAUNCH(D, 6, true, 8, true); break;   /* level 4 */ \
        case 8: ZPQ_LAUNCH(D, 8, true, 16, true); break;      /* level 5 */                              \
        default:                                                                                         \
            if (cfg.g == 8) ZPQ_LAUNCH(D, 0, false, 8, true); else ZPQ_LAUNCH(D, 0, false, 16, true);    \
            break;                                                                                       \
        }                                                                                                \
    } while (0)
    if (decode) ZPQ_DISPATCH(true); else ZPQ_DISPATCH(false);
#undef ZPQ_DISPATCH
#undef ZPQ_LAUNCH_SP
#undef ZPQ_LAUNCH_HIO
#undef ZPQ_LAUNCH
    return ZPQ_OK;
}
