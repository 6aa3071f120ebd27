// zpq_common.h -- structures shared by the host side of libzpaq_hip.so and its
// gfx950 kernels.  A zpq_model is the device-independent result of walking a
// COMP/HCOMP header the way Predictor.init + ZPAQL.inith/initp do
// (reference zpaq/predictor.v:292-470, zpaq/zpaql.v:74-95): which tables a block
// owns, where they sit inside the block's HBM state slot, and how they start out.
#pragma once
#include <stdint.h>

#define ZPQ_MAX_HDR 4096  // header bytes kept on the device (hsize is 16-bit in the format)
#define ZPQ_MAX_COMP 255

// Component type codes (reference zpaq/types.v:6-17).
enum { ZT_NONE = 0, ZT_CONST = 1, ZT_CM = 2, ZT_ICM = 3, ZT_MATCH = 4, ZT_AVG = 5,
       ZT_MIX2 = 6, ZT_MIX = 7, ZT_ISSE = 8, ZT_SSE = 9 };

// How a u32 table starts out (predictor.v:343-464).
enum { ZF_ZERO = 0, ZF_CONST = 1, ZF_PATTERN = 2 };

struct DComp {
    int32_t type;
    int32_t a, b, c, limit;      // Component fields as Predictor.init leaves them
    int32_t j, k, rate, mask;    // MIX2 cm[0..3] / MIX ht[0..1] / SSE start (in `rate`)
    uint32_t cm_len, ht_len, a16_len;
    uint32_t cm_fill, cm_fill_val, cm_pat_len;  // ZF_*, constant or image offset (u32 words), pattern words
    uint32_t a16_fill;           // initial u16 value of every a16 entry
    uint32_t sp_cap;             // 0 = dense hash table; else capacity (lines, a multiple of 4) of the compact line store
    uint64_t cm_off, ht_off, a16_off;  // byte offsets inside the state slot
    uint64_t tb_off;             // 0, or: "touched" bitmap of this dense hash table (one bit per 16-byte row, inside the zeroed part of
                                 // the slot); the table itself then lies BEHIND the zeroed part and is never cleared -- a row whose
                                 // bit is clear reads as zeros (zpq_touch_layout, zpq_chain.hip's two-hypothesis decoder)
    uint64_t sp_tag_off, sp_line_off;  // compact line store: u32 tags[cap] (dense line index + 1, 0 = free) inside the zeroed
                                       // part of the slot, 64-B lines[cap] behind it (a line is zeroed when it is claimed)
};

// Mutable per-component scalars that outlive a segment (MATCH: a=len b=offset
// c=predicted bit cxt=bit position limit=buffer position; predictor.v:371-380,710-741).
struct DCompScal {
    int32_t a, b, c, limit;
    uint32_t cxt;
    uint32_t pad_[3];
};

// ZPAQL registers that persist across bytes and segments (zpaql.v:6-31).
struct DVmRegs {
    uint32_t a, b, c, d;
    int32_t f, pc;
    uint32_t pad_[2];
};

struct DModel {
    int32_t n;                   // components (header[4]); 0 = no model
    int32_t hdr_len, cend, hbegin, hend;
    uint32_t mlen, hlen;         // M bytes, H words (0 = array absent: reads 0, writes ignored)
    uint32_t fast_kind;          // 0 = generic only, 1 = ICM/ISSE/MIX2 chain kernel applies
    uint64_t regs_off, r_off, scal_off, h_off, m_off;  // inside the slot
    uint64_t zero_bytes;         // leading part of the slot that must be zeroed (== slot_bytes here)
    uint64_t slot_bytes;         // one block's whole state, 256-B aligned
    uint32_t img_words;          // init image length (u32)
    uint32_t pad_;
    uint8_t header[ZPQ_MAX_HDR];
    DComp comp[ZPQ_MAX_COMP];
};

// Kernel launch parameters for one batch.
struct DBatch {
    const DModel *model;
    const uint32_t *img;         // init image (ICM cminit / ISSE weights / SSE ramps)
    uint8_t *slots;              // nslots * slot_bytes
    int32_t nslots, nblocks;
    uint32_t flags;              // ZPQ_FLAG_* | ZB_*
    uint32_t ntrace;
    const uint8_t *in;
    const uint64_t *in_off;
    uint8_t *out;
    const uint64_t *out_off;
    uint32_t *out_len;
    uint32_t *consumed;          // decode only (may be null)
    uint32_t *final_code;        // decode only (may be null)
    uint32_t *first_byte;        // decode only (may be null)
    int32_t *status;
    int32_t *trace;              // predict() per modelled bit of block 0 (may be null)
    uint32_t *ctx_out;           // debug: H[0..n) per byte of block 0 (may be null)
    // read-only tables in HBM (L2-resident)
    const int16_t *squash;       // [4096]   predictor.v:21-49 (entry 4095 unused)
    const int16_t *stretch;      // [32768]  predictor.v:73-96
    const uint32_t *dt;          // [1024]   predictor.v:111-166
    const int16_t *dt2k;         // [256]    predictor.v:99-106
    const uint8_t *ns;           // [1024]   statetable.v:15-57
    const uint32_t *stretch_c;   // [2048+64] compact stretch: (base<<16)|step bitmap, then exact ends
    // Encode, striped upload (host_pipeline): bytes of a block's input at offset >= gate_pos may still be on their way
    // over PCIe when the kernel starts; a lane reads them only once *gate_flag (pinned host memory) is non-zero.
    const uint32_t *gate_flag;   // null = the whole input is resident
    uint32_t gate_pos;
    // Decode, early download (host_pipeline): a block that has STORED its first prog_pos output bytes (or has ended)
    // makes them visible to the copy engines (release at system scope) and adds one to *prog_counter (pinned host
    // memory); when the count reaches the number of blocks the host copies that stripe out beside the running kernel.
    uint32_t prog_pos;
    uint32_t *prog_counter;      // null = no progress reports
};

#define ZB_KEEP_STATE 0x100u     // do not re-initialise the slot (later segments of one block)
#define ZB_CTX_ONLY 0x200u       // debug: run only the ZPAQL VM and dump contexts
