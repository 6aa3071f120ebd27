// zpq_chain_cfg.h -- what the kernels for "chain" models (zpq_chain.hip: lane = component; zpq_pipe.hip: wave =
// component) share: the LDS layout of the read-only tables and of a block's counters / weights, and the launch
// configuration the host derives from a model (zpq_chain_build_cfg in zpq_chain.hip).
#pragma once
#include <stdint.h>

#include "zpq_common.h"

namespace zpqc {

typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int G = 16;          // lanes per ZPAQ block (= one DPP row)
// (blocks per wave = 64 / lanes-per-block, fixed per kernel instantiation)
constexpr int MAXW = 8;        // waves per workgroup upper bound

// LDS layout of the shared read-only tables
constexpr int LDS_STRETCH = 0;                   // u32[2048+128]
constexpr int LDS_SQUASH = (2048 + 128) * 4;     // u16[4096]
constexpr int LDS_NS = LDS_SQUASH + 4096 * 2;    // u8[1024]
constexpr int LDS_STATE = LDS_NS + 1024;         // per-block state follows (16-B aligned)
constexpr int LDS_DST_BYTES = 32768 * 2;         // the whole stretch table as i16 (level 1's decoders, in place of the packing)

// HCOMP program shapes the kernel evaluates in registers instead of interpreting
enum { VM_GENERIC = 0, VM_HASHCHAIN = 1, VM_LEVEL1 = 2 };

struct Cfg {
    int32_t n;                 // components
    int32_t nisse_end;         // components 1..nisse_end-1 are ISSE (chain length incl. ICM)
    int32_t has_mix2;          // last component is MIX2
    int32_t blocks_per_wg;
    int32_t lds_per_block;     // bytes
    int32_t vm_kind;
    int32_t g;                 // lanes per block chosen on the host (8 or 16)
    int32_t nch_spec;          // compile-time specialisation picked on the host: chain length (0 = runtime path)
    int32_t sparse;            // some component uses a compact line store
    int64_t split_enc;         // zpq_pipe.hip, k_pipe2: 0, or wave -> role (4 bits each) | split components << 32 | waves << 40 | 1 << 48
    uint32_t dbg_ht_and;       // timing experiments only: AND-mask on hash-table offsets (0xFFFFFFFF = off)
    int32_t lds_dummy;         // byte offset (from LDS_STATE) of the per-workgroup dummy tables idle lanes use
    int32_t lds_mixw;          // byte offset inside the block's LDS state of u16[16]: the nibble's candidate MIX2 weights (decode)
    uint16_t lds_off32[G];     // component c's u32 table inside the block's LDS state (cm | w0 + w1 low bits)
    uint16_t lds_off8[G];      // ISSE c's u8 table (w1 bits 12..19); 0xFFFF = none
};

}  // namespace zpqc

// false = the model is not a chain the kernels handle (-> lanes / generic kernel)
bool zpq_chain_build_cfg(const DModel *M, zpqc::Cfg *cfg);
