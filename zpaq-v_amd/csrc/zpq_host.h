// zpq_host.h -- host-only internals of libzpaq_hip.so.
#pragma once
#include <stdint.h>

#include <vector>

#include "zpq_common.h"

namespace zpq {

// Start-up tables exactly as the reference builds them (predictor.v:7-18).
struct Tables {
    int32_t squash[4096];
    int32_t stretch[32768];
    int32_t dt[1024];
    int32_t dt2k[256];
    uint8_t ns[1024];
    uint32_t stretch_c[2048 + 128];  // LDS-sized packing of stretch (see zpq_model.cpp)
};
const Tables &tables(int *status);
uint64_t tables_fnv(int which);

}  // namespace zpq

// Re-lay a model's state slot with a compact line store (capacity `cap` lines, a multiple of 4) for every
// ICM/ISSE hash table that is larger than the store; false if none qualifies.  The zeroed prefix of the slot
// (zero_bytes) then ends before the stores' line arrays: a line is cleared when it is claimed.
bool zpq_sparse_layout(const DModel &dense, uint32_t cap, DModel *out);

// Re-lay a model's state slot so that every ICM/ISSE hash table of at least 64 KiB needs no clearing: the table moves behind
// the zeroed part of the slot and gets a "touched" bitmap (one bit per 16-byte row = 1/128 of the table) inside it.  A
// level-2 block then clears 0.4 MiB instead of 12 MiB.  false if no table qualifies.
bool zpq_touch_layout(const DModel &dense, DModel *out);

// The HCOMP shape every shipped level >= 2 uses (levels.v:126-141 and on):
//   b=c c-- *c=a d=0 (hash *d=a d++) x K  hash *d=a halt
// leaves H[k] = hash^(k+1)(byte, previous byte) for k <= K and never touches H beyond.  Returns the
// number of hashes K+1 when the model's program is exactly that (and M, H are large enough), else 0.
int zpq_vm_hashchain(const DModel *M);

struct zpq_model {
    DModel d;
    std::vector<uint32_t> img;  // initial table contents (ICM/ISSE/SSE), uploaded per ctx
    uint64_t id;                // unique, for the per-ctx device cache
};
