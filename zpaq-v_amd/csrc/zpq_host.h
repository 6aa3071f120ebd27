// zpq_host.h -- host-only internals of libzpaq_hip.so.
#pragma once
#include <stdint.h>

#include <atomic>
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
    mutable std::atomic<int> refs{1};   // the creator's handle + one per zpq_block built on it: zpq_model_destroy only drops the creator's
};
void zpq_model_retain(const zpq_model *m);
void zpq_model_release(const zpq_model *m);   // deletes the model with its last reference

// ---- the C boundary never lets a C++ exception out (include/zpaq_hip.h: "never aborts, no exceptions").  Every
// extern "C" definition is a function-try-block:   extern "C" int f(...) try { ... } ZPQ_CATCH(return ZPQ_E_INTERNAL)
// std::bad_alloc and std::system_error can come from the containers and threads of the host code; the HIP runtime itself
// has been seen to throw (std::bad_variant_access out of a call on a stream whose owner was gone).
void zpq_note_exception(const char *where) noexcept;
#define ZPQ_CATCH(on_error) catch (...) { zpq_note_exception(__func__); on_error; }

// the same for one-line entry points:  return zpq_guard<int>(0, __func__, [&] { ... });
template <class R, class F> static inline R zpq_guard(R on_error, const char *where, F &&f) noexcept
{
    try { return f(); } catch (...) { zpq_note_exception(where); return on_error; }
}
template <class F> static inline void zpq_guard_v(const char *where, F &&f) noexcept
{
    try { f(); } catch (...) { zpq_note_exception(where); }
}
