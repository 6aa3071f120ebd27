// zpq_api.hip -- the GPU half of the C ABI in include/zpaq_hip.h: per-device
// context, state-slot management, kernel selection and launch.
//
// Data layout in HBM (DESIGN.md has the full picture):
//   * read-only tables (squash/stretch/dt/dt2k/ns, ~80 KiB) uploaded once per ctx;
//   * one DModel + init image per (ctx, model);
//   * a pool of per-block STATE SLOTS, each DModel::slot_bytes (12.07 MiB at level 2):
//     [VM regs | R[256] | MATCH scalars | H | M | per-component cm/ht/a16 tables].
//     A slot is owned by one resident workgroup (generic kernel) or one lane group
//     (chain kernel) and re-initialised in-kernel between blocks;
//   * caller-provided in/out slabs addressed by in_off/out_off.
// There is no CPU fallback anywhere in this file: if HIP is unusable every
// compute entry point returns ZPQ_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_host.h"

extern "C" void zpq_launch_generic(const DBatch *B, const DModel *hostM, int decode, int grid, hipStream_t stream);
extern "C" int zpq_generic_blocks_per_cu(const DModel *M);
extern "C" int zpq_lanes_supported(const DModel *M);
extern "C" int zpq_lanes_blocks_per_cu(const DModel *M);
extern "C" int zpq_launch_lanes(const DBatch *B, const DModel *hostM, int decode, int nslots, hipStream_t stream);
extern "C" const char *zpq_lanes_kernel_name(const DModel *M, int decode);   // k_rows (four blocks per wave) or k_lanes
// zpq_gpipe.hip: the ENCODER of general models as a pipeline of waves (wave = component, lane = block)
extern "C" int zpq_gpipe_applies(const DModel *M);
extern "C" int zpq_gpipe_blocks_per_cu(const DModel *M);
extern "C" int zpq_launch_gpipe(const DBatch *B, const DModel *hostM, int nslots, hipStream_t stream);
extern "C" int zpq_gdec_applies(const DModel *M);               // ... and their DECODER, bit-synchronous, a barrier per level of the prediction chain
extern "C" int zpq_gdec_blocks_per_cu(const DModel *M);
extern "C" int zpq_launch_gdec(const DBatch *B, const DModel *hostM, int nslots, hipStream_t stream);
extern "C" int zpq_chain_blocks_per_wg(const DModel *M);   // 0 = model not supported by the chain kernel
extern "C" int zpq_chain_max_wgs(const DModel *M, int cus);
extern "C" int zpq_chain_plan(const DModel *M, int nblocks, int cus, int *blocks_per_wg);
extern "C" int zpq_launch_chain(const DBatch *B, const DModel *hostM, int decode, int nwg, int blocks_per_wg,
                                hipStream_t stream, const char **name_out);
extern "C" int zpq_chain_has_hio(const DModel *M);   // kernels that take part in striped host transfers exist for this model
extern "C" int zpq_launch_sha1(const uint8_t *in, const uint64_t *beg, const uint64_t *end, int n, uint8_t *out20, hipStream_t stream);

#define HIPCK(x)                                                              \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "[zpaq_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? ZPQ_E_NOMEM : ZPQ_E_NODEVICE; \
        }                                                                     \
    } while (0)

struct DevModel {
    DModel *d_model = nullptr;
    uint32_t *d_img = nullptr;
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return ZPQ_OK;
        // Grow: take the new buffer first so that a failed regrow keeps the old one usable.  Pools of ~100 GiB
        // cannot coexist with their successor; only then is the old one given up before the second attempt.
        size_t want = n + n / 8 + 4096;
        void *q = nullptr;
        if (hipMalloc(&q, want) != hipSuccess) {
            (void)hipGetLastError();
            q = nullptr;
            if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
            if (hipMalloc(&q, want) != hipSuccess) {
                (void)hipGetLastError();
                want = n + 4096;
                if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); return ZPQ_E_NOMEM; }
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return ZPQ_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct zpq_ctx {
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    int16_t *d_squash = nullptr, *d_stretch = nullptr, *d_dt2k = nullptr;
    uint32_t *d_dt = nullptr, *d_stretch_c = nullptr;
    uint8_t *d_ns = nullptr;
    std::map<uint64_t, DevModel> models;   // key = model id << 32 | compact-store capacity (0 = dense layout)
    uint32_t sparse_cap = 0;               // line-store capacity (lines) for the largest block the caller submits
    uint32_t sparse_pct = 112;             // capacity = this percentage of the lines a largest block can touch (125: measured 4-7 % slower at levels 4-5, fewer blocks fit)
    DevBuf slots;
    uint64_t budget = 0;
    int last_slots = 0;
    uint32_t last_sp = 0;                  // line-store capacity the last launch ran with (0 = dense)
    const char *last_name = "";
    // staging for the host-pointer entry points
    DevBuf s_in, s_out, s_inoff, s_outoff, s_u32[4], s_status, s_misc;
    // pipelined host batches (host_pipeline): two staging sets, a copy-in and a copy-out stream beside the compute stream
    struct PipeSet {
        DevBuf in, out, inoff, outoff, dstoff, meta, status;
        hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
        // pinned host side of the small arrays: a copy to or from PAGEABLE memory blocks the calling thread until it
        // has run, i.e. until the round's kernel is done -- which would keep the next round's upload from being queued
        uint64_t *h_off = nullptr;         // [3][n + 1]: relative in / out offsets, absolute destination offsets
        uint32_t *h_meta = nullptr;        // [4][n] out_len, consumed, final_code, first_byte  + [n] status
        size_t h_cap = 0;                  // blocks the pinned arrays are sized for
        int b0 = 0, n = 0;                 // the round whose results sit in h_meta (not yet handed to the caller)
        int ensure_host(size_t nblk)
        {
            if (nblk <= h_cap) return ZPQ_OK;
            if (h_off) (void)hipHostFree(h_off);
            if (h_meta) (void)hipHostFree(h_meta);
            h_off = nullptr; h_meta = nullptr; h_cap = 0;
            const size_t want = nblk + nblk / 8 + 16;
            if (hipHostMalloc((void **)&h_off, 3 * (want + 1) * 8, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&h_meta, 5 * want * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return ZPQ_E_NOMEM; }
            h_cap = want;
            return ZPQ_OK;
        }
    } pipe[2];
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    uint32_t *h_gate = nullptr;            // pinned, device-visible: "the rest of the striped upload has arrived"
    std::mutex mu;
    std::vector<zpq_block *> children;     // blocks created on this ctx and not yet destroyed (guarded by g_reg_mu)
};

// ------------------------------------------------------------------ handle lifetime
// A front end may drop its handles in any order (a garbage-collected host language does): a zpq_block may outlive its
// zpq_ctx and its zpq_model.  The ctx keeps the list of its blocks and, when destroyed, releases their device state and
// ORPHANS them (b->ctx = nullptr: every later call on the block returns ZPQ_E_CLOSED, zpq_block_destroy only frees the
// host struct); a block holds a reference on its model.  A zpq_ctx* is checked against the set of living contexts at
// every entry point, so that a call on a destroyed ctx is an error code, not a read of freed memory.
static std::mutex g_reg_mu;
static std::unordered_set<const zpq_ctx *> &live_set()
{
    static std::unordered_set<const zpq_ctx *> *s = new std::unordered_set<const zpq_ctx *>();   // never destroyed: handles may be dropped from atexit handlers
    return *s;
}
static bool ctx_live(const zpq_ctx *c)
{
    if (!c) return false;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return live_set().count(c) != 0;
}

void zpq_note_exception(const char *where) noexcept
{
    try {
        try { throw; }
        catch (const std::exception &e) { fprintf(stderr, "[zpaq_hip] %s: C++ exception stopped at the C boundary: %s\n", where, e.what()); }
        catch (...) { fprintf(stderr, "[zpaq_hip] %s: unknown C++ exception stopped at the C boundary\n", where); }
    } catch (...) {
    }
}

struct zpq_block {
    zpq_ctx *ctx;              // nullptr once the ctx has been destroyed (orphaned block)
    const zpq_model *model;    // the block holds a reference (zpq_model_retain)
    uint8_t *slot;
    bool fresh;
    // A block's FIRST segment runs on the fast batch kernels (their state lives in LDS / the shared pool and is
    // gone afterwards).  Most blocks have exactly one segment; if a second one arrives, the persistent state in
    // `slot` is materialised first by replaying segment 1's symbols through the generic kernel (the model state
    // depends only on the symbol sequence, not on which direction coded it).
    bool lazy = false;
    std::vector<uint8_t> lazy_in;
    uint32_t lazy_flags = 0;
};

// ------------------------------------------------------------------ ctx
static int ctx_init(zpq_ctx *c, const zpq::Tables &T);
static int set_max_block_bytes(zpq_ctx *c, uint64_t bytes);
extern "C" int zpq_ctx_create(int device, zpq_ctx **out) try
{
    if (!out) return ZPQ_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZPQ_E_NODEVICE;
    if (device < 0 || device >= ndev) return ZPQ_E_ARG;
    int tst = ZPQ_OK;
    const zpq::Tables &T = zpq::tables(&tst);
    if (tst != ZPQ_OK) return tst;
    HIPCK(hipSetDevice(device));
    zpq_ctx *c = new (std::nothrow) zpq_ctx();
    if (!c) return ZPQ_E_NOMEM;
    c->device = device;
    const int rc = ctx_init(c, T);
    { std::lock_guard<std::mutex> lk(g_reg_mu); live_set().insert(c); }
    if (rc != ZPQ_OK) { zpq_ctx_destroy(c); return rc; }   // releases whatever the failed step left behind
    *out = c;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

static int ctx_init(zpq_ctx *c, const zpq::Tables &T)
{
    const int device = c->device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cus = prop.multiProcessorCount;
    HIPCK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCK(hipEventCreate(&c->ev0));
    HIPCK(hipEventCreate(&c->ev1));
    HIPCK(hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
    HIPCK(hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
    HIPCK(hipHostMalloc((void **)&c->h_gate, 64, hipHostMallocPortable | hipHostMallocMapped));
    *c->h_gate = 1;
    for (auto &ps : c->pipe) {
        HIPCK(hipEventCreateWithFlags(&ps.ev_in, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&ps.ev_k, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&ps.ev_out, hipEventDisableTiming));
    }
    // device tables: squash/stretch/dt2k narrowed to i16 (all values fit)
    std::vector<int16_t> sq(4096), st(32768), d2(256);
    for (int i = 0; i < 4096; i++) sq[i] = (int16_t)T.squash[i];
    for (int i = 0; i < 32768; i++) st[i] = (int16_t)T.stretch[i];
    for (int i = 0; i < 256; i++) d2[i] = (int16_t)T.dt2k[i];
    HIPCK(hipMalloc((void **)&c->d_squash, 4096 * 2));
    HIPCK(hipMalloc((void **)&c->d_stretch, 32768 * 2));
    HIPCK(hipMalloc((void **)&c->d_dt2k, 256 * 2));
    HIPCK(hipMalloc((void **)&c->d_dt, 1024 * 4));
    HIPCK(hipMalloc((void **)&c->d_ns, 1024));
    HIPCK(hipMalloc((void **)&c->d_stretch_c, sizeof T.stretch_c));
    HIPCK(hipMemcpy(c->d_squash, sq.data(), 4096 * 2, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_stretch, st.data(), 32768 * 2, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_dt2k, d2.data(), 256 * 2, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_dt, T.dt, 1024 * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_ns, T.ns, 1024, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->d_stretch_c, T.stretch_c, sizeof T.stretch_c, hipMemcpyHostToDevice));
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) c->budget = (uint64_t)fr / 100 * 85;
    else c->budget = 64ull << 30;
    {
        const char *ev = getenv("ZPQ_SPARSE_PCT");             // tuning knob: store capacity as % of the touched-line bound
        const int pct = ev ? atoi(ev) : 0;
        if (pct >= 101 && pct <= 400) c->sparse_pct = (uint32_t)pct;
    }
    set_max_block_bytes(c, 65536);
    return ZPQ_OK;
}

extern "C" void zpq_ctx_destroy(zpq_ctx *c) try
{
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!c || !live_set().erase(c)) return;            // null, destroyed already, or never a ctx of this library
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    {
        // blocks that outlive their ctx: their device state goes with it, the host structs stay for zpq_block_destroy
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (zpq_block *b : c->children) {
            if (b->slot) (void)hipFree(b->slot);
            b->slot = nullptr;
            b->ctx = nullptr;
        }
        c->children.clear();
    }
    for (auto &kv : c->models) { (void)hipFree(kv.second.d_model); (void)hipFree(kv.second.d_img); }
    c->slots.release();
    c->s_in.release(); c->s_out.release(); c->s_inoff.release(); c->s_outoff.release();
    for (auto &b : c->s_u32) b.release();
    c->s_status.release(); c->s_misc.release();
    for (auto &ps : c->pipe) {
        ps.in.release(); ps.out.release(); ps.inoff.release(); ps.outoff.release(); ps.dstoff.release(); ps.meta.release(); ps.status.release();
        if (ps.h_off) (void)hipHostFree(ps.h_off);
        if (ps.h_meta) (void)hipHostFree(ps.h_meta);
        if (ps.ev_in) (void)hipEventDestroy(ps.ev_in);
        if (ps.ev_k) (void)hipEventDestroy(ps.ev_k);
        if (ps.ev_out) (void)hipEventDestroy(ps.ev_out);
    }
    if (c->h_gate) (void)hipHostFree(c->h_gate);
    if (c->s_h2d) { (void)hipStreamSynchronize(c->s_h2d); (void)hipStreamDestroy(c->s_h2d); }
    if (c->s_d2h) { (void)hipStreamSynchronize(c->s_d2h); (void)hipStreamDestroy(c->s_d2h); }
    (void)hipFree(c->d_squash); (void)hipFree(c->d_stretch); (void)hipFree(c->d_dt2k);
    (void)hipFree(c->d_dt); (void)hipFree(c->d_ns); (void)hipFree(c->d_stretch_c);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
} ZPQ_CATCH(return)

extern "C" int zpq_ctx_sync(zpq_ctx *c) try
{
    if (!ctx_live(c)) return ZPQ_E_ARG;
    HIPCK(hipStreamSynchronize(c->stream));
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)
extern "C" int zpq_ctx_device(const zpq_ctx *c) { return ctx_live(c) ? c->device : -1; }
extern "C" void *zpq_ctx_stream(zpq_ctx *c) { return ctx_live(c) ? (void *)c->stream : nullptr; }
extern "C" int zpq_ctx_set_state_budget(zpq_ctx *c, uint64_t bytes) try
{
    if (!ctx_live(c)) return ZPQ_E_ARG;
    c->budget = bytes;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)
extern "C" int zpq_ctx_set_max_block_bytes(zpq_ctx *c, uint64_t bytes) try
{
    if (!ctx_live(c)) return ZPQ_E_ARG;
    return set_max_block_bytes(c, bytes);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)
static int set_max_block_bytes(zpq_ctx *c, uint64_t bytes)
{
    // A block of N bytes (+ PP byte) probes each hash table 2(N+1) times (predictor.v:558-560: once per nibble),
    // so it touches at most that many 64-byte lines.  The store holds sparse_pct % of that bound: probing is
    // linear over 4-slot groups, a worst-case block (every probe a new line) ends at 89 % load.
    const uint64_t probes = 2 * (bytes + 2);
    uint64_t cap = (probes * c->sparse_pct + 99) / 100 + 16;
    cap = (cap + 3) & ~3ull;
    if (cap > (1ull << 25)) cap = 1ull << 25;               // line offsets are 32-bit in the kernel
    c->sparse_cap = (uint32_t)cap;
    return ZPQ_OK;
}
extern "C" int zpq_ctx_last_slots(const zpq_ctx *c) { return ctx_live(c) ? c->last_slots : 0; }
extern "C" unsigned zpq_ctx_last_line_store(const zpq_ctx *c) { return ctx_live(c) ? c->last_sp : 0u; }
extern "C" const char *zpq_ctx_last_kernel_name(const zpq_ctx *c) { return ctx_live(c) ? c->last_name : ""; }
extern "C" float zpq_ctx_last_kernel_ms(const zpq_ctx *c) try
{
    if (!ctx_live(c) || !c->ev_valid) return -1.f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
    return ms;
} ZPQ_CATCH(return -1.f)

static int get_dev_model(zpq_ctx *c, const zpq_model *m, const DModel &layout, uint32_t sparse_cap, DevModel *out)
{
    const uint64_t key = (m->id << 32) | (uint64_t)sparse_cap;
    auto it = c->models.find(key);
    if (it != c->models.end()) { *out = it->second; return ZPQ_OK; }
    DevModel dm;
    HIPCK(hipMalloc((void **)&dm.d_model, sizeof(DModel)));
    HIPCK(hipMalloc((void **)&dm.d_img, m->img.size() * 4 + 4));
    HIPCK(hipMemcpy(dm.d_model, &layout, sizeof(DModel), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dm.d_img, m->img.data(), m->img.size() * 4, hipMemcpyHostToDevice));
    c->models[key] = dm;
    *out = dm;
    return ZPQ_OK;
}

// ------------------------------------------------------------------ batch core (device pointers)
struct BatchArgs {
    int nblocks;
    const uint8_t *in; const uint64_t *in_off;
    uint32_t flags;
    uint8_t *out; const uint64_t *out_off;
    uint32_t *out_len, *consumed, *final_code, *first_byte;
    int32_t *status;
    int32_t *trace; uint32_t ntrace;
    uint32_t *ctx_out;
    uint8_t *own_slot;   // zpq_block: use this slot instead of the pool
    const uint32_t *gate_flag;   // striped upload (chain encode only): see DBatch
    uint32_t gate_pos;
    uint32_t *prog_counter;      // early download (chain decode only): see DBatch
    uint32_t prog_pos;
};

// What a batch call will launch: kernel family, slot layout, resident slots and grid.
struct Plan {
    bool chain = false, lanes = false;
    bool gpipe = false;          // lanes family, encode: the wave-per-component pipeline (zpq_gpipe.hip)
    bool gdec = false;           // lanes family, decode: the wave-per-component decoder (zpq_gpipe.hip, k_gdec)
    bool touch = false;          // dense tables that are not cleared: "touched" bitmaps (zpq_touch_layout)
    uint32_t sp = 0;             // compact line store capacity (lines), 0 = dense tables
    const DModel *M = nullptr;   // layout the kernels see (dense or compact)
    int nslots = 0, grid = 0, bpw = 0;
};

// resident slots / grid of the chain kernel for one slot layout
static int plan_chain(zpq_ctx *c, const DModel &M, int nblocks, int *nslots_out, int *grid_out, int *bpw_out)
{
    const uint64_t max_by_mem = M.slot_bytes ? (c->budget / M.slot_bytes) : (uint64_t)nblocks;
    if (max_by_mem == 0) return ZPQ_E_NOMEM;
    int bpw = 0;
    const uint64_t can_hold = max_by_mem < (uint64_t)nblocks ? max_by_mem : (uint64_t)nblocks;
    if (!zpq_chain_plan(&M, (int)can_hold, c->cus, &bpw)) return ZPQ_E_INTERNAL;
    int nwg = (nblocks + bpw - 1) / bpw;
    const int cap_wg = zpq_chain_max_wgs(&M, c->cus);
    if (nwg > cap_wg) nwg = cap_wg;
    int nslots = nwg * bpw;
    if ((uint64_t)nslots > max_by_mem) nslots = (int)max_by_mem;      // last workgroup partly idle
    if (nslots > nblocks) nslots = nblocks;
    *nslots_out = nslots;
    *grid_out = (nslots + bpw - 1) / bpw;
    *bpw_out = bpw;
    return ZPQ_OK;
}

extern "C" int zpq_chain_touch_decode(const DModel *M);   // the decoder of this (dense) model reads rows through "touched" bitmaps
extern "C" int zpq_pipe_applies(const DModel *M, int blocks_per_wg, int nslots);   // the wave-pipelined encoder codes this plan
extern "C" int zpq_pipe_touch(void);                                                // ... and reads rows through the bitmaps (timing builds only)

static int plan_batch(zpq_ctx *c, const zpq_model *m, uint32_t flags, int nblocks, bool trace, bool own_slot, Plan *P, int decode = 0) try
{
    P->chain = m->d.fast_kind && !(flags & (ZPQ_FLAG_GENERIC | ZPQ_FLAG_LANES | ZPQ_FLAG_NOEOF | ZB_CTX_ONLY)) &&
               !trace && !own_slot && zpq_chain_blocks_per_wg(&m->d) > 0;
    static thread_local DModel sparse_layout, touch_layout;
    P->sp = 0;
    P->touch = false;
    P->M = &m->d;
    // everything else with up to 64 components: one block per wave, lane i = component i
    P->lanes = !P->chain && !(flags & (ZPQ_FLAG_GENERIC | ZPQ_FLAG_NOEOF | ZB_CTX_ONLY | ZB_KEEP_STATE)) &&
               !trace && !own_slot && zpq_lanes_supported(&m->d);
    int nslots, grid, bpw = 0;
    if (P->chain) {
        // Dense tables or the compact line store?  The store costs ~120 instructions per nibble and hashed
        // component, so it is used where it pays: when it lets the batch finish in fewer rounds of resident
        // blocks (levels 3-5 at scale, level 1 beyond ~6900 blocks), or when a dense slot is so large
        // (levels 4-5: 385 MiB / 2 GiB) that clearing it per block costs more than the store does.
        int dn = 0, dg = 0, db = 0;
        const int rcd = plan_chain(c, m->d, nblocks, &dn, &dg, &db);
        const char *ev = getenv("ZPQ_SPARSE_FORCE_LOG2");        // tests: exercise the store on small models
        const int lg = ev ? atoi(ev) : 0;
        const bool forced = lg >= 8 && lg <= 25;
        const uint32_t cap = forced ? (1u << lg) : c->sparse_cap;
        const char *mode = getenv("ZPQ_SPARSE_MODE");            // tuning knob: "never" / "always"
        const bool never = mode && !strcmp(mode, "never"), always = forced || (mode && !strcmp(mode, "always"));
        int sn = 0, sg = 0, sb = 0;
        bool use_sparse = false;
        if (!never && zpq_sparse_layout(m->d, cap, &sparse_layout) && plan_chain(c, sparse_layout, nblocks, &sn, &sg, &sb) == ZPQ_OK) {
            if (rcd != ZPQ_OK || always || m->d.slot_bytes > (256ull << 20)) use_sparse = true;
            else {
                const int rounds_d = (nblocks + dn - 1) / dn, rounds_s = (nblocks + sn - 1) / sn;
                use_sparse = rounds_s < rounds_d;
            }
        }
        if (use_sparse) { P->sp = cap; P->M = &sparse_layout; nslots = sn; grid = sg; bpw = sb; }
        else {
            if (rcd != ZPQ_OK) return rcd;
            nslots = dn; grid = dg; bpw = db;
            // Dense tables WITHOUT clearing (12 MiB per level-2 block), one "touched" bit per row instead: built for the
            // wave-pipelined encoder and the two-hypothesis decoder, parity-green, measured slower than the clearing it saves
            // (EXPERIMENTS.md 4.1) -- only the -DZPP_TOUCH / -DZPQ_TOUCH_DEC timing builds take this path (ZPQ_TOUCH=0: not even they).
            const char *tv = getenv("ZPQ_TOUCH");
            int tn = 0, tg = 0, tb = 0;
            const bool kernel_ok = decode ? zpq_chain_touch_decode(&m->d) != 0 : (zpq_pipe_touch() && zpq_pipe_applies(&m->d, db, dn) != 0);
            if (!(tv && atoi(tv) == 0) && kernel_ok && zpq_touch_layout(m->d, &touch_layout) &&
                plan_chain(c, touch_layout, nblocks, &tn, &tg, &tb) == ZPQ_OK && (decode || zpq_pipe_applies(&touch_layout, tb, tn))) {
                P->touch = true; P->M = &touch_layout; nslots = tn; grid = tg; bpw = tb;
            }
        }
    } else {
        const DModel &M = *P->M;
        const uint64_t max_by_mem = M.slot_bytes ? (c->budget / M.slot_bytes) : (uint64_t)nblocks;
        if (max_by_mem == 0 && !own_slot) return ZPQ_E_NOMEM;
        nslots = nblocks;
        P->gpipe = P->lanes && !decode && zpq_gpipe_applies(&M) != 0;
        P->gdec = P->lanes && decode && zpq_gdec_applies(&M) != 0;
        const int cap_res = c->cus * (P->gpipe ? zpq_gpipe_blocks_per_cu(&M) : P->gdec ? zpq_gdec_blocks_per_cu(&M) : P->lanes ? zpq_lanes_blocks_per_cu(&M) : zpq_generic_blocks_per_cu(&M));
        if (nslots > cap_res) nslots = cap_res;
        if (!own_slot && (uint64_t)nslots > max_by_mem) nslots = (int)max_by_mem;
        grid = nslots;
    }
    if (own_slot) { nslots = 1; grid = 1; }
    P->nslots = nslots; P->grid = grid; P->bpw = bpw;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_ctx_resident_capacity(zpq_ctx *c, const zpq_model *m, uint32_t flags) try
{
    if (!ctx_live(c) || !m) return ZPQ_E_ARG;
    // the largest batch that both directions code in ONE round (a general model's encoder -- a wave per component -- holds more
    // blocks per CU than its decoder)
    Plan P, Q;
    int rc = plan_batch(c, m, flags & 0xffu, 1 << 30, false, false, &P, 0);
    if (rc == ZPQ_OK) rc = plan_batch(c, m, flags & 0xffu, 1 << 30, false, false, &Q, 1);
    return rc != ZPQ_OK ? rc : (P.nslots < Q.nslots ? P.nslots : Q.nslots);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

static int run_batch(zpq_ctx *c, const zpq_model *m, int decode, const BatchArgs &a)
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (!m) return ZPQ_E_ARG;
    if (a.nblocks < 0) return ZPQ_E_ARG;
    if (a.nblocks == 0) return ZPQ_OK;
    if (!a.in_off || !a.out_off || !a.out_len || !a.status) return ZPQ_E_ARG;
    if (a.own_slot && a.nblocks != 1) return ZPQ_E_ARG;
    HIPCK(hipSetDevice(c->device));
    Plan P;
    int rc = plan_batch(c, m, a.flags, a.nblocks, a.trace != nullptr, a.own_slot != nullptr, &P, decode);
    if (rc != ZPQ_OK) return rc;
    const DModel &M = *P.M;
    const bool want_chain = P.chain, want_lanes = P.lanes;
    const int nslots = P.nslots, grid = P.grid, bpw = P.bpw;
    DevModel dm;
    rc = get_dev_model(c, m, M, P.touch ? 0xFFFFFFFEu : P.sp, &dm);
    if (rc != ZPQ_OK) return rc;

    DBatch B;
    memset(&B, 0, sizeof B);
    B.model = dm.d_model; B.img = dm.d_img;
    B.nblocks = a.nblocks; B.flags = a.flags; B.ntrace = a.ntrace;
    B.in = a.in; B.in_off = a.in_off; B.out = a.out; B.out_off = a.out_off;
    B.out_len = a.out_len; B.consumed = a.consumed; B.final_code = a.final_code;
    B.first_byte = a.first_byte; B.status = a.status; B.trace = a.trace; B.ctx_out = a.ctx_out;
    B.squash = c->d_squash; B.stretch = c->d_stretch; B.dt = c->d_dt; B.dt2k = c->d_dt2k;
    B.ns = c->d_ns; B.stretch_c = c->d_stretch_c;
    B.gate_flag = (want_chain && !decode) ? a.gate_flag : nullptr;
    B.gate_pos = a.gate_pos;
    if (a.gate_flag && !B.gate_flag) return ZPQ_E_INTERNAL;      // (a gated upload must meet a kernel that honours the gate)
    B.prog_counter = (want_chain && decode) ? a.prog_counter : nullptr;   // (other kernels do not report: the host then copies after the kernel)
    B.prog_pos = a.prog_pos;

    if (a.own_slot) {
        B.slots = a.own_slot;
    } else {
        rc = c->slots.ensure((size_t)nslots * M.slot_bytes + 256);   // (the slack a MIX row read needs is part of slot_bytes: zpq_model.cpp mix_row_pad)
        if (rc != ZPQ_OK) return rc;
        B.slots = (uint8_t *)c->slots.p;
    }
    B.nslots = nslots;
    c->last_slots = nslots;
    c->last_sp = P.sp;

    HIPCK(hipEventRecord(c->ev0, c->stream));
    if (want_chain && !a.own_slot) {
        const char *name = nullptr;
        rc = zpq_launch_chain(&B, &M, decode, grid, bpw, c->stream, &name);
        if (rc != ZPQ_OK) return rc;
        c->last_name = name;
    } else if (want_lanes && P.gpipe) {
        rc = zpq_launch_gpipe(&B, &M, nslots, c->stream);
        if (rc != ZPQ_OK) return rc;
        c->last_name = "k_gpipe<encode>";
    } else if (want_lanes && P.gdec) {
        rc = zpq_launch_gdec(&B, &M, nslots, c->stream);
        if (rc != ZPQ_OK) return rc;
        c->last_name = "k_gdec<decode>";
    } else if (want_lanes) {
        rc = zpq_launch_lanes(&B, &M, decode, nslots, c->stream);
        if (rc != ZPQ_OK) return rc;
        c->last_name = zpq_lanes_kernel_name(&M, decode);
    } else {
        zpq_launch_generic(&B, &M, decode, grid, c->stream);
        c->last_name = decode ? "k_generic<decode>" : "k_generic<encode>";
    }
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    return ZPQ_OK;
}

extern "C" int zpq_encode_blocks_dev(zpq_ctx *c, const zpq_model *m, int nblocks, const uint8_t *in,
                                     const uint64_t *in_off, uint32_t flags, uint8_t *out,
                                     const uint64_t *out_off, uint32_t *out_len, int32_t *status) try
{
    BatchArgs a = {nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, nullptr, nullptr, nullptr,
                   status, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, 0};
    return run_batch(c, m, 0, a);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_decode_blocks_dev(zpq_ctx *c, const zpq_model *m, int nblocks, const uint8_t *in,
                                     const uint64_t *in_off, uint32_t flags, uint8_t *out,
                                     const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                                     uint32_t *final_code, uint32_t *first_byte, int32_t *status) try
{
    BatchArgs a = {nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, consumed, final_code,
                   first_byte, status, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, 0};
    return run_batch(c, m, 1, a);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ host-pointer forms
static int host_pipeline(zpq_ctx *c, const zpq_model *m, int decode, int nblocks, const uint8_t *in, const uint64_t *in_off,
                         uint32_t flags, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                         uint32_t *final_code, uint32_t *first_byte, int32_t *status);
static int host_batch(zpq_ctx *c, const zpq_model *m, int decode, int nblocks, const uint8_t *in,
                      const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off,
                      uint32_t *out_len, uint32_t *consumed, uint32_t *final_code,
                      uint32_t *first_byte, int32_t *status, uint8_t *own_slot, int32_t *trace,
                      uint32_t ntrace, uint32_t *ctx_out, size_t ctx_words)
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (!m || nblocks < 0) return ZPQ_E_ARG;
    if (nblocks == 0) return ZPQ_OK;
    if (!in_off || !out_off || !out_len || !status) return ZPQ_E_ARG;
    for (int b = 0; b < nblocks; b++)
        if (in_off[b + 1] < in_off[b] || out_off[b + 1] < out_off[b] ||
            in_off[b + 1] - in_off[b] > 0xFFFFFFF0ull || out_off[b + 1] - out_off[b] > 0xFFFFFFF0ull)
            return ZPQ_E_ARG;
    const size_t in_bytes = (size_t)in_off[nblocks], out_bytes = (size_t)out_off[nblocks];
    if ((in_bytes && !in) || (out_bytes && !out)) return ZPQ_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCK(hipSetDevice(c->device));
    if (nblocks >= 2 && !own_slot && !trace && !ctx_out)
        return host_pipeline(c, m, decode, nblocks, in, in_off, flags, out, out_off, out_len, consumed, final_code, first_byte, status);
    int rc;
    const size_t offb = (size_t)(nblocks + 1) * 8, u32b = (size_t)nblocks * 4;
    if ((rc = c->s_in.ensure(in_bytes + 16)) || (rc = c->s_out.ensure(out_bytes + 16)) ||
        (rc = c->s_inoff.ensure(offb)) || (rc = c->s_outoff.ensure(offb)) ||
        (rc = c->s_status.ensure(u32b)))
        return rc;
    for (auto &b : c->s_u32) if ((rc = b.ensure(u32b))) return rc;
    const size_t misc_bytes = (trace ? (size_t)ntrace * 4 : 0) + ctx_words * 4 + 16;
    if ((rc = c->s_misc.ensure(misc_bytes))) return rc;
    hipStream_t s = c->stream;
    if (in_bytes) HIPCK(hipMemcpyAsync(c->s_in.p, in, in_bytes, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(c->s_inoff.p, in_off, offb, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(c->s_outoff.p, out_off, offb, hipMemcpyHostToDevice, s));
    HIPCK(hipMemsetAsync(c->s_status.p, 0xff, u32b, s));   // -1: "kernel never reported"
    HIPCK(hipMemsetAsync(c->s_u32[0].p, 0, u32b, s));
    if (misc_bytes > 16) HIPCK(hipMemsetAsync(c->s_misc.p, 0, misc_bytes, s));
    BatchArgs a;
    memset(&a, 0, sizeof a);
    a.nblocks = nblocks; a.in = (const uint8_t *)c->s_in.p; a.in_off = (const uint64_t *)c->s_inoff.p;
    a.flags = flags; a.out = (uint8_t *)c->s_out.p; a.out_off = (const uint64_t *)c->s_outoff.p;
    a.out_len = (uint32_t *)c->s_u32[0].p;
    a.consumed = decode ? (uint32_t *)c->s_u32[1].p : nullptr;
    a.final_code = decode ? (uint32_t *)c->s_u32[2].p : nullptr;
    a.first_byte = decode ? (uint32_t *)c->s_u32[3].p : nullptr;
    a.status = (int32_t *)c->s_status.p;
    a.own_slot = own_slot;
    if (trace) { a.trace = (int32_t *)c->s_misc.p; a.ntrace = ntrace; }
    if (ctx_out) a.ctx_out = (uint32_t *)c->s_misc.p;
    rc = run_batch(c, m, decode, a);
    if (rc != ZPQ_OK) return rc;
    HIPCK(hipMemcpyAsync(out_len, a.out_len, u32b, hipMemcpyDeviceToHost, s));
    HIPCK(hipMemcpyAsync(status, a.status, u32b, hipMemcpyDeviceToHost, s));
    if (decode && consumed) HIPCK(hipMemcpyAsync(consumed, a.consumed, u32b, hipMemcpyDeviceToHost, s));
    if (decode && final_code) HIPCK(hipMemcpyAsync(final_code, a.final_code, u32b, hipMemcpyDeviceToHost, s));
    if (decode && first_byte) HIPCK(hipMemcpyAsync(first_byte, a.first_byte, u32b, hipMemcpyDeviceToHost, s));
    if (out_bytes && nblocks == 1) {
        // one segment (zpq_block_*): its slab is sized for the worst case; move only what was produced
        HIPCK(hipStreamSynchronize(s));
        const size_t got = out_len[0] < out_bytes ? out_len[0] : out_bytes;
        if (got) HIPCK(hipMemcpyAsync(out, c->s_out.p, got, hipMemcpyDeviceToHost, s));
    } else if (out_bytes) HIPCK(hipMemcpyAsync(out, c->s_out.p, out_bytes, hipMemcpyDeviceToHost, s));
    if (trace) HIPCK(hipMemcpyAsync(trace, c->s_misc.p, (size_t)ntrace * 4, hipMemcpyDeviceToHost, s));
    if (ctx_out) HIPCK(hipMemcpyAsync(ctx_out, c->s_misc.p, ctx_words * 4, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    return ZPQ_OK;
}

// ------------------------------------------------------------------ pinned host memory + pipelined host batches
extern "C" void *zpq_host_alloc(size_t bytes) try
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
} ZPQ_CATCH(return nullptr)
extern "C" void zpq_host_free(void *p) try
{
    if (p) (void)hipHostFree(p);
} ZPQ_CATCH(return)

// device-visible address of a host buffer if it is pinned (zpq_host_alloc / hipHostMalloc / hipHostRegister), else null
static void *pinned_device_ptr(const void *p)
{
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

__global__ void __launch_bounds__(256) k_gather(const uint8_t *src, const uint64_t *src_off, const uint32_t *len,
                                                uint8_t *dst, const uint64_t *dst_off, int n, uint32_t skip);

// Batches of two or more blocks through host pointers, in ROUNDS of at most the resident capacity: round r+1 is
// uploaded and round r-1 downloaded while round r is coded (three streams, two staging sets).  What replaces the
// reference's sequential loop over files (cmd/main.v:283-311: read, compress, write, next file) keeps that order per
// block but moves the three phases of different rounds in parallel.  With PINNED caller buffers (zpq_host_alloc)
// uploads are plain DMA and the output is packed by the GPU straight into the caller's slabs -- only bytes that
// were produced cross PCIe; with pageable buffers the runtime's staged copies are used and whole slabs come back.
static int host_pipeline(zpq_ctx *c, const zpq_model *m, int decode, int nblocks, const uint8_t *in, const uint64_t *in_off,
                         uint32_t flags, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                         uint32_t *final_code, uint32_t *first_byte, int32_t *status)
{
    HIPCK(hipSetDevice(c->device));
    Plan P0;
    int rc = plan_batch(c, m, flags, nblocks, false, false, &P0, decode);
    if (rc != ZPQ_OK) return rc;
    // Rounds only where overlapping transfers with coding can pay: a batch that exceeds the resident capacity AND
    // moves a sizeable amount of data.  Smaller batches go as one launch (blocks beyond the capacity are worked off
    // by the resident groups in turn, inside the kernel).
    uint64_t min_bytes = 64ull << 20;
    if (const char *ev = getenv("ZPQ_PIPE_MIN_BYTES")) min_bytes = strtoull(ev, nullptr, 0);   // tests: force rounds on small data
    const bool big = in_off[nblocks] - in_off[0] + out_off[nblocks] - out_off[0] >= min_bytes;
    const int per_round = (big && P0.nslots > 0) ? P0.nslots : nblocks;
    uint8_t *const out_dev_view = (uint8_t *)pinned_device_ptr(out);           // non-null: the GPU can write the caller's slabs
    const bool in_pinned = pinned_device_ptr(in) != nullptr;
    (void)in_pinned;                                                            // (hipMemcpyAsync takes either kind; pinned = true DMA)
    const int nrounds = (nblocks + per_round - 1) / per_round;
    // hand a finished round's small results (in the set's pinned arrays) to the caller's arrays
    auto deliver = [&](zpq_ctx::PipeSet &S) {
        if (!S.n) return;
        const size_t nn = (size_t)S.n;
        memcpy(out_len + S.b0, S.h_meta, nn * 4);
        if (decode && consumed) memcpy(consumed + S.b0, S.h_meta + nn, nn * 4);
        if (decode && final_code) memcpy(final_code + S.b0, S.h_meta + 2 * nn, nn * 4);
        if (decode && first_byte) memcpy(first_byte + S.b0, S.h_meta + 3 * nn, nn * 4);
        memcpy(status + S.b0, S.h_meta + 4 * nn, nn * 4);
        S.n = 0;
    };
    c->pipe[0].n = c->pipe[1].n = 0;
    const uint32_t *gate_flag_dev = nullptr;
    uint64_t stripe_L = 0, early_L = 0;
    for (int r = 0; r < nrounds; r++) {
        zpq_ctx::PipeSet &S = c->pipe[r & 1];
        const int b0 = r * per_round, b1 = (b0 + per_round < nblocks) ? b0 + per_round : nblocks, n = b1 - b0;
        if (r >= 2) { HIPCK(hipEventSynchronize(S.ev_out)); deliver(S); }       // the set's previous round has left the device
        const uint64_t in_base = in_off[b0], in_bytes = in_off[b1] - in_base;
        const uint64_t out_base = out_off[b0], out_bytes = out_off[b1] - out_base;
        const size_t offb = (size_t)(n + 1) * 8, u32b = (size_t)n * 4;
        if ((rc = S.in.ensure(in_bytes + 16)) || (rc = S.out.ensure(out_bytes + 16)) || (rc = S.inoff.ensure(offb)) ||
            (rc = S.outoff.ensure(offb)) || (rc = S.dstoff.ensure(offb)) || (rc = S.meta.ensure(u32b * 4)) || (rc = S.status.ensure(u32b)) ||
            (rc = S.ensure_host((size_t)n)))
            return rc;
        uint64_t *h_in = S.h_off, *h_out = S.h_off + (n + 1), *h_dst = S.h_off + 2 * (size_t)(n + 1);
        for (int k = 0; k <= n; k++) { h_in[k] = in_off[b0 + k] - in_base; h_out[k] = out_off[b0 + k] - out_base; h_dst[k] = out_off[b0 + k]; }
        // ---- upload on the copy-in stream
        // One round cannot overlap its upload with its own coding by rounds -- every block starts at once.  But an
        // encoder needs a block's bytes only as it gets to them (64 KiB take ~200 ms), so for a single round of equally
        // long blocks from pinned memory the upload is STRIPED: the first 8 KiB of every block go first, the kernel
        // starts, the rest follows while it runs; a lane that reaches the end of the first stripe waits for *h_gate.
        bool striped = false;
        const uint64_t stripe = 8192;
        if (!decode && nrounds == 1 && in_pinned && P0.chain && !P0.sp && zpq_chain_has_hio(&m->d) && n >= 64 && !getenv("ZPQ_NO_STRIPE")) {
            const uint64_t L = h_in[1] - h_in[0];
            striped = L >= 4 * stripe && (L & 3u) == 0;
            for (int k = 1; k < n && striped; k++) striped = h_in[k + 1] - h_in[k] == L;
            if (striped) {
                void *gate_dev = nullptr;
                if (hipHostGetDevicePointer(&gate_dev, c->h_gate, 0) != hipSuccess) { (void)hipGetLastError(); striped = false; }
                else {
                    __atomic_store_n(c->h_gate, 0u, __ATOMIC_RELEASE);
                    HIPCK(hipMemcpy2DAsync(S.in.p, L, in + in_base, L, stripe, (size_t)n, hipMemcpyHostToDevice, c->s_h2d));
                    gate_flag_dev = (const uint32_t *)gate_dev;
                    stripe_L = L;
                }
            }
        }
        if (!striped && in_bytes) HIPCK(hipMemcpyAsync(S.in.p, in + in_base, in_bytes, hipMemcpyHostToDevice, c->s_h2d));
        HIPCK(hipMemcpyAsync(S.inoff.p, h_in, offb, hipMemcpyHostToDevice, c->s_h2d));
        HIPCK(hipMemcpyAsync(S.outoff.p, h_out, offb, hipMemcpyHostToDevice, c->s_h2d));
        HIPCK(hipMemcpyAsync(S.dstoff.p, h_dst, offb, hipMemcpyHostToDevice, c->s_h2d));
        HIPCK(hipMemsetAsync(S.status.p, 0xff, u32b, c->s_h2d));              // -1: "kernel never reported"
        HIPCK(hipMemsetAsync(S.meta.p, 0, u32b * 4, c->s_h2d));
        HIPCK(hipEventRecord(S.ev_in, c->s_h2d));
        // ---- code on the compute stream
        HIPCK(hipStreamWaitEvent(c->stream, S.ev_in, 0));
        uint32_t *d_len = (uint32_t *)S.meta.p, *d_cons = d_len + n, *d_code = d_cons + n, *d_first = d_code + n;
        BatchArgs a;
        memset(&a, 0, sizeof a);
        a.nblocks = n; a.in = (const uint8_t *)S.in.p; a.in_off = (const uint64_t *)S.inoff.p; a.flags = flags;
        a.out = (uint8_t *)S.out.p; a.out_off = (const uint64_t *)S.outoff.p; a.out_len = d_len;
        a.consumed = decode ? d_cons : nullptr; a.final_code = decode ? d_code : nullptr; a.first_byte = decode ? d_first : nullptr;
        a.status = (int32_t *)S.status.p;
        if (striped) { a.gate_flag = gate_flag_dev; a.gate_pos = (uint32_t)stripe; }
        // The mirror image for a decoder's output: every block reports when its first `early` bytes are stored (and
        // visible to the copy engines); once all have, that stripe is copied out beside the kernel, which still has
        // the last quarter of every block to decode.  Single round, pinned slabs of equal length and stride.
        uint32_t early = 0;
        uint32_t *prog_dev = nullptr;
        if (decode && nrounds == 1 && out_dev_view && P0.chain && !P0.sp && zpq_chain_has_hio(&m->d) && n >= 64 && !getenv("ZPQ_NO_STRIPE")) {
            const uint64_t L = h_out[1] - h_out[0];
            bool uni = L >= 16384 && L <= 0x7FFFFFFFull;
            for (int k = 1; k < n && uni; k++) uni = h_out[k + 1] - h_out[k] == L;
            void *pd = nullptr;
            if (uni && hipHostGetDevicePointer(&pd, c->h_gate + 8, 0) == hipSuccess) {
                early = (uint32_t)(L / 4 * 3) & ~255u;
                prog_dev = (uint32_t *)pd;
                __atomic_store_n(c->h_gate + 8, 0u, __ATOMIC_RELEASE);
                a.prog_counter = prog_dev; a.prog_pos = early;
                early_L = L;
            } else (void)hipGetLastError();
        }
        rc = run_batch(c, m, decode, a);
        if (striped) {
            // the rest of every block, beside the running kernel; then the signal (whatever happened: a kernel left
            // waiting for it would only end at its spin limit)
            hipError_t e2 = rc == ZPQ_OK ? hipMemcpy2DAsync((uint8_t *)S.in.p + stripe, stripe_L, in + in_base + stripe, stripe_L, stripe_L - stripe,
                                                            (size_t)n, hipMemcpyHostToDevice, c->s_h2d) : hipSuccess;
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->s_h2d);
            __atomic_store_n(c->h_gate, 1u, __ATOMIC_RELEASE);
            if (e2 != hipSuccess && rc == ZPQ_OK) { (void)hipDeviceSynchronize(); return ZPQ_E_NODEVICE; }
        }
        if (rc != ZPQ_OK) { (void)hipDeviceSynchronize(); return rc; }
        HIPCK(hipEventRecord(S.ev_k, c->stream));
        uint32_t skip = 0;
        if (prog_dev) {
            // wait (politely) until every block has reported, or the kernel has ended without all reports (a kernel
            // family that does not report, an error): then nothing was copied early and everything goes the usual way
            for (;;) {
                if (__atomic_load_n(c->h_gate + 8, __ATOMIC_ACQUIRE) >= (uint32_t)n) { skip = early; break; }
                if (hipEventQuery(S.ev_k) == hipSuccess) { skip = __atomic_load_n(c->h_gate + 8, __ATOMIC_ACQUIRE) >= (uint32_t)n ? early : 0; break; }
                (void)hipGetLastError();
                struct timespec ts = {0, 50000};
                nanosleep(&ts, nullptr);
            }
            if (skip) HIPCK(hipMemcpy2DAsync(out + out_base, early_L, S.out.p, early_L, skip, (size_t)n, hipMemcpyDeviceToHost, c->s_d2h));
        }
        // ---- download on the copy-out stream
        HIPCK(hipStreamWaitEvent(c->s_d2h, S.ev_k, 0));
        HIPCK(hipMemcpyAsync(S.h_meta, d_len, u32b * 4, hipMemcpyDeviceToHost, c->s_d2h));
        HIPCK(hipMemcpyAsync(S.h_meta + 4 * (size_t)n, S.status.p, u32b, hipMemcpyDeviceToHost, c->s_d2h));
        S.b0 = b0; S.n = n;
        if (out_bytes) {
            if (out_dev_view) {
                // the GPU packs each block's produced bytes straight into the caller's (pinned) slab
                hipLaunchKernelGGL(k_gather, dim3(n), dim3(256), 0, c->s_d2h, (const uint8_t *)S.out.p, (const uint64_t *)S.outoff.p,
                                   (const uint32_t *)d_len, out_dev_view, (const uint64_t *)S.dstoff.p, n, skip);
                HIPCK(hipGetLastError());
            } else {
                // pageable destination: the runtime stages this copy and the call returns only when it has run, so
                // with pageable slabs rounds do not overlap (use zpq_host_alloc)
                HIPCK(hipMemcpyAsync(out + out_base, S.out.p, out_bytes, hipMemcpyDeviceToHost, c->s_d2h));
            }
        }
        HIPCK(hipEventRecord(S.ev_out, c->s_d2h));
    }
    HIPCK(hipStreamSynchronize(c->s_d2h));
    HIPCK(hipStreamSynchronize(c->stream));
    deliver(c->pipe[nrounds & 1]);                                             // (older round first)
    deliver(c->pipe[(nrounds + 1) & 1]);
    return ZPQ_OK;
}

extern "C" int zpq_encode_blocks(zpq_ctx *c, const zpq_model *m, int nblocks, const uint8_t *in,
                                 const uint64_t *in_off, uint32_t flags, uint8_t *out,
                                 const uint64_t *out_off, uint32_t *out_len, int32_t *status) try
{
    return host_batch(c, m, 0, nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, nullptr, nullptr,
                      nullptr, status, nullptr, nullptr, 0, nullptr, 0);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_decode_blocks(zpq_ctx *c, const zpq_model *m, int nblocks, const uint8_t *in,
                                 const uint64_t *in_off, uint32_t flags, uint8_t *out,
                                 const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                                 uint32_t *final_code, uint32_t *first_byte, int32_t *status) try
{
    return host_batch(c, m, 1, nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, consumed,
                      final_code, first_byte, status, nullptr, nullptr, 0, nullptr, 0);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ several GPUs behind one call
// Blocks are independent (compressor.v:90,147-148: fresh Predictor/ZPAQL per block), so any static deal gives the same
// bytes.  Context g takes the g-th contiguous share -- contiguous so that its PCIe transfers are -- on its own host
// thread (one host thread per device, no cross-device step but the caller's own buffers).
static int multi_batch(zpq_ctx *const *ctxs, int nctx, const zpq_model *m, int decode, int nblocks, const uint8_t *in,
                       const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                       uint32_t *consumed, uint32_t *final_code, uint32_t *first_byte, int32_t *status)
{
    if (!ctxs || nctx <= 0 || !m || nblocks < 0) return ZPQ_E_ARG;
    for (int g = 0; g < nctx; g++) if (!ctxs[g]) return ZPQ_E_ARG;
    if (nblocks == 0) return ZPQ_OK;
    if (!in_off || !out_off || !out_len || !status) return ZPQ_E_ARG;
    const int G = nctx < nblocks ? nctx : nblocks;
    std::vector<int> rcs((size_t)G, ZPQ_OK);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        const int b0 = (int)((int64_t)nblocks * g / G), b1 = (int)((int64_t)nblocks * (g + 1) / G);
        th.emplace_back([=, &rcs]() {
            rcs[(size_t)g] = host_batch(ctxs[g], m, decode, b1 - b0, in, in_off + b0, flags, out, out_off + b0, out_len + b0,
                                        consumed ? consumed + b0 : nullptr, final_code ? final_code + b0 : nullptr,
                                        first_byte ? first_byte + b0 : nullptr, status + b0, nullptr, nullptr, 0, nullptr, 0);
        });
    }
    for (std::thread &t : th) t.join();
    for (int r : rcs) if (r != ZPQ_OK) return r;
    return ZPQ_OK;
}

extern "C" int zpq_encode_blocks_multi(zpq_ctx *const *ctxs, int nctx, const zpq_model *m, int nblocks, const uint8_t *in,
                                       const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off,
                                       uint32_t *out_len, int32_t *status) try
{
    return multi_batch(ctxs, nctx, m, 0, nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, nullptr, nullptr, nullptr, status);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_decode_blocks_multi(zpq_ctx *const *ctxs, int nctx, const zpq_model *m, int nblocks, const uint8_t *in,
                                       const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off,
                                       uint32_t *out_len, uint32_t *consumed, uint32_t *final_code, uint32_t *first_byte,
                                       int32_t *status) try
{
    return multi_batch(ctxs, nctx, m, 1, nblocks, in, in_off, flags & 0xffu, out, out_off, out_len, consumed, final_code, first_byte, status);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ slab compaction
// out_len[b] bytes of each capacity-strided output slab -> one dense buffer, so that only real
// payload bytes cross PCIe.  One workgroup per block: bytes up to the destination's first 16-byte
// boundary, then 16-byte stores fed by two aligned 16-byte loads funnel-shifted to the source's
// misalignment, then the tail.
// skip: leave out the first `skip` bytes of every block (they have been copied already)
__global__ void __launch_bounds__(256) k_gather(const uint8_t *src, const uint64_t *src_off, const uint32_t *len,
                                                uint8_t *dst, const uint64_t *dst_off, int n, uint32_t skip)
{
    const int b = blockIdx.x;
    if (b >= n) return;
    if (len[b] <= skip) return;
    const uint8_t *s = src + src_off[b] + skip;
    uint8_t *d = dst + dst_off[b] + skip;
    const uint32_t L = len[b] - skip;
    uint32_t head = (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u);
    if (head > L) head = L;
    for (uint32_t i = threadIdx.x; i < head; i += 256) d[i] = s[i];
    const uint32_t nvec = (L - head) / 16u;
    const uint8_t *sv = s + head;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(sv) & 3u);      // source byte offset inside its dword
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(sv - mis);
    uint4 *d16 = reinterpret_cast<uint4 *>(d + head);
    const uint32_t sh = mis * 8u;
    for (uint32_t v = threadIdx.x; v < nvec; v += 256) {
        const uint32_t *q = s4 + 4u * v;
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
        uint4 o;
        if (mis == 0) { o = make_uint4(w0, w1, w2, w3); }
        else {
            const uint32_t w4 = q[4];                       // in bounds: mis > 0 means bytes of the 5th dword belong to this vector
            o.x = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh);
            o.y = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
            o.z = (uint32_t)((((uint64_t)w3 << 32) | w2) >> sh);
            o.w = (uint32_t)((((uint64_t)w4 << 32) | w3) >> sh);
        }
        d16[v] = o;
    }
    for (uint32_t i = head + nvec * 16u + threadIdx.x; i < L; i += 256) d[i] = s[i];
}

extern "C" int zpq_gather_dev(zpq_ctx *c, int nblocks, const uint8_t *src, const uint64_t *src_off, const uint32_t *len,
                              uint8_t *dst, const uint64_t *dst_off) try
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (nblocks < 0) return ZPQ_E_ARG;
    if (nblocks == 0) return ZPQ_OK;
    if (!src_off || !len || !dst_off) return ZPQ_E_ARG;
    HIPCK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_gather, dim3(nblocks), dim3(256), 0, c->stream, src, src_off, len, dst, dst_off, nblocks, 0u);
    return hipGetLastError() == hipSuccess ? ZPQ_OK : ZPQ_E_INTERNAL;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ SHA-1 side kernel (sha1.v:6-146)
extern "C" int zpq_sha1_blocks_dev(zpq_ctx *c, int nblocks, const uint8_t *in, const uint64_t *in_off, uint8_t *out20) try
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (nblocks < 0) return ZPQ_E_ARG;
    if (nblocks == 0) return ZPQ_OK;
    if (!in_off || !out20) return ZPQ_E_ARG;
    HIPCK(hipSetDevice(c->device));
    return zpq_launch_sha1(in, in_off, in_off + 1, nblocks, out20, c->stream);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_sha1_ranges_dev(zpq_ctx *c, int nranges, const uint8_t *in, const uint64_t *begin, const uint64_t *end, uint8_t *out20) try
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (nranges < 0) return ZPQ_E_ARG;
    if (nranges == 0) return ZPQ_OK;
    if (!begin || !end || !out20) return ZPQ_E_ARG;
    HIPCK(hipSetDevice(c->device));
    return zpq_launch_sha1(in, begin, end, nranges, out20, c->stream);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_sha1_blocks(zpq_ctx *c, int nblocks, const uint8_t *in, const uint64_t *in_off, uint8_t *out20) try
{
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (nblocks < 0) return ZPQ_E_ARG;
    if (nblocks == 0) return ZPQ_OK;
    if (!in_off || !out20) return ZPQ_E_ARG;
    for (int b = 0; b < nblocks; b++) if (in_off[b + 1] < in_off[b]) return ZPQ_E_ARG;
    const size_t base = (size_t)in_off[0], in_bytes = (size_t)in_off[nblocks] - base;
    if (in_bytes && !in) return ZPQ_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCK(hipSetDevice(c->device));
    int rc;
    const size_t offb = (size_t)(nblocks + 1) * 8;
    if ((rc = c->s_in.ensure(in_bytes + 16)) || (rc = c->s_inoff.ensure(offb)) || (rc = c->s_out.ensure((size_t)nblocks * 20)))
        return rc;
    std::vector<uint64_t> rel((size_t)nblocks + 1);
    for (int b = 0; b <= nblocks; b++) rel[b] = in_off[b] - base;
    hipStream_t s = c->stream;
    if (in_bytes) HIPCK(hipMemcpyAsync(c->s_in.p, in + base, in_bytes, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(c->s_inoff.p, rel.data(), offb, hipMemcpyHostToDevice, s));
    rc = zpq_launch_sha1((const uint8_t *)c->s_in.p, (const uint64_t *)c->s_inoff.p, (const uint64_t *)c->s_inoff.p + 1, nblocks, (uint8_t *)c->s_out.p, s);
    if (rc != ZPQ_OK) return rc;
    HIPCK(hipMemcpyAsync(out20, c->s_out.p, (size_t)nblocks * 20, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ one block, many segments
extern "C" int zpq_block_create(zpq_ctx *c, const zpq_model *m, zpq_block **out) try
{
    if (!out) return ZPQ_E_ARG;
    *out = nullptr;
    if (!ctx_live(c)) return c ? ZPQ_E_CLOSED : ZPQ_E_ARG;
    if (!m) return ZPQ_E_ARG;
    HIPCK(hipSetDevice(c->device));
    zpq_block *b = new (std::nothrow) zpq_block();
    if (!b) return ZPQ_E_NOMEM;
    b->ctx = c; b->model = m; b->fresh = true; b->slot = nullptr;
    if (hipMalloc((void **)&b->slot, m->d.slot_bytes) != hipSuccess) { (void)hipGetLastError(); delete b; return ZPQ_E_NOMEM; }
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!live_set().count(c)) { (void)hipFree(b->slot); delete b; return ZPQ_E_CLOSED; }
        c->children.push_back(b);
    }
    zpq_model_retain(m);
    *out = b;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" void zpq_block_destroy(zpq_block *b) try
{
    if (!b) return;
    zpq_ctx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        c = b->ctx;                                         // nullptr: the ctx went first and took the slot with it
        if (c) c->children.erase(std::remove(c->children.begin(), c->children.end(), b), c->children.end());
        b->ctx = nullptr;
    }
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (b->slot) (void)hipFree(b->slot);
    }
    zpq_model_release(b->model);
    delete b;
} ZPQ_CATCH(return)

// replay the first segment into b->slot (see zpq_block): encode with no output room -- the coder only counts
static int block_materialise(zpq_block *b)
{
    if (!b->ctx) return ZPQ_E_CLOSED;
    if (!b->lazy) return ZPQ_OK;
    const uint64_t in_off[2] = {0, b->lazy_in.size()}, out_off[2] = {0, 0};
    uint32_t olen = 0;
    int32_t st = 0;
    uint8_t dummy = 0;
    const int rc = host_batch(b->ctx, b->model, 0, 1, b->lazy_in.data(), in_off, b->lazy_flags | ZPQ_FLAG_GENERIC, &dummy, out_off, &olen,
                              nullptr, nullptr, nullptr, &st, b->slot, nullptr, 0, nullptr, 0);
    if (rc != ZPQ_OK) return rc;
    if (st != ZPQ_OK && st != ZPQ_E_OVERFLOW) return st;
    b->lazy = false;
    b->lazy_in.clear();
    b->lazy_in.shrink_to_fit();
    return ZPQ_OK;
}

extern "C" int zpq_block_encode_segment(zpq_block *b, const uint8_t *in, size_t n, uint32_t flags,
                                        uint8_t *out, size_t cap, size_t *out_len) try
{
    if (!b || !out_len || (n && !in) || (cap && !out)) return ZPQ_E_ARG;
    if (!b->ctx) return ZPQ_E_CLOSED;                   // the block's ctx has been destroyed
    const uint64_t in_off[2] = {0, n}, out_off[2] = {0, cap};
    uint32_t olen = 0;
    int32_t st = 0;
    if (b->fresh && !(flags & ZPQ_FLAG_GENERIC)) {
        int rc = host_batch(b->ctx, b->model, 0, 1, in, in_off, flags & 0xffu, out, out_off, &olen, nullptr, nullptr,
                            nullptr, &st, nullptr, nullptr, 0, nullptr, 0);
        if (rc != ZPQ_OK) return rc;
        *out_len = olen;
        if (st != ZPQ_OK) return st;                    // nothing persisted: the block is still fresh, the caller may retry
        b->fresh = false;
        b->lazy = true;
        b->lazy_in.assign(in, in + n);
        b->lazy_flags = flags & 0xffu;
        return ZPQ_OK;
    }
    if (int rcm = block_materialise(b)) return rcm;
    const uint32_t f = (flags & 0xffu) | ZPQ_FLAG_GENERIC | (b->fresh ? 0u : ZB_KEEP_STATE);
    int rc = host_batch(b->ctx, b->model, 0, 1, in, in_off, f, out, out_off, &olen, nullptr, nullptr,
                        nullptr, &st, b->slot, nullptr, 0, nullptr, 0);
    if (rc != ZPQ_OK) return rc;
    b->fresh = false;
    *out_len = olen;
    return st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_block_decode_segment(zpq_block *b, const uint8_t *in, size_t n, uint32_t flags,
                                        uint8_t *out, size_t cap, size_t *out_len, size_t *consumed,
                                        uint32_t *final_code, uint32_t *first_byte) try
{
    if (!b || !out_len || (n && !in) || (cap && !out)) return ZPQ_E_ARG;
    if (!b->ctx) return ZPQ_E_CLOSED;                   // the block's ctx has been destroyed
    const uint64_t in_off[2] = {0, n}, out_off[2] = {0, cap};
    uint32_t olen = 0, cons = 0, code = 0, first = 0xFFFFFFFFu;
    int32_t st = 0;
    if (b->fresh && !(flags & ZPQ_FLAG_GENERIC)) {
        int rc = host_batch(b->ctx, b->model, 1, 1, in, in_off, flags & 0xffu, out, out_off, &olen, &cons, &code, &first,
                            &st, nullptr, nullptr, 0, nullptr, 0);
        if (rc != ZPQ_OK) return rc;
        *out_len = olen;
        if (consumed) *consumed = cons;
        if (final_code) *final_code = code;
        if (first_byte) *first_byte = first;
        if (st != ZPQ_OK) return st;                    // still fresh
        // the symbols this segment coded: [PP byte] + output.  With ZPQ_FLAG_PP the first symbol came back in `first`
        // (0xFFFFFFFF = the stream ended before it); a later encode replays 0 as the PP byte, anything else literally.
        b->fresh = false;
        b->lazy = true;
        b->lazy_in.clear();
        b->lazy_flags = flags & 0xffu;
        if (flags & ZPQ_FLAG_PP) {
            if (first == 0xFFFFFFFFu) b->lazy_flags &= ~(uint32_t)ZPQ_FLAG_PP;
            else if (first != 0) { b->lazy_flags &= ~(uint32_t)ZPQ_FLAG_PP; b->lazy_in.push_back((uint8_t)first); }
        }
        b->lazy_in.insert(b->lazy_in.end(), out, out + olen);
        return ZPQ_OK;
    }
    if (int rcm = block_materialise(b)) return rcm;
    const uint32_t f = (flags & 0xffu) | ZPQ_FLAG_GENERIC | (b->fresh ? 0u : ZB_KEEP_STATE);
    int rc = host_batch(b->ctx, b->model, 1, 1, in, in_off, f, out, out_off, &olen, &cons, &code, &first,
                        &st, b->slot, nullptr, 0, nullptr, 0);
    if (rc != ZPQ_OK) return rc;
    b->fresh = false;
    *out_len = olen;
    if (consumed) *consumed = cons;
    if (final_code) *final_code = code;
    if (first_byte) *first_byte = first;
    return st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// ------------------------------------------------------------------ test hooks
extern "C" int zpq_debug_contexts(zpq_ctx *c, const zpq_model *m, const uint8_t *in, size_t n, uint32_t *h_out) try
{
    if (!ctx_live(c) || !m || !h_out || (n && !in)) return ZPQ_E_ARG;
    if (m->d.n == 0 || n == 0) return ZPQ_OK;
    const uint64_t in_off[2] = {0, n}, out_off[2] = {0, 0};
    uint32_t olen = 0;
    int32_t st = 0;
    uint8_t dummy = 0;
    int rc = host_batch(c, m, 0, 1, in, in_off, ZPQ_FLAG_GENERIC | ZB_CTX_ONLY, &dummy, out_off, &olen,
                        nullptr, nullptr, nullptr, &st, nullptr, nullptr, 0, h_out, n * (size_t)m->d.n);
    return rc != ZPQ_OK ? rc : st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_debug_encode_trace(zpq_ctx *c, const zpq_model *m, const uint8_t *in, size_t n,
                                      uint32_t flags, uint8_t *out, size_t cap, size_t *out_len,
                                      int32_t *p_trace, size_t ntrace) try
{
    if (!ctx_live(c) || !m || !out_len || !p_trace || (n && !in) || (cap && !out)) return ZPQ_E_ARG;
    const uint64_t in_off[2] = {0, n}, out_off[2] = {0, cap};
    uint32_t olen = 0;
    int32_t st = 0;
    int rc = host_batch(c, m, 0, 1, in, in_off, (flags & 0xffu) | ZPQ_FLAG_GENERIC, out, out_off, &olen,
                        nullptr, nullptr, nullptr, &st, nullptr, p_trace, (uint32_t)ntrace, nullptr, 0);
    if (rc != ZPQ_OK) return rc;
    *out_len = olen;
    return st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)
