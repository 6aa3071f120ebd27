/*
 * zpaq_hip.h -- C ABI of libzpaq_hip.so: the MI355X (gfx950) ZPAQ block codec.
 *
 * This is the drop-in boundary for ONE hot path of dy-tea/zpaq-v: the bit-serial
 * context-mixing coder.  Each entry point cites the reference interface it
 * replaces (file:line under the reference's zpaq/ directory).  The reference's
 * front end (compressor.v / decompressor.v) keeps its API; its per-byte calls
 *     c.enc.compress(ch)            compressor.v:272,287,375-378
 *     d.dec.decompress()            decompressor.v:470,485
 * are lifted to one call per SEGMENT (zpq_block_*) or per BATCH of independent
 * blocks (zpq_encode_blocks / zpq_decode_blocks).  INTEGRATION.md shows the V
 * `fn C.…` binding a maintainer would add.
 *
 * Conventions: extern "C", plain pointers and sizes, caller-owned buffers that
 * must stay valid for the duration of the call, int return = ZPQ_OK or a
 * negative ZPQ_E_* code, never aborts, no exceptions.  All results are
 * bit-identical to the reference's V CPU path (as restated by oracle/).
 * There is NO CPU fallback: without a HIP device every compute entry point
 * returns ZPQ_E_NODEVICE.
 */
#ifndef ZPAQ_HIP_H
#define ZPAQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (library-level returns and per-block status[]) ---- */
#define ZPQ_OK 0
#define ZPQ_E_NODEVICE (-1)  /* no HIP device / HIP runtime failure */
#define ZPQ_E_ARG (-2)       /* NULL pointer, negative count, bad offsets */
#define ZPQ_E_HEADER (-3)    /* malformed COMP/HCOMP header (V would panic on an index) */
#define ZPQ_E_TOOBIG (-4)    /* a table the header asks for exceeds the supported size */
#define ZPQ_E_MIX_M0 (-5)    /* MIX with m==0: the reference divides by m (predictor.v:426) */
#define ZPQ_E_NOMEM (-6)     /* host or device allocation failed */
#define ZPQ_E_OVERFLOW (-7)  /* per block: output slab too small (out_len holds the needed size if known) */
#define ZPQ_E_VMSTEPS (-8)   /* per block: HCOMP exceeded ZPQ_VM_STEP_CAP steps in one run (reference would hang) */
#define ZPQ_E_INTERNAL (-9)  /* self-check failed, or a C++ exception (host allocation, the HIP runtime) was stopped at this boundary */
#define ZPQ_E_CLOSED (-10)   /* the zpq_ctx this handle (a ctx pointer, a zpq_block) belongs to has been destroyed */

#define ZPQ_VM_STEP_CAP (1u << 20)

/* ---- flags ---- */
#define ZPQ_FLAG_PP 1u        /* encode: code a leading PP-mode byte 0 first (compressor.v:271-274).
                                 decode: drop the first decoded byte (the PP byte) from the output
                                 and report it in first_byte[] (decompressor.v:469-475). */
#define ZPQ_FLAG_GENERIC 2u   /* force the generic all-component interpreter kernel */
#define ZPQ_FLAG_LANES 8u     /* force the lane-per-component kernel (any model with <= 64 components) */
#define ZPQ_FLAG_NOEOF 4u     /* encode: stop after the last data byte, no compress(-1)/flush().  Only the
                                 reference's component-less end_segment path needs it (compressor.v:364). */

/* ---- header helpers: levels.v:26-375 (get_compression_level) and the scan
 *      that defines cend/hbegin/hend, compressor.v:96-145 ---- */
int zpq_level_header(int level, uint8_t *buf, int cap, int *len, int *cend, int *hbegin, int *hend);
int zpq_scan_header(const uint8_t *hdr, int len, int *cend, int *hbegin, int *hend);

/* ---- model: what Predictor.init(&z) + ZPAQL.inith/initp derive from a header
 *      (predictor.v:292-470, zpaql.v:74-95).  Device independent. ---- */
typedef struct zpq_model zpq_model;
int zpq_model_create(const uint8_t *hdr, int len, int cend, int hbegin, int hend, zpq_model **out);
int zpq_model_create_level(int level, zpq_model **out);
void zpq_model_destroy(zpq_model *);
int zpq_model_ncomp(const zpq_model *);
uint64_t zpq_model_state_bytes(const zpq_model *); /* device bytes of one block's tables */
int zpq_model_has_fast_path(const zpq_model *);    /* 1 if the LDS-resident chain kernel applies */

/* ---- per-device context: stream, read-only tables, per-block state slots ---- */
/* Handle lifetime: handles may be destroyed in ANY order (a garbage-collected host language will).  zpq_ctx_destroy
 * releases the device state of every zpq_block still alive on the ctx and orphans it: later calls on such a block
 * return ZPQ_E_CLOSED, zpq_block_destroy still has to be called and frees the host struct only.  A zpq_block keeps
 * its zpq_model alive (zpq_model_destroy drops the creator's reference).  Calls on a destroyed zpq_ctx* return
 * ZPQ_E_CLOSED / ZPQ_E_ARG (the pointer is checked against the set of living contexts), zpq_ctx_destroy twice is a
 * no-op.  What stays the caller's duty: no call may be RUNNING on a ctx while another thread destroys it. */
typedef struct zpq_ctx zpq_ctx;
int zpq_ctx_create(int device, zpq_ctx **out);
void zpq_ctx_destroy(zpq_ctx *);
int zpq_ctx_sync(zpq_ctx *);
void *zpq_ctx_stream(zpq_ctx *); /* the hipStream_t all launches of this ctx go to */
int zpq_ctx_device(const zpq_ctx *); /* HIP device index the ctx was created on */
/* Upper bound on state-slot memory this ctx may hold (default: 85% of the HBM free at zpq_ctx_create). */
int zpq_ctx_set_state_budget(zpq_ctx *, uint64_t bytes);
/* Largest block (bytes) the caller will submit to the chain kernel: sizes the compact line store
 * that stands in for every hash table larger than it (a block of N bytes touches at most 2(N+2)
 * lines per table, predictor.v:495-532,558-560; the store holds 1.12x that).  Default 65536: level 1's
 * ISSE and all tables of levels 3-5 use the store, level 2's 4 MiB tables stay dense.  A bigger block gets
 * ZPQ_E_TOOBIG in status[] instead of wrong output. */
int zpq_ctx_set_max_block_bytes(zpq_ctx *, uint64_t bytes);
/* Resident blocks (state slots) the last batch call used; for reporting. */
int zpq_ctx_last_slots(const zpq_ctx *);
/* Capacity (64-byte lines per hash table) of the compact line store the last batch call's kernel ran with;
 * 0 = dense tables.  For reporting and for tests that must know which instantiation they exercised. */
unsigned zpq_ctx_last_line_store(const zpq_ctx *);
/* How many blocks of this model a single launch keeps resident on this ctx (the smaller of what
 * the CUs' LDS/wave slots hold and what the state budget holds); a larger batch is worked off by
 * the resident groups in turn.  Where the encoder and the decoder of a model hold different numbers
 * (general models: a wave per component either way, different register budgets) the smaller one is
 * reported: a batch of this size is one round in both directions.  flags as for the batch calls
 * (kernel choice).  < 0 = ZPQ_E_*. */
int zpq_ctx_resident_capacity(zpq_ctx *, const zpq_model *, uint32_t flags);
/* Time of the last batch's coding kernel alone, from HIP events on the ctx stream (ms). */
float zpq_ctx_last_kernel_ms(const zpq_ctx *);
const char *zpq_ctx_last_kernel_name(const zpq_ctx *);

/*
 * Batch of independent ZPAQ blocks, one segment per block, a FRESH
 * Predictor/ZPAQL per block (compressor.v:90,147-148,184-185).  Replaces, per
 * block b, the loop
 *     enc := Encoder.new(); pr.reset()                compressor.v:238-245
 *     [enc.compress(0)]  if ZPQ_FLAG_PP               compressor.v:271-274
 *     for ch in in[in_off[b] .. in_off[b+1]): enc.compress(ch)   compressor.v:277-290, encoder.v:93-120
 *     enc.compress(-1); enc.flush()                   compressor.v:375-378, encoder.v:130-139
 * and writes exactly the bytes the Encoder would have put() to
 * out[out_off[b] ...], at most out_off[b+1]-out_off[b] of them; out_len[b] =
 * byte count.  status[b] = ZPQ_OK / ZPQ_E_OVERFLOW / ZPQ_E_VMSTEPS.
 * Host-pointer form: copies in/out over PCIe around the kernel.
 */
int zpq_encode_blocks(zpq_ctx *, const zpq_model *, int nblocks, const uint8_t *in,
                      const uint64_t *in_off, uint32_t flags, uint8_t *out,
                      const uint64_t *out_off, uint32_t *out_len, int32_t *status);
/*
 * Mirror: per block  pr.reset(); dec := Decoder.new(); dec.init()  decompressor.v:413-418, decoder.v:29-47
 *                    loop c := dec.decompress() until -1           decompressor.v:470,485, decoder.v:122-145
 * out_len[b] = decoded bytes stored; consumed[b] = bytes the Decoder pulled from
 * its Reader (clipped to the segment length) so the caller can run
 * Decoder.skip() (decoder.v:151-196) from there; final_code[b] = Decoder.code at
 * EOF (skip() starts from it); first_byte[b] = the PP byte when ZPQ_FLAG_PP
 * (else 0xFFFFFFFF).  consumed/final_code/first_byte may be NULL.
 */
int zpq_decode_blocks(zpq_ctx *, const zpq_model *, int nblocks, const uint8_t *in,
                      const uint64_t *in_off, uint32_t flags, uint8_t *out,
                      const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                      uint32_t *final_code, uint32_t *first_byte, int32_t *status);

/*
 * Host-pointer batches are worked off in rounds of at most the resident capacity; the upload of round
 * r+1 and the download of round r-1 run beside the coding of round r (three streams).  Buffers from
 * zpq_host_alloc (pinned, device-visible) make the transfers plain DMA and let the GPU pack each block's
 * produced bytes straight into the caller's slab -- only bytes that exist cross PCIe.  Pageable buffers
 * work too (staged copies by the runtime, whole slabs come back).  One slab may be at most 4 GiB - 16.
 */
void *zpq_host_alloc(size_t bytes);
void zpq_host_free(void *);

/*
 * The same batch dealt over several contexts (one per GPU; several on one GPU also work): context g
 * codes the g-th contiguous share of the blocks on its own host thread.  Replaces the reference's
 * single-threaded loop over files (cmd/main.v:283-311; its -threads option is parsed and dropped,
 * cmd/main.v:35,97,156).  Blocks are independent, so the result is bit-identical for any nctx.
 */
int zpq_encode_blocks_multi(zpq_ctx *const *ctxs, int nctx, const zpq_model *, int nblocks, const uint8_t *in,
                            const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off,
                            uint32_t *out_len, int32_t *status);
int zpq_decode_blocks_multi(zpq_ctx *const *ctxs, int nctx, const zpq_model *, int nblocks, const uint8_t *in,
                            const uint64_t *in_off, uint32_t flags, uint8_t *out, const uint64_t *out_off,
                            uint32_t *out_len, uint32_t *consumed, uint32_t *final_code, uint32_t *first_byte,
                            int32_t *status);

/* Same, but every pointer is a DEVICE pointer and the call only enqueues work on
 * the ctx stream (no host sync, no PCIe).  This is the form bench.py times.
 * Ordering contract: the ctx stream is created hipStreamNonBlocking, so it does NOT
 * synchronise with the legacy default stream (or with torch's current stream).  The caller
 * must make sure the producers of in/in_off/out_off have finished (device or stream
 * synchronize, or an event the ctx stream waits on) before calling, and must zpq_ctx_sync()
 * (or wait on zpq_ctx_stream()) before consuming the results on another stream.
 * Threading contract: the host-pointer entry points (zpq_encode_blocks, zpq_decode_blocks,
 * zpq_sha1_blocks, zpq_block_*) serialise on a per-ctx mutex and may be called from several
 * threads; the _dev forms take no lock -- one thread per ctx at a time (they share the ctx's
 * slot pool and stream), which is how the one-thread-per-device design of SURVEY 8(e) uses them. */
int zpq_encode_blocks_dev(zpq_ctx *, const zpq_model *, int nblocks, const uint8_t *in,
                          const uint64_t *in_off, uint32_t flags, uint8_t *out,
                          const uint64_t *out_off, uint32_t *out_len, int32_t *status);
int zpq_decode_blocks_dev(zpq_ctx *, const zpq_model *, int nblocks, const uint8_t *in,
                          const uint64_t *in_off, uint32_t flags, uint8_t *out,
                          const uint64_t *out_off, uint32_t *out_len, uint32_t *consumed,
                          uint32_t *final_code, uint32_t *first_byte, int32_t *status);

/*
 * Compaction of capacity-strided output slabs: dst[dst_off[b] ..] = src[src_off[b] .. + len[b]).
 * Device pointers, enqueue only.  Lets a front end download exactly the coded bytes that
 * Writer.put() would have received (encoder.v:76-83) instead of whole slabs.
 */
int zpq_gather_dev(zpq_ctx *, int nblocks, const uint8_t *src, const uint64_t *src_off, const uint32_t *len,
                   uint8_t *dst, const uint64_t *dst_off);

/*
 * SHA-1 of nblocks independent byte ranges in[in_off[b] .. in_off[b+1]) -> out20[20*b ..]:
 * the digest Compressor/Decompresser accumulate with sha1.put() per uncompressed byte
 * (compressor.v:284, decompressor.v:493,505; sha1.v:6-146) and store behind marker 253 in the
 * segment trailer (compressor.v:389-395).  One message per GPU lane, so it pays for batches
 * (hundreds of ranges); a front end handling a single large segment should hash on the host.
 * _dev: device pointers, enqueue only on the ctx stream.  zpq_sha1_ranges_dev takes explicit
 * [begin[b], end[b]) pairs (decoded blocks sit in capacity-strided slabs).
 */
int zpq_sha1_ranges_dev(zpq_ctx *, int nranges, const uint8_t *in, const uint64_t *begin, const uint64_t *end, uint8_t *out20);
int zpq_sha1_blocks(zpq_ctx *, int nblocks, const uint8_t *in, const uint64_t *in_off, uint8_t *out20);
int zpq_sha1_blocks_dev(zpq_ctx *, int nblocks, const uint8_t *in, const uint64_t *in_off, uint8_t *out20);

/*
 * One ZPAQ block whose model state persists across segments -- what
 * Compressor{z,pr} / Decompresser{z,pr} hold between start_block and end_block
 * (compressor.v:79-188, decompressor.v:219-346).  zpq_block_create ==
 * "z.clear(); inith(); initp(); pr = Predictor.new(); pr.init(&z)".
 * Each segment call == new coder + pr.reset() + the per-byte loop above.
 */
typedef struct zpq_block zpq_block;
int zpq_block_create(zpq_ctx *, const zpq_model *, zpq_block **out);
void zpq_block_destroy(zpq_block *);
int zpq_block_encode_segment(zpq_block *, const uint8_t *in, size_t n, uint32_t flags,
                             uint8_t *out, size_t cap, size_t *out_len);
int zpq_block_decode_segment(zpq_block *, const uint8_t *in, size_t n, uint32_t flags,
                             uint8_t *out, size_t cap, size_t *out_len, size_t *consumed,
                             uint32_t *final_code, uint32_t *first_byte);

/* ---- inspection / test hooks ---- */
/* squash_table / stretch_table as built at start-up (predictor.v:11-15,21-96). */
int zpq_tables(int32_t *squash4096, int32_t *stretch32768);
/* dt_table (predictor.v:111-166), dt2k (predictor.v:99-106), StateTable.ns (statetable.v:15-57); any pointer may be NULL. */
int zpq_tables_ex(int32_t *dt1024, int32_t *dt2k256, uint8_t *ns1024);
/* Run the device ZPAQL VM over `in` on a fresh block and return, for every input
 * byte, the n context hashes Predictor.update copies out (predictor.v:809-816):
 * h_out[i*ncomp + k].  Exercises zpaql.v:167-954 alone. */
int zpq_debug_contexts(zpq_ctx *, const zpq_model *, const uint8_t *in, size_t n, uint32_t *h_out);
/* Encode one fresh block with the generic kernel and record predict()'s return
 * value for the first ntrace modelled bits (predictor.v:536-668). */
int zpq_debug_encode_trace(zpq_ctx *, const zpq_model *, const uint8_t *in, size_t n,
                           uint32_t flags, uint8_t *out, size_t cap, size_t *out_len,
                           int32_t *p_trace, size_t ntrace);
const char *zpq_status_string(int code);
const char *zpq_version(void);

#ifdef __cplusplus
}
#endif
#endif
