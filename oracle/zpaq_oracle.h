/*
 * zpaq_oracle.h -- CPU oracle for the ZPAQ context-mixing block codec path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * (dy-tea/zpaq-v, V language) algorithm for the hot path: Predictor
 * predict/update, ZPAQL VM, arithmetic Encoder/Decoder, StateTable, level
 * headers.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product path (zpaq-v_amd/) never links or calls it.
 *
 * Parity status: the reference is V source and no V compiler exists in this
 * environment, so the oracle cannot be run against a reference binary.  It is
 * pinned by (0) the reference's own literal data on this path -- state_table_data,
 * dt_table, the six level headers, compsize, the block locator -- read from the V
 * source into tests/golden/reference_literals.json and compared entry by entry
 * (tests/test_reference_literals.py), (1) every known-answer test the reference's own zpaq_test.v holds
 * for this path (StateTable/cminit/oplen/squash/stretch ranges/coder initial
 * state/level-0 header/level-1 "Hello World!" round trip) and (2) byte-for-byte
 * agreement with a second, independently written Python restatement
 * (oracle/pyref/zpaq_pyref.py) on tables, coded streams and per-bit traces.
 * The reference holds NO golden vector for any compressed byte, so
 * coded-stream parity is "unpinned by the reference" beyond (1)+(2).
 *
 * All citations are file:line in /root/reference/zpaq/.
 */
#ifndef ZPAQ_ORACLE_H
#define ZPAQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tables (predictor.v:7-214, statetable.v:15-100) ---- */
int zo_squash(int d);                 /* predictor.v:193-202 */
int zo_stretch(int p);                /* predictor.v:205-214 */
int zo_clamp2k(int x);                /* predictor.v:217-225 */
int zo_clamp512k(int x);              /* predictor.v:228-236 */
int zo_ns_next(int state, int y);     /* statetable.v:75-84  */
int zo_cminit(int state);             /* statetable.v:90-100 */
int zo_ns_n0(int state);              /* statetable.v:103-108 */
int zo_ns_n1(int state);              /* statetable.v:111-116 */
int zo_oplen(int op);                 /* types.v:51-64 */
int zo_iserr(int op);                 /* types.v:67-69 */
int zo_compsize(int ctype);           /* types.v:74-85 (-1 if ctype>9) */
/* Copy the raw tables out (any pointer may be NULL). */
void zo_tables(int32_t *squash4096, int32_t *stretch32768, int32_t *dt1024,
               int32_t *dt2k256, uint8_t *ns1024);

/* ---- level headers (levels.v:26-375) and the header scan that defines
 *      cend/hbegin/hend (compressor.v:96-145) ---- */
int zo_level_header(int level, uint8_t *buf, int cap); /* returns length */
const char *zo_level_name(int level);
void zo_scan_header(const uint8_t *hdr, int len, int *cend, int *hbegin, int *hend);

/* ---- ZPAQL VM (zpaql.v) ---- */
typedef struct zo_vm zo_vm;
zo_vm *zo_vm_new(const uint8_t *hdr, int len, int cend, int hbegin, int hend);
void zo_vm_free(zo_vm *);
void zo_vm_run(zo_vm *, uint32_t input);           /* zpaql.v:167-175 */
uint32_t zo_vm_reg(const zo_vm *, int which);      /* 0=a 1=b 2=c 3=d 4=f 5=pc */
void zo_vm_set_reg(zo_vm *, int which, uint32_t v);
uint32_t zo_vm_h(const zo_vm *, uint32_t i);       /* masked h_get, zpaql.v:196-202 */
uint32_t zo_vm_m(const zo_vm *, uint32_t i);       /* masked m_get, zpaql.v:178-184 */
uint32_t zo_vm_r(const zo_vm *, int i);
int zo_vm_hlen(const zo_vm *);
int zo_vm_mlen(const zo_vm *);

/* ---- one ZPAQ block's model state: ZPAQL + Predictor
 *      (compressor.v:147-148,184-185; predictor.v:292-470) ---- */
typedef struct zo_codec zo_codec;
/* Returns NULL and sets *err (if non-NULL) on a header the oracle refuses
 * (table too large, MIX with m==0, truncated record). */
zo_codec *zo_codec_new(const uint8_t *hdr, int len, int cend, int hbegin, int hend, int *err);
zo_codec *zo_codec_new_level(int level, int *err);
void zo_codec_free(zo_codec *);
int zo_codec_ncomp(const zo_codec *);
size_t zo_codec_state_bytes(const zo_codec *);

/* Predictor-level access for unit tests and traces. */
void zo_pred_reset(zo_codec *);        /* predictor.v:827-833 */
int zo_pred_predict(zo_codec *);       /* predictor.v:536-668 */
void zo_pred_update(zo_codec *, int y);/* predictor.v:672-824 */
int zo_pred_p(const zo_codec *, int i);
uint32_t zo_pred_h(const zo_codec *, int i);
uint32_t zo_pred_c8(const zo_codec *);
uint32_t zo_pred_hmap4(const zo_codec *);

/* One traced bit: what the coder saw. */
typedef struct {
    int32_t p;      /* predict() return, 1..32767 */
    int32_t y;      /* coded bit */
    uint32_t low;   /* coder low after the bit */
    uint32_t high;  /* coder high after the bit */
} zo_trace_bit;

#define ZO_FLAG_PP 1u /* code a leading PP-mode byte 0 (compressor.v:271-274) */

/*
 * Encode one segment on this block's model state: new Encoder (low=1,
 * high=0xFFFFFFFF; encoder.v:27-34), Predictor.reset(), optional PP byte,
 * every input byte through Encoder.compress (encoder.v:93-120), then
 * compress(-1) and flush() (compressor.v:375-378, encoder.v:130-139).
 * Returns the number of bytes the Encoder would have put(), or -1 if `cap`
 * is too small.  `trace` (may be NULL) receives up to ntrace modelled bits.
 */
int64_t zo_encode_segment(zo_codec *, const uint8_t *in, size_t n, unsigned flags,
                          uint8_t *out, size_t cap, zo_trace_bit *trace, size_t ntrace);

/*
 * Decode one segment: Predictor.reset(), Decoder.init (decoder.v:29-47), then
 * Decoder.decompress() until it returns -1 (decoder.v:122-145).  Every decoded
 * byte (the PP byte included) is stored to out.  Returns the decoded byte
 * count, or -1 if `cap` is too small.  *consumed = Reader position afterwards
 * (bytes the decoder pulled, clipped to n).
 */
int64_t zo_decode_segment(zo_codec *, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                          size_t *consumed, zo_trace_bit *trace, size_t ntrace);

/*
 * Batch helpers (fresh model per block, one segment per block) used as the
 * CPU baseline and for bulk parity checks.  Block b reads
 * in[in_off[b]..in_off[b+1]) and writes out[out_off[b]..out_off[b+1]);
 * out_len[b] = bytes written or -1 on overflow.  nthreads>=1 (pthreads,
 * blocks divided statically).  Each block allocates and zero-fills its own
 * tables like Predictor.init does (predictor.v:325-470).
 */
int zo_encode_blocks(const uint8_t *hdr, int len, int cend, int hbegin, int hend,
                     int nblocks, const uint8_t *in, const uint64_t *in_off, unsigned flags,
                     uint8_t *out, const uint64_t *out_off, int64_t *out_len, int nthreads);
int zo_decode_blocks(const uint8_t *hdr, int len, int cend, int hbegin, int hend,
                     int nblocks, const uint8_t *in, const uint64_t *in_off,
                     uint8_t *out, const uint64_t *out_off, int64_t *out_len, int nthreads);

/* ---- framing (compressor.v:63-75,150-181,212-235,357-413; sha1.v:6-146) ---- */
void zo_sha1(const uint8_t *data, size_t n, uint8_t out20[20]);
/*
 * Whole-archive writer the way cmd/main.v:298-311 drives Compressor: one block,
 * one segment.  called_compress=0 reproduces "compress() never called" (no PP
 * byte).  Returns bytes written or -1 on overflow.
 */
int64_t zo_compress_archive(int level, const char *filename, const char *comment,
                            const uint8_t *data, size_t n, int called_compress,
                            uint8_t *out, size_t cap);
/*
 * Reader the way cmd/main.v:349-380 drives Decompresser: find_block,
 * find_filename, decompress(-1), read_segment_end for the FIRST segment of the
 * first block at or after *pos.  Returns decoded length, -1 if no block, -2 on
 * overflow.  *pos advances past the segment end; sha_ok reports the (ignored
 * by the reference, decompressor.v:619-628) SHA-1 comparison.
 */
int64_t zo_decompress_archive(const uint8_t *arc, size_t n, size_t *pos, char *filename,
                              size_t fncap, char *comment, size_t cmcap, uint8_t *out,
                              size_t cap, int *sha_ok);

#ifdef __cplusplus
}
#endif
#endif
