// zpaq_frontend.hpp -- C++ mirror of the reference's block/segment front end
// (zpaq/compressor.v, zpaq/decompressor.v, zpaq/io.v) on top of the C ABI in
// zpaq_hip.h.  Same method names, argument meaning, call order and the same silent
// state-machine guards (compressor.v:80,213,260,358,403); the per-byte coder calls are
// replaced by one zpq_block_* call per segment.  Framing (block locator, header,
// segment header/trailer, SHA-1, store mode) is done on the host and is byte-identical
// to the reference's writer.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "zpaq_hip.h"

namespace zpaq {

// io.v:6-21
struct Reader {
    virtual ~Reader() {}
    virtual int get() = 0;                       // one byte, -1 at EOF
    virtual int read(uint8_t *buf, int n);       // default: loop over get()
};
struct Writer {
    virtual ~Writer() {}
    virtual void put(int c) = 0;
    virtual void write(const uint8_t *buf, int n);
};

// io.v:40-108
class FileReader : public Reader {
public:
    explicit FileReader(std::vector<uint8_t> data) : data_(std::move(data)), pos_(0) {}
    int get() override { return pos_ < data_.size() ? data_[pos_++] : -1; }
    size_t position() const { return pos_; }
private:
    std::vector<uint8_t> data_;
    size_t pos_;
};
class FileWriter : public Writer {
public:
    void put(int c) override { data_.push_back((uint8_t)c); }
    const std::vector<uint8_t> &bytes() const { return data_; }
private:
    std::vector<uint8_t> data_;
};

// io.v:113-185: a byte buffer that is both a Reader and a Writer
class StringBuffer : public Reader, public Writer {
public:
    StringBuffer() : read_pos_(0) {}
    explicit StringBuffer(std::vector<uint8_t> data) : data_(std::move(data)), read_pos_(0) {}       // from_bytes
    int get() override { return read_pos_ < data_.size() ? data_[read_pos_++] : -1; }
    void put(int c) override { data_.push_back((uint8_t)c); }
    const std::vector<uint8_t> &bytes() const { return data_; }
    int len() const { return (int)data_.size(); }
    void reset_read() { read_pos_ = 0; }
    void clear() { data_.clear(); read_pos_ = 0; }
private:
    std::vector<uint8_t> data_;
    size_t read_pos_;
};

// io.v:24-37 (to_u32 keeps the reference's Go-style precedence: p[0] + (p[1] << 8) + ...)
inline int to_u16(const uint8_t *p, size_t n) { return n < 2 ? 0 : (int)p[0] + (int)p[1] * 256; }
inline uint32_t to_u32(const uint8_t *p, size_t n)
{
    return n < 4 ? 0u : (uint32_t)p[0] + ((uint32_t)p[1] << 8) + ((uint32_t)p[2] << 16) + ((uint32_t)p[3] << 24);
}

// sha1.v:6-146
class SHA1 {
public:
    SHA1() { init(); }
    void init();
    void put(int c);
    void write_bytes(const uint8_t *p, size_t n);
    std::vector<uint8_t> result();
private:
    void process_block();
    uint64_t len0_;
    uint32_t h_[5];
    uint8_t buf_[64];
    int bufn_;
    bool final_;
};

// The framing bytes around a coded payload, shared by Compressor and the batch archive writer.
namespace framing {
void block_header(Writer &w, const uint8_t *hdr, int len, int cend, int hbegin, int hend);   // compressor.v:63-75,150-181
void segment_header(Writer &w, const std::string &filename, const std::string &comment);     // compressor.v:217-235
void segment_trailer(Writer &w, const uint8_t sha1[20]);       // 00 00 00 00, 253, digest   // compressor.v:382-395
void block_end(Writer &w);                                                                   // compressor.v:407-410
}

// compressor.v:16-418
class Compressor {
public:
    explicit Compressor(zpq_ctx *ctx);           // Compressor.new(); ctx may be null for store mode only
    ~Compressor();
    void set_input(Reader *r) { input_ = r; }
    void set_output(Writer *w) { output_ = w; }
    void start_block(int level);                                   // compressor.v:79-188
    void start_block_hcomp(const std::string &hcomp);              // compressor.v:191-209 (quirk Q14 kept)
    void start_segment(const std::string &filename, const std::string &comment);   // :212-255
    bool compress(int n);                                          // :259-293  true iff n bytes were read
    void end_segment();                                            // :357-399
    void end_block();                                              // :402-413
    std::vector<uint8_t> get_sha1() { return sha1_.result(); }     // :416-418
    int last_error() const { return err_; }                        // ZPQ_* of the last GPU call (the reference has none)
private:
    bool compress_store(int n);
    void flush_store_buffer();
    void drop_block();
    enum { kBlock = 0, kSegment = 1, kStart = 2 };
    int state_;
    zpq_ctx *ctx_;
    zpq_model *model_;
    zpq_block *block_;
    Reader *input_;
    Writer *output_;
    SHA1 sha1_;
    int level_;
    int ncomp_;
    std::vector<uint8_t> header_;
    std::vector<uint8_t> stage_;       // bytes of the open segment awaiting the coder
    bool pp_coded_;                    // compress() ran at least once: PP byte goes first (:271-274)
    std::vector<uint8_t> store_buf_;
    bool first_byte_;
    int segs_in_block_;                // segments coded in the open block (the first one may be retried, see end_segment)
    int err_;
};

// decompressor.v:169-640
class Decompresser {
public:
    explicit Decompresser(zpq_ctx *ctx);
    ~Decompresser();
    void set_input(Reader *r);
    void set_output(Writer *w) { output_ = w; }
    bool find_block();                                             // decompressor.v:219-346
    bool find_filename();                                          // :350-429
    std::string get_filename() const { return filename_; }
    std::string get_comment() const { return comment_; }
    bool decompress(int n);                                        // :443-515  n < 0 = all
    void read_segment_end();                                       // :590-635
    std::vector<uint8_t> get_sha1() { return sha1_.result(); }
    int last_error() const { return err_; }
    size_t position() const { return pos_; }                       // bytes of the input stream consumed so far
    bool block_ended() const { return state_ == kStart; }          // find_filename met the end-of-block marker (:364-368)
    // digest stored behind marker 253 of the segment just ended (the reference reads and drops it, :608-628)
    bool stored_sha1(uint8_t out20[20]) const { if (has_stored_sha1_) for (int i = 0; i < 20; i++) out20[i] = stored_sha1_[i]; return has_stored_sha1_; }
private:
    int get();
    bool decompress_store(int n);
    bool decode_segment();
    void drop_block();
    enum { kBlock = 0, kSegment = 1, kFilename = 2, kStart = 3 };
    int state_;
    zpq_ctx *ctx_;
    zpq_model *model_;
    zpq_block *block_;
    Reader *input_;
    Writer *output_;
    std::vector<uint8_t> in_;          // the Reader's bytes, pulled once (the coder needs a flat buffer)
    size_t pos_;
    bool slurped_;
    SHA1 sha1_;
    std::string filename_, comment_;
    int ncomp_;
    uint32_t store_count_;
    bool first_seg_;
    // decoded segment
    bool decoded_;
    std::vector<uint8_t> seg_;         // decoded bytes after the PP byte
    size_t seg_pos_;
    bool seg_empty_;                   // the coder hit EOF before any byte (compress() never called)
    uint32_t final_code_;
    int segs_in_block_;
    int err_;
    bool has_stored_sha1_;
    uint8_t stored_sha1_[20];
    int header_n_, header_t0_;         // header[4], header[5]: handed to the PostProcessor as ph, pm
};

// ---- batch form of the reference CLI's loops (cmd/main.v:239-470) -------------------------------
// The CLI writes ONE BLOCK WITH ONE SEGMENT PER FILE (cmd/main.v:283-311), and blocks are
// independent, so a set of files is a batch: all blocks are coded by one zpq_encode_blocks call
// (and hashed by one zpq_sha1_blocks call) instead of a Compressor per file.  The archive bytes
// are identical to what the per-file Compressor loop writes.
struct ArchiveFile {
    std::string name, comment;
    std::vector<uint8_t> data;
    uint64_t size = 0;        // uncompressed bytes (filled by archive_extract even when data is not wanted)
    bool sha1_ok = true;      // stored digest == digest of the extracted bytes (the reference reads it and ignores it, decompressor.v:608-628)
    int status = ZPQ_OK;      // per-file ZPQ_* code
};
// run_add (cmd/main.v:283-311): appends one block per file to *archive.  level 0..5.
// fragment_bytes > 0 (NOT reference behaviour; the reference parses -fragment and ignores it): a file
// longer than that is cut into blocks of fragment_bytes, so that one big file is thousands of
// independent blocks instead of one serial one.  The first block carries the file's name and comment,
// the others an empty name -- libzpaq's convention for "more of the previous file".  Still a valid ZPAQ
// level-1 stream; the reference's extractor would see the extra blocks as files without a name.
int archive_add(zpq_ctx *ctx, int level, const std::vector<ArchiveFile> &files, std::vector<uint8_t> *archive,
                size_t fragment_bytes = 0);
// run_extract / run_list (cmd/main.v:342-380,440-465): every segment of every block, in archive
// order.  Single-segment modelled blocks are decoded together in one batch; anything else
// (store mode, several segments per block) goes through Decompresser.  want_data = false keeps
// only names, comments and sizes (list).
// Several GPUs: block b -> ctxs[b mod G], one host thread per context, no collective; results are in
// archive order and identical for any G (ctxs[0] also serves the sequential replay path).
int archive_add(const std::vector<zpq_ctx *> &ctxs, int level, const std::vector<ArchiveFile> &files, std::vector<uint8_t> *archive,
                size_t fragment_bytes = 0);
// archive_add over caller-owned bytes: no copy of the file contents on the way to the GPU (what the flat C surface uses)
int archive_add_views(const std::vector<zpq_ctx *> &ctxs, int level, int nfiles, const char *const *names, const char *const *comments,
                      const uint8_t *const *data, const uint64_t *lens, std::vector<uint8_t> *archive, size_t fragment_bytes = 0);
// join_unnamed: a segment without a name is appended to the file before it (archives written with fragment_bytes).
int archive_extract(zpq_ctx *ctx, const uint8_t *arc, size_t n, bool want_data, std::vector<ArchiveFile> *files,
                    bool join_unnamed = false);
int archive_extract(const std::vector<zpq_ctx *> &ctxs, const uint8_t *arc, size_t n, bool want_data, std::vector<ArchiveFile> *files,
                    bool join_unnamed = false);

}  // namespace zpaq

// Flat C surface over the two classes (in-memory Reader/Writer), used by the ctypes tests.
struct zpqf_comp;
struct zpqf_decomp;
extern "C" {
zpqf_comp *zpqf_compressor_new(zpq_ctx *ctx);
void zpqf_compressor_free(zpqf_comp *);
void zpqf_compressor_set_input(zpqf_comp *, const uint8_t *p, size_t n);
void zpqf_compressor_start_block(zpqf_comp *, int level);
void zpqf_compressor_start_block_hcomp(zpqf_comp *, const uint8_t *p, size_t n);
void zpqf_compressor_start_segment(zpqf_comp *, const char *filename, const char *comment);
int zpqf_compressor_compress(zpqf_comp *, int n);
void zpqf_compressor_end_segment(zpqf_comp *);
void zpqf_compressor_end_block(zpqf_comp *);
int zpqf_compressor_last_error(zpqf_comp *);
size_t zpqf_compressor_output(zpqf_comp *, const uint8_t **p);
void zpqf_compressor_sha1(zpqf_comp *, uint8_t out20[20]);
zpqf_decomp *zpqf_decompresser_new(zpq_ctx *ctx);
void zpqf_decompresser_free(zpqf_decomp *);
void zpqf_decompresser_set_input(zpqf_decomp *, const uint8_t *p, size_t n);
int zpqf_decompresser_find_block(zpqf_decomp *);
int zpqf_decompresser_find_filename(zpqf_decomp *);
size_t zpqf_decompresser_filename(zpqf_decomp *, char *buf, size_t cap);
size_t zpqf_decompresser_comment(zpqf_decomp *, char *buf, size_t cap);
int zpqf_decompresser_decompress(zpqf_decomp *, int n);
void zpqf_decompresser_read_segment_end(zpqf_decomp *);
int zpqf_decompresser_last_error(zpqf_decomp *);
size_t zpqf_decompresser_output(zpqf_decomp *, const uint8_t **p);
void zpqf_decompresser_sha1(zpqf_decomp *, uint8_t out20[20]);
/* archive_add / archive_extract for ctypes: names/comments are NUL-terminated, data[i] has lens[i] bytes.
 * The result lives in the returned handle until zpqf_archive_free. */
struct zpqf_archive;
zpqf_archive *zpqf_archive_add_fragmented(zpq_ctx *ctx, int level, int nfiles, const char *const *names, const char *const *comments,
                                          const uint8_t *const *data, const uint64_t *lens, uint64_t fragment_bytes, int *rc);
zpqf_archive *zpqf_archive_add(zpq_ctx *ctx, int level, int nfiles, const char *const *names, const char *const *comments,
                               const uint8_t *const *data, const uint64_t *lens, int *rc);
/* several GPUs: block b -> ctxs[b mod nctx] */
zpqf_archive *zpqf_archive_add_multi(zpq_ctx *const *ctxs, int nctx, int level, int nfiles, const char *const *names,
                                     const char *const *comments, const uint8_t *const *data, const uint64_t *lens,
                                     uint64_t fragment_bytes, int *rc);
zpqf_archive *zpqf_archive_extract_multi(zpq_ctx *const *ctxs, int nctx, const uint8_t *arc, size_t n, int want_data, int *rc);
size_t zpqf_archive_bytes(zpqf_archive *, const uint8_t **p);
zpqf_archive *zpqf_archive_extract(zpq_ctx *ctx, const uint8_t *arc, size_t n, int want_data /* bit 0: data, bit 1: join unnamed */, int *rc);
int zpqf_archive_nfiles(zpqf_archive *);
const char *zpqf_archive_name(zpqf_archive *, int i);
const char *zpqf_archive_comment(zpqf_archive *, int i);
uint64_t zpqf_archive_size(zpqf_archive *, int i);
int zpqf_archive_sha1_ok(zpqf_archive *, int i);
int zpqf_archive_status(zpqf_archive *, int i);
const uint8_t *zpqf_archive_data(zpqf_archive *, int i);
void zpqf_archive_free(zpqf_archive *);
}
