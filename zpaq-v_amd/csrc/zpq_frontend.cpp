// zpq_frontend.cpp -- host front end: zpaq::Compressor / zpaq::Decompresser
// (include/zpaq_frontend.hpp) over the C ABI, plus a flat C surface (zpqf_*) so that
// ctypes tests can drive it the way cmd/main.v:298-311,349-380 drives the V classes.
//
// Everything here is framing and bookkeeping on the host: block locator + header
// (compressor.v:63-75,150-181), segment header (:217-235), trailer 00 00 00 00 / 253 /
// SHA-1 (:380-396), end-of-block 0xFF (:407-410), store mode (:297-354), the locator
// scan (decompressor.v:227-254), header read (:277-334), Decoder.skip (decoder.v:151-196)
// and the PostProcessor PASS state (decompressor.v:56-82).  All modelled coding goes to
// the GPU through zpq_block_encode_segment / zpq_block_decode_segment.
#include "../../include/zpaq_frontend.hpp"
#include "zpq_vm.h"
#include "zpq_host.h"

#include <string.h>

namespace zpaq {

int Reader::read(uint8_t *buf, int n)
{
    int k = 0;
    for (; k < n; k++) {
        const int c = get();
        if (c < 0) break;
        buf[k] = (uint8_t)c;
    }
    return k;
}
void Writer::write(const uint8_t *buf, int n)
{
    for (int i = 0; i < n; i++) put(buf[i]);
}

// ------------------------------------------------------------------ SHA-1 (sha1.v:6-146)
static inline uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

void SHA1::init()
{
    len0_ = 0; bufn_ = 0; final_ = false;
    h_[0] = 0x67452301u; h_[1] = 0xEFCDAB89u; h_[2] = 0x98BADCFEu; h_[3] = 0x10325476u; h_[4] = 0xC3D2E1F0u;
}
void SHA1::process_block()
{
    uint32_t w[80];
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf_[4 * i] << 24 | (uint32_t)buf_[4 * i + 1] << 16 | (uint32_t)buf_[4 * i + 2] << 8 | buf_[4 * i + 3];
    for (int i = 16; i < 80; i++) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
    uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4];
    for (int i = 0; i < 80; i++) {
        uint32_t f, k;
        if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
        else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
        else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
        else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
        const uint32_t t = rol(a, 5) + f + e + k + w[i];
        e = d; d = c; c = rol(b, 30); b = a; a = t;
    }
    h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e;
}
void SHA1::put(int c)
{
    if (final_) return;
    buf_[bufn_++] = (uint8_t)c;
    len0_ += 8;
    if (bufn_ == 64) { process_block(); bufn_ = 0; }
}
void SHA1::write_bytes(const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; i++) put(p[i]);
}
std::vector<uint8_t> SHA1::result()
{
    if (!final_) {
        buf_[bufn_++] = 0x80;
        if (bufn_ > 56) {
            while (bufn_ < 64) buf_[bufn_++] = 0;
            process_block();
            bufn_ = 0;
        }
        while (bufn_ < 56) buf_[bufn_++] = 0;
        for (int i = 7; i >= 0; i--) buf_[bufn_++] = (uint8_t)(len0_ >> (i * 8));
        process_block();
        final_ = true;
    }
    std::vector<uint8_t> out(20);
    for (int i = 0; i < 5; i++) {
        out[4 * i] = (uint8_t)(h_[i] >> 24); out[4 * i + 1] = (uint8_t)(h_[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h_[i] >> 8); out[4 * i + 3] = (uint8_t)h_[i];
    }
    return out;
}

// ------------------------------------------------------------------ Compressor
static const uint8_t kLocator[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

namespace framing {
void block_header(Writer &w, const uint8_t *hdr, int len, int cend, int hbegin, int hend)
{
    for (uint8_t b : kLocator) w.put(b);                            // compressor.v:63-75
    w.put(0x7a); w.put(0x50); w.put(0x51);
    w.put((len >= 5 && hdr[4] != 0) ? 1 : 2);                       // :157-158
    w.put(1);
    const int hsize = (cend + 1) + (hend - hbegin + 1);             // :167
    w.put(hsize & 0xFF);
    w.put((hsize >> 8) & 0xFF);
    for (int i = 0; i <= cend && i < len; i++) w.put(hdr[i]);
    for (int i = hbegin; i <= hend && i < len; i++) w.put(hdr[i]);
}
void segment_header(Writer &w, const std::string &filename, const std::string &comment)
{
    w.put(1);                                                       // compressor.v:217-235
    for (unsigned char c : filename) w.put(c);
    w.put(0);
    for (unsigned char c : comment) w.put(c);
    w.put(0);
    w.put(0);
}
void segment_trailer(Writer &w, const uint8_t sha1[20])
{
    for (int i = 0; i < 4; i++) w.put(0);                           // compressor.v:382-385
    w.put(253);                                                     // :389-395
    for (int i = 0; i < 20; i++) w.put(sha1[i]);
}
void block_end(Writer &w) { w.put(0xFF); }                          // compressor.v:407-410
}  // namespace framing

Compressor::Compressor(zpq_ctx *ctx)
    : state_(kStart), ctx_(ctx), model_(nullptr), block_(nullptr), input_(nullptr), output_(nullptr),
      level_(1), ncomp_(0), pp_coded_(false), first_byte_(true), segs_in_block_(0), err_(ZPQ_OK)
{
}
Compressor::~Compressor() { drop_block(); }

void Compressor::drop_block()
{
    segs_in_block_ = 0;
    if (block_) { zpq_block_destroy(block_); block_ = nullptr; }
    if (model_) { zpq_model_destroy(model_); model_ = nullptr; }
}

void Compressor::start_block(int level)
{
    if (state_ != kStart) return;
    level_ = level;
    uint8_t hdr[256];
    int len = 0, cend = 0, hbegin = 0, hend = 0;
    zpq_level_header(level, hdr, (int)sizeof hdr, &len, &cend, &hbegin, &hend);   // levels.v + compressor.v:96-145
    header_.assign(hdr, hdr + len);
    ncomp_ = len >= 5 ? hdr[4] : 0;
    if (output_) framing::block_header(*output_, hdr, len, cend, hbegin, hend);
    drop_block();
    err_ = zpq_model_create(hdr, len, cend, hbegin, hend, &model_); // Predictor.init(&z) (:184-185)
    if (err_ == ZPQ_OK && ncomp_ > 0 && ctx_) err_ = zpq_block_create(ctx_, model_, &block_);
    else if (ncomp_ > 0 && !ctx_) err_ = ZPQ_E_NODEVICE;
    state_ = kBlock;
}

void Compressor::start_block_hcomp(const std::string &hcomp)
{
    if (state_ != kStart) return;
    // compressor.v:191-209: header bytes taken verbatim, NOTHING written to the output, and
    // cend/hbegin/hend keep the values of a fresh ZPAQL (0): every component stays type 0.
    header_.assign(hcomp.begin(), hcomp.end());
    ncomp_ = header_.size() >= 5 ? header_[4] : 0;
    drop_block();
    err_ = zpq_model_create(header_.data(), (int)header_.size(), 0, 0, 0, &model_);
    if (err_ == ZPQ_OK && ctx_) err_ = zpq_block_create(ctx_, model_, &block_);
    else if (!ctx_) err_ = ZPQ_E_NODEVICE;
    state_ = kBlock;
}

void Compressor::start_segment(const std::string &filename, const std::string &comment)
{
    if (state_ != kBlock) return;
    if (output_) framing::segment_header(*output_, filename, comment);
    sha1_.init();
    stage_.clear();
    pp_coded_ = false;
    store_buf_.clear();
    first_byte_ = true;
    state_ = kSegment;
}

bool Compressor::compress(int n)
{
    if (state_ != kSegment || !input_) return false;
    if (level_ == 0) return compress_store(n);                      // compressor.v:265-267
    if (first_byte_) { pp_coded_ = true; first_byte_ = false; }     // :271-274 PP byte 0 goes through the model
    for (int count = 0; count < n; count++) {
        const int ch = input_->get();
        if (ch < 0) return false;
        sha1_.put(ch);
        stage_.push_back((uint8_t)ch);                              // was: c.enc.compress(ch)
    }
    return true;
}

bool Compressor::compress_store(int n)
{
    if (!input_ || !output_) return false;
    if (first_byte_) { store_buf_.push_back(0); first_byte_ = false; }
    for (int count = 0; count < n; count++) {
        const int ch = input_->get();
        if (ch < 0) return false;
        sha1_.put(ch);
        store_buf_.push_back((uint8_t)ch);
        if (store_buf_.size() >= 65536) flush_store_buffer();
    }
    return true;
}

void Compressor::flush_store_buffer()
{
    if (!output_ || store_buf_.empty()) return;
    const uint32_t sz = (uint32_t)store_buf_.size();
    output_->put((sz >> 24) & 0xFF); output_->put((sz >> 16) & 0xFF);
    output_->put((sz >> 8) & 0xFF); output_->put(sz & 0xFF);
    output_->write(store_buf_.data(), (int)store_buf_.size());
    store_buf_.clear();
}

void Compressor::end_segment()
{
    if (state_ != kSegment) return;
    if (output_) {
        const bool modeled = ncomp_ > 0;
        if (level_ == 0) {                                          // compressor.v:364-372
            flush_store_buffer();
            for (int i = 0; i < 4; i++) output_->put(0);
        } else {
            // was: c.enc.compress(-1); c.enc.flush()  (:375-378).  With no components the
            // reference takes the store branch here and never writes EOF/flush although
            // compress() did code the bytes (:364): keep that with ZPQ_FLAG_NOEOF.
            uint32_t flags = pp_coded_ ? ZPQ_FLAG_PP : 0u;
            if (!modeled) flags |= ZPQ_FLAG_NOEOF;
            if (block_) {
                // Worst-case expansion is ~16x (p16 >= 3/65536); real data needs input + a little.  A block's first
                // segment leaves the block untouched on ZPQ_E_OVERFLOW (zpq_block_encode_segment), so it starts with
                // a tight buffer and retries once with the worst case; later segments get the worst case at once.
                const size_t worst = stage_.size() * 17 + 4096, tight = stage_.size() + stage_.size() / 8 + 4096;
                std::vector<uint8_t> out(segs_in_block_ == 0 ? tight : worst);
                size_t n = 0;
                err_ = zpq_block_encode_segment(block_, stage_.data(), stage_.size(), flags, out.data(), out.size(), &n);
                if (err_ == ZPQ_E_OVERFLOW && segs_in_block_ == 0) {
                    out.resize(worst);
                    err_ = zpq_block_encode_segment(block_, stage_.data(), stage_.size(), flags, out.data(), out.size(), &n);
                }
                if (err_ == ZPQ_OK) { output_->write(out.data(), (int)n); segs_in_block_++; }
            } else if (err_ == ZPQ_OK) err_ = ZPQ_E_NODEVICE;
        }
        const std::vector<uint8_t> h = sha1_.result();
        if (level_ == 0) { output_->put(253); for (uint8_t b : h) output_->put(b); }
        else framing::segment_trailer(*output_, h.data());
    }
    stage_.clear();
    state_ = kBlock;
}

void Compressor::end_block()
{
    if (state_ != kBlock) return;
    if (output_) framing::block_end(*output_);
    state_ = kStart;
}

// ------------------------------------------------------------------ Decompresser
Decompresser::Decompresser(zpq_ctx *ctx)
    : state_(kStart), ctx_(ctx), model_(nullptr), block_(nullptr), input_(nullptr), output_(nullptr),
      pos_(0), slurped_(false), ncomp_(0), store_count_(0), first_seg_(true), decoded_(false),
      seg_pos_(0), seg_empty_(false), final_code_(0), segs_in_block_(0), err_(ZPQ_OK), has_stored_sha1_(false), header_n_(0), header_t0_(0)
{
}
Decompresser::~Decompresser() { drop_block(); }
void Decompresser::drop_block()
{
    if (block_) { zpq_block_destroy(block_); block_ = nullptr; }
    if (model_) { zpq_model_destroy(model_); model_ = nullptr; }
}
void Decompresser::set_input(Reader *r)
{
    input_ = r; in_.clear(); pos_ = 0; slurped_ = false;
}
int Decompresser::get()
{
    if (!slurped_) {                      // the GPU coder needs the stream as a flat buffer
        if (input_) {
            uint8_t tmp[65536];
            for (;;) {
                const int k = input_->read(tmp, (int)sizeof tmp);
                if (k <= 0) break;
                in_.insert(in_.end(), tmp, tmp + k);
            }
        }
        slurped_ = true;
    }
    return pos_ < in_.size() ? in_[pos_++] : -1;
}

static const int kCompSize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};

bool Decompresser::find_block()
{
    if (!input_) return false;
    uint32_t h1 = 0x3D49B113u, h2 = 0x29EB7F93u, h3 = 0x2614BE13u, h4 = 0x3828EB13u;   // decompressor.v:227-236
    for (;;) {
        const int c = get();
        if (c < 0) return false;
        h1 = h1 * 12 + (uint32_t)c; h2 = h2 * 20 + (uint32_t)c; h3 = h3 * 28 + (uint32_t)c; h4 = h4 * 44 + (uint32_t)c;
        if (h1 == 0xB16B88F1u && h2 == 0xFF5376F1u && h3 == 0x72AC5BF1u && h4 == 0x2F909AF1u) break;
    }
    const int level = get();
    if (level != 1 && level != 2) return false;
    if (get() != 1) return false;
    const int lo = get(), hi = get();
    if (lo < 0 || hi < 0) return false;
    const int hsize = lo + hi * 256;
    std::vector<uint8_t> hdr;
    for (int i = 0; i < 5; i++) { const int b = get(); if (b < 0) return false; hdr.push_back((uint8_t)b); }
    const int n = hdr[4];
    for (int i = 0; i < n; i++) {
        const int t = get();
        if (t < 0 || t >= 10) return false;
        hdr.push_back((uint8_t)t);
        for (int j = 1; j < kCompSize[t]; j++) { const int b = get(); if (b < 0) return false; hdr.push_back((uint8_t)b); }
    }
    if (get() != 0) return false;
    hdr.push_back(0);
    const int cend = (int)hdr.size() - 1, hbegin = (int)hdr.size();
    const int hcomp_len = hsize - (int)hdr.size();
    for (int i = 0; i < hcomp_len; i++) { const int b = get(); if (b < 0) return false; hdr.push_back((uint8_t)b); }
    const int hend = (int)hdr.size() - 1;
    ncomp_ = n;
    header_n_ = hdr.size() > 4 ? hdr[4] : 0;                // what the reference passes as "ph, pm": header[4], header[5]
    header_t0_ = hdr.size() > 5 ? hdr[5] : 0;               // (decompressor.v:456-463); stored by the PostProcessor, never used
    segs_in_block_ = 0;
    drop_block();
    err_ = zpq_model_create(hdr.data(), (int)hdr.size(), cend, hbegin, hend, &model_);
    if (err_ != ZPQ_OK) return false;
    if (n > 0) {
        if (!ctx_) { err_ = ZPQ_E_NODEVICE; return false; }
        err_ = zpq_block_create(ctx_, model_, &block_);
        if (err_ != ZPQ_OK) return false;
    }
    state_ = kBlock;
    return true;
}

bool Decompresser::find_filename()
{
    if (state_ != kBlock || !input_) return false;
    const int marker = get();
    if (marker < 0) return false;
    if (marker == 0xFF) { state_ = kStart; return false; }
    std::string fn, cm;
    for (;;) {
        const int c = get();
        if (c < 0) return false;
        if (c == 0) break;
        if (c == 0xFF) { state_ = kStart; return false; }
        fn.push_back((char)c);
    }
    filename_ = fn;
    for (;;) {
        const int c = get();
        if (c < 0) return false;
        if (c == 0) break;
        cm.push_back((char)c);
    }
    comment_ = cm;
    if (get() < 0) return false;
    sha1_.init();
    store_count_ = 0;
    first_seg_ = true;
    decoded_ = false;
    seg_.clear(); seg_pos_ = 0; seg_empty_ = false;
    state_ = kSegment;
    return true;
}

// PostProcessor.write() for a whole decoded segment in PROG mode (decompressor.v:55-152): `in` is
// everything the decoder produced after the mode byte 1: psize lo, psize hi, psize PCOMP bytes, then the
// data bytes, each of which is one run of the PCOMP program; its OUT bytes are the segment's output.
// The reference's PostProcessor VM is set up in a peculiar way, kept as is: header = psize + 300 zero
// bytes with cend = 8, hbegin = 136 and the program at 136..; header[0..1] = (6 + psize) lo/hi;
// initp() then sizes M from header[1] (2^((6+psize) >> 8) bytes if that is 1..31, else no M at all) and
// inith() is never called, so H stays empty (reads 0, writes ignored); ph/pm are stored and never used.
// A size < 1 falls back to PASS (:97-100).  Returns a ZPQ_* status (ZPQ_E_VMSTEPS if a run does not end).
static int postprocess_prog(const std::vector<uint8_t> &in, int ph, int pm, std::vector<uint8_t> *out)
{
    out->clear();
    if (in.size() < 2) return ZPQ_OK;                   // EOS while reading the size: nothing is output (:85-96)
    const int psize = in[0] + in[1] * 256;
    if (psize < 1) { out->assign(in.begin() + 2, in.end()); return ZPQ_OK; }
    if (in.size() < (size_t)psize + 2) return ZPQ_OK;   // EOS while loading the program (:115-118)
    std::vector<uint8_t> header((size_t)psize + 300, 0);
    const int hbegin = 8 + 128, hend = hbegin + psize, total = 8 - 2 + psize;
    header[0] = (uint8_t)(total & 255);
    header[1] = (uint8_t)(total >> 8);
    header[4] = (uint8_t)ph;
    header[5] = (uint8_t)pm;
    memcpy(header.data() + hbegin, in.data() + 2, (size_t)psize);
    const int hm = header[1];
    std::vector<uint8_t> m((hm > 0 && hm < 32) ? ((size_t)1 << hm) : 0, 0);
    std::vector<uint32_t> r(256, 0);
    zpqvm::Vm z;
    z.a = z.b = z.c = z.d = 0; z.f = 0; z.pc = hbegin;
    z.m = m.data(); z.mlen = (uint32_t)m.size();
    z.h = nullptr; z.hlen = 0;
    z.r = r.data();
    z.hdr = header.data(); z.hdr_len = (int32_t)header.size(); z.hbegin = hbegin; z.hend = hend;
    z.out = out;
    for (size_t i = (size_t)psize + 2; i < in.size(); i++)
        if (!zpqvm::vm_run(z, in[i])) return ZPQ_E_VMSTEPS;
    return ZPQ_OK;
}

// One GPU call decodes the whole segment (was: dec.decompress() per byte).
bool Decompresser::decode_segment()
{
    decoded_ = true;
    const size_t remain = in_.size() - pos_;
    // The segment's end is only known once it is decoded.  Its coded bytes cannot contain the 16-byte block
    // locator (a 2^-128 event), so the next locator bounds the input; should the decoder nevertheless run to
    // the end of that window, the call is repeated on everything that follows (what Decoder's Reader would see).
    static const uint8_t tag16[16] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3, 0x7a, 0x50, 0x51};
    size_t window = remain;
    if (remain > 16) {
        const void *m = memmem(in_.data() + pos_, remain, tag16, sizeof tag16);
        if (m) window = (size_t)(static_cast<const uint8_t *>(m) - (in_.data() + pos_));
    }
    std::vector<uint8_t> out;
    // first try: 8x the coded size (the batch entry points move at most 4 GiB per slab), grown on ZPQ_E_OVERFLOW
    const size_t cap_max = 0xFFFFFF00ull;
    size_t cap = std::min<size_t>(window * 8 + 65536, (size_t)256 << 20), n = 0, consumed = 0;
    uint32_t first = 0xFFFFFFFFu;
    for (int attempt = 0; attempt < 12; attempt++) {
        out.resize(cap);
        zpq_block *blk = block_;
        err_ = zpq_block_decode_segment(blk, in_.data() + pos_, window, ZPQ_FLAG_PP, out.data(), cap, &n, &consumed,
                                        &final_code_, &first);
        const bool ran_out = err_ == ZPQ_OK && window < remain && consumed >= window;
        if (err_ != ZPQ_E_OVERFLOW && !ran_out) break;
        // a retry must start from the same model state: that is only possible on a
        // fresh block, which is the case for the first segment; otherwise report it
        if (segs_in_block_ != 0) { if (ran_out) err_ = ZPQ_E_HEADER; break; }
        if (ran_out) window = remain;
        else {
            if (cap >= cap_max) break;
            cap = std::min<size_t>(cap * 8, cap_max);
        }
        zpq_block_destroy(block_); block_ = nullptr;
        if (zpq_block_create(ctx_, model_, &block_) != ZPQ_OK) { err_ = ZPQ_E_NOMEM; break; }
    }
    if (err_ != ZPQ_OK) return false;
    segs_in_block_++;
    pos_ += consumed;
    seg_empty_ = (first == 0xFFFFFFFFu);
    if (first == 1) {                                       // PostProcessor PROG mode (decompressor.v:63,84-148)
        const std::vector<uint8_t> coded_out(out.begin(), out.begin() + (ptrdiff_t)n);
        err_ = postprocess_prog(coded_out, header_n_, header_t0_, &seg_);
        if (err_ != ZPQ_OK) return false;
    } else {
        seg_.assign(out.begin(), out.begin() + (ptrdiff_t)n);  // PASS; a mode byte > 1 also means PASS (:64-66)
    }
    seg_pos_ = 0;
    return true;
}

bool Decompresser::decompress(int n)
{
    if (state_ != kSegment) return false;
    if (ncomp_ == 0) return decompress_store(n);
    if (!decoded_ && !decode_segment()) return false;
    first_seg_ = false;
    if (seg_empty_) return false;                       // decompressor.v:469-475: EOF before the PP byte
    long limit = n < 0 ? 0x7FFFFFFFL : n;               // :478-482
    long count = 0;
    while (count < limit) {
        if (seg_pos_ >= seg_.size()) return false;      // dec.decompress() == -1 (:488-500)
        const int b = seg_[seg_pos_++];
        sha1_.put(b);
        if (output_) output_->put(b);
        count++;
    }
    return true;
}

bool Decompresser::decompress_store(int n)              // decompressor.v:518-587
{
    if (!input_) return false;
    long limit = n < 0 ? 0x7FFFFFFFL : n, count = 0;
    while (count < limit) {
        if (store_count_ == 0) {
            const int b0 = get(), b1 = get(), b2 = get(), b3 = get();
            if (b0 < 0 || b1 < 0 || b2 < 0 || b3 < 0) { err_ = ZPQ_E_HEADER; return false; }   // stream ends inside the segment
            store_count_ = ((uint32_t)b0 << 24) | ((uint32_t)b1 << 16) | ((uint32_t)b2 << 8) | (uint32_t)b3;
            if (store_count_ == 0) return false;
            if (first_seg_) {
                if (get() < 0) { err_ = ZPQ_E_HEADER; return false; }
                store_count_--;
                first_seg_ = false;
                if (store_count_ == 0) continue;
            }
        }
        const int c = get();
        if (c < 0) { err_ = ZPQ_E_HEADER; return false; }   // truncated chunk: same bool as the reference, but not "OK"
        sha1_.put(c);
        if (output_) output_->put(c);
        store_count_--;
        count++;
    }
    return true;
}

void Decompresser::read_segment_end()
{
    if (state_ != kSegment) return;
    int marker = 0;
    if (ncomp_ > 0) {
        // Decoder.skip() (decoder.v:151-196), starting from the decoder's 4-byte window.
        uint32_t curr;
        if (decoded_) curr = final_code_;
        else {                                           // Decoder.init read 4 bytes (decoder.v:38-46)
            curr = 0;
            for (int i = 0; i < 4; i++) { const int c = get(); curr = c < 0 ? (curr << 8) : ((curr << 8) | (uint32_t)c); }
        }
        marker = -1;
        bool ok = true;
        if (curr == 0) { const int c = get(); if (c < 0) ok = false; else curr = (uint32_t)c; }
        while (ok && curr != 0) { const int c = get(); if (c < 0) { ok = false; break; } curr = (curr << 8) | (uint32_t)c; }
        while (ok) { const int c = get(); if (c < 0) break; if (c != 0) { marker = c; break; } }
    } else {
        marker = get();
        // a store segment must end in 253 (+ SHA-1) or 254; the reference reads on regardless ("robustness",
        // decompressor.v:629-632) -- so does this, but the segment is not reported as OK
        if (marker != 253 && marker != 254 && err_ == ZPQ_OK) err_ = ZPQ_E_HEADER;
    }
    has_stored_sha1_ = false;
    if (marker == 253) {                                 // stored SHA-1: read, compared, result unused (:608-628)
        for (int i = 0; i < 20; i++) { const int c = get(); stored_sha1_[i] = (uint8_t)(c < 0 ? 0 : c); }
        has_stored_sha1_ = true;
    }
    state_ = kBlock;
}

}  // namespace zpaq

// ------------------------------------------------------------------ flat C surface for ctypes
// Every entry point stops C++ exceptions here (std::bad_alloc from the staging vectors, anything the layers below let
// through): a boolean call then answers "false" and last_error reports ZPQ_E_INTERNAL.
using namespace zpaq;

struct zpqf_comp {
    Compressor c;
    FileWriter out;
    FileReader *in;
    int guard_err = ZPQ_OK;
    explicit zpqf_comp(zpq_ctx *ctx) : c(ctx), in(nullptr) { c.set_output(&out); }
    ~zpqf_comp() { delete in; }
};
struct zpqf_decomp {
    Decompresser d;
    FileWriter out;
    FileReader *in;
    int guard_err = ZPQ_OK;
    explicit zpqf_decomp(zpq_ctx *ctx) : d(ctx), in(nullptr) { d.set_output(&out); }
    ~zpqf_decomp() { delete in; }
};

namespace {
template <class H, class F> void call_v(H *h, const char *where, F &&f) noexcept
{
    if (!h) return;
    try { f(); } catch (...) { zpq_note_exception(where); h->guard_err = ZPQ_E_INTERNAL; }
}
template <class H, class F> int call_b(H *h, const char *where, F &&f) noexcept
{
    if (!h) return 0;
    try { return f() ? 1 : 0; } catch (...) { zpq_note_exception(where); h->guard_err = ZPQ_E_INTERNAL; return 0; }
}
size_t copy_str(const std::string &s, char *buf, size_t cap)
{
    if (buf && cap) { const size_t k = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(buf, s.data(), k); buf[k] = 0; }
    return s.size();
}
}  // namespace

extern "C" {
zpqf_comp *zpqf_compressor_new(zpq_ctx *ctx) { return zpq_guard<zpqf_comp *>(nullptr, __func__, [&] { return new zpqf_comp(ctx); }); }
void zpqf_compressor_free(zpqf_comp *h) { zpq_guard_v(__func__, [&] { delete h; }); }
void zpqf_compressor_set_input(zpqf_comp *h, const uint8_t *p, size_t n)
{
    call_v(h, __func__, [&] {
        delete h->in;
        h->in = nullptr;
        h->in = new FileReader(std::vector<uint8_t>(p, p + n));
        h->c.set_input(h->in);
    });
}
void zpqf_compressor_start_block(zpqf_comp *h, int level) { call_v(h, __func__, [&] { h->c.start_block(level); }); }
void zpqf_compressor_start_block_hcomp(zpqf_comp *h, const uint8_t *p, size_t n)
{
    call_v(h, __func__, [&] { h->c.start_block_hcomp(std::string((const char *)p, n)); });
}
void zpqf_compressor_start_segment(zpqf_comp *h, const char *fn, const char *cm) { call_v(h, __func__, [&] { h->c.start_segment(fn ? fn : "", cm ? cm : ""); }); }
int zpqf_compressor_compress(zpqf_comp *h, int n) { return call_b(h, __func__, [&] { return h->c.compress(n); }); }
void zpqf_compressor_end_segment(zpqf_comp *h) { call_v(h, __func__, [&] { h->c.end_segment(); }); }
void zpqf_compressor_end_block(zpqf_comp *h) { call_v(h, __func__, [&] { h->c.end_block(); }); }
int zpqf_compressor_last_error(zpqf_comp *h) { return !h ? ZPQ_E_ARG : h->guard_err != ZPQ_OK ? h->guard_err : h->c.last_error(); }
size_t zpqf_compressor_output(zpqf_comp *h, const uint8_t **p)
{
    if (!h || !p) return 0;
    *p = h->out.bytes().data();
    return h->out.bytes().size();
}
void zpqf_compressor_sha1(zpqf_comp *h, uint8_t out20[20])
{
    call_v(h, __func__, [&] {
        const std::vector<uint8_t> r = h->c.get_sha1();
        memcpy(out20, r.data(), 20);
    });
}

zpqf_decomp *zpqf_decompresser_new(zpq_ctx *ctx) { return zpq_guard<zpqf_decomp *>(nullptr, __func__, [&] { return new zpqf_decomp(ctx); }); }
void zpqf_decompresser_free(zpqf_decomp *h) { zpq_guard_v(__func__, [&] { delete h; }); }
void zpqf_decompresser_set_input(zpqf_decomp *h, const uint8_t *p, size_t n)
{
    call_v(h, __func__, [&] {
        delete h->in;
        h->in = nullptr;
        h->in = new FileReader(std::vector<uint8_t>(p, p + n));
        h->d.set_input(h->in);
    });
}
int zpqf_decompresser_find_block(zpqf_decomp *h) { return call_b(h, __func__, [&] { return h->d.find_block(); }); }
int zpqf_decompresser_find_filename(zpqf_decomp *h) { return call_b(h, __func__, [&] { return h->d.find_filename(); }); }
size_t zpqf_decompresser_filename(zpqf_decomp *h, char *buf, size_t cap)
{
    if (!h) return 0;
    return zpq_guard<size_t>(0, __func__, [&] { return copy_str(h->d.get_filename(), buf, cap); });
}
size_t zpqf_decompresser_comment(zpqf_decomp *h, char *buf, size_t cap)
{
    if (!h) return 0;
    return zpq_guard<size_t>(0, __func__, [&] { return copy_str(h->d.get_comment(), buf, cap); });
}
int zpqf_decompresser_decompress(zpqf_decomp *h, int n) { return call_b(h, __func__, [&] { return h->d.decompress(n); }); }
void zpqf_decompresser_read_segment_end(zpqf_decomp *h) { call_v(h, __func__, [&] { h->d.read_segment_end(); }); }
int zpqf_decompresser_last_error(zpqf_decomp *h) { return !h ? ZPQ_E_ARG : h->guard_err != ZPQ_OK ? h->guard_err : h->d.last_error(); }
size_t zpqf_decompresser_output(zpqf_decomp *h, const uint8_t **p)
{
    if (!h || !p) return 0;
    *p = h->out.bytes().data();
    return h->out.bytes().size();
}
void zpqf_decompresser_sha1(zpqf_decomp *h, uint8_t out20[20])
{
    call_v(h, __func__, [&] {
        const std::vector<uint8_t> r = h->d.get_sha1();
        memcpy(out20, r.data(), 20);
    });
}
}
