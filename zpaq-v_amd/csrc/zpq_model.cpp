// zpq_model.cpp -- host side of libzpaq_hip.so that needs no GPU: the start-up
// tables, the level headers, the header scan and the walk that turns a COMP/HCOMP
// header into a zpq_model (state-slot layout + initial values).
//
// Reference interfaces mirrored (file:line under the reference's zpaq/):
//   init_squash_table/exp_approx  predictor.v:21-70    init_stretch_table/ln_approx predictor.v:73-96,169-190
//   init_dt2k_table predictor.v:99-106   dt_table predictor.v:109-166   StateTable statetable.v:15-100
//   get_compression_level levels.v:26-375   header scan compressor.v:96-145
//   Predictor.init predictor.v:292-470      ZPAQL.inith/initp zpaql.v:74-95
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
#include "zpq_host.h"

namespace zpq {

static Tables g_tab;
static std::once_flag g_tab_once;
static int g_tab_status = ZPQ_OK;

// The two series are evaluated in IEEE double, one rounding per V operation, in the
// V source's order (this file is compiled with -ffp-contract=off).
static double series_exp(double x)
{
    if (x < -20.0) return 0.0;
    if (x > 20.0) return 485165195.4;
    double sum = 1.0, t = 1.0;
    for (int k = 1; k < 40; k++) {
        const double ratio = x / (double)k;
        t = t * ratio;
        sum = sum + t;
        if (t < 1e-15 && t > -1e-15) break;
    }
    return sum;
}

static double series_ln(double x)
{
    if (x <= 0.0) return -20.0;
    if (x > 1e9) return 20.0;
    const double y = (x - 1.0) / (x + 1.0);
    const double yy = y * y;
    double sum = y, t = y;
    for (int k = 1; k < 50; k++) {
        t = t * yy;
        const double add = t / (double)(2 * k + 1);
        sum = sum + add;
        if (t < 1e-15 && t > -1e-15) break;
    }
    return 2.0 * sum;
}

// libzpaq's bit-history state machine, as published; the reference carries its
// output as a 1024-byte literal (statetable.v:15-57).
namespace sm {
static int states_for(int n0, int n1)
{
    static const int cap[6] = {20, 48, 15, 8, 6, 5};
    if (n0 < n1) return states_for(n1, n0);
    if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > cap[n1]) return 0;
    return (n1 > 0 && n0 + n1 <= 17) ? 2 : 1;
}
static int decay(int n)
{
    static const int steps[7] = {1, 2, 3, 4, 5, 7, 8};
    int r = 0;
    for (int s : steps) r += n >= s;
    return r;
}
static void step(int &n0, int &n1, int y)
{
    if (n0 < n1) { step(n1, n0, 1 - y); return; }
    if (y) { n1++; n0 = decay(n0); }
    else { n0++; n1 = decay(n1); }
    while (!states_for(n0, n1)) {
        if (n1 < 2) n0--;
        else { n0 = (n0 * (n1 - 1) + n1 / 2) / n1; n1--; }
    }
}
static void build(uint8_t *ns)
{
    const int N = 50;
    std::vector<uint8_t> id(N * N * 2, 0);
    auto at = [&](int a, int b, int y) -> uint8_t & { return id[(a * N + b) * 2 + y]; };
    int next = 0;
    for (int tot = 0; tot < N; tot++)
        for (int n1 = 0; n1 <= tot; n1++) {
            const int n0 = tot - n1, k = states_for(n0, n1);
            if (!k) continue;
            at(n0, n1, 0) = (uint8_t)next;
            at(n0, n1, 1) = (uint8_t)(next + k - 1);
            next += k;
        }
    memset(ns, 0, 1024);
    for (int n0 = 0; n0 < N; n0++)
        for (int n1 = 0; n1 < N; n1++)
            for (int y = 0; y < states_for(n0, n1); y++) {
                const int s = at(n0, n1, y);
                int a = n0, b = n1;
                step(a, b, 0);
                ns[s * 4] = at(a, b, 0);
                a = n0; b = n1;
                step(a, b, 1);
                ns[s * 4 + 1] = at(a, b, 1);
                ns[s * 4 + 2] = (uint8_t)n0;
                ns[s * 4 + 3] = (uint8_t)n1;
            }
}
}  // namespace sm

static uint64_t fnv64(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static void build_tables()
{
    Tables &T = g_tab;
    memset(&T, 0, sizeof T);
    for (int i = -2047; i <= 2047; i++) {          // predictor.v:21-49
        double d = (double)i / 64.0;
        if (d < -20.0) d = -20.0;
        if (d > 20.0) d = 20.0;
        double e;
        if (d >= 0) {
            const double den = 1.0 + series_exp(-d);
            e = 1.0 / den;
        } else {
            const double t = series_exp(d);
            const double den = 1.0 + t;
            e = t / den;
        }
        double s = 32767.0 * e;
        s = s + 0.5;
        const int v = (int)s;
        T.squash[i + 2047] = v < 1 ? 1 : (v > 32767 ? 32767 : v);
    }
    for (int i = 0; i < 32768; i++) {              // predictor.v:73-96
        const double p = (double)i / 32767.0;
        int v;
        if (p <= 0.0) v = -2047;
        else if (p >= 1.0) v = 2047;
        else {
            const double odds = p / (1.0 - p);
            const double l = series_ln(odds) * 64.0;
            v = (int)l;
            v = v < -2047 ? -2047 : (v > 2047 ? 2047 : v);
        }
        T.stretch[i] = v;
    }
    for (int i = 0; i < 256; i++) T.dt2k[i] = 2048 - 2048 / (i + 1);          // predictor.v:103
    for (int i = 0; i < 1024; i++) T.dt[i] = (1 << 17) / (i * 2 + 3) * 2;     // predictor.v:109
    sm::build(T.ns);

    // Self-check against fingerprints of the tables the oracle tests pin
    // (tests/test_oracle.py::test_tables_match_golden_fingerprints).  A mismatch
    // means this build's floating point differs (FMA contraction, fast-math):
    // refuse to run rather than emit streams that differ from the reference.
    if (fnv64(T.squash, sizeof T.squash) != 0x9a6ae5c2591e9955ull ||
        fnv64(T.stretch, sizeof T.stretch) != 0x5f5cf55be5e5c3a4ull ||
        fnv64(T.ns, sizeof T.ns) != 0x723688bb92cf8054ull)
        g_tab_status = ZPQ_E_INTERNAL;

    // Compact stretch for LDS: for 64 <= p < 32704 the table rises by at most 1 per
    // step, so 16 entries pack into base(i16) + 15 step bits; the 128 end entries
    // are kept exactly.  (Built here, checked exhaustively in build_tables_check.)
    for (int b = 0; b < 2048; b++) {
        const int base = T.stretch[b * 16];
        uint32_t bits = 0;
        for (int k = 1; k < 16; k++)
            if (T.stretch[b * 16 + k] != T.stretch[b * 16 + k - 1]) bits |= 1u << k;
        T.stretch_c[b] = ((uint32_t)(uint16_t)(int16_t)base << 16) | bits;
    }
    for (int k = 0; k < 64; k++) {
        T.stretch_c[2048 + k] = (uint32_t)(uint16_t)(int16_t)T.stretch[k];
        T.stretch_c[2048 + 64 + k] = (uint32_t)(uint16_t)(int16_t)T.stretch[32704 + k];
    }
    for (int p = 64; p < 32704; p++) {
        const int step = T.stretch[p] - T.stretch[p - 1];
        if ((p & 15) && step != 0 && step != 1) g_tab_status = ZPQ_E_INTERNAL;
    }
    // k_chain decodes EVERY index below 32767 from the packed words (the table's only step above 1 is its last one, and no
    // ICM counter reaches it): word 0 starts at stretch(1), which is what the reference returns for index 0 as well
    // (predictor.v:205-214).  The decode, exhaustively:
    {
        const int base = T.stretch[1];
        uint32_t bits = 0;
        for (int k = 2; k < 16; k++)
            if (T.stretch[k] != T.stretch[k - 1]) bits |= 1u << k;
        T.stretch_c[0] = ((uint32_t)(uint16_t)(int16_t)base << 16) | bits;
    }
    for (int q = 0; q < 32767; q++) {
        const uint32_t wv = T.stretch_c[q >> 4];
        const uint32_t k = q & 15, field = (wv >> 1) & ((1u << k) - 1u);
        const int v = ((int32_t)wv >> 16) + __builtin_popcount(field);
        if (v != T.stretch[q < 1 ? 1 : q]) g_tab_status = ZPQ_E_INTERNAL;
    }
    for (int st = 0; st < 256; st++) {                                           // an ICM counter starts below index 32767 (and
        const uint32_t n0 = T.ns[st * 4 + 2], n1 = T.ns[st * 4 + 3];              // no update carries it there: zpq_chain.hip)
        if (((((n1 * 2 + 1) << 22) / (n0 + n1 + 1)) >> 8) >= 32767u) g_tab_status = ZPQ_E_INTERNAL;
    }
}

const Tables &tables(int *status)
{
    std::call_once(g_tab_once, build_tables);
    if (status) *status = g_tab_status;
    return g_tab;
}

uint64_t tables_fnv(int which)
{
    const Tables &T = tables(nullptr);
    switch (which) {
    case 0: return fnv64(T.squash, sizeof T.squash);
    case 1: return fnv64(T.stretch, sizeof T.stretch);
    default: return fnv64(T.ns, sizeof T.ns);
    }
}

static inline int sq(const Tables &T, int d)
{
    int i = d + 2047;
    if (i < 0) i = 0;
    if (i >= 4094) i = 4093;
    return T.squash[i];
}
static inline int str(const Tables &T, int p)
{
    if (p < 1) p = 1;
    if (p >= 32768) p = 32767;
    return T.stretch[p];
}
static inline int c512(int x) { return x < -262144 ? -262144 : (x > 262143 ? 262143 : x); }
static inline int cminit(const Tables &T, int s)   // statetable.v:90-100
{
    const uint32_t n0 = T.ns[s * 4 + 2], n1 = T.ns[s * 4 + 3];
    return (int)(((n1 * 2 + 1) << 22) / (n0 + n1 + 1));
}

static const int kCompSize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};   // types.v:74-85

}  // namespace zpq

using namespace zpq;

// ---------------------------------------------------------------- C ABI (no GPU needed)

extern "C" int zpq_scan_header(const uint8_t *h, int len, int *cend, int *hbegin, int *hend) try
{
    if (!h || len < 0 || !cend || !hbegin || !hend) return ZPQ_E_ARG;
    if (len < 5) { *cend = *hbegin = *hend = len; return ZPQ_OK; }   // compressor.v:141-145
    int pos = 5;
    for (int i = 0; i < h[4] && pos < len; i++) {                    // compressor.v:97-110
        const int t = h[pos];
        if (t >= 10) break;
        pos += kCompSize[t];
    }
    *cend = pos;
    if (pos < len && h[pos] == 0) pos++;
    *hbegin = pos;
    while (pos < len && h[pos] != 0) {                               // compressor.v:122-139
        const uint8_t op = h[pos++];
        if ((op & 7) == 7) pos += (op == 63) ? 2 : 1;
    }
    *hend = pos;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" int zpq_level_header(int level, uint8_t *buf, int cap, int *len, int *cend, int *hbegin, int *hend) try
{
    // levels.v:40-375: the six literal headers.  Levels 2..5 share one shape:
    // ICM + ISSE chain (+ MIX2), contexts from "b=c c-- *c=a d=0 (hash *d=a d++)* hash *d=a halt".
    struct Shape { int hh, hm, bits, isse, mix2; };
    static const Shape shapes[6] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {9, 16, 16, 2, 0},
                                    {10, 18, 18, 4, 0}, {12, 20, 20, 5, 16}, {14, 22, 22, 7, 18}};
    std::vector<uint8_t> h;
    if (level == 0) h.assign(7, 0);
    else if (level >= 2 && level <= 5) {
        const Shape &s = shapes[level];
        const int n = 1 + s.isse + (s.mix2 ? 1 : 0);
        h = {(uint8_t)s.hh, (uint8_t)s.hm, 0, 0, (uint8_t)n, 3, (uint8_t)s.bits};
        for (int j = 0; j < s.isse; j++) { h.push_back(8); h.push_back((uint8_t)s.bits); h.push_back((uint8_t)j); }
        if (s.mix2) for (int v : {6, s.mix2, s.isse - 1, s.isse, 24, 255}) h.push_back((uint8_t)v);
        for (int v : {0, 74, 18, 104, 95, 0}) h.push_back((uint8_t)v);
        for (int i = 0; i < n - 1; i++) for (int v : {59, 112, 25}) h.push_back((uint8_t)v);
        for (int v : {59, 112, 56, 0, 0}) h.push_back((uint8_t)v);
    } else {   // 1 and anything else (levels.v:34)
        h = {1, 2, 0, 0, 2, 3, 16, 8, 19, 0, 0, 96, 4, 28, 59, 10, 59, 112, 25, 10, 59, 10, 59, 112, 56, 0};
    }
    if (len) *len = (int)h.size();
    if (buf) {
        if (cap < (int)h.size()) return ZPQ_E_ARG;
        memcpy(buf, h.data(), h.size());
    }
    int a, b, c;
    zpq_scan_header(h.data(), (int)h.size(), &a, &b, &c);
    if (cend) *cend = a;
    if (hbegin) *hbegin = b;
    if (hend) *hend = c;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

static std::atomic<uint64_t> g_model_ids{1};

static uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
// zpq_gpipe.hip reads every MIX weight row as eight words whatever m is (no branch around a load): up to 28 bytes past the
// row.  The spare bytes belong to the LAYOUT, so that any slot provider (the ctx pool, a zpq_block's own slot) has them.
static uint64_t mix_row_pad(const DModel &D)
{
    for (int i = 0; i < D.n; i++) if (D.comp[i].type == ZT_MIX) return 32;
    return 0;
}

extern "C" int zpq_model_create(const uint8_t *hdr, int len, int cend, int hbegin, int hend, zpq_model **out) try
{
    if (!out || len < 0 || (len > 0 && !hdr)) return ZPQ_E_ARG;
    *out = nullptr;
    if (len > ZPQ_MAX_HDR) return ZPQ_E_TOOBIG;
    // Offsets index the header on the host (component walk, program-shape compare) and on the device (program
    // fetch): anything outside the header is refused.  hend < hbegin = an empty program, as ZPAQL.run treats it
    // (pc outside [hbegin, hend) stops at once, zpaql.v:170-174).
    if (cend < 0 || hbegin < 0) return ZPQ_E_ARG;
    if (cend > len || hbegin > len || hend > len) return ZPQ_E_HEADER;
    if (hend < hbegin) hend = hbegin;
    int tst = ZPQ_OK;
    const Tables &T = tables(&tst);
    if (tst != ZPQ_OK) return tst;
    zpq_model *m = new (std::nothrow) zpq_model();
    if (!m) return ZPQ_E_NOMEM;
    m->id = g_model_ids.fetch_add(1);
    DModel &D = m->d;
    memset(&D, 0, sizeof D);
    D.hdr_len = len; D.cend = cend; D.hbegin = hbegin; D.hend = hend;
    if (len) memcpy(D.header, hdr, (size_t)len);

    uint64_t off = 0;
    D.regs_off = off; off += sizeof(DVmRegs);
    D.r_off = off; off += 256 * 4;
    D.scal_off = off; off += sizeof(DCompScal) * ZPQ_MAX_COMP;
    off = align_up(off, 256);
    if (len >= 2) {                                   // zpaql.v:74-95
        const int hh = hdr[0], hm = hdr[1];
        if (hh > 0 && hh < 32) { if (hh > 24) { delete m; return ZPQ_E_TOOBIG; } D.hlen = 1u << hh; }
        if (hm > 0 && hm < 32) { if (hm > 28) { delete m; return ZPQ_E_TOOBIG; } D.mlen = 1u << hm; }
    }
    D.h_off = off; off = align_up(off + 4ull * D.hlen, 256);
    D.m_off = off; off = align_up(off + D.mlen, 256);

    // init image: [0,256) ICM cminit, [256,768) ISSE weight pairs, then 32 words per SSE
    std::vector<uint32_t> &img = m->img;
    img.resize(768);
    for (int s = 0; s < 256; s++) {
        img[s] = (uint32_t)cminit(T, s);                                        // predictor.v:366-368
        img[256 + s * 2] = 1u << 15;                                            // predictor.v:443
        img[256 + s * 2 + 1] = (uint32_t)c512(str(T, cminit(T, s) >> 8) * 1024); // predictor.v:444-445
    }

    int n = (len >= 5) ? hdr[4] : 0;                  // predictor.v:300-323
    D.n = n;
    bool fast = n > 0;
    int cp = 5;
    int i = 0;
    for (; i < n && cp < cend; i++) {                 // predictor.v:331
        DComp &c = D.comp[i];
        const int t = hdr[cp];
        c.type = t;
        if (t >= 1 && t <= 9 && cp + kCompSize[t] > len) { delete m; return ZPQ_E_HEADER; }
        auto table = [&](uint64_t bytes) { const uint64_t o = off; off = align_up(off + bytes, 256); return o; };
        switch (t) {
        case ZT_CONST: c.a = hdr[cp + 1]; fast = false; break;
        case ZT_CM:
            c.a = hdr[cp + 1]; c.limit = hdr[cp + 2] * 4;
            if (c.a > 28) { delete m; return ZPQ_E_TOOBIG; }
            c.cm_len = 1u << c.a; c.cm_off = table(4ull * c.cm_len);
            c.cm_fill = ZF_CONST; c.cm_fill_val = 0x80000000u;                   // predictor.v:352-354
            fast = false;
            break;
        case ZT_ICM:
            c.a = hdr[cp + 1];
            if (c.a > 24) { delete m; return ZPQ_E_TOOBIG; }
            c.ht_len = 16u << (c.a + 2); c.cm_len = 256;
            c.cm_off = table(1024); c.ht_off = table(c.ht_len);
            c.cm_fill = ZF_PATTERN; c.cm_fill_val = 0; c.cm_pat_len = 256;
            break;
        case ZT_MATCH:
            c.a = hdr[cp + 1]; c.b = hdr[cp + 2];
            if (c.a > 28 || c.b > 30) { delete m; return ZPQ_E_TOOBIG; }
            c.cm_len = 1u << c.a; c.ht_len = 1u << c.b;
            c.cm_off = table(4ull * c.cm_len); c.ht_off = table(c.ht_len);
            fast = false;
            break;
        case ZT_AVG: c.a = hdr[cp + 1]; c.b = hdr[cp + 2]; c.c = hdr[cp + 3]; fast = false; break;
        case ZT_MIX2:
            c.a = hdr[cp + 1];
            if (c.a > 28) { delete m; return ZPQ_E_TOOBIG; }
            c.b = hdr[cp + 2]; c.c = 1 << c.a;
            c.j = hdr[cp + 2]; c.k = hdr[cp + 3]; c.rate = hdr[cp + 4]; c.mask = hdr[cp + 5];
            c.a16_len = 1u << c.a; c.a16_off = table(2ull * c.a16_len); c.a16_fill = 32768;
            if (c.j >= i || c.k >= i) fast = false;
            break;
        case ZT_MIX: {
            c.a = hdr[cp + 1];
            if (c.a > 24) { delete m; return ZPQ_E_TOOBIG; }
            const int mm = hdr[cp + 3];
            if (mm == 0) { delete m; return ZPQ_E_MIX_M0; }
            c.b = hdr[cp + 2]; c.c = 1 << c.a; c.limit = mm;
            c.rate = hdr[cp + 4]; c.mask = hdr[cp + 5];
            c.cm_len = (uint32_t)c.c * (uint32_t)mm; c.cm_off = table(4ull * c.cm_len);
            c.cm_fill = ZF_CONST; c.cm_fill_val = (uint32_t)(65536 / mm) << 8;   // predictor.v:426 (quirk Q9)
            fast = false;
            break;
        }
        case ZT_ISSE:
            c.a = hdr[cp + 1]; c.b = hdr[cp + 2];
            if (c.a > 24) { delete m; return ZPQ_E_TOOBIG; }
            c.ht_len = 16u << (c.a + 2); c.cm_len = 512;
            c.cm_off = table(2048); c.ht_off = table(c.ht_len);
            c.cm_fill = ZF_PATTERN; c.cm_fill_val = 256; c.cm_pat_len = 512;
            if (c.b >= i) fast = false;
            break;
        case ZT_SSE: {
            c.a = hdr[cp + 1]; c.b = hdr[cp + 2];
            if (c.a > 24) { delete m; return ZPQ_E_TOOBIG; }
            c.cm_len = (1u << c.a) * 32; c.cm_off = table(4ull * c.cm_len);
            c.limit = hdr[cp + 4] * 4; c.rate = hdr[cp + 3];
            c.cm_fill = ZF_PATTERN; c.cm_fill_val = (uint32_t)img.size(); c.cm_pat_len = 32;
            for (int k = 0; k < 32; k++)                                          // predictor.v:459-462
                img.push_back(((uint32_t)sq(T, k * 64 - 992) << 17) | (uint32_t)hdr[cp + 3]);
            fast = false;
            break;
        }
        default: fast = false; break;
        }
        cp += (t >= 1 && t <= 9) ? kCompSize[t] : 1;                              // predictor.v:465-467
    }
    if (i < n) fast = false;   // components past cend stay type 0 (quirk Q14)
    if (n > 0 && D.comp[0].type != ZT_ICM) fast = false;
    D.fast_kind = fast ? 1u : 0u;
    D.slot_bytes = align_up(off + mix_row_pad(D), 256);
    D.zero_bytes = D.slot_bytes;
    D.img_words = (uint32_t)img.size();
    *out = m;
    return ZPQ_OK;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// Compact line store layout.  A block of N bytes probes each hash table 2(N+1) times, so it
// can touch at most that many 64-byte lines; the dense table (64 << sizebits bytes: 256 MiB at
// level 5) is replaced by `cap` line slots plus a tag per slot (line index + 1).  Untouched
// lines are zero in the dense table and lines are zeroed before use here, so the two are
// indistinguishable to the coder ("hard parts").
bool zpq_sparse_layout(const DModel &dense, uint32_t cap, DModel *out)
{
    *out = dense;
    DModel &D = *out;
    if (cap < 16 || (cap & 3u)) return false;
    const uint64_t store = (64ull + 4ull) * cap;
    bool any = false;
    uint64_t off = D.m_off;
    off = align_up(off + D.mlen, 256);
    auto take = [&](uint64_t bytes) { const uint64_t o = off; off = align_up(off + bytes, 256); return o; };
    // everything that has to start out zero (or is initialised in-kernel) first ...
    for (int i = 0; i < D.n; i++) {
        DComp &c = D.comp[i];
        c.sp_cap = 0;
        if (c.cm_len) c.cm_off = take(4ull * c.cm_len);
        if (c.ht_len) {
            const bool hashed = c.type == ZT_ICM || c.type == ZT_ISSE;
            if (hashed && (uint64_t)c.ht_len > store) {
                c.sp_cap = cap;
                c.sp_tag_off = take(4ull * cap);
                c.ht_off = 0;
                any = true;
            } else {
                c.ht_off = take(c.ht_len);
            }
        }
        if (c.a16_len) c.a16_off = take(2ull * c.a16_len);
    }
    D.zero_bytes = align_up(off, 256);
    // ... then the line arrays, which need no clearing: claiming a free slot clears its line
    for (int i = 0; i < D.n; i++) {
        DComp &c = D.comp[i];
        if (c.sp_cap) c.sp_line_off = take(64ull * cap);
    }
    D.slot_bytes = align_up(off + mix_row_pad(D), 256);
    return any;
}

bool zpq_touch_layout(const DModel &dense, DModel *out)
{
    *out = dense;
    DModel &D = *out;
    bool any = false;
    uint64_t off = D.m_off;
    off = align_up(off + D.mlen, 256);
    auto take = [&](uint64_t bytes) { const uint64_t o = off; off = align_up(off + bytes, 256); return o; };
    for (int i = 0; i < D.n; i++) {
        DComp &c = D.comp[i];
        c.sp_cap = 0; c.tb_off = 0;
        if (c.cm_len) c.cm_off = take(4ull * c.cm_len);
        if (c.ht_len) {
            const bool hashed = c.type == ZT_ICM || c.type == ZT_ISSE;
            if (hashed && c.ht_len >= 65536u) { c.tb_off = take(c.ht_len / 128u); c.ht_off = 0; any = true; }
            else c.ht_off = take(c.ht_len);
        }
        if (c.a16_len) c.a16_off = take(2ull * c.a16_len);
    }
    D.zero_bytes = align_up(off, 256);
    for (int i = 0; i < D.n; i++) {
        DComp &c = D.comp[i];
        if (c.tb_off) c.ht_off = take(c.ht_len);
    }
    D.slot_bytes = align_up(off + mix_row_pad(D), 256);
    return any;
}

extern "C" int zpq_model_create_level(int level, zpq_model **out) try
{
    uint8_t h[128];
    int len = 0, a = 0, b = 0, c = 0;
    const int rc = zpq_level_header(level, h, (int)sizeof h, &len, &a, &b, &c);
    if (rc != ZPQ_OK) return rc;
    return zpq_model_create(h, len, a, b, c, out);
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

void zpq_model_retain(const zpq_model *m) { if (m) m->refs.fetch_add(1, std::memory_order_relaxed); }
void zpq_model_release(const zpq_model *m)
{
    if (m && m->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete m;
}
// Drops the creator's reference; a zpq_block built on the model keeps it alive until the block goes (handles may be
// dropped in any order).
extern "C" void zpq_model_destroy(zpq_model *m) try
{
    zpq_model_release(m);
} ZPQ_CATCH(return)
extern "C" int zpq_model_ncomp(const zpq_model *m) { return m ? m->d.n : 0; }
int zpq_vm_hashchain(const DModel *M)
{
    static const uint8_t head[] = {74, 18, 104, 95, 0}, tail[] = {59, 112, 56};
    const uint8_t *p = M->header + M->hbegin;
    const int plen = M->hend - M->hbegin;
    if (plen < (int)(sizeof head + sizeof tail) || (plen - (int)(sizeof head + sizeof tail)) % 3 != 0) return 0;
    if (memcmp(p, head, sizeof head) != 0 || memcmp(p + plen - sizeof tail, tail, sizeof tail) != 0) return 0;
    const int K = (plen - (int)(sizeof head + sizeof tail)) / 3;
    for (int k = 0; k < K; k++) {
        const uint8_t *q = p + sizeof head + 3 * k;
        if (q[0] != 59 || q[1] != 112 || q[2] != 25) return 0;
    }
    if (M->mlen < 2 || M->hlen < (uint32_t)(K + 1)) return 0;
    return K + 1;
}

extern "C" uint64_t zpq_model_state_bytes(const zpq_model *m) { return m ? m->d.slot_bytes : 0; }
extern "C" int zpq_model_has_fast_path(const zpq_model *m) { return m ? (int)m->d.fast_kind : 0; }

extern "C" int zpq_tables(int32_t *squash4096, int32_t *stretch32768) try
{
    int st = ZPQ_OK;
    const Tables &T = tables(&st);
    if (squash4096) memcpy(squash4096, T.squash, sizeof T.squash);
    if (stretch32768) memcpy(stretch32768, T.stretch, sizeof T.stretch);
    return st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

// dt (predictor.v:111-166), dt2k (predictor.v:99-106) and the StateTable's ns[1024] (statetable.v:15-57) as this library
// builds them; tests hold them equal to the reference's literals (tests/golden/reference_literals.json).
extern "C" int zpq_tables_ex(int32_t *dt1024, int32_t *dt2k256, uint8_t *ns1024) try
{
    int st = ZPQ_OK;
    const Tables &T = tables(&st);
    if (dt1024) memcpy(dt1024, T.dt, sizeof T.dt);
    if (dt2k256) memcpy(dt2k256, T.dt2k, sizeof T.dt2k);
    if (ns1024) memcpy(ns1024, T.ns, sizeof T.ns);
    return st;
} ZPQ_CATCH(return ZPQ_E_INTERNAL)

extern "C" const char *zpq_status_string(int code) try
{
    switch (code) {
    case ZPQ_OK: return "ok";
    case ZPQ_E_NODEVICE: return "no HIP device / HIP runtime error";
    case ZPQ_E_ARG: return "bad argument";
    case ZPQ_E_HEADER: return "malformed header";
    case ZPQ_E_TOOBIG: return "table too large";
    case ZPQ_E_MIX_M0: return "MIX with m == 0";
    case ZPQ_E_NOMEM: return "out of memory";
    case ZPQ_E_OVERFLOW: return "output slab too small";
    case ZPQ_E_VMSTEPS: return "ZPAQL step cap exceeded";
    case ZPQ_E_INTERNAL: return "internal error (self-check failed, or a C++ exception stopped at the C boundary)";
    case ZPQ_E_CLOSED: return "the handle's context has been destroyed";
    default: return "unknown";
    }
} ZPQ_CATCH(return "")

extern "C" const char *zpq_version(void) { return "zpaq-v_amd 0.1 (gfx950)"; }
