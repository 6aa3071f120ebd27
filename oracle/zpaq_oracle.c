/*
 * zpaq_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see zpaq_oracle.h).
 *
 * Plain-C restatement of dy-tea/zpaq-v's context-mixing codec path.  Every
 * function cites the V source it follows (file:line under
 * /root/reference/zpaq/).  V `int` is 32-bit two's complement with wrapping
 * arithmetic and arithmetic `>>`; V follows Go operator precedence.  Those
 * semantics are made explicit here (wadd/wmul, compile with -fwrapv) and the
 * float-built tables are computed with plain IEEE double ops in source order
 * (compile with -ffp-contract=off, no fast-math).
 *
 * Parity status: see the header.  "parity unpinned by the reference" for coded
 * bytes; pinned by the reference's KATs + an independent second restatement.
 */
#include "zpaq_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;

/* V int arithmetic wraps. */
static inline i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
static inline i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }
static inline i32 wmul(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }
static inline i32 sar(i32 a, int s) { return a >> s; } /* gcc: arithmetic */

/* ------------------------------------------------------------------ */
/* Tables                                                             */
/* ------------------------------------------------------------------ */

static i32 g_squash[4096];   /* predictor.v:21-49  */
static i32 g_stretch[32768]; /* predictor.v:73-96  */
static i32 g_dt2k[256];      /* predictor.v:99-106 */
static i32 g_dt[1024];       /* predictor.v:111-166: literal libzpaq table,
                                dt[i] = (1<<17)/(i*2+3)*2 (checked equal in
                                tests/test_oracle_tables.py via SHA-256) */
static u8 g_ns[1024];        /* statetable.v:15-57: libzpaq sns[], regenerated
                                by libzpaq's published construction; SHA-256
                                fixture pins it to the reference's literal */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* predictor.v:52-70 -- 40-term Taylor series with early break. */
static double exp_approx(double x)
{
    if (x < -20.0) return 0.0;
    if (x > 20.0) return 485165195.4;
    double result = 1.0, term = 1.0;
    for (int i = 1; i < 40; i++) {
        double q = x / (double)i;
        term = term * q;
        result = result + term;
        if (term < 1e-15 && term > -1e-15) break;
    }
    return result;
}

/* predictor.v:169-190 -- <=50-term atanh series. */
static double ln_approx(double x)
{
    if (x <= 0.0) return -20.0;
    if (x > 1e9) return 20.0;
    double y = (x - 1.0) / (x + 1.0);
    double y2 = y * y;
    double result = y, term = y;
    for (int i = 1; i < 50; i++) {
        term = term * y2;
        double q = term / (double)(2 * i + 1);
        result = result + q;
        if (term < 1e-15 && term > -1e-15) break;
    }
    return 2.0 * result;
}

/* libzpaq's published StateTable construction (the reference stores its
 * output as a literal, statetable.v:15-57). */
static int st_num_states(int n0, int n1)
{
    static const int bound[6] = {20, 48, 15, 8, 6, 5};
    if (n0 < n1) return st_num_states(n1, n0);
    if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > bound[n1]) return 0;
    return 1 + (n1 > 0 && n0 + n1 <= 17);
}
static void st_discount(int *n0)
{
    int n = *n0;
    *n0 = (n >= 1) + (n >= 2) + (n >= 3) + (n >= 4) + (n >= 5) + (n >= 7) + (n >= 8);
}
static void st_next_state(int *n0, int *n1, int y)
{
    if (*n0 < *n1) {
        st_next_state(n1, n0, 1 - y);
        return;
    }
    if (y) {
        ++*n1;
        st_discount(n0);
    } else {
        ++*n0;
        st_discount(n1);
    }
    while (!st_num_states(*n0, *n1)) {
        if (*n1 < 2) --*n0;
        else {
            *n0 = (*n0 * (*n1 - 1) + (*n1 / 2)) / *n1;
            --*n1;
        }
    }
}
static void build_ns(void)
{
    enum { N = 50 };
    static u8 t[N][N][2];
    int state = 0;
    memset(t, 0, sizeof t);
    for (int i = 0; i < N; ++i)
        for (int n1 = 0; n1 <= i; ++n1) {
            int n0 = i - n1, n = st_num_states(n0, n1);
            if (n) {
                t[n0][n1][0] = (u8)state;
                t[n0][n1][1] = (u8)(state + n - 1);
                state += n;
            }
        }
    memset(g_ns, 0, sizeof g_ns);
    for (int n0 = 0; n0 < N; ++n0)
        for (int n1 = 0; n1 < N; ++n1)
            for (int y = 0; y < st_num_states(n0, n1); ++y) {
                int s = t[n0][n1][y], s0 = n0, s1 = n1;
                st_next_state(&s0, &s1, 0);
                g_ns[s * 4 + 0] = t[s0][s1][0];
                s0 = n0, s1 = n1;
                st_next_state(&s0, &s1, 1);
                g_ns[s * 4 + 1] = t[s0][s1][1];
                g_ns[s * 4 + 2] = (u8)n0;
                g_ns[s * 4 + 3] = (u8)n1;
            }
}

static void init_tables(void)
{
    /* predictor.v:21-49 */
    memset(g_squash, 0, sizeof g_squash); /* entry 4095 stays 0 (unused) */
    for (int i = -2047; i <= 2047; i++) {
        double d = (double)i / 64.0;
        if (d < -20.0) d = -20.0;
        if (d > 20.0) d = 20.0;
        double e;
        if (d >= 0) {
            e = 1.0 / (1.0 + exp_approx(-d));
        } else {
            double tmp = exp_approx(d);
            e = tmp / (1.0 + tmp);
        }
        double vv = 32767.0 * e;
        vv = vv + 0.5;
        int v = (int)vv; /* V int(f64): truncation */
        g_squash[i + 2047] = v < 1 ? 1 : (v > 32767 ? 32767 : v);
    }
    /* predictor.v:73-96 */
    for (int i = 0; i < 32768; i++) {
        double p = (double)i / 32767.0;
        if (p <= 0.0) g_stretch[i] = -2047;
        else if (p >= 1.0) g_stretch[i] = 2047;
        else {
            double ln_odds = ln_approx(p / (1.0 - p));
            int v = (int)(ln_odds * 64.0);
            g_stretch[i] = v < -2047 ? -2047 : (v > 2047 ? 2047 : v);
        }
    }
    /* predictor.v:99-106: dt2k[i] = 2048 - 2048/(i+1) (NOT libzpaq's) */
    for (int i = 0; i < 256; i++) g_dt2k[i] = 2048 - 2048 / (i + 1);
    /* predictor.v:109 formula; literal at :111-166 */
    for (int i = 0; i < 1024; i++) g_dt[i] = (1 << 17) / (i * 2 + 3) * 2;
    build_ns();
}
static inline void tables_ready(void) { pthread_once(&g_once, init_tables); }

void zo_tables(int32_t *sq, int32_t *st, int32_t *dt, int32_t *dt2k, uint8_t *ns)
{
    tables_ready();
    if (sq) memcpy(sq, g_squash, sizeof g_squash);
    if (st) memcpy(st, g_stretch, sizeof g_stretch);
    if (dt) memcpy(dt, g_dt, sizeof g_dt);
    if (dt2k) memcpy(dt2k, g_dt2k, sizeof g_dt2k);
    if (ns) memcpy(ns, g_ns, sizeof g_ns);
}

/* predictor.v:193-202 */
static inline i32 squash(i32 d)
{
    i32 idx = wadd(d, 2047);
    if (idx < 0) idx = 0;
    if (idx >= 4094) idx = 4093;
    return g_squash[idx];
}
/* predictor.v:205-214 */
static inline i32 stretch(i32 p)
{
    i32 idx = p;
    if (idx < 1) idx = 1;
    if (idx >= 32768) idx = 32767;
    return g_stretch[idx];
}
/* predictor.v:217-225 */
static inline i32 clamp2k(i32 x) { return x < -2048 ? -2048 : (x > 2047 ? 2047 : x); }
/* predictor.v:228-236 */
static inline i32 clamp512k(i32 x) { return x < -262144 ? -262144 : (x > 262143 ? 262143 : x); }

/* statetable.v:75-84 */
static inline i32 ns_next(i32 state, i32 y)
{
    if (state < 0 || state >= 256) return 0;
    i32 idx = state * 4 + y;
    if (idx < 0 || idx >= 1024) return 0;
    return g_ns[idx];
}
/* statetable.v:90-100 */
static inline i32 cminit(i32 state)
{
    if (state < 0 || state >= 256) return 1 << 22;
    u32 n0 = g_ns[state * 4 + 2], n1 = g_ns[state * 4 + 3];
    return (i32)(((n1 * 2 + 1) << 22) / (n0 + n1 + 1));
}

int zo_squash(int d) { tables_ready(); return squash(d); }
int zo_stretch(int p) { tables_ready(); return stretch(p); }
int zo_clamp2k(int x) { return clamp2k(x); }
int zo_clamp512k(int x) { return clamp512k(x); }
int zo_ns_next(int s, int y) { tables_ready(); return ns_next(s, y); }
int zo_cminit(int s) { tables_ready(); return cminit(s); }
int zo_ns_n0(int s) { tables_ready(); return (s < 0 || s >= 256) ? 0 : g_ns[s * 4 + 2]; }
int zo_ns_n1(int s) { tables_ready(); return (s < 0 || s >= 256) ? 0 : g_ns[s * 4 + 3]; }

/* types.v:51-64 */
static inline int oplen(u8 op)
{
    if (op == 255) return 3;
    if ((op & 7) == 7) return 2;
    return 1;
}
int zo_oplen(int op) { return oplen((u8)op); }
int zo_iserr(int op) { return (u8)op == 56; }
/* types.v:74-85 */
static const int g_compsize[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};
int zo_compsize(int t) { return (t < 0 || t > 9) ? -1 : g_compsize[t]; }

/* ------------------------------------------------------------------ */
/* Level headers (levels.v:40-375).  Data, byte for byte.             */
/* ------------------------------------------------------------------ */

static int chain_header(u8 *b, int hh, int hm, int bits, int n_isse, int mix2_bits)
{
    /* levels.v:100-141,154-205,217-282,294-371: ICM + ISSE chain (+MIX2),
     * HCOMP "b=c c-- *c=a d=0 (hash *d=a d++)* hash *d=a halt", then the
     * HCOMP terminator and one extra 0 ("end of PCOMP"). */
    int n = 1 + n_isse + (mix2_bits ? 1 : 0), k = 0;
    b[k++] = (u8)hh; b[k++] = (u8)hm; b[k++] = 0; b[k++] = 0; b[k++] = (u8)n;
    b[k++] = 3; b[k++] = (u8)bits;
    for (int i = 0; i < n_isse; i++) { b[k++] = 8; b[k++] = (u8)bits; b[k++] = (u8)i; }
    if (mix2_bits) {
        b[k++] = 6; b[k++] = (u8)mix2_bits; b[k++] = (u8)(n_isse - 1); b[k++] = (u8)n_isse;
        b[k++] = 24; b[k++] = 255;
    }
    b[k++] = 0;
    b[k++] = 74; b[k++] = 18; b[k++] = 104; b[k++] = 95; b[k++] = 0;
    for (int i = 0; i < n - 1; i++) { b[k++] = 59; b[k++] = 112; b[k++] = 25; }
    b[k++] = 59; b[k++] = 112; b[k++] = 56;
    b[k++] = 0; b[k++] = 0;
    return k;
}

int zo_level_header(int level, uint8_t *buf, int cap)
{
    u8 b[128];
    int k = 0;
    switch (level) {
    case 0: /* levels.v:40-49 */
        memset(b, 0, 7); k = 7; break;
    case 2: k = chain_header(b, 9, 16, 16, 2, 0); break;
    case 3: k = chain_header(b, 10, 18, 18, 4, 0); break;
    case 4: k = chain_header(b, 12, 20, 20, 5, 16); break;
    case 5: k = chain_header(b, 14, 22, 22, 7, 18); break;
    default: { /* level 1 and "else" (levels.v:34,53-92) */
        static const u8 l1[] = {1, 2, 0, 0, 2, 3, 16, 8, 19, 0, 0,
                                96, 4, 28, 59, 10, 59, 112, 25, 10, 59, 10, 59, 112, 56, 0};
        k = (int)sizeof l1; memcpy(b, l1, sizeof l1); break;
    }
    }
    if (buf && cap >= k) memcpy(buf, b, (size_t)k);
    return k;
}

const char *zo_level_name(int level)
{
    static const char *names[6] = {"store", "fast", "normal", "high", "max", "ultra"};
    return (level >= 0 && level <= 5) ? names[level] : names[1];
}

/* compressor.v:96-145 (same scan in zpaq_test.v:446-476). */
void zo_scan_header(const uint8_t *h, int len, int *cend, int *hbegin, int *hend)
{
    if (len >= 5) {
        int n = h[4], pos = 5;
        for (int i = 0; i < n && pos < len; i++) {
            int ctype = h[pos];
            if (ctype >= 10) break;
            pos += g_compsize[ctype];
        }
        *cend = pos;
        if (pos < len && h[pos] == 0) pos++;
        *hbegin = pos;
        while (pos < len) {
            u8 op = h[pos];
            if (op == 0) break;
            pos++;
            if ((op & 7) == 7) pos += (op == 63) ? 2 : 1; /* quirk: 63 gets 2, 255 gets 1 */
        }
        *hend = pos;
    } else {
        *cend = *hbegin = *hend = len;
    }
}

/* ------------------------------------------------------------------ */
/* ZPAQL VM (zpaql.v)                                                 */
/* ------------------------------------------------------------------ */

#define ZO_VM_STEP_CAP (1u << 20) /* the reference has no cap (zpaql.v:170-174) */

struct zo_vm {
    u32 a, b, c, d;
    i32 f, pc;
    u8 *m; u32 mlen;
    u32 *h; u32 hlen;
    u32 r[256];
    u8 *header; i32 header_len;
    i32 cend, hbegin, hend;
    int step_overflow;
};

static int vm_init(zo_vm *z, const u8 *hdr, int len, int cend, int hbegin, int hend)
{
    memset(z, 0, sizeof *z);
    z->header = (u8 *)malloc(len > 0 ? (size_t)len : 1);
    if (!z->header) return -1;
    if (len > 0) memcpy(z->header, hdr, (size_t)len);
    z->header_len = len; z->cend = cend; z->hbegin = hbegin; z->hend = hend;
    /* zpaql.v:74-95 inith/initp */
    if (len >= 2) {
        int hh = z->header[0], hm = z->header[1];
        if (hh > 0 && hh < 32) {
            if (hh > 28) return -2;
            z->hlen = 1u << hh;
            z->h = (u32 *)calloc(z->hlen, 4);
            if (!z->h) return -1;
        }
        if (hm > 0 && hm < 32) {
            if (hm > 30) return -2;
            z->mlen = 1u << hm;
            z->m = (u8 *)calloc(z->mlen, 1);
            if (!z->m) return -1;
        }
        z->pc = hbegin;
    }
    return 0;
}
static void vm_destroy(zo_vm *z) { free(z->header); free(z->m); free(z->h); }

/* zpaql.v:178-211 */
static inline u8 m_get(const zo_vm *z, u32 i) { return z->mlen ? z->m[i & (z->mlen - 1)] : 0; }
static inline void m_set(zo_vm *z, u32 i, u8 v) { if (z->mlen) z->m[i & (z->mlen - 1)] = v; }
static inline u32 h_get(const zo_vm *z, u32 i) { return z->hlen ? z->h[i & (z->hlen - 1)] : 0; }
static inline void h_set(zo_vm *z, u32 i, u32 v) { if (z->hlen) z->h[i & (z->hlen - 1)] = v; }

/* operand sources 0..7 = A B C D *B *C *D N (zpaql.v:394-940 column order) */
static inline u32 vm_src(const zo_vm *z, int s, i32 operand)
{
    switch (s) {
    case 0: return z->a;
    case 1: return z->b;
    case 2: return z->c;
    case 3: return z->d;
    case 4: return m_get(z, z->b);
    case 5: return m_get(z, z->c);
    case 6: return h_get(z, z->d);
    default: return (u32)operand;
    }
}
/* targets 0..6 = A B C D *B *C *D */
static inline u32 vm_tget(const zo_vm *z, int t)
{
    return vm_src(z, t, 0);
}
static inline void vm_tset(zo_vm *z, int t, u32 v)
{
    switch (t) {
    case 0: z->a = v; break;
    case 1: z->b = v; break;
    case 2: z->c = v; break;
    case 3: z->d = v; break;
    case 4: m_set(z, z->b, (u8)v); break;
    case 5: m_set(z, z->c, (u8)v); break;
    default: h_set(z, z->d, v); break;
    }
}

/* zpaql.v:215-954.  Returns 0 to stop the run. */
static int vm_execute(zo_vm *z)
{
    if (z->pc < z->hbegin || z->pc >= z->hend) return 0;
    u8 op = z->header[z->pc];
    z->pc++;
    i32 operand = 0;
    /* zpaql.v:224-231: bounds are against header.len, not hend */
    if (oplen(op) == 2 && z->pc < z->header_len) {
        operand = z->header[z->pc];
        z->pc++;
    } else if (oplen(op) == 3 && z->pc + 1 < z->header_len) {
        operand = z->header[z->pc] + z->header[z->pc + 1] * 256;
        z->pc += 2;
    }
    if (op < 64) {
        int t = op >> 3, k = op & 7;
        if (t == 7) { /* 56..63 break the row pattern */
            switch (op) {
            case 56: return 0;                                                    /* HALT  :379 */
            case 57: return 1;                                                    /* OUT   :382 only grows a host buffer */
            case 59: z->a = (z->a + m_get(z, z->b) + 512) * 773; return 1;         /* HASH  :385 */
            case 60: h_set(z, z->d, (h_get(z, z->d) + z->a + 512) * 773); return 1; /* HASHD :388 */
            case 63: z->pc += ((operand + 128) & 255) - 127; return 1;             /* JMP   :391 */
            default: return 0;                                                    /* 58,61,62 */
            }
        }
        switch (k) {
        case 0: { /* X<>A  :236,256,275,294,313,336,359 */
            if (t == 0) return 1;
            u32 tmp = vm_tget(z, t);
            vm_tset(z, t, z->a);
            z->a = tmp;
            return 1;
        }
        case 1: vm_tset(z, t, vm_tget(z, t) + 1); return 1; /* X++ (u8 wrap for *B,*C via the store) */
        case 2: vm_tset(z, t, vm_tget(z, t) - 1); return 1; /* X-- */
        case 3: vm_tset(z, t, ~vm_tget(z, t)); return 1;    /* X!  */
        case 4: vm_tset(z, t, 0); return 1;                 /* X=0 */
        case 7:
            if (t <= 3) { vm_tset(z, t, z->r[operand & 255]); return 1; } /* X=R N :252,271,290,309 */
            if (t == 4) { if (z->f != 0) z->pc += ((operand + 128) & 255) - 127; return 1; } /* JT :330 */
            if (t == 5) { if (z->f == 0) z->pc += ((operand + 128) & 255) - 127; return 1; } /* JF :353 */
            z->r[operand & 255] = z->a; return 1;                                          /* R=A :376 */
        default: return 0; /* 5,6,13,14,... undefined :947-950 */
        }
    }
    if (op < 120) { /* assignments :394-562 */
        int t = (op - 64) >> 3, s = op & 7;
        vm_tset(z, t, vm_src(z, s, operand));
        return 1;
    }
    if (op < 128) return 0;
    if (op < 216) { /* a op= src :564-867 */
        u32 v = vm_src(z, op & 7, operand);
        switch ((op - 128) >> 3) {
        case 0: z->a += v; break;
        case 1: z->a -= v; break;
        case 2: z->a *= v; break;
        case 3: if (v != 0) z->a /= v; break;
        case 4: if (v != 0) z->a %= v; break;
        case 5: z->a &= v; break;
        case 6: z->a &= ~v; break;
        case 7: z->a |= v; break;
        case 8: z->a ^= v; break;
        case 9: z->a <<= (v & 31); break;
        default: z->a >>= (v & 31); break;
        }
        return 1;
    }
    if (op < 240) { /* compares :869-940 */
        u32 v = vm_src(z, op & 7, operand);
        switch ((op - 216) >> 3) {
        case 0: z->f = z->a == v; break;
        case 1: z->f = z->a < v; break;
        default: z->f = z->a > v; break;
        }
        return 1;
    }
    if (op == 255) { /* LJ :941-946 */
        if (z->pc < 2) return 0;
        z->pc = z->hbegin + z->header[z->pc - 2] + z->header[z->pc - 1] * 256;
        if (z->pc >= z->hend) return 0;
        return 1;
    }
    return 0; /* 240..254 */
}

/* zpaql.v:167-175 */
static void vm_run(zo_vm *z, u32 input)
{
    z->a = input;
    z->pc = z->hbegin;
    u32 steps = 0;
    while (z->pc < z->hend && z->pc >= z->hbegin) {
        if (!vm_execute(z)) break;
        if (++steps >= ZO_VM_STEP_CAP) { z->step_overflow = 1; break; }
    }
}

zo_vm *zo_vm_new(const uint8_t *hdr, int len, int cend, int hbegin, int hend)
{
    zo_vm *z = (zo_vm *)malloc(sizeof *z);
    if (!z) return NULL;
    if (vm_init(z, hdr, len, cend, hbegin, hend) != 0) { vm_destroy(z); free(z); return NULL; }
    return z;
}
void zo_vm_free(zo_vm *z) { if (z) { vm_destroy(z); free(z); } }
void zo_vm_run(zo_vm *z, uint32_t input) { vm_run(z, input); }
uint32_t zo_vm_reg(const zo_vm *z, int w)
{
    switch (w) { case 0: return z->a; case 1: return z->b; case 2: return z->c; case 3: return z->d;
                 case 4: return (u32)z->f; default: return (u32)z->pc; }
}
void zo_vm_set_reg(zo_vm *z, int w, uint32_t v)
{
    switch (w) { case 0: z->a = v; break; case 1: z->b = v; break; case 2: z->c = v; break;
                 case 3: z->d = v; break; case 4: z->f = (i32)v; break; default: z->pc = (i32)v; }
}
uint32_t zo_vm_h(const zo_vm *z, uint32_t i) { return h_get(z, i); }
uint32_t zo_vm_m(const zo_vm *z, uint32_t i) { return m_get(z, i); }
uint32_t zo_vm_r(const zo_vm *z, int i) { return z->r[i & 255]; }
int zo_vm_hlen(const zo_vm *z) { return (int)z->hlen; }
int zo_vm_mlen(const zo_vm *z) { return (int)z->mlen; }

/* ------------------------------------------------------------------ */
/* Predictor (predictor.v:239-833)                                    */
/* ------------------------------------------------------------------ */

typedef struct {
    i32 ctype;
    u32 *cm; u32 cm_len;
    u8 *ht; u32 ht_len;
    u16 *a16; u32 a16_len;
    i32 a, b, c;
    u32 cxt;
    i32 limit;
} comp_t;

struct zo_codec {
    zo_vm z;
    u32 c8, hmap4;
    i32 n;
    u32 *h;
    i32 *p;
    comp_t *comp;
    size_t state_bytes;
};

static void *zalloc(zo_codec *c, size_t n, size_t sz)
{
    c->state_bytes += n * sz;
    return calloc(n ? n : 1, sz);
}

/* predictor.v:292-470 */
static int pred_init(zo_codec *pc)
{
    zo_vm *z = &pc->z;
    pc->c8 = 1; pc->hmap4 = 1; pc->n = 0;
    if (z->header_len < 5) return 0;
    int n = z->header[4];
    if (n == 0) return 0;
    pc->n = n;
    pc->comp = (comp_t *)calloc((size_t)n, sizeof(comp_t));
    pc->p = (i32 *)calloc((size_t)n, sizeof(i32));
    pc->h = (u32 *)calloc((size_t)n, sizeof(u32));
    if (!pc->comp || !pc->p || !pc->h) return -1;
    const u8 *hd = z->header;
    int hl = z->header_len;
    int cp = 5;
    for (int i = 0; i < n && cp < z->cend; i++) {
        comp_t *cr = &pc->comp[i];
        int ctype = hd[cp];
        cr->ctype = ctype;
        if (ctype >= 1 && ctype <= 9 && cp + g_compsize[ctype] > hl) return -3; /* V: index panic */
        switch (ctype) {
        case 1: cr->a = hd[cp + 1]; break;
        case 2: {
            cr->a = hd[cp + 1];
            cr->limit = hd[cp + 2] * 4;
            if (cr->a > 28) return -2;
            cr->cm_len = 1u << cr->a;
            cr->cm = (u32 *)zalloc(pc, cr->cm_len, 4);
            if (!cr->cm) return -1;
            for (u32 j = 0; j < cr->cm_len; j++) cr->cm[j] = 0x80000000u;
            break;
        }
        case 3: {
            cr->a = hd[cp + 1];
            if (cr->a > 24) return -2;
            cr->ht_len = 16u << (cr->a + 2);
            cr->cm_len = 256;
            cr->cm = (u32 *)zalloc(pc, 256, 4);
            cr->ht = (u8 *)zalloc(pc, cr->ht_len, 1);
            if (!cr->cm || !cr->ht) return -1;
            for (int j = 0; j < 256; j++) cr->cm[j] = (u32)cminit(j);
            break;
        }
        case 4: {
            cr->a = hd[cp + 1];
            cr->b = hd[cp + 2];
            if (cr->a > 28 || cr->b > 30) return -2;
            cr->cm_len = 1u << cr->a;
            cr->ht_len = 1u << cr->b;
            cr->cm = (u32 *)zalloc(pc, cr->cm_len, 4);
            cr->ht = (u8 *)zalloc(pc, cr->ht_len, 1);
            if (!cr->cm || !cr->ht) return -1;
            cr->limit = 0; cr->c = 0; cr->cxt = 0;
            break;
        }
        case 5: cr->a = hd[cp + 1]; cr->b = hd[cp + 2]; cr->c = hd[cp + 3]; break;
        case 6: {
            cr->a = hd[cp + 1];
            if (cr->a > 28) return -2;
            u32 size = 1u << cr->a;
            cr->b = hd[cp + 2];
            cr->c = (i32)size;
            cr->a16_len = size;
            cr->a16 = (u16 *)zalloc(pc, size, 2);
            cr->cm_len = 4;
            cr->cm = (u32 *)zalloc(pc, 4, 4);
            if (!cr->a16 || !cr->cm) return -1;
            for (u32 j = 0; j < size; j++) cr->a16[j] = 32768;
            cr->cm[0] = hd[cp + 2]; cr->cm[1] = hd[cp + 3];
            cr->cm[2] = hd[cp + 4]; cr->cm[3] = hd[cp + 5];
            break;
        }
        case 7: {
            cr->a = hd[cp + 1];
            if (cr->a > 24) return -2;
            u32 size = 1u << cr->a;
            int m = hd[cp + 3];
            if (m == 0) return -4; /* predictor.v:426 divides by m */
            cr->b = hd[cp + 2];
            cr->c = (i32)size;
            cr->limit = m;
            cr->ht_len = 2;
            cr->ht = (u8 *)zalloc(pc, 2, 1);
            if (!cr->ht) return -1;
            cr->ht[0] = hd[cp + 4]; cr->ht[1] = hd[cp + 5];
            cr->cm_len = size * (u32)m;
            cr->cm = (u32 *)zalloc(pc, cr->cm_len, 4);
            if (!cr->cm) return -1;
            for (u32 k = 0; k < cr->cm_len; k++) cr->cm[k] = (u32)(65536 / m) << 8;
            break;
        }
        case 8: {
            cr->a = hd[cp + 1];
            cr->b = hd[cp + 2];
            if (cr->a > 24) return -2;
            cr->ht_len = 16u << (cr->a + 2);
            cr->ht = (u8 *)zalloc(pc, cr->ht_len, 1);
            cr->cm_len = 512;
            cr->cm = (u32 *)zalloc(pc, 512, 4);
            if (!cr->ht || !cr->cm) return -1;
            for (int k = 0; k < 256; k++) {
                cr->cm[k * 2] = 1u << 15;
                i32 st_init = cminit(k);
                cr->cm[k * 2 + 1] = (u32)clamp512k(wmul(stretch(st_init >> 8), 1024));
            }
            break;
        }
        case 9: {
            cr->a = hd[cp + 1];
            cr->b = hd[cp + 2];
            if (cr->a > 24) return -2;
            u32 size = 1u << cr->a;
            cr->cm_len = size * 32;
            cr->cm = (u32 *)zalloc(pc, cr->cm_len, 4);
            if (!cr->cm) return -1;
            cr->limit = hd[cp + 4] * 4;
            i32 start = hd[cp + 3];
            for (u32 k = 0; k < cr->cm_len; k++) {
                i32 q = (i32)(k & 31) * 64 - 992;
                cr->cm[k] = ((u32)squash(q) << 17) | (u32)start;
            }
            break;
        }
        default: break;
        }
        cp += (ctype >= 1 && ctype <= 9) ? g_compsize[ctype] : 1; /* :465-467 */
    }
    return 0;
}

/* predictor.v:495-532 */
static i32 find_ht(u8 *ht, u32 ht_len, int sizebits, u32 cxt)
{
    i32 chk = (i32)((cxt >> sizebits) & 255);
    i32 h0 = (i32)((cxt * 16) & (ht_len - 16));
    if (ht[h0] == (u8)chk) return h0;
    i32 h1 = h0 ^ 16;
    if (ht[h1] == (u8)chk) return h1;
    i32 h2 = h0 ^ 32;
    if (ht[h2] == (u8)chk) return h2;
    i32 r;
    if (ht[h0 + 1] <= ht[h1 + 1] && ht[h0 + 1] <= ht[h2 + 1]) r = h0;
    else if (ht[h1 + 1] < ht[h2 + 1]) r = h1;
    else r = h2;
    memset(ht + r, 0, 16);
    ht[r] = (u8)chk;
    return r;
}

/* predictor.v:536-668 */
static i32 pred_predict(zo_codec *pc)
{
    i32 n = pc->n;
    if (n == 0) return 16384;
    i32 *p = pc->p;
    for (i32 i = 0; i < n; i++) {
        comp_t *cr = &pc->comp[i];
        switch (cr->ctype) {
        case 1: p[i] = (cr->a - 128) * 16; break;
        case 2: {
            cr->cxt = pc->h[i] ^ pc->hmap4;
            i32 idx = (i32)cr->cxt & (i32)(cr->cm_len - 1);
            p[i] = stretch((i32)(cr->cm[idx] >> 17));
            break;
        }
        case 3: {
            if (pc->c8 == 1 || (pc->c8 & 0xf0) == 16)
                cr->c = find_ht(cr->ht, cr->ht_len, cr->a + 2, pc->h[i] + 16 * pc->c8);
            cr->cxt = cr->ht[cr->c + (i32)(pc->hmap4 & 15)];
            p[i] = stretch((i32)(cr->cm[cr->cxt] >> 8));
            break;
        }
        case 4: {
            if (cr->a == 0) p[i] = 0;
            else {
                i32 idx = wsub(cr->limit, cr->b) & (i32)(cr->ht_len - 1);
                cr->c = (cr->ht[idx] >> (7 - (i32)cr->cxt)) & 1;
                i32 weight = g_dt2k[cr->a & 255];
                p[i] = stretch((weight * (cr->c * -2 + 1)) & 32767);
            }
            break;
        }
        case 5: {
            i32 j = cr->a, k = cr->b, wt = cr->c;
            p[i] = (j < n && k < n) ? sar(wadd(wmul(p[j], wt), wmul(p[k], 256 - wt)), 8) : 0;
            break;
        }
        case 6: {
            i32 j = (i32)cr->cm[0], k = (i32)cr->cm[1], mask = (i32)cr->cm[3];
            cr->cxt = (pc->h[i] + (pc->c8 & (u32)mask)) & (u32)(cr->c - 1);
            i32 w = cr->a16[cr->cxt];
            p[i] = (j < n && k < n)
                       ? clamp2k(sar(wadd(wmul(w, p[j]), wmul(65536 - w, p[k])), 16))
                       : 0;
            break;
        }
        case 7: {
            i32 j = cr->b, m = cr->limit, mask = cr->ht[1];
            cr->cxt = (u32)(wadd((i32)pc->h[i], (i32)pc->c8 & mask) & (cr->c - 1));
            i32 idx = (i32)cr->cxt * m;
            i32 sum = 0;
            for (i32 l = 0; l < m && (j + l) < n; l++) {
                i32 wt = sar((i32)cr->cm[idx + l], 8);
                sum = wadd(sum, wmul(wt, p[j + l]));
            }
            p[i] = clamp2k(sar(sum, 8));
            break;
        }
        case 8: {
            if (pc->c8 == 1 || (pc->c8 & 0xf0) == 16)
                cr->c = find_ht(cr->ht, cr->ht_len, cr->a + 2, pc->h[i] + 16 * pc->c8);
            cr->cxt = cr->ht[cr->c + (i32)(pc->hmap4 & 15)];
            i32 wt0 = (i32)cr->cm[cr->cxt * 2], wt1 = (i32)cr->cm[cr->cxt * 2 + 1];
            i32 j = cr->b;
            if (j < n) p[i] = clamp2k(sar(wadd(wmul(wt0, p[j]), wmul(wt1, 64)), 16));
            else p[i] = clamp2k(sar(wt1, 10));
            break;
        }
        case 9: {
            i32 j = cr->b;
            cr->cxt = (pc->h[i] + pc->c8) * 32;
            i32 pq = 992;
            if (j < n) pq = wadd(p[j], 992);
            if (pq < 0) pq = 0;
            if (pq > 1983) pq = 1983;
            i32 wt = pq & 63;
            pq >>= 6;
            i32 idx = wadd((i32)cr->cxt, pq);
            i32 idx2 = wadd(idx, 1);
            if (idx >= 0 && idx2 < (i32)cr->cm_len) {
                i32 p1 = (i32)(cr->cm[idx] >> 10), p2 = (i32)(cr->cm[idx2] >> 10);
                p[i] = stretch(sar(wadd(wmul(p1, 64 - wt), wmul(p2, wt)), 13));
            } else p[i] = 0;
            cr->cxt = (u32)idx + (u32)(wt >> 5);
            break;
        }
        default: p[i] = 0; break;
        }
    }
    return squash(p[n - 1]);
}

/* predictor.v:672-824 */
static void pred_update(zo_codec *pc, i32 y)
{
    i32 n = pc->n;
    i32 *p = pc->p;
    for (i32 i = 0; i < n; i++) {
        comp_t *cr = &pc->comp[i];
        switch (cr->ctype) {
        case 2: {
            i32 idx = (i32)cr->cxt & (i32)(cr->cm_len - 1);
            u32 pn = cr->cm[idx];
            i32 count = (i32)(pn & 0x3ff);
            i32 err = y * 32767 - (i32)(pn >> 17);
            i32 dt_val = g_dt[count];
            i32 upd = wmul(err, dt_val) & -1024;
            i32 inc = count < cr->limit ? 1 : 0;
            cr->cm[idx] = (u32)wadd(wadd((i32)pn, upd), inc);
            break;
        }
        case 3: {
            i32 slot = cr->c + (i32)(pc->hmap4 & 15);
            cr->ht[slot] = (u8)ns_next(cr->ht[slot], y);
            u32 v = cr->cm[cr->cxt];
            cr->cm[cr->cxt] = (u32)wadd((i32)v, sar(y * 32767 - (i32)(v >> 8), 2));
            break;
        }
        case 4: {
            if (cr->c != y) cr->a = 0;
            i32 mask = (i32)(cr->ht_len - 1);
            i32 idx = cr->limit & mask;
            cr->ht[idx] = (u8)(((u32)cr->ht[idx] << 1) | (u32)y);
            cr->cxt++;
            if (cr->cxt >= 8) {
                cr->cxt = 0;
                cr->limit = wadd(cr->limit, 1);
                cr->limit &= mask;
                u32 hi = pc->h[i];
                i32 cmi = (i32)hi & (i32)(cr->cm_len - 1);
                if (cr->a == 0) {
                    cr->b = wsub(cr->limit, (i32)cr->cm[cmi]);
                    if ((cr->b & mask) != 0) { /* Go precedence (Q4), predictor.v:725 */
                        while (cr->a < 255) {
                            i32 i1 = wsub(wsub(cr->limit, cr->a), 1) & mask;
                            i32 i2 = wsub(wsub(wsub(cr->limit, cr->a), cr->b), 1) & mask;
                            if (cr->ht[i1] != cr->ht[i2]) break;
                            cr->a++;
                        }
                    }
                } else if (cr->a < 255) cr->a++;
                cr->cm[cmi] = (u32)cr->limit;
            }
            break;
        }
        case 6: {
            i32 j = (i32)cr->cm[0], k = (i32)cr->cm[1], rate = (i32)cr->cm[2];
            i32 err = sar(wmul(y * 32767 - squash(p[i]), rate), 5);
            if (j < n && k < n) {
                i32 w = cr->a16[cr->cxt];
                w = wadd(w, sar(wadd(wmul(err, wsub(p[j], p[k])), 1 << 12), 13));
                if (w < 0) w = 0;
                if (w > 65535) w = 65535;
                cr->a16[cr->cxt] = (u16)w;
            }
            break;
        }
        case 7: {
            i32 jj = cr->b, m = cr->limit, rate = cr->ht[0];
            i32 err = sar(wmul(y * 32767 - squash(p[i]), rate), 4);
            i32 idx = (i32)cr->cxt * m;
            for (i32 l = 0; l < m && (jj + l) < n; l++) {
                i32 wt = clamp512k(wadd((i32)cr->cm[idx + l],
                                        sar(wadd(wmul(err, p[jj + l]), 1 << 12), 13)));
                cr->cm[idx + l] = (u32)wt;
            }
            break;
        }
        case 8: {
            i32 j = cr->b;
            i32 err = y * 32767 - squash(p[i]);
            if (j < n) {
                i32 wt0 = clamp512k(wadd((i32)cr->cm[cr->cxt * 2],
                                         sar(wadd(wmul(err, p[j]), 1 << 12), 13)));
                i32 wt1 = clamp512k(wadd((i32)cr->cm[cr->cxt * 2 + 1], sar(err + 16, 5)));
                cr->cm[cr->cxt * 2] = (u32)wt0;
                cr->cm[cr->cxt * 2 + 1] = (u32)wt1;
            }
            cr->ht[cr->c + (i32)(pc->hmap4 & 15)] = (u8)ns_next((i32)cr->cxt, y);
            break;
        }
        case 9: {
            i32 idx = (i32)cr->cxt & (i32)(cr->cm_len - 1);
            u32 v = cr->cm[idx];
            i32 err = y * 32767 - (i32)(v >> 17);
            i32 count = (i32)v & 1023;
            if (count < cr->limit)
                v = (u32)wadd(wadd((i32)v, sar(wadd(wmul(err, cr->limit - count), 1 << 12), 13)), 1);
            cr->cm[idx] = v;
            break;
        }
        default: break; /* CONST, AVG, unknown: no update */
        }
    }
    /* predictor.v:807-823 */
    pc->c8 = (pc->c8 << 1) | (u32)y;
    if (pc->c8 >= 256) {
        vm_run(&pc->z, pc->c8 - 256);
        for (i32 i = 0; i < n && (u32)i < pc->z.hlen; i++) pc->h[i] = pc->z.h[i];
        pc->hmap4 = 1;
        pc->c8 = 1;
    } else if (pc->c8 >= 16 && pc->c8 < 32) {
        pc->hmap4 = ((pc->hmap4 & 0xf) << 5) | ((u32)y << 4) | 1;
    } else {
        pc->hmap4 = (pc->hmap4 & 0x1f0) | (((pc->hmap4 & 0xf) * 2 + (u32)y) & 0xf);
    }
}

/* predictor.v:827-833 */
static void pred_reset(zo_codec *pc)
{
    pc->c8 = 1; pc->hmap4 = 1;
    for (i32 i = 0; i < pc->n; i++) pc->h[i] = 0;
}

zo_codec *zo_codec_new(const uint8_t *hdr, int len, int cend, int hbegin, int hend, int *err)
{
    tables_ready();
    zo_codec *pc = (zo_codec *)calloc(1, sizeof *pc);
    int rc = pc ? 0 : -1;
    if (pc) rc = vm_init(&pc->z, hdr, len, cend, hbegin, hend);
    if (rc == 0) {
        pc->state_bytes = (size_t)pc->z.mlen + 4 * (size_t)pc->z.hlen;
        rc = pred_init(pc);
    }
    if (err) *err = rc;
    if (rc != 0) { zo_codec_free(pc); return NULL; }
    return pc;
}

zo_codec *zo_codec_new_level(int level, int *err)
{
    u8 hdr[128];
    int len = zo_level_header(level, hdr, sizeof hdr), cend, hbegin, hend;
    zo_scan_header(hdr, len, &cend, &hbegin, &hend);
    return zo_codec_new(hdr, len, cend, hbegin, hend, err);
}

void zo_codec_free(zo_codec *pc)
{
    if (!pc) return;
    for (i32 i = 0; i < pc->n && pc->comp; i++) {
        free(pc->comp[i].cm); free(pc->comp[i].ht); free(pc->comp[i].a16);
    }
    free(pc->comp); free(pc->p); free(pc->h);
    vm_destroy(&pc->z);
    free(pc);
}
int zo_codec_ncomp(const zo_codec *pc) { return pc->n; }
size_t zo_codec_state_bytes(const zo_codec *pc) { return pc->state_bytes; }
void zo_pred_reset(zo_codec *pc) { pred_reset(pc); }
int zo_pred_predict(zo_codec *pc) { return pred_predict(pc); }
void zo_pred_update(zo_codec *pc, int y) { pred_update(pc, y); }
int zo_pred_p(const zo_codec *pc, int i) { return (i >= 0 && i < pc->n) ? pc->p[i] : 0; }
uint32_t zo_pred_h(const zo_codec *pc, int i) { return (i >= 0 && i < pc->n) ? pc->h[i] : 0; }
uint32_t zo_pred_c8(const zo_codec *pc) { return pc->c8; }
uint32_t zo_pred_hmap4(const zo_codec *pc) { return pc->hmap4; }

/* ------------------------------------------------------------------ */
/* Arithmetic coder (encoder.v, decoder.v)                            */
/* ------------------------------------------------------------------ */

typedef struct {
    u32 low, high;
    u8 *out; size_t cap, pos;
    int overflow;
} enc_t;

static inline void enc_put(enc_t *e, u32 b)
{
    if (e->pos < e->cap) e->out[e->pos] = (u8)b;
    else e->overflow = 1;
    e->pos++;
}

/* encoder.v:48-89 */
static inline void enc_encode(enc_t *e, i32 y, i32 p)
{
    i32 pr = p;
    if (pr < 0) pr = 0;
    if (pr > 65535) pr = 65535;
    u32 range = e->high - e->low;
    u32 mid = e->low + (u32)(((u64)range * (u64)(u32)pr) >> 16);
    if (y != 0) e->high = mid;
    else e->low = mid + 1;
    while ((e->high ^ e->low) < 0x1000000u) {
        enc_put(e, e->high >> 24);
        e->low <<= 8;
        e->high = (e->high << 8) | 0xFF;
        if (e->low == 0) e->low = 1;
    }
}

typedef struct {
    zo_trace_bit *t; size_t n, cap;
} tracer_t;

/* encoder.v:93-120 */
static void enc_compress(enc_t *e, zo_codec *pc, i32 c, tracer_t *tr)
{
    if (c == -1) { enc_encode(e, 1, 0); return; }
    enc_encode(e, 0, 0);
    for (int i = 7; i >= 0; i--) {
        i32 y = (c >> i) & 1;
        i32 p = pred_predict(pc);
        enc_encode(e, y, p * 2 + 1);
        pred_update(pc, y);
        if (tr && tr->n < tr->cap) {
            zo_trace_bit *b = &tr->t[tr->n++];
            b->p = p; b->y = y; b->low = e->low; b->high = e->high;
        }
    }
}

int64_t zo_encode_segment(zo_codec *pc, const uint8_t *in, size_t n, unsigned flags,
                          uint8_t *out, size_t cap, zo_trace_bit *trace, size_t ntrace)
{
    enc_t e = {1, 0xFFFFFFFFu, out, cap, 0, 0}; /* encoder.v:19-34 */
    tracer_t tr = {trace, 0, trace ? ntrace : 0};
    pred_reset(pc);                                     /* compressor.v:245 */
    if (flags & ZO_FLAG_PP) enc_compress(&e, pc, 0, &tr); /* compressor.v:271-274 */
    for (size_t i = 0; i < n; i++) enc_compress(&e, pc, in[i], &tr);
    enc_compress(&e, pc, -1, &tr);                      /* compressor.v:375 */
    /* encoder.v:130-139 flush */
    enc_put(&e, e.high >> 24); enc_put(&e, e.high >> 16);
    enc_put(&e, e.high >> 8); enc_put(&e, e.high);
    return e.overflow ? -1 : (int64_t)e.pos;
}

typedef struct {
    u32 low, high, code;
    const u8 *in; size_t n, pos;
} dec_t;

/* decoder.v:57-62 + io.v:55-62: Reader.get returns -1 at EOF */
static inline i32 dec_get(dec_t *d)
{
    if (d->pos >= d->n) return -1;
    return d->in[d->pos++];
}
static inline void dec_shift(dec_t *d)
{
    i32 c = dec_get(d);
    d->code = c < 0 ? (d->code << 8) : ((d->code << 8) | (u32)c);
}

/* decoder.v:73-118 */
static inline i32 dec_decode(dec_t *d, i32 p)
{
    i32 pr = p;
    if (pr < 0) pr = 0;
    if (pr > 65535) pr = 65535;
    u32 range = d->high - d->low;
    u32 mid = d->low + (u32)(((u64)range * (u64)(u32)pr) >> 16);
    i32 y;
    if (d->code <= mid) { y = 1; d->high = mid; }
    else { y = 0; d->low = mid + 1; }
    while ((d->high ^ d->low) < 0x1000000u) {
        d->low <<= 8;
        d->high = (d->high << 8) | 0xFF;
        if (d->low == 0) d->low = 1;
        dec_shift(d);
    }
    return y;
}

static int64_t decode_impl(zo_codec *pc, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                           size_t *consumed, u32 *final_code, zo_trace_bit *trace, size_t ntrace)
{
    dec_t d = {1, 0xFFFFFFFFu, 0, in, n, 0};
    size_t nt = 0, pos = 0;
    int overflow = 0;
    pred_reset(pc);                              /* decompressor.v:415 */
    for (int i = 0; i < 4; i++) dec_shift(&d);   /* decoder.v:38-46 */
    for (;;) {                                   /* decoder.v:122-145 */
        if (dec_decode(&d, 0) != 0) break;
        u32 c = 1;
        while (c < 256) {
            i32 p = pred_predict(pc);
            i32 y = dec_decode(&d, p * 2 + 1);
            pred_update(pc, y);
            c = (c << 1) | (u32)y;
            if (trace && nt < ntrace) {
                zo_trace_bit *b = &trace[nt++];
                b->p = p; b->y = y; b->low = d.low; b->high = d.high;
            }
        }
        if (pos < cap) out[pos] = (u8)(c - 256);
        else overflow = 1;
        pos++;
        if (overflow && pos > cap + (1u << 20)) break; /* runaway guard on garbage input */
    }
    if (consumed) *consumed = d.pos;
    if (final_code) *final_code = d.code;
    return overflow ? -1 : (int64_t)pos;
}

int64_t zo_decode_segment(zo_codec *pc, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                          size_t *consumed, zo_trace_bit *trace, size_t ntrace)
{
    return decode_impl(pc, in, n, out, cap, consumed, NULL, trace, ntrace);
}

/* ------------------------------------------------------------------ */
/* Batch helpers (CPU baseline)                                       */
/* ------------------------------------------------------------------ */

typedef struct {
    const u8 *hdr; int len, cend, hbegin, hend;
    int nblocks, tid, nthreads, decode;
    const u8 *in; const u64 *in_off; unsigned flags;
    u8 *out; const u64 *out_off; int64_t *out_len;
    int err;
} batch_t;

static void *batch_worker(void *arg)
{
    batch_t *b = (batch_t *)arg;
    int lo = (int)((int64_t)b->nblocks * b->tid / b->nthreads);
    int hi = (int)((int64_t)b->nblocks * (b->tid + 1) / b->nthreads);
    for (int k = lo; k < hi; k++) {
        int err = 0;
        zo_codec *pc = zo_codec_new(b->hdr, b->len, b->cend, b->hbegin, b->hend, &err);
        if (!pc) { b->err = err ? err : -1; b->out_len[k] = -1; continue; }
        const u8 *src = b->in + b->in_off[k];
        size_t n = (size_t)(b->in_off[k + 1] - b->in_off[k]);
        u8 *dst = b->out + b->out_off[k];
        size_t cap = (size_t)(b->out_off[k + 1] - b->out_off[k]);
        if (b->decode) b->out_len[k] = zo_decode_segment(pc, src, n, dst, cap, NULL, NULL, 0);
        else b->out_len[k] = zo_encode_segment(pc, src, n, b->flags, dst, cap, NULL, 0);
        zo_codec_free(pc);
    }
    return NULL;
}

static int run_batch(batch_t proto, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    tables_ready();
    batch_t *w = (batch_t *)calloc((size_t)nthreads, sizeof *w);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    if (!w || !th) { free(w); free(th); return -1; }
    for (int t = 0; t < nthreads; t++) {
        w[t] = proto; w[t].tid = t; w[t].nthreads = nthreads;
        if (t > 0) pthread_create(&th[t], NULL, batch_worker, &w[t]);
    }
    batch_worker(&w[0]);
    int err = w[0].err;
    for (int t = 1; t < nthreads; t++) { pthread_join(th[t], NULL); if (w[t].err) err = w[t].err; }
    free(w); free(th);
    return err;
}

int zo_encode_blocks(const uint8_t *hdr, int len, int cend, int hbegin, int hend,
                     int nblocks, const uint8_t *in, const uint64_t *in_off, unsigned flags,
                     uint8_t *out, const uint64_t *out_off, int64_t *out_len, int nthreads)
{
    batch_t b = {hdr, len, cend, hbegin, hend, nblocks, 0, 1, 0, in, in_off, flags, out, out_off, out_len, 0};
    return run_batch(b, nthreads);
}

int zo_decode_blocks(const uint8_t *hdr, int len, int cend, int hbegin, int hend,
                     int nblocks, const uint8_t *in, const uint64_t *in_off,
                     uint8_t *out, const uint64_t *out_off, int64_t *out_len, int nthreads)
{
    batch_t b = {hdr, len, cend, hbegin, hend, nblocks, 0, 1, 1, in, in_off, 0, out, out_off, out_len, 0};
    return run_batch(b, nthreads);
}

/* ------------------------------------------------------------------ */
/* SHA-1 (sha1.v:6-146) and block/segment framing                      */
/* ------------------------------------------------------------------ */

static inline u32 rotl(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

static void sha1_block(u32 h[5], const u8 *buf)
{
    u32 w[80];
    for (int i = 0; i < 16; i++)
        w[i] = (u32)buf[i * 4] << 24 | (u32)buf[i * 4 + 1] << 16 | (u32)buf[i * 4 + 2] << 8 | buf[i * 4 + 3];
    for (int i = 16; i < 80; i++) w[i] = rotl(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int i = 0; i < 80; i++) {
        u32 f, k;
        if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999; }
        else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1; }
        else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
        else { f = b ^ c ^ d; k = 0xCA62C1D6; }
        u32 t = rotl(a, 5) + f + e + k + w[i];
        e = d; d = c; c = rotl(b, 30); b = a; a = t;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

void zo_sha1(const uint8_t *data, size_t n, uint8_t out[20])
{
    u32 h[5] = {0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0};
    size_t i = 0;
    for (; i + 64 <= n; i += 64) sha1_block(h, data + i);
    u8 buf[128];
    size_t rem = n - i, tot = rem < 56 ? 64 : 128;
    memset(buf, 0, sizeof buf);
    if (rem) memcpy(buf, data + i, rem);
    buf[rem] = 0x80;
    u64 bits = (u64)n * 8;
    for (int k = 0; k < 8; k++) buf[tot - 1 - k] = (u8)(bits >> (8 * k));
    sha1_block(h, buf);
    if (tot == 128) sha1_block(h, buf + 64);
    for (int k = 0; k < 5; k++) {
        out[k * 4] = (u8)(h[k] >> 24); out[k * 4 + 1] = (u8)(h[k] >> 16);
        out[k * 4 + 2] = (u8)(h[k] >> 8); out[k * 4 + 3] = (u8)h[k];
    }
}

typedef struct { u8 *p; size_t cap, pos; int ovf; } wr_t;
static inline void wput(wr_t *w, u32 b)
{
    if (w->pos < w->cap) w->p[w->pos] = (u8)b; else w->ovf = 1;
    w->pos++;
}

/* compressor.v:12-13 */
static const u8 g_locator[13] = {0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3};

int64_t zo_compress_archive(int level, const char *filename, const char *comment,
                            const uint8_t *data, size_t n, int called_compress,
                            uint8_t *out, size_t cap)
{
    tables_ready();
    u8 hdr[128];
    int len = zo_level_header(level, hdr, sizeof hdr), cend, hbegin, hend, err = 0;
    zo_scan_header(hdr, len, &cend, &hbegin, &hend);
    wr_t w = {out, cap, 0, 0};
    /* start_block: compressor.v:150-181 */
    for (int i = 0; i < 13; i++) wput(&w, g_locator[i]);
    wput(&w, 0x7a); wput(&w, 0x50); wput(&w, 0x51);
    wput(&w, (len >= 5 && hdr[4] != 0) ? 1 : 2);
    wput(&w, 1);
    int hsize = (cend + 1) + (hend - hbegin + 1);
    wput(&w, hsize & 0xFF); wput(&w, (hsize >> 8) & 0xFF);
    for (int i = 0; i <= cend && i < len; i++) wput(&w, hdr[i]);
    for (int i = hbegin; i <= hend && i < len; i++) wput(&w, hdr[i]);
    /* start_segment: compressor.v:217-235 */
    wput(&w, 1);
    for (const char *s = filename; *s; s++) wput(&w, (u8)*s);
    wput(&w, 0);
    for (const char *s = comment; *s; s++) wput(&w, (u8)*s);
    wput(&w, 0);
    wput(&w, 0);
    int modeled = (len >= 5 && hdr[4] != 0);
    if (level == 0 || !modeled) {
        /* store mode: compressor.v:297-354,364-372 */
        size_t i = 0;
        u8 first = called_compress ? 1 : 0;
        size_t total = n + first, done = 0;
        while (done < total) {
            size_t chunk = total - done < 65536 ? total - done : 65536;
            wput(&w, (u32)(chunk >> 24)); wput(&w, (u32)(chunk >> 16));
            wput(&w, (u32)(chunk >> 8)); wput(&w, (u32)chunk);
            size_t k = 0;
            if (done == 0 && first) { wput(&w, 0); k = 1; }
            for (; k < chunk; k++) wput(&w, data[i++]);
            done += chunk;
        }
        wput(&w, 0); wput(&w, 0); wput(&w, 0); wput(&w, 0);
    } else {
        zo_codec *pc = zo_codec_new(hdr, len, cend, hbegin, hend, &err);
        if (!pc) return -3;
        size_t room = w.pos < cap ? cap - w.pos : 0;
        int64_t k = zo_encode_segment(pc, data, called_compress ? n : 0,
                                      called_compress ? ZO_FLAG_PP : 0, out + (room ? w.pos : 0),
                                      room, NULL, 0);
        zo_codec_free(pc);
        if (k < 0) return -1;
        w.pos += (size_t)k;
        wput(&w, 0); wput(&w, 0); wput(&w, 0); wput(&w, 0); /* compressor.v:382-385 */
    }
    /* compressor.v:389-395 */
    u8 sha[20];
    zo_sha1(data, called_compress ? n : 0, sha);
    wput(&w, 253);
    for (int i = 0; i < 20; i++) wput(&w, sha[i]);
    wput(&w, 0xFF); /* end_block: compressor.v:407-410 */
    return w.ovf ? -1 : (int64_t)w.pos;
}

int64_t zo_decompress_archive(const uint8_t *arc, size_t n, size_t *ppos, char *filename,
                              size_t fncap, char *comment, size_t cmcap, uint8_t *out,
                              size_t cap, int *sha_ok)
{
    tables_ready();
    size_t pos = *ppos;
#define GET() (pos < n ? (int)arc[pos++] : -1)
    /* find_block: decompressor.v:227-254 */
    u32 h1 = 0x3D49B113, h2 = 0x29EB7F93, h3 = 0x2614BE13, h4 = 0x3828EB13;
    for (;;) {
        int c = GET();
        if (c < 0) { *ppos = pos; return -1; }
        h1 = h1 * 12 + (u32)c; h2 = h2 * 20 + (u32)c; h3 = h3 * 28 + (u32)c; h4 = h4 * 44 + (u32)c;
        if (h1 == 0xB16B88F1 && h2 == 0xFF5376F1 && h3 == 0x72AC5BF1 && h4 == 0x2F909AF1) break;
    }
    int level = GET();
    if (level != 1 && level != 2) { *ppos = pos; return -1; }
    if (GET() != 1) { *ppos = pos; return -1; }
    int lo = GET(), hi = GET();
    if (lo < 0 || hi < 0) { *ppos = pos; return -1; }
    int hsize = lo + hi * 256;
    /* decompressor.v:277-334 */
    u8 hdr[65536 + 8];
    int hl = 0;
    for (int i = 0; i < 5; i++) { int b = GET(); if (b < 0) { *ppos = pos; return -1; } hdr[hl++] = (u8)b; }
    int ncomp = hdr[4];
    for (int i = 0; i < ncomp; i++) {
        int ct = GET();
        if (ct < 0 || ct >= 10) { *ppos = pos; return -1; }
        hdr[hl++] = (u8)ct;
        for (int j = 1; j < g_compsize[ct]; j++) { int b = GET(); if (b < 0) { *ppos = pos; return -1; } hdr[hl++] = (u8)b; }
    }
    if (GET() != 0) { *ppos = pos; return -1; }
    hdr[hl++] = 0;
    int cend = hl - 1, hbegin = hl;
    int hcomp_len = hsize - hl;
    for (int i = 0; i < hcomp_len; i++) { int b = GET(); if (b < 0) { *ppos = pos; return -1; } hdr[hl++] = (u8)b; }
    int hend = hl - 1;
    /* find_filename: decompressor.v:356-408 */
    int marker = GET();
    if (marker < 0 || marker == 0xFF) { *ppos = pos; return -1; }
    size_t k = 0;
    for (;;) { int c = GET(); if (c < 0) { *ppos = pos; return -1; } if (c == 0) break;
               if (c == 0xFF) { *ppos = pos; return -1; }
               if (filename && k + 1 < fncap) filename[k++] = (char)c; }
    if (filename && fncap) filename[k] = 0;
    k = 0;
    for (;;) { int c = GET(); if (c < 0) { *ppos = pos; return -1; } if (c == 0) break;
               if (comment && k + 1 < cmcap) comment[k++] = (char)c; }
    if (comment && cmcap) comment[k] = 0;
    if (GET() < 0) { *ppos = pos; return -1; }
    int64_t nout = 0;
    int ovf = 0;
    if (ncomp == 0) {
        /* store mode: decompressor.v:518-587 */
        int first = 1;
        for (;;) {
            int b0 = GET(), b1 = GET(), b2 = GET(), b3 = GET();
            if (b0 < 0 || b1 < 0 || b2 < 0 || b3 < 0) break;
            u32 cnt = ((u32)b0 << 24) | ((u32)b1 << 16) | ((u32)b2 << 8) | (u32)b3;
            if (cnt == 0) break;
            if (first) { if (GET() < 0) break; cnt--; first = 0; }
            while (cnt) { int c = GET(); if (c < 0) { cnt = 0; break; }
                          if ((size_t)nout < cap) out[nout] = (u8)c; else ovf = 1;
                          nout++; cnt--; }
        }
        marker = GET(); /* decompressor.v:605 */
    } else {
        int err = 0;
        zo_codec *pc = zo_codec_new(hdr, hl, cend, hbegin, hend, &err);
        if (!pc) { *ppos = pos; return -3; }
        /* decode everything incl. the PP byte, then PostProcessor PASS
         * (decompressor.v:56-82): first byte selects the mode, the rest passes. */
        size_t consumed = 0, tmpcap = cap + 1;
        u8 *tmp = (u8 *)malloc(tmpcap ? tmpcap : 1);
        u32 curr = 0;
        int64_t nd = decode_impl(pc, arc + pos, n - pos, tmp, tmpcap, &consumed, &curr, NULL, 0);
        zo_codec_free(pc);
        if (nd < 0) { free(tmp); *ppos = pos; return -2; }
        if (nd > 0 && tmp[0] == 1) { free(tmp); *ppos = pos; return -4; } /* PROG mode: out of scope */
        if (nd > 0) { nout = nd - 1; memcpy(out, tmp + 1, (size_t)nout); }
        free(tmp);
        /* Decoder.skip(): decoder.v:151-196, starting from the decoder's `code`. */
        pos += consumed;
        marker = -1;
        if (curr == 0) { int c = GET(); if (c >= 0) curr = (u32)c; else goto done; }
        while (curr != 0) { int c = GET(); if (c < 0) goto done; curr = (curr << 8) | (u32)c; }
        for (;;) { int c = GET(); if (c < 0) goto done; if (c != 0) { marker = c; break; } }
    }
done:
    if (sha_ok) *sha_ok = -1;
    if (marker == 253) { /* decompressor.v:608-628 */
        u8 stored[20], calc[20];
        memset(stored, 0, 20);
        for (int i = 0; i < 20; i++) { int c = GET(); if (c >= 0) stored[i] = (u8)c; }
        if (!ovf) { zo_sha1(out, (size_t)nout, calc); if (sha_ok) *sha_ok = memcmp(stored, calc, 20) == 0; }
    }
#undef GET
    *ppos = pos;
    return ovf ? -2 : nout;
}
