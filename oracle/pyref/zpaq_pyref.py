"""Second, independent restatement of dy-tea/zpaq-v's codec path (pure Python).

TEST INFRASTRUCTURE ONLY.  Written directly from the V sources (file:line cites
are under /root/reference/zpaq/), NOT from oracle/zpaq_oracle.c, so that the two
restatements check each other.  It is slow (pure-Python loops) and is used only
for small cases: cross-checking the C oracle and generating tests/golden/*.json
(tests/golden/make_golden.py).  Nothing in the product path imports it.

V semantics made explicit: `int` = 32-bit two's complement with wrap-around
(i32()), `u32` wraps (u32()), `>>` on int is arithmetic, Go operator precedence.
"""

M32 = 0xFFFFFFFF


def u32(x):
    return x & M32


def i32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


def u8(x):
    return x & 0xFF


# ---------------------------------------------------------------- tables

def exp_approx(x):  # predictor.v:52-70
    if x < -20.0:
        return 0.0
    if x > 20.0:
        return 485165195.4
    result = 1.0
    term = 1.0
    for i in range(1, 40):
        term *= x / float(i)
        result += term
        if term < 1e-15 and term > -1e-15:
            break
    return result


def ln_approx(x):  # predictor.v:169-190
    if x <= 0.0:
        return -20.0
    if x > 1e9:
        return 20.0
    y = (x - 1.0) / (x + 1.0)
    y2 = y * y
    result = y
    term = y
    for i in range(1, 50):
        term *= y2
        result += term / float(2 * i + 1)
        if term < 1e-15 and term > -1e-15:
            break
    return 2.0 * result


def init_squash_table():  # predictor.v:21-49
    t = [0] * 4096
    for i in range(-2047, 2048):
        d = float(i) / 64.0
        if d < -20.0:
            d = -20.0
        if d > 20.0:
            d = 20.0
        if d >= 0:
            e = 1.0 / (1.0 + exp_approx(-d))
        else:
            tmp = exp_approx(d)
            e = tmp / (1.0 + tmp)
        v = int(32767.0 * e + 0.5)  # truncation toward zero, like V int(f64)
        if v < 1:
            t[i + 2047] = 1
        elif v > 32767:
            t[i + 2047] = 32767
        else:
            t[i + 2047] = v
    return t


def init_stretch_table():  # predictor.v:73-96
    t = [0] * 32768
    for i in range(32768):
        p = float(i) / 32767.0
        if p <= 0.0:
            t[i] = -2047
        elif p >= 1.0:
            t[i] = 2047
        else:
            v = int(ln_approx(p / (1.0 - p)) * 64.0)
            t[i] = -2047 if v < -2047 else (2047 if v > 2047 else v)
    return t


SQUASH = init_squash_table()
STRETCH = init_stretch_table()
DT2K = [2048 - 2048 // (i + 1) for i in range(256)]       # predictor.v:99-106
DT = [(1 << 17) // (i * 2 + 3) * 2 for i in range(1024)]  # predictor.v:109 (literal :111-166)


def squash(d):  # predictor.v:193-202
    idx = i32(d + 2047)
    if idx < 0:
        idx = 0
    if idx >= 4094:
        idx = 4093
    return SQUASH[idx]


def stretch(p):  # predictor.v:205-214
    idx = p
    if idx < 1:
        idx = 1
    if idx >= 32768:
        idx = 32767
    return STRETCH[idx]


def clamp2k(x):  # predictor.v:217-225
    return -2048 if x < -2048 else (2047 if x > 2047 else x)


def clamp512k(x):  # predictor.v:228-236
    return -262144 if x < -262144 else (262143 if x > 262143 else x)


def _build_ns():
    """libzpaq's StateTable construction; the reference stores the result as a
    literal (statetable.v:15-57).  Its SHA-256 is pinned in tests/golden."""
    bound = [20, 48, 15, 8, 6, 5]

    def num_states(n0, n1):
        if n0 < n1:
            return num_states(n1, n0)
        if n0 < 0 or n1 < 0 or n1 >= 6 or n0 > bound[n1]:
            return 0
        return 1 + (1 if (n1 > 0 and n0 + n1 <= 17) else 0)

    def discount(n):
        return sum(1 for k in (1, 2, 3, 4, 5, 7, 8) if n >= k)

    def next_state(n0, n1, y):
        if n0 < n1:
            b, a = next_state(n1, n0, 1 - y)
            return a, b
        if y:
            n1 += 1
            n0 = discount(n0)
        else:
            n0 += 1
            n1 = discount(n1)
        while not num_states(n0, n1):
            if n1 < 2:
                n0 -= 1
            else:
                n0 = (n0 * (n1 - 1) + (n1 // 2)) // n1
                n1 -= 1
        return n0, n1

    t = {}
    state = 0
    for i in range(50):
        for n1 in range(i + 1):
            n0 = i - n1
            n = num_states(n0, n1)
            if n:
                t[(n0, n1, 0)] = state
                t[(n0, n1, 1)] = state + n - 1
                state += n
    ns = [0] * 1024
    for n0 in range(50):
        for n1 in range(50):
            for y in range(num_states(n0, n1)):
                s = t[(n0, n1, y)]
                a, b = next_state(n0, n1, 0)
                ns[s * 4] = t[(a, b, 0)]
                a, b = next_state(n0, n1, 1)
                ns[s * 4 + 1] = t[(a, b, 1)]
                ns[s * 4 + 2] = n0
                ns[s * 4 + 3] = n1
    return ns


NS = _build_ns()


def ns_next(state, y):  # statetable.v:75-84
    if state < 0 or state >= 256:
        return 0
    idx = state * 4 + y
    if idx < 0 or idx >= 1024:
        return 0
    return NS[idx]


def cminit(state):  # statetable.v:90-100
    if state < 0 or state >= 256:
        return 1 << 22
    n0 = NS[state * 4 + 2]
    n1 = NS[state * 4 + 3]
    return i32(u32((n1 * 2 + 1) << 22) // (n0 + n1 + 1))


def oplen(op):  # types.v:51-64
    if op == 255:
        return 3
    if (op & 7) == 7:
        return 2
    return 1


COMPSIZE = [0, 2, 3, 2, 3, 4, 6, 6, 3, 5]  # types.v:74-85

# ---------------------------------------------------------------- levels


def level_header(level):  # levels.v:26-375 (data)
    def chain(hh, hm, bits, isse, mix2):
        n = 1 + isse + (1 if mix2 else 0)
        b = [hh, hm, 0, 0, n, 3, bits]
        for j in range(isse):
            b += [8, bits, j]
        if mix2:
            b += [6, mix2, isse - 1, isse, 24, 255]
        b += [0, 74, 18, 104, 95, 0]
        b += [59, 112, 25] * (n - 1)
        b += [59, 112, 56, 0, 0]
        return bytes(b)

    if level == 0:
        return bytes(7)
    if level == 2:
        return chain(9, 16, 16, 2, 0)
    if level == 3:
        return chain(10, 18, 18, 4, 0)
    if level == 4:
        return chain(12, 20, 20, 5, 16)
    if level == 5:
        return chain(14, 22, 22, 7, 18)
    return bytes([1, 2, 0, 0, 2, 3, 16, 8, 19, 0, 0,
                  96, 4, 28, 59, 10, 59, 112, 25, 10, 59, 10, 59, 112, 56, 0])


def scan_header(h):  # compressor.v:96-145
    if len(h) >= 5:
        n = h[4]
        pos = 5
        i = 0
        while i < n and pos < len(h):
            ctype = h[pos]
            if ctype >= len(COMPSIZE):
                break
            pos += COMPSIZE[ctype]
            i += 1
        cend = pos
        if pos < len(h) and h[pos] == 0:
            pos += 1
        hbegin = pos
        while pos < len(h):
            op = h[pos]
            if op == 0:
                break
            pos += 1
            if (op & 7) == 7:
                pos += 2 if op == 63 else 1
        return cend, hbegin, pos
    return len(h), len(h), len(h)


# ---------------------------------------------------------------- ZPAQL


class ZPAQL:
    STEP_CAP = 1 << 20

    def __init__(self, header, cend, hbegin, hend):
        self.a = self.b = self.c = self.d = 0
        self.f = 0
        self.pc = 0
        self.header = bytes(header)
        self.cend, self.hbegin, self.hend = cend, hbegin, hend
        self.r = [0] * 256
        self.m = bytearray()
        self.h = []
        if len(self.header) >= 2:  # zpaql.v:74-95
            hh, hm = self.header[0], self.header[1]
            if 0 < hh < 32:
                self.h = [0] * (1 << hh)
            if 0 < hm < 32:
                self.m = bytearray(1 << hm)
            self.pc = hbegin

    def m_get(self, i):  # zpaql.v:178-184
        return self.m[i & (len(self.m) - 1)] if self.m else 0

    def m_set(self, i, v):  # zpaql.v:187-193
        if self.m:
            self.m[i & (len(self.m) - 1)] = v & 255

    def h_get(self, i):  # zpaql.v:196-202
        return self.h[i & (len(self.h) - 1)] if self.h else 0

    def h_set(self, i, v):  # zpaql.v:205-211
        if self.h:
            self.h[i & (len(self.h) - 1)] = v & M32

    def run(self, inp):  # zpaql.v:167-175
        self.a = u32(inp)
        self.pc = self.hbegin
        steps = 0
        while self.pc < self.hend and self.pc >= self.hbegin:
            if not self.execute():
                break
            steps += 1
            if steps >= self.STEP_CAP:
                raise RuntimeError("ZPAQL step cap")

    def execute(self):  # zpaql.v:215-954
        if self.pc < self.hbegin or self.pc >= self.hend:
            return False
        hd = self.header
        op = hd[self.pc]
        self.pc += 1
        operand = 0
        if oplen(op) == 2 and self.pc < len(hd):
            operand = hd[self.pc]
            self.pc += 1
        elif oplen(op) == 3 and self.pc + 1 < len(hd):
            operand = hd[self.pc] + hd[self.pc + 1] * 256
            self.pc += 2
        z = self
        if op == 0:
            pass
        elif op in (1, 9, 17, 25):       # X++
            n = "abcd"[op >> 3]
            setattr(z, n, u32(getattr(z, n) + 1))
        elif op in (2, 10, 18, 26):      # X--
            n = "abcd"[op >> 3]
            setattr(z, n, u32(getattr(z, n) - 1))
        elif op in (3, 11, 19, 27):      # X!
            n = "abcd"[op >> 3]
            setattr(z, n, u32(~getattr(z, n)))
        elif op in (4, 12, 20, 28):      # X=0
            setattr(z, "abcd"[op >> 3], 0)
        elif op in (7, 15, 23, 31):      # X=R N
            setattr(z, "abcd"[op >> 3], z.r[operand & 255])
        elif op == 8:
            z.a, z.b = z.b, z.a
        elif op == 16:
            z.a, z.c = z.c, z.a
        elif op == 24:
            z.a, z.d = z.d, z.a
        elif op in (32, 40):             # *B<>A, *C<>A
            p = z.b if op == 32 else z.c
            tmp = z.m_get(p)
            z.m_set(p, u8(z.a))
            z.a = tmp
        elif op in (33, 41):
            p = z.b if op == 33 else z.c
            z.m_set(p, u8(z.m_get(p) + 1))
        elif op in (34, 42):
            p = z.b if op == 34 else z.c
            z.m_set(p, u8(z.m_get(p) - 1))
        elif op in (35, 43):
            p = z.b if op == 35 else z.c
            z.m_set(p, u8(~z.m_get(p)))
        elif op in (36, 44):
            z.m_set(z.b if op == 36 else z.c, 0)
        elif op == 39:                   # JT
            if z.f != 0:
                z.pc += ((operand + 128) & 255) - 127
        elif op == 47:                   # JF
            if z.f == 0:
                z.pc += ((operand + 128) & 255) - 127
        elif op == 48:
            tmp = z.h_get(z.d)
            z.h_set(z.d, z.a)
            z.a = tmp
        elif op == 49:
            z.h_set(z.d, u32(z.h_get(z.d) + 1))
        elif op == 50:
            z.h_set(z.d, u32(z.h_get(z.d) - 1))
        elif op == 51:
            z.h_set(z.d, u32(~z.h_get(z.d)))
        elif op == 52:
            z.h_set(z.d, 0)
        elif op == 55:
            z.r[operand & 255] = z.a
        elif op == 56:
            return False
        elif op == 57:  # OUT: appends to a host buffer only (zpaql.v:151-159,382-384)
            if getattr(z, "outbuf", None) is not None:
                z.outbuf.append(z.a & 255)
        elif op == 59:
            z.a = u32((z.a + z.m_get(z.b) + 512) * 773)
        elif op == 60:
            z.h_set(z.d, u32((z.h_get(z.d) + z.a + 512) * 773))
        elif op == 63:
            z.pc += ((operand + 128) & 255) - 127
        elif 64 <= op < 120:             # assignments
            src = op & 7
            v = [z.a, z.b, z.c, z.d, z.m_get(z.b), z.m_get(z.c), z.h_get(z.d), operand][src]
            dst = (op - 64) >> 3
            if dst == 0:
                z.a = u32(v)
            elif dst == 1:
                z.b = u32(v)
            elif dst == 2:
                z.c = u32(v)
            elif dst == 3:
                z.d = u32(v)
            elif dst == 4:
                z.m_set(z.b, u8(v))
            elif dst == 5:
                z.m_set(z.c, u8(v))
            else:
                z.h_set(z.d, u32(v))
        elif 128 <= op < 240:
            src = op & 7
            v = [z.a, z.b, z.c, z.d, z.m_get(z.b), z.m_get(z.c), z.h_get(z.d), operand][src]
            g = (op - 128) >> 3
            a = z.a
            if g == 0:
                a = a + v
            elif g == 1:
                a = a - v
            elif g == 2:
                a = a * v
            elif g == 3:
                a = a // v if v != 0 else a
            elif g == 4:
                a = a % v if v != 0 else a
            elif g == 5:
                a = a & v
            elif g == 6:
                a = a & ~v
            elif g == 7:
                a = a | v
            elif g == 8:
                a = a ^ v
            elif g == 9:
                a = a << (v & 31)
            elif g == 10:
                a = a >> (v & 31)
            elif g == 11:
                z.f = 1 if z.a == v else 0
            elif g == 12:
                z.f = 1 if z.a < v else 0
            else:
                z.f = 1 if z.a > v else 0
            if g <= 10:
                z.a = u32(a)
        elif op == 255:
            z.pc = z.hbegin + hd[z.pc - 2] + hd[z.pc - 1] * 256
            if z.pc >= z.hend:
                return False
        else:
            return False
        return True


# ---------------------------------------------------------------- Predictor


class Component:
    def __init__(self):
        self.ctype = 0
        self.cm = []
        self.ht = bytearray()
        self.a16 = []
        self.a = self.b = self.c = 0
        self.cxt = 0
        self.limit = 0


class Predictor:
    def __init__(self, z):  # predictor.v:292-470
        self.z = z
        self.c8 = 1
        self.hmap4 = 1
        self.comp = []
        self.p = []
        self.h = []
        hd = z.header
        if len(hd) < 5 or hd[4] == 0:
            return
        n = hd[4]
        self.comp = [Component() for _ in range(n)]
        self.p = [0] * n
        self.h = [0] * n
        cp = 5
        i = 0
        while i < n and cp < z.cend:
            cr = self.comp[i]
            t = hd[cp]
            cr.ctype = t
            if t == 1:
                cr.a = hd[cp + 1]
                cp += 2
            elif t == 2:
                cr.a = hd[cp + 1]
                cr.limit = hd[cp + 2] * 4
                cr.cm = [0x80000000] * (1 << cr.a)
                cp += 3
            elif t == 3:
                cr.a = hd[cp + 1]
                cr.cm = [u32(cminit(j)) for j in range(256)]
                cr.ht = bytearray(16 << (cr.a + 2))
                cp += 2
            elif t == 4:
                cr.a = hd[cp + 1]
                cr.b = hd[cp + 2]
                cr.cm = [0] * (1 << cr.a)
                cr.ht = bytearray(1 << cr.b)
                cp += 3
            elif t == 5:
                cr.a, cr.b, cr.c = hd[cp + 1], hd[cp + 2], hd[cp + 3]
                cp += 4
            elif t == 6:
                cr.a = hd[cp + 1]
                size = 1 << cr.a
                cr.b = hd[cp + 2]
                cr.c = size
                cr.a16 = [32768] * size
                cr.cm = [hd[cp + 2], hd[cp + 3], hd[cp + 4], hd[cp + 5]]
                cp += 6
            elif t == 7:
                cr.a = hd[cp + 1]
                size = 1 << cr.a
                m = hd[cp + 3]
                cr.b = hd[cp + 2]
                cr.c = size
                cr.limit = m
                cr.ht = bytearray([hd[cp + 4], hd[cp + 5]])
                cr.cm = [u32((65536 // m) << 8)] * (size * m)
                cp += 6
            elif t == 8:
                cr.a = hd[cp + 1]
                cr.b = hd[cp + 2]
                cr.ht = bytearray(16 << (cr.a + 2))
                cr.cm = [0] * 512
                for k in range(256):
                    cr.cm[k * 2] = 1 << 15
                    cr.cm[k * 2 + 1] = u32(clamp512k(i32(stretch(cminit(k) >> 8) * 1024)))
                cp += 3
            elif t == 9:
                cr.a = hd[cp + 1]
                cr.b = hd[cp + 2]
                size = 1 << cr.a
                cr.limit = hd[cp + 4] * 4
                start = hd[cp + 3]
                cr.cm = [u32((squash((k & 31) * 64 - 992) << 17) | start) for k in range(size * 32)]
                cp += 5
            else:
                cp += 1
            i += 1

    def reset(self):  # predictor.v:827-833
        self.c8 = 1
        self.hmap4 = 1
        for i in range(len(self.h)):
            self.h[i] = 0

    @staticmethod
    def find_ht(ht, sizebits, cxt):  # predictor.v:495-532
        chk = (cxt >> sizebits) & 255
        h0 = u32(cxt * 16) & u32(len(ht) - 16)
        if ht[h0] == chk:
            return h0
        h1 = h0 ^ 16
        if ht[h1] == chk:
            return h1
        h2 = h0 ^ 32
        if ht[h2] == chk:
            return h2
        if ht[h0 + 1] <= ht[h1 + 1] and ht[h0 + 1] <= ht[h2 + 1]:
            r = h0
        elif ht[h1 + 1] < ht[h2 + 1]:
            r = h1
        else:
            r = h2
        for k in range(16):
            ht[r + k] = 0
        ht[r] = chk
        return r

    def predict(self):  # predictor.v:536-668
        n = len(self.comp)
        if n == 0:
            return 16384
        p = self.p
        for i in range(n):
            cr = self.comp[i]
            t = cr.ctype
            if t == 1:
                p[i] = (cr.a - 128) * 16
            elif t == 2:
                cr.cxt = u32(self.h[i]) ^ self.hmap4
                idx = i32(cr.cxt) & (len(cr.cm) - 1)
                p[i] = stretch(cr.cm[idx] >> 17)
            elif t == 3:
                if self.c8 == 1 or (self.c8 & 0xf0) == 16:
                    cr.c = self.find_ht(cr.ht, cr.a + 2, u32(self.h[i] + 16 * self.c8))
                cr.cxt = cr.ht[cr.c + (self.hmap4 & 15)]
                p[i] = stretch(cr.cm[cr.cxt] >> 8)
            elif t == 4:
                if cr.a == 0:
                    p[i] = 0
                else:
                    idx = i32(cr.limit - cr.b) & (len(cr.ht) - 1)
                    cr.c = (cr.ht[idx] >> (7 - cr.cxt)) & 1
                    weight = DT2K[cr.a & 255]
                    p[i] = stretch((weight * (cr.c * -2 + 1)) & 32767)
            elif t == 5:
                j, k, wt = cr.a, cr.b, cr.c
                if j < n and k < n:
                    p[i] = i32(p[j] * wt + p[k] * (256 - wt)) >> 8
                else:
                    p[i] = 0
            elif t == 6:
                j, k, mask = cr.cm[0], cr.cm[1], cr.cm[3]
                cr.cxt = u32(self.h[i] + (self.c8 & mask)) & u32(cr.c - 1)
                w = cr.a16[cr.cxt]
                if j < n and k < n:
                    p[i] = clamp2k(i32(w * p[j] + (65536 - w) * p[k]) >> 16)
                else:
                    p[i] = 0
            elif t == 7:
                j, m, mask = cr.b, cr.limit, cr.ht[1]
                cr.cxt = u32(i32(i32(self.h[i]) + (i32(self.c8) & mask)) & (cr.c - 1))
                idx = cr.cxt * m
                s = 0
                l = 0
                while l < m and (j + l) < n:
                    wt = i32(cr.cm[idx + l]) >> 8
                    s = i32(s + i32(wt * p[j + l]))
                    l += 1
                p[i] = clamp2k(s >> 8)
            elif t == 8:
                if self.c8 == 1 or (self.c8 & 0xf0) == 16:
                    cr.c = self.find_ht(cr.ht, cr.a + 2, u32(self.h[i] + 16 * self.c8))
                cr.cxt = cr.ht[cr.c + (self.hmap4 & 15)]
                wt0 = i32(cr.cm[cr.cxt * 2])
                wt1 = i32(cr.cm[cr.cxt * 2 + 1])
                j = cr.b
                if j < n:
                    p[i] = clamp2k(i32(i32(wt0 * p[j]) + i32(wt1 * 64)) >> 16)
                else:
                    p[i] = clamp2k(wt1 >> 10)
            elif t == 9:
                j = cr.b
                cr.cxt = u32((self.h[i] + self.c8) * 32)
                pq = 992
                if j < n:
                    pq = p[j] + 992
                if pq < 0:
                    pq = 0
                if pq > 1983:
                    pq = 1983
                wt = pq & 63
                pq >>= 6
                idx = i32(i32(cr.cxt) + pq)
                idx2 = i32(idx + 1)
                if idx >= 0 and idx2 < len(cr.cm):
                    p1 = cr.cm[idx] >> 10
                    p2 = cr.cm[idx2] >> 10
                    p[i] = stretch(i32(p1 * (64 - wt) + p2 * wt) >> 13)
                else:
                    p[i] = 0
                cr.cxt = u32(u32(idx) + (wt >> 5))
            else:
                p[i] = 0
        return squash(p[n - 1])

    def update(self, y):  # predictor.v:672-824
        n = len(self.comp)
        p = self.p
        for i in range(n):
            cr = self.comp[i]
            t = cr.ctype
            if t == 2:
                idx = i32(cr.cxt) & (len(cr.cm) - 1)
                pn = cr.cm[idx]
                count = pn & 0x3ff
                err = y * 32767 - (pn >> 17)
                upd = i32(err * DT[count]) & -1024
                inc = 1 if count < cr.limit else 0
                cr.cm[idx] = u32(i32(pn) + upd + inc)
            elif t == 3:
                s = cr.c + (self.hmap4 & 15)
                cr.ht[s] = u8(ns_next(cr.ht[s], y))
                v = cr.cm[cr.cxt]
                cr.cm[cr.cxt] = u32(i32(v) + ((y * 32767 - (v >> 8)) >> 2))
            elif t == 4:
                if cr.c != y:
                    cr.a = 0
                mask = len(cr.ht) - 1
                idx = cr.limit & mask
                cr.ht[idx] = u8((cr.ht[idx] << 1) | y)
                cr.cxt += 1
                if cr.cxt >= 8:
                    cr.cxt = 0
                    cr.limit += 1
                    cr.limit &= mask
                    hi = self.h[i]
                    ci = i32(hi) & (len(cr.cm) - 1)
                    if cr.a == 0:
                        cr.b = i32(cr.limit - i32(cr.cm[ci]))
                        if (cr.b & mask) != 0:
                            while cr.a < 255:
                                i1 = (cr.limit - cr.a - 1) & mask
                                i2 = (cr.limit - cr.a - cr.b - 1) & mask
                                if cr.ht[i1] != cr.ht[i2]:
                                    break
                                cr.a += 1
                    elif cr.a < 255:
                        cr.a += 1
                    cr.cm[ci] = u32(cr.limit)
            elif t == 6:
                j, k, rate = cr.cm[0], cr.cm[1], cr.cm[2]
                err = i32((y * 32767 - squash(p[i])) * rate) >> 5
                if j < n and k < n:
                    w = cr.a16[cr.cxt]
                    w += i32(i32(err * (p[j] - p[k])) + (1 << 12)) >> 13
                    if w < 0:
                        w = 0
                    if w > 65535:
                        w = 65535
                    cr.a16[cr.cxt] = w
            elif t == 7:
                jj, m, rate = cr.b, cr.limit, cr.ht[0]
                err = i32((y * 32767 - squash(p[i])) * rate) >> 4
                idx = cr.cxt * m
                l = 0
                while l < m and (jj + l) < n:
                    wt = clamp512k(i32(i32(cr.cm[idx + l]) + (i32(i32(err * p[jj + l]) + (1 << 12)) >> 13)))
                    cr.cm[idx + l] = u32(wt)
                    l += 1
            elif t == 8:
                j = cr.b
                err = y * 32767 - squash(p[i])
                if j < n:
                    wt0 = clamp512k(i32(i32(cr.cm[cr.cxt * 2]) + (i32(i32(err * p[j]) + (1 << 12)) >> 13)))
                    wt1 = clamp512k(i32(i32(cr.cm[cr.cxt * 2 + 1]) + ((err + 16) >> 5)))
                    cr.cm[cr.cxt * 2] = u32(wt0)
                    cr.cm[cr.cxt * 2 + 1] = u32(wt1)
                cr.ht[cr.c + (self.hmap4 & 15)] = u8(ns_next(cr.cxt, y))
            elif t == 9:
                idx = i32(cr.cxt) & (len(cr.cm) - 1)
                v = cr.cm[idx]
                err = y * 32767 - (v >> 17)
                count = i32(v) & 1023
                if count < cr.limit:
                    v = u32(i32(v) + (i32(i32(err * (cr.limit - count)) + (1 << 12)) >> 13) + 1)
                cr.cm[idx] = v
        # predictor.v:807-823
        self.c8 = u32((self.c8 << 1) | y)
        if self.c8 >= 256:
            self.z.run(self.c8 - 256)
            i = 0
            while i < n and i < len(self.z.h):
                self.h[i] = self.z.h[i]
                i += 1
            self.hmap4 = 1
            self.c8 = 1
        elif 16 <= self.c8 < 32:
            self.hmap4 = ((self.hmap4 & 0xf) << 5) | (y << 4) | 1
        else:
            self.hmap4 = (self.hmap4 & 0x1f0) | (((self.hmap4 & 0xf) * 2 + y) & 0xf)


# ---------------------------------------------------------------- coder


class Encoder:  # encoder.v
    def __init__(self, pr):
        self.low = 1
        self.high = 0xFFFFFFFF
        self.pr = pr
        self.out = bytearray()
        self.trace = None

    def encode(self, y, p):  # encoder.v:48-89
        pr = 0 if p < 0 else (65535 if p > 65535 else p)
        rng = u32(self.high - self.low)
        mid = u32(self.low + ((rng * pr) >> 16))
        if y != 0:
            self.high = mid
        else:
            self.low = u32(mid + 1)
        while (self.high ^ self.low) < 0x1000000:
            self.out.append(self.high >> 24)
            self.low = u32(self.low << 8)
            self.high = u32(self.high << 8) | 0xFF
            if self.low == 0:
                self.low = 1

    def compress(self, c):  # encoder.v:93-120
        if c == -1:
            self.encode(1, 0)
            return
        self.encode(0, 0)
        for i in range(7, -1, -1):
            y = (c >> i) & 1
            p = self.pr.predict()
            self.encode(y, p * 2 + 1)
            self.pr.update(y)
            if self.trace is not None:
                self.trace.append((p, y, self.low, self.high))

    def flush(self):  # encoder.v:130-139
        for s in (24, 16, 8, 0):
            self.out.append((self.high >> s) & 255)


class Decoder:  # decoder.v
    def __init__(self, pr, data):
        self.low = 1
        self.high = 0xFFFFFFFF
        self.code = 0
        self.pr = pr
        self.data = data
        self.pos = 0
        for _ in range(4):  # decoder.v:38-46
            self._shift()

    def _get(self):
        if self.pos >= len(self.data):
            return -1
        c = self.data[self.pos]
        self.pos += 1
        return c

    def _shift(self):
        c = self._get()
        self.code = u32(self.code << 8) if c < 0 else (u32(self.code << 8) | c)

    def decode(self, p):  # decoder.v:73-118
        pr = 0 if p < 0 else (65535 if p > 65535 else p)
        rng = u32(self.high - self.low)
        mid = u32(self.low + ((rng * pr) >> 16))
        if self.code <= mid:
            y = 1
            self.high = mid
        else:
            y = 0
            self.low = u32(mid + 1)
        while (self.high ^ self.low) < 0x1000000:
            self.low = u32(self.low << 8)
            self.high = u32(self.high << 8) | 0xFF
            if self.low == 0:
                self.low = 1
            self._shift()
        return y

    def decompress(self):  # decoder.v:122-145
        if self.decode(0) != 0:
            return -1
        c = 1
        while c < 256:
            p = self.pr.predict()
            y = self.decode(p * 2 + 1)
            self.pr.update(y)
            c = (c << 1) | y
        return c - 256


# ---------------------------------------------------------------- drivers


class PostProcessor:
    """decompressor.v:14-167, byte for byte: (PASS=0 | PROG=1 psize[0..1] pcomp[0..psize-1]) data..."""

    def __init__(self, ph=0, pm=0):  # PostProcessor.new + init (:26-43)
        self.state = 0
        self.hsize = 0
        self.ph, self.pm = ph, pm
        self.z = None
        self.outbuf = bytearray()

    def write(self, c):  # :55-152
        if self.state == 0:
            if c < 0:
                return self.state
            self.state = c + 1
            if self.state > 2:
                self.state = 1
        elif self.state == 1:
            if c >= 0:
                self.outbuf.append(c)
        elif self.state == 2:
            if c < 0:
                return self.state
            self.hsize = c
            self.state = 3
        elif self.state == 3:
            if c < 0:
                return self.state
            self.hsize += c * 256
            if self.hsize < 1:
                self.state = 1
                return self.state
            hdr = bytearray(self.hsize + 300)
            hdr[4], hdr[5] = self.ph & 255, self.pm & 255
            z = ZPAQL(b"", 8, 8 + 128, 8 + 128)      # fresh VM: no H, no M, registers 0
            z.header = hdr
            z.outbuf = []
            self.z = z
            self.state = 4
        elif self.state == 4:
            if c < 0:
                return self.state
            z = self.z
            if z.hend < len(z.header):
                z.header[z.hend] = c
                z.hend += 1
            if z.hend - z.hbegin == self.hsize:
                total = z.cend - 2 + z.hend - z.hbegin
                z.header[0] = total & 255
                z.header[1] = (total >> 8) & 255
                hm = z.header[1]                      # initp() (zpaql.v:86-95): M from header[1]; inith() is never called
                if 0 < hm < 32:
                    z.m = bytearray(1 << hm)
                z.pc = z.hbegin
                z.header = bytes(z.header)
                self.state = 5
        elif self.state == 5:
            if c >= 0:
                self.z.run(c)
                self.outbuf.extend(self.z.outbuf)
                self.z.outbuf = []
        return self.state


def postprocess(decoded, ph=0, pm=0):
    """Everything Decoder.decompress() yields for one segment (mode byte first) -> the segment's output, the way
    Decompresser.decompress drives the PostProcessor (decompressor.v:466-512)."""
    pp = PostProcessor(ph, pm)
    it = iter(decoded)
    while (pp.state & 3) != 1:
        c = next(it, -1)
        if c < 0:
            return b""
        pp.write(c)
    for c in it:
        pp.write(c)
    return bytes(pp.outbuf)


def new_model(header, offsets=None):
    cend, hbegin, hend = offsets if offsets else scan_header(header)
    z = ZPAQL(header, cend, hbegin, hend)
    return Predictor(z)


def encode_segment(pr, data, pp=True, trace=None):
    """New Encoder + Predictor.reset() + [PP byte] + data + EOF + flush
    (compressor.v:238-245,271-290,375-378)."""
    pr.reset()
    e = Encoder(pr)
    e.trace = trace
    if pp:
        e.compress(0)
    for b in data:
        e.compress(b)
    e.compress(-1)
    e.flush()
    return bytes(e.out)


def decode_segment(pr, coded):
    pr.reset()
    d = Decoder(pr, coded)
    out = bytearray()
    while True:
        c = d.decompress()
        if c < 0:
            break
        out.append(c)
    return bytes(out), d.pos
