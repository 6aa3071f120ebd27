#!/usr/bin/env python3
"""tools/isa_budget.py [KERNEL_SUBSTRING] [--src zpq_chain.hip]: a per-role instruction budget of one kernel's byte loop.

Compiles the kernel source for gfx950 with -gline-tables-only -save-temps (no GPU needed), takes the kernel's assembly,
finds its byte loop (the outermost loop with the largest body), and attributes every instruction of the loop body to
the SOURCE LINE it was generated from (the innermost .loc, i.e. inside inlined lambdas) and -- through the inlined-at
chain -- to the bit step (K, nibble) whose code it is.  Source lines are grouped into ROLES by the line ranges of the
lambdas in the source (found by name below, not hard-coded numbers).  Static counts: an instruction inside a nested loop
(the coder's renormalisation, the line store's walk) is counted once and flagged.

Default kernel: the level-2 decoder k_chain<decode, plain, NCH=3, no MIX2, G=8, dense, no HIO>.
Output: a table role x {VALU, SALU, branch, LDS, VMEM, wait/nop} per BYTE and per BIT (/ 8), written to stdout."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zpaq-v_amd", "csrc")


def compile_asm(src, extra=()):
    td = tempfile.mkdtemp(prefix="isa_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-gline-tables-only", "-save-temps=obj", "-c", "-x", "hip", os.path.join(CSRC, src), "-o", os.path.join(td, "k.o")] + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(td):
        if f.endswith("gfx950.s"):
            return os.path.join(td, f)
    raise SystemExit("no device assembly produced")


def kind(op):
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_call"):
        return "branch"
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def source_ranges(src_path):
    """Line ranges of the named lambdas / sections in the source -> role."""
    L = open(src_path).read().split("\n")
    marks = []

    def find(pat, start=0):
        for i in range(start, len(L)):
            if pat in L[i]:
                return i + 1
        return None
    names = [
        ("stretch_of (LDS-packed stretch)", "auto stretch_of = [&]"),
        ("enc_byte / window", "auto enc_byte = [&]"),
        ("dec window request/adopt", "auto dec_request = [&]"),
        ("in_byte (coded input)", "auto in_byte = [&]"),
        ("bcast_down (decoded bit)", "auto bcast_down = [&]"),
        ("select_row (find_ht)", "auto select_row = [&]"),
        ("row request (prefetch)", "#define ZPQ_LOAD_ROWS"),
        ("swz_addr", "auto swz_addr = [&]"),
        ("row request (prefetch)", "auto load_rows = [&]"),
        ("xchg (copy hand-over)", "auto xchg = [&]"),
        ("take_prefetched (rows arrive, forwarding, write-back)", "auto take_prefetched = [&]"),
        ("vm_hash / vm_commit (context hashes)", "auto vm_hash = [&]"),
        ("coded-byte queue (encode)", "auto oq_flush = [&]"),
        ("run_vm", "auto run_vm = [&]"),
        ("nibble_begin (first entry)", "auto nibble_begin = [&]"),
        ("MIX2 weights", "auto mix_byte_begin = [&]"),
        ("bitstep (plain / pipelined)", "auto bitstep = [&]"),
        ("hyp: chain p0 -> p1 -> ... (chain_of)", "auto chain_of = [&]"),
        ("bitstep_hyp", "auto bitstep_hyp = [&]"),
        ("step dispatch", "auto step = [&]"),
        ("byte loop (EOF flag, window, bookkeeping)", "prefetch_rows(0u, 1u);"),
    ]
    for role, pat in names:
        ln = find(pat)
        if ln:
            marks.append((ln, role))
    marks.sort()
    # finer roles inside bitstep_hyp, by its own comments
    hyp0 = find("auto bitstep_hyp = [&]")
    if hyp0:
        sub = [("hyp: entry unpack, chain call", "const u32 s = cur_s;", hyp0),
               ("hyp: squash lookup", "const i32 sq = s_squash[p + 2048];", hyp0),
               ("hyp: own outcome: next state + entry fetch", "// ---- this copy's outcome", hyp0),
               ("hyp: own outcome: train counter / weights", "const u32 cmn = (u32)wadd((i32)cmv", hyp0),
               ("hyp: arithmetic decoder", "// ---- the bit (both copies decode it", hyp0),
               ("hyp: bit broadcast", "y = bcast_down(y);", hyp0),
               ("hyp: commit / take over", "// ---- commit / take over", hyp0),
               ("hyp: next state into the row, c8, slot", "// next bit-history state into the row (statetable.v:75-84), with the decoded bit", hyp0),
               ("hyp: row requests (HYP4 / last-bit)", "if (HYP4 && bit == 6) {", hyp0)]
        for role, pat, st in sub:
            ln = find(pat, st)
            if ln:
                marks.append((ln, role))
        marks.sort()

    def role_of(line):
        r = "line 0 (optimiser-merged code, exec-mask save/restore, mask algebra)"
        for ln, role in marks:
            if ln <= line:
                r = role
            else:
                break
        return r
    return role_of


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src = "zpq_chain.hip"
    extra = []
    for i, a in enumerate(sys.argv[1:]):
        if a == "--src":
            src = sys.argv[i + 2]
        if a.startswith("-D"):
            extra.append(a)
    args = [a for a in args if a != src and not a.startswith("-D")]
    want = args[0] if args else "k_chainILb1ELb0ELi3ELb0ELi8ELb0ELb0EE"
    asm = compile_asm(src, extra)
    lines = open(asm, errors="replace").read().split("\n")
    # the kernel's body
    start = next(i for i, l in enumerate(lines) if want in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S*:", l) and want in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    role_of = source_ranges(os.path.join(CSRC, src))
    file_ids = {}
    for l in lines[:start]:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            file_ids[int(m.group(1))] = m.group(3)
    HELPER_END = next((i + 1 for i, l in enumerate(open(os.path.join(CSRC, src)).read().split("\n")) if "__global__" in l), 0)
    insts = []          # (index, op, kind, line, chain_lines, label_before)
    labels = {}
    cur_line, cur_chain, cur_file = 0, (), ""
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)\s+(\d+)(.*)", s)
        if m:
            cur_file = file_ids.get(int(m.group(1)), "")
            cur_line = int(m.group(2))
            cur_chain = tuple(int(x) for x in re.findall(r"zpq_chain\.hip:(\d+):\d+", m.group(4))[1:]) if src in m.group(4) else ()
            if src not in cur_file and src not in m.group(4):
                pass
            # the small helpers in front of the kernel (wadd, clamp2k, mul_shr16, row_shr1 ...) count for their caller
            if src in (cur_file or "") and cur_line < HELPER_END and cur_chain:
                cur_line, cur_chain = cur_chain[0], cur_chain[1:]
            # when the innermost location is in a header (hip runtime), use the first zpq line of the chain
            if src not in (cur_file or ""):
                z = re.findall(re.escape(src) + r":(\d+):\d+", m.group(4))
                if z:
                    cur_line = int(z[0])
                    cur_chain = tuple(int(x) for x in z[1:])
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            mm = re.search(r"(\.LBB\d+_\d+)", s)
            tgt = mm.group(1) if mm else None
        insts.append((op, kind(op), cur_line, cur_chain, tgt))
    # loops = backward branches
    loops = []
    for i, (op, k, ln, ch, tgt) in enumerate(insts):
        if tgt and tgt in labels and labels[tgt] <= i:
            loops.append((labels[tgt], i))
    if not loops:
        raise SystemExit("no loop found")
    # the byte loop = the innermost loop that contains every instruction generated from the bit steps
    bits = [i for i, (op, k, ln, ch, tgt) in enumerate(insts) if role_of(ln).startswith(("hyp:", "bitstep")) and "chain_of" not in role_of(ln)]
    if not bits:
        raise SystemExit("no bit-step instructions found in this kernel")
    cover = [(x, y) for (x, y) in loops if x <= min(bits) and max(bits) <= y]
    outer = min(cover, key=lambda ab: ab[1] - ab[0]) if cover else max(loops, key=lambda ab: ab[1] - ab[0])
    a, b = outer
    inner = [(x, y) for (x, y) in loops if a <= x and y <= b and (x, y) != outer]
    in_inner = lambda i: any(x <= i <= y for x, y in inner)
    table = collections.defaultdict(lambda: collections.Counter())
    inner_tbl = collections.defaultdict(lambda: collections.Counter())
    for i in range(a, b + 1):
        op, k, ln, ch, tgt = insts[i]
        r = role_of(ln)
        (inner_tbl if in_inner(i) else table)[r][k] += 1
    kinds = ["VALU", "SALU", "branch", "LDS", "VMEM", "wait", "other"]
    print("kernel: %s" % lines[start].split(":")[0])
    print("byte loop: %d static instructions (%d of them inside %d nested loops: renormalisation / walks, counted once)" % (
        b - a + 1, sum(sum(c.values()) for c in inner_tbl.values()), len(inner)))
    print("%-62s %6s %6s %6s %6s %6s %6s | %7s %8s" % ("role (static, outside nested loops)", *kinds[:6], "per byte", "per bit"))
    tot = collections.Counter()
    for r in sorted(table, key=lambda r: -sum(table[r].values())):
        c = table[r]
        n = sum(c.values())
        tot.update(c)
        print("%-62s %6d %6d %6d %6d %6d %6d | %7d %8.1f" % (r, *[c[k] for k in kinds[:6]], n, n / 8.0))
    n = sum(tot.values())
    print("%-62s %6d %6d %6d %6d %6d %6d | %7d %8.1f" % ("TOTAL", *[tot[k] for k in kinds[:6]], n, n / 8.0))
    if inner_tbl:
        print("\ninside nested loops (per pass):")
        for r in sorted(inner_tbl, key=lambda r: -sum(inner_tbl[r].values())):
            c = inner_tbl[r]
            print("%-62s %6d %6d %6d %6d %6d %6d | %7d" % (r, *[c[k] for k in kinds[:6]], sum(c.values())))
    # scalar + branch instructions: which opcodes
    sal = collections.Counter()
    for i in range(a, b + 1):
        op, k, ln, ch, tgt = insts[i]
        if k in ("SALU", "branch", "wait") and not in_inner(i):
            sal[op] += 1
    print("\nscalar / branch / wait opcodes in the byte loop (static, per byte):")
    print("  " + ", ".join("%s %d" % kv for kv in sal.most_common()))


if __name__ == "__main__":
    main()
