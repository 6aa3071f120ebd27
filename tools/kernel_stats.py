"""Per-kernel resource table of libzpaq_hip.so's gfx950 code objects: VGPRs, SGPRs, scratch bytes.
Splits the .hip_fatbin section into its uncompressed clang offload bundles and reads each gfx950 ELF's
AMDGPU metadata note with llvm-readelf (no GPU needed).  Used by tests/test_kernel_resources.py: a chain
kernel that spills to scratch is a 5x slowdown that parity tests cannot see."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_table(so=None):
    so = so or os.path.join(ROOT, "zpaq-v_amd", "lib", "libzpaq_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        d = open(fat, "rb").read()
        for m in re.finditer(re.escape(MAGIC), d):
            base = m.start()
            n = struct.unpack_from("<Q", d, base + len(MAGIC))[0]
            p = base + len(MAGIC) + 8
            for _ in range(n):
                off, size, tlen = struct.unpack_from("<QQQ", d, p)
                triple = d[p + 24:p + 24 + tlen].decode()
                p += 24 + tlen
                if "gfx950" not in triple or size == 0:
                    continue
                co = os.path.join(td, "dev.co")
                open(co, "wb").write(d[base + off:base + off + size])
                txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
                for k in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, flags=re.S):
                    body = k.group(2)
                    g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", body).group(1))
                    out[k.group(1)] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size")}
    return out


if __name__ == "__main__":
    t = kernel_table(sys.argv[1] if len(sys.argv) > 1 else None)
    for k in sorted(t):
        print("%-100s %s" % (k[:100], t[k]))
