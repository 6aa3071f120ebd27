"""the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

GPU sanitizers are not available on this pool, so the CPU restatement -- the thing every GPU result is
compared with -- is the part that gets them: `make -C oracle asan` builds zpaq_oracle.c with
-fsanitize=address,undefined, and the golden stream set (tests/test_oracle.py: every reference KAT, every
level's golden coded streams, traces, multi-segment streams, the ZPAQL known answers) runs against that build
in a child interpreter with the sanitizer runtime preloaded.  Any report makes the child exit non-zero."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_golden_stream_set_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update({
        "ZPQ_ORACLE_SO": "libzpaq_oracle_asan.so",
        "LD_PRELOAD": asan,
        # CPython itself is not leak-clean; everything else is fatal
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=97:allocator_may_return_null=1",
        "UBSAN_OPTIONS": "halt_on_error=1:exitcode=98:print_stacktrace=1",
    })
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    # the sanitizer build was really the one loaded
    probe = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); import oracle_lib as O; O.lib(); "
                            "print(any('libzpaq_oracle_asan' in l for l in open('/proc/self/maps')))" % os.path.join(ROOT, "tests")],
                           env=env, capture_output=True, text=True, cwd=ROOT)
    assert probe.stdout.strip() == "True", probe.stdout + probe.stderr
