"""A spill to scratch in the bit loop is a 5x slowdown that no parity test sees (it happened twice while the
round-2 kernels were written: a four-way register select turned into a stack array, and `cond ? vec4 : vec4` on
ext-vector types lowered through memory).  The gfx950 code objects carry their resource usage in metadata, readable
without a GPU: the kernels that code the shipped levels must use no scratch at all."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_specialised_kernels_use_no_scratch(zpq):
    zpq.lib()                                                # built
    from kernel_stats import kernel_table
    t = kernel_table()
    chain = {k: v for k, v in t.items() if "k_chain" in k}
    assert len(chain) >= 20, sorted(t)
    for name, r in chain.items():
        runtime_loop = "Li0E" in name                         # NCH = 0: carries the ZPAQL interpreter (its byte array M needs a stack)
        if not runtime_loop:
            assert r["scratch"] == 0, (name, r)
        assert r["vgpr"] <= 256, (name, r)                    # two waves of one workgroup may share a SIMD
    pipe = {k: v for k, v in t.items() if "k_pipeI" in k}
    assert len(pipe) >= 8, sorted(t)                          # levels 1-3 x dense / line store (+ striped-upload forms)
    for name, r in pipe.items():
        assert r["scratch"] == 0 and r["vgpr"] <= 256, (name, r)   # (level 3: six waves of one workgroup on four SIMDs)
    # round 3: the level-1 encoder with split stages (k_pipe2) and the wave-split decoder (k_dpipe, opt-in)
    extra = {k: v for k, v in t.items() if "k_pipe2" in k or "k_dpipe" in k}
    assert len(extra) >= 7, sorted(t)
    for name, r in extra.items():
        assert r["scratch"] == 0 and r["vgpr"] <= 256, (name, r)
    # round 3: the wave-per-component encoder of general models: up to sixteen waves of ONE workgroup, four per SIMD
    gd = {k: v for k, v in t.items() if "k_gdec" in k}
    assert len(gd) == 1 and all(r["scratch"] == 0 and r["vgpr"] <= 128 for r in gd.values()), gd   # (its decoder)
    gp = {k: v for k, v in t.items() if "k_gpipe" in k}
    assert len(gp) == 2, sorted(t)                            # byte-batched table accesses / bit-serial stages
    for name, r in gp.items():
        assert r["vgpr"] <= 128 and r["scratch"] <= 16, (name, r)   # (batched: two registers spilled once per byte, measured harmless)
    for name, r in t.items():
        if "k_lanes" in name and name.endswith("Lb1EEEv6DBatchNS_4LCfgE"):   # the hash-chain instantiation (no interpreter)
            assert r["scratch"] == 0, (name, r)
