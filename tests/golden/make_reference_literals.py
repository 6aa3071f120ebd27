#!/usr/bin/env python3
"""Extract the reference's own LITERAL data on the hot path into tests/golden/reference_literals.json.

Run in the build container (the only place /root/reference exists):

    python tests/golden/make_reference_literals.py [/root/reference]

The reference (dy-tea/zpaq-v, V source) cannot be compiled here, but three kinds of data on the coder's path are
written out in its source as literals and can be read as text -- this is the only reference-held data that
covers the hot path's tables:

  * state_table_data[1024]  zpaq/statetable.v:15-57   (the bit-history state machine `ns`)
  * dt_table[1024]          zpaq/predictor.v:111-166  (the CM adaptation-rate table `dt`)
  * the `hcomp:` byte arrays of levels 0-5 with their hh/hm fields   zpaq/levels.v:40-375
  * compsize[10]            zpaq/types.v:74-85        (COMP record lengths)
  * zpaq_block_locator      zpaq/compressor.v:12-13   (the 13-byte block locator of the framing)

The oracle (oracle/zpaq_oracle.c), the second restatement (oracle/pyref) and the product (zpq_model.cpp) all
REGENERATE ns / dt / the level headers from constructions and formulas; tests/test_reference_literals.py holds
all three equal to what this file extracted.  The fixture is DATA (numbers), no source text is stored.
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def int_list(body):
    """Numbers of a V array literal body: `u8(1), 2, 0x37, int(87380), ...`."""
    body = re.sub(r"\b(?:u8|int|u32|u16|i32)\(\s*([^)]+?)\s*\)", r"\1", body)
    out = []
    for tok in body.replace("\n", " ").split(","):
        tok = tok.strip()
        if not tok:
            continue
        out.append(int(tok, 0))
    return out


def const_array(text, name):
    m = re.search(r"const\s+" + re.escape(name) + r"\s*=\s*\[(.*?)\]", text, re.S)
    if not m:
        raise SystemExit("literal %s not found" % name)
    return int_list(m.group(1))


def level_functions(text):
    """{level: function name} from get_compression_level's match (levels.v:26-36)."""
    m = re.search(r"fn get_compression_level\(.*?\{(.*?)\n\}", text, re.S)
    out = {}
    for lv, fn in re.findall(r"(\d)\s*\{\s*(\w+)\(\)\s*\}", m.group(1)):
        out[int(lv)] = fn
    return out


def level_literal(text, fn):
    m = re.search(r"fn\s+" + re.escape(fn) + r"\(\)\s*CompressionLevel\s*\{(.*?)\n\}", text, re.S)
    if not m:
        raise SystemExit("function %s not found" % fn)
    body = m.group(1)
    h = re.search(r"hcomp:\s*\[(.*?)\]", body, re.S)
    name = re.search(r"name:\s*'([^']*)'", body)
    hh = re.search(r"\bhh:\s*(\d+)", body)
    hm = re.search(r"\bhm:\s*(\d+)", body)
    return {"name": name.group(1), "hcomp": int_list(h.group(1)), "hh": int(hh.group(1)), "hm": int(hm.group(1))}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    z = os.path.join(ref, "zpaq")

    def read(fn):
        with open(os.path.join(z, fn), "r", encoding="utf-8") as f:
            return strip_comments(f.read())

    st, pr, lv, ty, co = read("statetable.v"), read("predictor.v"), read("levels.v"), read("types.v"), read("compressor.v")
    ns = const_array(st, "state_table_data")
    dt = const_array(pr, "dt_table")
    compsize = const_array(ty, "compsize")
    locator = const_array(co, "zpaq_block_locator")
    assert len(ns) == 1024 and all(0 <= v <= 255 for v in ns), len(ns)
    assert len(dt) == 1024, len(dt)
    assert len(compsize) == 10 and len(locator) == 13
    fns = level_functions(lv)
    assert sorted(fns) == [0, 1, 2, 3, 4, 5], fns
    levels = {str(k): dict(level_literal(lv, fn), function=fn) for k, fn in sorted(fns.items())}
    out = {
        "note": "literal data read from the reference's V source by tests/golden/make_reference_literals.py; data only",
        "sources": {"state_table_data": "zpaq/statetable.v:15-57", "dt_table": "zpaq/predictor.v:111-166",
                    "levels": "zpaq/levels.v:26-375", "compsize": "zpaq/types.v:74-85",
                    "block_locator": "zpaq/compressor.v:12-13"},
        "state_table_data": ns,
        "state_table_sha256": hashlib.sha256(bytes(ns)).hexdigest(),
        "dt_table": dt,
        "compsize": compsize,
        "block_locator": locator,
        "levels": levels,
    }
    path = os.path.join(HERE, "reference_literals.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, "ns", len(ns), "dt", len(dt), "levels", {k: len(v["hcomp"]) for k, v in levels.items()})


if __name__ == "__main__":
    main()
