"""tools/ab_balanced.py (EXPERIMENTS R4.11): the balanced order, the parser of quickbench's lines and the comparison at equal
reference time -- host logic, no GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ab_balanced", os.path.join(ROOT, "tools", "ab_balanced.py"))
AB = importlib.util.module_from_spec(spec)
spec.loader.exec_module(AB)

REC = """== lib=shipped
rep1 L2 8192x65536: enc 0.123s (4371.0 MB/s, kernel 199.9 ms k_pipe<encode>)  dec 0.235s (2285.7 MB/s, kernel 299.9 ms)  slots 8192
rep2 L2 8192x65536: enc 0.123s (4375.7 MB/s, kernel 122.7 ms k_pipe<encode>)  dec 0.235s (2286.4 MB/s, kernel 237.1 ms)  slots 8192
status ok/roundtrip equal: True  ratio 0.3983
== lib=var
rep2 L2 8192x65536: enc 0.120s (4463.0 MB/s, kernel 120.3 ms k_pipe<encode>)  dec 0.232s (2312.6 MB/s, kernel 232.1 ms)  slots 8192
== lib=var
rep2 L2 8192x65536: enc 0.123s (4366.6 MB/s, kernel 122.9 ms k_pipe<encode>)  dec 0.236s (2277.9 MB/s, kernel 237.5 ms)  slots 8192
== lib=shipped
rep2 L2 8192x65536: enc 0.120s (4457.4 MB/s, kernel 120.4 ms k_pipe<encode>)  dec 0.232s (2313.1 MB/s, kernel 232.3 ms)  slots 8192
"""


def test_balanced_order_gives_every_arm_odd_and_even_places():
    assert AB.balanced_order(["a", "b"], 4) == ["a", "b", "b", "a", "a", "b", "b", "a"]
    order = AB.balanced_order(["a", "b", "c"], 2)
    assert order == ["a", "b", "c", "c", "b", "a"]
    for arm in "abc":
        places = [i % 2 for i, x in enumerate(order) if x == arm]
        assert sorted(places) == [0, 1]


def test_parse_takes_the_last_rep_of_every_process():
    s = AB.parse(REC)
    assert s == [("shipped", 122.7, 237.1), ("var", 120.3, 232.1), ("var", 122.9, 237.5), ("shipped", 120.4, 232.3)]


def test_arms_are_compared_at_equal_reference_time():
    """A naive mean would call 'var' 0 % different here only by luck of the order; an alternating order (shipped always in the slow
    place) would have called it 2 % faster.  Split by the unchanged encoder's time, the two arms are equal in both places."""
    res, text = AB.report(AB.parse(REC), "enc")
    assert res["shipped"]["fast"] == (232.3, 1) and res["shipped"]["slow"] == (237.1, 1)
    assert res["var"]["fast"] == (232.1, 1) and res["var"]["slow"] == (237.5, 1)
    assert "median" in text
    res, _ = AB.report(AB.parse(REC), "dec")                   # the other way round: the encoder under test
    assert res["var"]["fast"][0] == 120.3 and res["shipped"]["slow"][0] == 122.7
