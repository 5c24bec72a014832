"""Extracts the INPUT DATA of the reference's BoxQP tests (tests/cddp_core/test_boxqp.cpp: the 5-variable QP of ComparisonTest and the
15-variable Q, q of LargeDimensionTest; the tests themselves assert nothing, they print) into tests/golden/ref_boxqp_inputs.json.
Run in the build container (needs /root/reference); the JSON is what travels."""
import json
import os
import re

SRC = "/root/reference/tests/cddp_core/test_boxqp.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_boxqp_inputs.json")
NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def block(text, name, start):
    """numbers of the first `name << ... ;` statement after offset start (comments stripped)"""
    m = re.compile(r"\b" + name + r"\s*<<(.*?);", re.S).search(text, start)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [float(v) for v in re.findall(NUM, body)], m.end()


text = open(SRC).read()
i5 = text.index("TEST(QPSolver, ComparisonTest)")
i15 = text.index("TEST(BoxQPSolver, LargeDimensionTest)")
Q5, e = block(text, "Q", i5); q5, _ = block(text, "q", e)
Q15, e = block(text, "Q", i15); q15, _ = block(text, "q", e)
assert len(Q5) == 25 and len(q5) == 5 and len(Q15) == 225 and len(q15) == 15
json.dump({"source": "tests/cddp_core/test_boxqp.cpp (input data only)",
           "cases": [{"name": "ComparisonTest_5", "n": 5, "Q": Q5, "q": q5, "lower": [0.0] * 5, "upper": [2.0] * 5},
                     {"name": "LargeDimensionTest_15", "n": 15, "Q": Q15, "q": q15, "lower": [-2.0] * 15, "upper": [2.0] * 15}]},
          open(OUT, "w"), indent=0)
print("wrote", OUT)
