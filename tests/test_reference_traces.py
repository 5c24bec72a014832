"""The oracle against traces of a cddp-cpp BINARY (tests/golden/ref_trace_*.json, written by oracle/ref_pin/compare_traces.py
--write-fixtures on a machine that has Eigen 3.4.0 + autodiff v1.1.2).  None exist yet: this image cannot build the reference, so
the test below reports "parity unpinned" by skipping -- it starts to bite the moment the fixtures are committed."""
import glob
import importlib.util
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_trace_*.json")))


def test_recipe_is_complete_and_refuses_to_build_without_the_real_dependencies():
    """The pin recipe exists, and build_ref.sh fails loudly -- it does not stand in for Eigen / autodiff -- when they are absent."""
    import subprocess
    d = os.path.join(os.path.dirname(HERE), "oracle", "ref_pin")
    for f in ("build_ref.sh", "dump_traces.cpp", "compare_traces.py"):
        assert os.path.exists(os.path.join(d, f)), f
    env = {k: v for k, v in os.environ.items() if k not in ("EIGEN3_INCLUDE_DIR", "AUTODIFF_INCLUDE_DIR")}
    out = subprocess.run(["bash", os.path.join(d, "build_ref.sh")], env=env, capture_output=True, text=True)
    assert out.returncode == 2 and ("EIGEN3_INCLUDE_DIR" in out.stderr or "reference sources not found" in out.stderr)


@pytest.mark.skipif(not FIXTURES, reason="parity unpinned: no cddp-cpp binary traces committed (oracle/ref_pin/ is the recipe; needs Eigen 3.4.0 + autodiff 1.1.2)")
@pytest.mark.parametrize("path", FIXTURES or ["none"])
def test_oracle_reproduces_the_reference_binary(api, oracle_built, path):
    spec = importlib.util.spec_from_file_location("compare_traces", os.path.join(os.path.dirname(HERE), "oracle", "ref_pin", "compare_traces.py"))
    ct = importlib.util.module_from_spec(spec); spec.loader.exec_module(ct)
    ref = json.load(open(path))
    for name, solver, p, x0s in ct.cases(api):
        if name == ref["case"] and solver == ref["solver"]:
            bad = ct.compare_one(api, name, solver, p, ref["x0"], ref)
            assert not bad, (path, bad)
            return
    raise AssertionError("no case for " + path)
