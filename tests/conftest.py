import importlib.util
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_api():
    """Import cddp-cpp_amd/pyapi.py (the directory name is not a valid Python identifier)."""
    name = "cddp_cpp_amd_pyapi"
    if name in sys.modules:
        mod = sys.modules[name]
        if not hasattr(mod, "ORACLE_LIB_PATH"):   # first imported by the product facade (pycddp_amd), which never loads the checker
            load_oracle_api().attach(mod)
        return mod
    # PyTorch bundles its own ROCm runtime: a process that uses both must import torch FIRST, so that libcddp_hip.so binds the
    # runtime torch already loaded (bench.py does the same).  Only when a GPU test session will want it.
    if os.path.exists("/dev/kfd"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "cddp-cpp_amd", "pyapi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    load_oracle_api().attach(mod)     # the checker: Oracle, oracle_solve_batch, ... on the same namespace
    return mod


def load_oracle_api():
    """Import oracle/oracle_api.py -- test infrastructure, never loaded by the product package."""
    name = "cddp_oracle_api"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "oracle", "oracle_api.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "host_arithmetic: gpu test of the host (plug-in) route -- the oracle stays in glibc arithmetic (trig_mode 0)")


@pytest.fixture(scope="session")
def api():
    return load_api()


@pytest.fixture(autouse=True)
def _oracle_arithmetic(request):
    """Every `-m gpu` test compares the HIP library with the oracle in the library's own arithmetic: the plants' sin / cos and the
    solver core's log / pow of cddp-cpp_amd/csrc/dev_trig.hpp (oracle trig_mode 1) -- since round 4 the ONE shipped library is built
    that way, so the comparison is strict.  CPU tests (oracle vs the numpy twin, golden fixtures, reference pins) keep mode 0: glibc,
    the reference's own arithmetic."""
    # ... except the tests of the library's HOST route (`host_arithmetic` marker: cddp_hip_plugin_solve and the stack-fed sweeps a host
    # loop drives -- user callbacks, host forward passes, host log / pow): host code calls the host libm, i.e. glibc, like the reference.
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("host_arithmetic") is not None:
        yield
        return
    load_api()                      # (a worker whose first gpu test does not use the `api` fixture: the checker binds to that module)
    oa = load_oracle_api()
    prev = oa.set_trig_mode(1)
    try:
        yield
    finally:
        oa.set_trig_mode(prev)


@pytest.fixture(scope="session")
def oracle_built(api):
    """Make sure the oracle shared library exists (built by __graft_entry__.build())."""
    if not os.path.exists(api.ORACLE_LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    return True
