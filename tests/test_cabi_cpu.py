"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/cddp_hip.h declares; host-only entry points behave; there is NO CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(api):
    if not os.path.exists(api.HIP_LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return api.load_hip()


def declared_symbols():
    txt = open(os.path.join(REPO, "include", "cddp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^#define[^\n]*(\\\n[^\n]*)*", "", txt, flags=re.M)   # function-like macros (cddp_hip_stacks_create -> _abi) are not symbols
    return sorted(set(re.findall(r"\b(cddp_hip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(api, lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libcddp_hip.so does not export %s" % s
    assert set(api.EXPORTED_SYMBOLS) == set(syms)


def test_abi_version_and_status_strings(api, lib):
    assert lib.cddp_hip_abi_version() == api.ABI_VERSION == 5
    want = ["Running", "OptimalSolutionFound", "AcceptableSolutionFound", "MaxIterationsReached",
            "RegularizationLimitReached_NotConverged", "MaxCpuTimeReached"]
    for i, w in enumerate(want):
        assert lib.cddp_hip_status_string(i).decode() == w   # status strings are API


def test_struct_sizes_match_header(api):
    assert C.sizeof(api.Result) == 96
    assert api.RESULT_DTYPE.itemsize == 96
    assert api.GATHER_DTYPE.itemsize == 16
    assert api.TRIAL_DTYPE.itemsize == 72
    assert C.sizeof(api.Stats) == 88


def test_default_options_match_reference_defaults(api, lib):
    o = api.Options()
    lib.cddp_hip_default_options(C.byref(o))
    d = api.default_options()
    for name, _ in api.Options._fields_:
        assert getattr(o, name) == getattr(d, name), name
    # include/cddp-cpp/cddp_core/options.hpp:41-251
    assert o.tolerance == 1e-5 and o.acceptable_tolerance == 1e-6 and o.max_iterations == 1
    assert o.ls_max_iterations == 11 and o.reg_initial_value == 1e-6 and o.reg_max_value == 1e7
    assert o.barrier_mu_initial == 1.0 and o.barrier_min_fraction_to_boundary == 0.99
    assert o.boxqp_max_iterations == 100 and o.boxqp_min_step_size == 1e-22


def test_build_alphas_matches_reference_ladder(api, lib):
    o = api.default_options()
    a = np.zeros(32)
    n = lib.cddp_hip_build_alphas(C.byref(o), a.ctypes.data_as(C.POINTER(C.c_double)), 32)
    assert n == 11 and np.allclose(a[:n], 0.5 ** np.arange(11))
    o.ls_max_iterations = 40
    n = lib.cddp_hip_build_alphas(C.byref(o), a.ctypes.data_as(C.POINTER(C.c_double)), 32)
    assert a[n - 1] == 1e-8 and n < 32
    # and it agrees with the oracle's ladder
    p = api.pendulum_problem(); p.options.ls_max_iterations = 40
    assert np.array_equal(api.Oracle(p).alphas(), a[:n])


def test_no_cpu_fallback(api, lib):
    """Without a GPU the product refuses to run (it must not route through the oracle)."""
    if lib.cddp_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    p = api.cartpole_problem()
    with pytest.raises(api.HipError) as e:
        api.HipBatchSolver(p, 4)
    assert "no CPU fallback" in str(e.value)


def test_product_sources_do_not_reference_oracle():
    """oracle/ is test infrastructure: nothing under the package may include, link or load it."""
    root = os.path.join(REPO, "cddp-cpp_amd")
    for dp, _, fns in os.walk(root):
        if "build" in dp or "__pycache__" in dp:
            continue
        for fn in fns:
            if not fn.endswith((".hip", ".hpp", ".h", ".cpp", ".py", "Makefile")):
                continue
            txt = open(os.path.join(dp, fn), errors="ignore").read()
            assert "oracle/" not in txt and "cddp_oracle" not in txt, os.path.join(dp, fn)


def test_plugin_and_stack_entry_points_refuse_another_abi(api, lib):
    """cddp_hip_plugin_solve and the stack handles copy a cddp_hip_options from a caller pointer; since ABI 5 the caller
    states the version and the struct size it was built against and a mismatch is refused BEFORE anything is read (no GPU needed)."""
    import ctypes as C
    h = C.c_void_p()
    for abi, nbytes in ((api.ABI_VERSION - 1, C.sizeof(api.Options)), (api.ABI_VERSION, C.sizeof(api.Options) - 8)):
        rc = lib.cddp_hip_stacks_create_abi(abi, nbytes, 0, 4, 2, 1, 0, 10, C.byref(h))
        assert rc == -2 and b"ABI mismatch" in lib.cddp_hip_last_error()
    ps = api.PluginStruct()
    ps.abi_version = api.ABI_VERSION - 1; ps.options_bytes = C.sizeof(api.Options)
    o = api.default_options()
    x0 = (C.c_double * 2)(0.0, 0.0)
    res = (C.c_char * 4096)()
    rc = lib.cddp_hip_plugin_solve(C.byref(ps), api.SOLVER_IPDDP, 10, C.c_double(0.1), C.byref(o), 0, 1, x0, None, None, res, None, None, None)
    assert rc == -2 and b"ABI mismatch" in lib.cddp_hip_last_error()


def test_rccl_declarations_restated_in_comm_hip_match_the_installed_header():
    """comm.hip declares the handful of RCCL types / entry points it binds with dlopen itself (the solver core builds without the RCCL headers).
    Where rccl.h is installed, tests/cpp/test_rccl_abi.cpp holds those restatements to it at compile time: id size, enum values, signatures."""
    import shutil, subprocess
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr) or shutil.which("g++") is None:
        pytest.skip("no RCCL header / g++ here")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(repo, "tests", "cpp", "test_rccl_abi.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # the restated values the check pins are the ones comm.hip actually uses
    src = open(os.path.join(repo, "cddp-cpp_amd", "csrc", "comm.hip")).read()
    assert "char internal[128]" in src and "ncclUint8 = 1" in src and "ncclSuccess = 0" in src
