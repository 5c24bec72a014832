"""Builds and runs the C++ tests of the host-side mirror of the reference interface
(cddp-cpp_amd/host/cddp_hip.hpp): registry / dispatch / error conventions on CPU, solves on the GPU."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "cddp-cpp_amd", "build", "test_host_api")


def build_exe():
    lib = os.path.join(REPO, "cddp-cpp_amd", "lib", "libcddp_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(REPO, "tests", "cpp", "test_host_api.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", src, "-o", EXE, "-L" + os.path.dirname(lib), "-lcddp_hip",
                               "-Wl,-rpath," + os.path.dirname(lib), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"])
    return EXE


def test_host_api_cpu():
    exe = build_exe()
    out = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_host_api_gpu():
    exe = build_exe()
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
