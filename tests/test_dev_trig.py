"""The device sin / cos of the plants (cddp-cpp_amd/csrc/dev_trig.hpp) compiled for the host: error against long-double libm
below 1 ulp on 1.4e6 arguments up to 1e9 rad, near the multiples of pi/2 and for tiny arguments; out-of-range and non-finite
arguments fall back to the libm.  (The reference calls std::sin / std::cos of glibc; the parity tests bridge <= 1 ulp.)"""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_sincos_is_within_one_ulp(tmp_path):
    exe = str(tmp_path / "test_dev_trig")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(REPO, "tests", "cpp", "test_dev_trig.cpp")])
    out = subprocess.run([exe, "200000"], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout
    ms, mc = (float(v) for v in out.stdout.split()[:2])
    assert ms < 0.85 and mc < 0.85


def test_shared_log_exp_pow_accuracy(tmp_path):
    """The parity build's log / exp / pow (dev_trig.hpp: log_shared, exp_fast, pow_shared) against long-double libm: log and exp
    below 1 ulp over the normal range / |x| <= 700, pow = exp(y log x) within 64 ulp for mu in [1e-10, 10], y in (0, 2) (a barrier
    parameter; what matters is that both sides of the parity comparison evaluate the SAME routine); out-of-range arguments fall
    back to the libm."""
    exe = str(tmp_path / "test_dev_elem")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(REPO, "tests", "cpp", "test_dev_elem.cpp")])
    out = subprocess.run([exe, "200000"], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout
    ml, me, mp = (float(v) for v in out.stdout.split()[:3])
    ma = float(out.stdout.split()[5])      # asin on |x| < 0.5 (the car's steering kinematics)
    assert ml < 0.9 and me < 0.95 and mp < 64.0 and ma < 0.8
