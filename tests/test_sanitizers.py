"""Sanitizer coverage on the CPU (no GPU AddressSanitizer on this pool): the library's untrusted-input path — descriptor parser, planners,
the circuit compiler's text generator — and the oracle, both under AddressSanitizer + UndefinedBehaviorSanitizer."""
import glob
import json
import os
import subprocess
import sys

from conftest import GOLDEN, ROOT


def test_descriptor_path_under_asan_and_ubsan():
    """make -C dusp_amd/csrc hostcheck: program.hpp + fused_plan.hpp + jit_codegen.hpp + ring_windows.hpp compiled by g++ with
    -fsanitize=address,undefined -fno-sanitize-recover=all, driven (tests/native/hostcheck.cpp) by every golden descriptor over a spread of
    workgroup geometries and knobs, by their truncations and by single- and double-word corruptions: every call ends in a verdict with a
    message (kernel text / malformed / unsupported), and the sanitizers — leak detection included — have nothing to report."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dusp_amd", "csrc"), "-s", "hostcheck"])
    exe = os.path.join(ROOT, "dusp_amd", "csrc", "build", "hostcheck")
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.desc.f64")))
    assert len(files) >= 200
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", HOSTCHECK_CORRUPTIONS="40")
    p = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, (out[-2000:], err[-4000:])
    rep = json.loads(out.strip().splitlines()[-1])
    assert rep["files"] == len(files) and rep["bad"] == 0
    assert rep["text"] > 10000 and rep["malformed"] > 40000 and rep["unsupported"] > 500  # (all three verdicts are exercised)


def test_oracle_under_asan_and_ubsan():
    """oracle/libdusp_oracle_asan.so (make -C oracle asan) renders every golden vector of tests/test_oracle_golden.py in a child
    interpreter with the sanitizer run time preloaded: same PCM, nothing reported."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               DUSP_ORACLE_LIB=os.path.join(ROOT, "oracle", "libdusp_oracle_asan.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=1500)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0 and "passed" in out, (out[-3000:], err[-3000:])
    assert "Sanitizer" not in err and "runtime error" not in err and "Sanitizer" not in out, (out[-2000:], err[-3000:])
