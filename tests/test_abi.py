"""The C-ABI library: builds, loads without a GPU, exports exactly what include/dusp_hip.h declares,
and fails loudly (no CPU fallback) when there is no device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dusp_amd import runtime

HEADER = os.path.join(ROOT, "include", "dusp_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dusp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(runtime.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), "libdusp_hip.so does not export " + name
    assert sorted(runtime.EXPORTS) == names


def test_version_and_abi():
    L = runtime.load()
    assert L.dusp_abi_version() == 7
    assert b"gfx950" in L.dusp_version()


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(runtime.DuspHipError, match="no usable HIP device|HIP error"):
        runtime.Context()


def test_repeat_add_equals_the_plain_loop(tmp_path):
    """dusp_amd/csrc/repeat_add.hpp (Timer's and Shape's running sums n steps at once; what lets the wave engine evaluate
    them lane-parallel and split a render in time) against sequential f64 additions, on the CPU: 40 000 random cases."""
    import json
    import subprocess
    exe = str(tmp_path / "repeat_add_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "repeat_add_check.cpp")])
    rep = json.loads(subprocess.check_output([exe, "40000"]).decode().strip().splitlines()[-1])
    assert rep["cases"] == 40000 and rep["bad"] == 0 and rep["linear_runs"] > 10000  # (linear_run: whole chunks inside one binade, checked sample by sample)


def test_descriptor_sizes_are_bounded_before_anything_is_allocated(tmp_path):
    """dusp_amd/csrc/program.hpp on the CPU: a 2^40-sample ring, a 2^40 maxDelay, delay lines that add up past 2^31 samples
    and unit / ring counts the descriptor cannot hold all come back as error strings (-> a dusp_status), no exception."""
    import json
    import subprocess
    exe = str(tmp_path / "compile_bounds_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "compile_bounds_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] == 6 and rep["bad"] == 0


def test_filter_coefficients_are_as_close_to_exact_as_the_reference_arithmetic(tmp_path):
    """dusp_amd/csrc/filter_lamda.hpp (what EVERY engine computes the Butterworth coefficients of Filter.js:66-84 with: one division, no
    tan() for cutoffs below Nyquist) against the reference's expressions in extended precision: 4.8 million cutoffs, every coefficient;
    zero / negative / above-Nyquist cutoffs against the expressions as written over the math library's tan."""
    import json
    import subprocess
    exe = str(tmp_path / "filter_lamda_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "filter_lamda_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] > 4_000_000 and rep["bad"] == 0
    assert rep["worst_ulp"] <= 6.0 and max(rep["worst_lp"][:3] + rep["worst_hp"][:3]) <= 4.5, rep
    assert rep["outside_rel"] <= 1e-9, rep


def test_the_lerp_in_its_delta_form_is_the_reference_expression(tmp_path):
    """dusp_amd/csrc/device_util.hpp lerp_delta / Table::pair_delta (one fma on T[i] and T[i+1] - T[i], what the compiled kernels' and the fused
    engine's oscillators evaluate when every phase is a multiple of 2^-28) against Osc.js:43-46 as written, at every index of the sine table
    at four sample rates, and the table classification (table_checks.hpp) the choice of form rests on: 48 kHz takes its differences in f64
    (class 1: two pairs next to the middle differ by 25 bits), 44.1 kHz in f32 (class 2)."""
    import json
    import subprocess
    exe = str(tmp_path / "lerp_delta_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "lerp_delta_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] > 2_000_000 and rep["bad"] == 0, rep
    assert rep["classes"] == [1, 2, 1, 2], rep


def test_increments_reach_fixed_point_through_the_f64_mantissa_exactly(tmp_path):
    """dusp_amd/csrc/jit_prelude.hpp jit_fix36 (a scanned oscillator's f32 increments as 2^-36 fixed point: trunc(f 2^36) read out of the
    mantissa field of |.| + 2^52) against the C cast it stands for, over every exponent below 2^16, both signs, edge and random mantissas."""
    import json
    import subprocess
    exe = str(tmp_path / "fix36_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "fix36_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] > 10_000 and rep["bad"] == 0, rep


def test_ring_windows_cover_what_the_units_touch(tmp_path):
    """dusp_amd/csrc/ring_windows.hpp (which slots of its delay rings a render that nothing continues zero-fills) on the CPU: 600 random
    Delays / MonoDelays / CircleBuffer nodes / ReadBackDelays, every slot the reference's unit touches enumerated sample by sample and
    found inside a window, every window inside its ring; per-instance, connected and out-of-range delays, the comb family and rings
    shorter than their window take the whole ring."""
    import json
    import subprocess
    exe = str(tmp_path / "ring_windows_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "ring_windows_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] >= 1200 and rep["bad"] == 0


def test_the_filter_scan_stays_within_its_stated_bound(tmp_path):
    """The arithmetic of jit_prelude.hpp JitFilterScan restated on the CPU (unrounded pairs in front of every lane's four samples, the
    lane's samples rounded as the reference rounds them) against Filter.js:40-46 as written: ten seconds of white noise and of a sine
    per admitted cutoff, both kinds — the deviation never exceeds 2^-24 (sum|h| + 2) max|y|, nor 1.9e-6 of the signal's scale."""
    import json
    import subprocess
    exe = str(tmp_path / "filter_scan_bound_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "native", "filter_scan_bound_check.cpp")])
    rep = json.loads(subprocess.check_output([exe]).decode().strip().splitlines()[-1])
    assert rep["cases"] >= 30 and rep["bad"] == 0 and rep["worst_of_bound"] <= 1.0 and rep["worst_of_scale"] <= 1.9e-6
    # ... and inside configs[3]'s feedback loop at gains 0.5 .. 0.99, resonant and other inputs: within the loop-gain bound of jit_filter_scan_ok (2)
    assert rep["loop_cases"] == 24 and rep["loop_bad"] == 0 and rep["loop_worst_of_bound"] <= 1.0 and rep["loop_worst_of_scale"] <= 1e-5
