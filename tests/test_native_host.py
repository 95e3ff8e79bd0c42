"""CPU checks of device-side arithmetic, compiled for the host from the same headers / the same text:
the wave-FFT index math (audiomod_amd/csrc/pv_wavefft.h, both the in-place and the prefetched-twiddle entry
points) and the divide-free princarg.  No GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def _build_and_run(tmp_path, sources, name):
    exe = str(tmp_path / name)
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{ROOT}/audiomod_amd/csrc"]
    cmd += [os.path.join(ROOT, s) for s in sources] + ["-o", exe]
    subprocess.run(cmd, check=True)
    return subprocess.run([exe], capture_output=True, text=True)


def test_wave_fft_bit_exact_on_host(tmp_path):
    r = _build_and_run(tmp_path, ["tests/native/host_wavefft.cc", "audiomod_amd/csrc/pv_plan.cc"], "host_wavefft")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("bit-exact") == 10, r.stdout     # 256, 512, 1024, 2048 complex points and 2048 on two waves x forward, inverse


def test_princarg_small_matches_reference_expression(tmp_path):
    r = _build_and_run(tmp_path, ["tests/native/host_princarg.cc"], "host_princarg")
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout, r.stdout
    # the device function must be the same text as the one swept on the host
    dev = open(os.path.join(ROOT, "audiomod_amd/csrc/pv_kernels.hip")).read()
    for line in ("const double yn = x > 0.0 ? (x > Y ? 2.0 * Y : Y) : 0.0;", "return (x - yn) + PV_PI;"):
        assert line in dev


def test_host_planner_under_sanitizers(tmp_path):
    """derive / Planner / plan_batch / carrier / whisper generators over ~900 configurations under ASan + UBSan"""
    exe = str(tmp_path / "host_plan_sanitize")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-ffp-contract=off", f"-I{ROOT}/include", f"-I{ROOT}/audiomod_amd/csrc",
           os.path.join(ROOT, "tests/native/host_plan_sanitize.cc"), os.path.join(ROOT, "audiomod_amd/csrc/pv_plan.cc"),
           "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower()):
        pytest.skip("sanitizer runtimes not installed")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failures" in r.stdout, r.stdout


def test_atan2f_restatement_is_bit_identical_to_libm(tmp_path):
    """The analysis kernels take their phases from pv_atan2f_fd (glibc <= 2.40's fdlibm atan2f restated), because a
    one-ulp phase difference can flip a princarg wrap in the phase propagation and change the output audibly.  The
    same header, compiled for the host, against the C library's atan2f: 40 M random pairs + every threshold."""
    r = _build_and_run(tmp_path, ["tests/native/host_atan2f.cc"], "host_atan2f")
    if r.returncode != 0 and " 0 mismatches" not in r.stdout:
        import platform
        libc = platform.libc_ver()
        if libc[0] == "glibc" and tuple(int(v) for v in libc[1].split(".")[:2]) >= (2, 41):
            pytest.skip("glibc >= 2.41 computes atan2f differently (correctly rounded): not the reference build's libm")
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
    dev = open(os.path.join(ROOT, "audiomod_amd/csrc/pv_kernels.hip")).read()
    code = "\n".join(ln.split("//")[0] for ln in dev.splitlines())  # comments may mention libm's atan2f
    assert "pv_atan2f_fd_finite(" in code and " atan2f(" not in code and "(atan2f(" not in code


def test_short_sincos_against_double(tmp_path):
    """The synthesis kernels' sine / cosine for wrapped phases (audiomod_amd/csrc/pv_sincos.h: two-term reduction, own
    polynomials, bit-operation quadrant step) against double-precision sin / cos: 40 M samples of [-pi, pi] and of the
    whole accepted range, every float near each multiple of pi/4, special values; bar 2 ulp or 1.2e-7 absolute."""
    r = _build_and_run(tmp_path, ["tests/native/host_sincos.cc"], "host_sincos")
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
    dev = open(os.path.join(ROOT, "audiomod_amd/csrc/pv_kernels.hip")).read()
    assert "pv_sincos_small(" in dev and "PV_SINCOS_MAX_ARG" in dev


def test_division_free_princarg_on_every_float(tmp_path):
    """princarg_f (pv_kernels.hip): the quotient of the reference's princarg formed with two fma's on the constant
    reciprocal instead of an IEEE division -- every call site passes a float, so quotient and result are compared with
    the reference expression on all 2^32 arguments (tests/native/host_princarg_all.cc; ~40 CPU-seconds, threaded)."""
    exe = str(tmp_path / "host_princarg_all")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread",
                    os.path.join(ROOT, "tests/native/host_princarg_all.cc"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 result mismatches, 0 quotient mismatches" in r.stdout, r.stdout + r.stderr
    dev = open(os.path.join(ROOT, "audiomod_amd/csrc/pv_kernels.hip")).read()
    for line in ("const double q0 = x * inv_y;", "const double r = __builtin_fma(-y, q0, x);",
                 "const double q1 = __builtin_fma(r, inv_y, q0);", "return (x - (y * floor(q1))) + PV_PI;",
                 "constexpr double inv_y = 1.0 / (-2.0 * PV_PI);"):
        assert line in dev, line
    code = "\n".join(ln.split("//")[0] for ln in dev.splitlines())
    assert "princarg_div(" not in code and " princarg(" not in code  # no other wrap of an unbounded argument is left
