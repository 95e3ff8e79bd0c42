import os
"""The oracle (oracle/pv_oracle.c) against the golden vectors captured from the compiled
reference (tools/make_golden.py).  Bit-exact: same x86-64 arithmetic model, same libm."""
import numpy as np
import pytest

from oracle import oracle_py as O
from tests.helpers import GOLD, bits_equal, e2e_cases, load_e2e


@pytest.mark.parametrize("name", e2e_cases())
def test_e2e_bit_exact(name):
    x, y, counts, meta = load_e2e(name)
    api = meta.pop("api", "offline")
    if api == "rt":
        got, cnt = O.run_realtime(x, **meta)
    else:
        got, cnt, _ = O.run_offline(x, **meta)
    assert cnt == counts
    assert bits_equal(got, y)


@pytest.fixture(scope="module")
def kat():
    return np.load(f"{GOLD}/kat_units.npz")


@pytest.mark.parametrize("N", [2048, 4096])
def test_hann(kat, N):
    L = O.lib()
    w = np.zeros(N, np.float32)
    area = np.zeros(1, np.float32)
    L.pvo_hann(N, w.ctypes.data, area.ctypes.data)
    assert bits_equal(w, kat[f"hann{N}"])
    assert bits_equal(area, kat[f"hann{N}_area"])


@pytest.mark.parametrize("N", [2048, 4096])
def test_forward_polar(kat, N):
    L = O.lib()
    H = N // 2 + 1
    for frame, want in zip(kat[f"fwd{N}_in"], kat[f"fwd{N}_magphase"]):
        mag = np.zeros(H, np.float32)
        ph = np.zeros(H, np.float32)
        L.pvo_forward_polar(N, np.ascontiguousarray(frame).ctypes.data, mag.ctypes.data, ph.ctypes.data)
        assert bits_equal(mag, want[0])
        assert bits_equal(ph, want[1])


@pytest.mark.parametrize("N", [2048, 4096])
def test_inverse_polar(kat, N):
    L = O.lib()
    for pol, want in zip(kat[f"inv{N}_in"], kat[f"inv{N}_out"]):
        out = np.zeros(N, np.float32)
        L.pvo_inverse_polar(N, np.ascontiguousarray(pol[0]).ctypes.data, np.ascontiguousarray(pol[1]).ctypes.data,
                            out.ctypes.data)
        assert bits_equal(out, want)


@pytest.mark.parametrize("tag", ["+4", "+7", "-7", "+12"])
def test_resampler(kat, tag):
    L = O.lib()
    sig = np.ascontiguousarray(kat["res_in"])
    ratio = float(kat[f"res{tag}_ratio"][0])
    r = L.pvo_res_create()
    outs = []
    pos = 0
    for c, want_n in kat[f"res{tag}_chunks"]:
        buf = np.zeros(int(c) * 4 + 64, np.float32)
        chunk = np.ascontiguousarray(sig[pos:pos + c])
        got = L.pvo_res_process(r, chunk.ctypes.data, int(c), ratio, buf.ctypes.data)
        assert got == want_n
        outs.append(buf[:got])
        pos += int(c)
    L.pvo_res_destroy(r)
    assert bits_equal(np.concatenate(outs), kat[f"res{tag}_out"])


@pytest.mark.parametrize("N", [2048, 4096])
@pytest.mark.parametrize("tag", ["+4", "-7", "1"])
def test_cepstral_formant_shift(kat, N, tag):
    """formantShiftSlice (dead code upstream, called directly by oracle/_ref/ref_formant): bit-exact"""
    L = O.lib()
    mags = kat[f"formant{N}_in"].copy()
    env = float(kat[f"formant{N}_{tag}_env"][0])
    for r in range(mags.shape[0]):
        L.pvo_formant_shift(N, mags[r].ctypes.data, env)
    assert bits_equal(mags, kat[f"formant{N}_{tag}_out"])


@pytest.mark.parametrize("N", [256, 512, 1024, 8192])
@pytest.mark.parametrize("tag", ["+5", "-9", "1"])
def test_cepstral_formant_shift_other_sizes(N, tag):
    """the same at the FFT sizes the engine serves through its generic kernels (tools/make_golden_formant_sizes.py)"""
    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_formant_sizes.npz"))
    L = O.lib()
    mags = kat[f"formant{N}_in"].copy()
    env = float(kat[f"formant{N}_{tag}_env"][0])
    for r in range(mags.shape[0]):
        L.pvo_formant_shift(N, mags[r].ctypes.data, env)
    assert bits_equal(mags, kat[f"formant{N}_{tag}_out"])


def test_princarg_range():
    L = O.lib()
    for a in np.linspace(-50, 50, 1001):
        v = L.pvo_princarg(float(a))
        assert -np.pi <= v <= np.pi + 1e-12
        assert abs(np.angle(np.exp(1j * v)) - np.angle(np.exp(1j * a))) < 1e-9 or abs(abs(v) - np.pi) < 1e-9
