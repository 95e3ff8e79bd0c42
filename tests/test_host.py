"""CPU-side tests of the product: the C-ABI library loads and exports every declared symbol, the host
planner (integer scheduling, increments, resampler set-up) agrees with the oracle, and the product
fails loudly without a GPU.  No compute kernels are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from audiomod_amd import engine as E
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "audiomod_pv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pv_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 19
    L = ctypes.CDLL(E.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/audiomod_pv.h but not exported"


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", E.LIB_PATH], capture_output=True, text=True).stdout
    assert "pv_oracle" not in out and "audiomod_ref" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", E.LIB_PATH], capture_output=True, text=True).stdout
    assert "pvo_" not in syms


PLAN_CASES = [
    ([480] * 200, dict(channels=2, semitones=4.0)),
    ([480] * 200, dict(channels=1, semitones=4.0)),
    ([480] * 200, dict(channels=2, semitones=7.0, mode="formant_pitchshift")),
    ([480] * 200, dict(channels=2, semitones=-7.0, mode="gender_change")),
    ([480] * 200, dict(channels=2, time_ratio=1.5, mode="time_stretch", fftsize=4096)),
    ([480] * 200, dict(channels=2, time_ratio=2.0, mode="time_stretch")),
    ([480] * 200, dict(channels=2, semitones=12.0)),
    ([480] * 200, dict(channels=2, semitones=-19.0)),
    ([480] * 200, dict(channels=2, mode="robotic")),
    ([64] * 1500, dict(channels=2, semitones=4.0)),
    ([4800, 1, 0, 17, 20000, 480, 0, 0, 5], dict(channels=2, semitones=-3.0)),
    ([441] * 300, dict(channels=2, semitones=4.0, sample_rate=44100, fftsize=1024)),
    ([480] * 200, dict(channels=2, semitones=4.0, hopsize=128)),
    ([480] * 100, dict(channels=2, mode="constant")),
    ([480] * 100, dict(channels=2, mode="whisper")),
    ([333] * 100, dict(channels=1, mode="constant", fftsize=1024)),
    ([480] * 100, dict(channels=2, mode="vocoder")),
    ([480] * 100, dict(channels=2, mode="vocoder_chord", fftsize=4096)),
]


@pytest.mark.parametrize("calls,kw", PLAN_CASES, ids=[str(i) for i in range(len(PLAN_CASES))])
def test_planner_matches_oracle(calls, kw):
    avail, shift, phase, info = E.plan_simulate(calls, **kw)
    ch = kw["channels"]
    o = O.Oracle(**kw)
    ref = []
    for n in calls:
        got = o.process(np.zeros((ch, n), np.float32))
        o.retrieve(got)
        ref.append(got)
    s, p = o.increments()
    oi = o.info()
    assert list(avail) == ref
    assert np.array_equal(s, shift) and np.array_equal(p, phase)
    for k in ("fftsize", "hop_in", "hop_out_nominal", "outbuf_capacity", "pitch_scale", "hs_ratio", "int_ratio",
              "resample", "slices"):
        assert info[k] == oi[k], k
    if info["resample"]:
        for k in ("res_num", "res_den", "res_filt_len", "res_oversample", "res_interp"):
            assert info[k] == oi[k], k


def test_derived_constants_table():
    """SURVEY.md section 8 table of derived constants."""
    rows = [
        (dict(semitones=4.0), 203, 256, (12382188, 9827749), 80),
        (dict(semitones=7.0), 170, 256, (272408136, 181810613), 96),
        (dict(semitones=-7.0), 455, 303, (1782457, 2670668), 64),
    ]
    for kw, h, oh, frac, fl in rows:
        _, _, _, info = E.plan_simulate([480], channels=2, **kw)
        assert (info["hop_in"], info["hop_out_nominal"]) == (h, oh)
        assert (info["res_num"], info["res_den"]) == frac
        assert info["res_filt_len"] == fl
    _, _, _, info = E.plan_simulate([480], channels=2, mode="time_stretch", time_ratio=1.5, fftsize=4096)
    assert (info["hop_in"], info["hop_out_nominal"], info["resample"]) == (341, 512, 0)
    assert info["bytes_per_slice"] == 108572
    _, _, _, info = E.plan_simulate([480], channels=2, semitones=4.0)
    assert info["bytes_per_slice"] == 56136


def test_first_calls_availability_sequence():
    """block 480, +4 st: 0,0,0,0,172,203,203,204,..."""
    avail, _, _, _ = E.plan_simulate([480] * 12, channels=2, semitones=4.0)
    assert list(avail[:8]) == [0, 0, 0, 0, 172, 203, 203, 204]


def test_errors():
    with pytest.raises(E.PvError):
        E.plan_simulate([480], channels=0, semitones=4.0)
    with pytest.raises(E.PvError):
        E.plan_simulate([480], channels=2, mode=42)


OVERRUN = [
    (dict(channels=2, semitones=-3.0), [100000, 480, 480, 30000, 480]),
    (dict(channels=2, mode="constant", fftsize=256), [4800, 4800, 64, 9000, 480]),
    (dict(channels=1, mode="vocoder", fftsize=512), [9000, 480, 9000]),
    (dict(channels=3, mode="time_stretch", time_ratio=2.5, fftsize=256, coremode=0), [6000, 6000, 100, 6000]),
    (dict(channels=2, mode="robotic", fftsize=256, semitones=-9.0), [5000, 5000]),
]


@pytest.mark.parametrize("kw,calls", OVERRUN, ids=[str(i) for i in range(len(OVERRUN))])
def test_overrun_drops_slices_like_the_reference(kw, calls):
    """A call that leaves more output pending than the reference's output ring holds makes the reference DROP
    slices (phasevocoderprocess.cc:337-364: the frame stays in the accumulators, nothing is written): the planner
    follows it -- same per-call availability, same increments (calculateIncrements runs for dropped slices too)."""
    avail, shift, phase, info = E.plan_simulate(calls, **kw)
    ch = kw["channels"]
    o = O.Oracle(**kw)
    ref = []
    for n in calls:
        got = o.process(np.zeros((ch, n), np.float32))
        o.retrieve(got)
        ref.append(got)
    s, p = o.increments()
    assert list(avail) == ref
    assert np.array_equal(s, shift) and np.array_equal(p, phase)
    assert max(ref) >= info["outbuf_capacity"] - 2 * info["fftsize"]  # the ring really filled up


def test_output_hop_above_the_fft_size_is_refused_on_both_sides():
    """A caller-chosen hop whose output hop exceeds N drives the reference's writeSlice into a memmove with a
    wrapped-around count (phasevocoderprocess.cc:1181-1190): undefined there.  The oracle stops instead of
    overflowing, and the engine's planner refuses the configuration with a reason."""
    from audiomod_amd import signals
    kw = dict(fftsize=2048, coremode=0, sample_rate=22050, semitones=15.58221435546875, hopsize=867)
    with pytest.raises(O.OracleUndefined):
        O.run_offline(signals.voice(6000, 2), **kw)
    with pytest.raises(E.PvError, match="undefined behaviour in the reference"):
        E.plan_simulate([480] * 12, channels=2, **kw)
    kw = dict(mode="time_stretch", fftsize=512, coremode=2, sample_rate=8000, time_ratio=2.838555335998535, hopsize=242)
    with pytest.raises(O.OracleUndefined):
        O.run_offline(signals.voice(6000, 2), flush=False, **kw)
    with pytest.raises(E.PvError, match="undefined behaviour in the reference"):
        E.plan_simulate([480] * 12, channels=2, **kw)
    # an error text never outlives the call that set it
    E.plan_simulate([480] * 4, channels=2, semitones=4.0)
    assert E.lib().pv_last_error() in (b"", None)


def test_whisper_phases_match_libc_rand():
    """The engine's own glibc-compatible generator against the C library's rand() from its default seed."""
    import ctypes.util
    libc = ctypes.CDLL(ctypes.util.find_library("c"))
    libc.srand(1)
    n = 5000
    raw = np.array([libc.rand() for _ in range(n)], dtype=np.float32)  # (float)rand()
    want = (np.float32(2 * np.pi) * raw) / np.float32(2147483647)
    got = E.whisper_phases(n)
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_fails_loudly_without_gpu():
    if E.lib().pv_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(E.PvError, match="no gfx950"):
        E.PhaseVocoder(48000, 2, 1.0, 4.0)
    with pytest.raises(E.PvError, match="no gfx950"):
        E.Batch(2, 48000, semitones=4.0)


@pytest.mark.parametrize("st", [0.1, 0.37, 1.0, 4.0, 7.0, -0.2, -7.0, 12.0, -12.0, 15.9, -15.9])
def test_planner_filter_table_is_the_oracles(st):
    """The Speex Q4 rate pair, filter geometry and every coefficient of the table the planner derives (pv_plan.cc
    make_resampler; uploaded for the resampling kernels) against the oracle's, which is pinned on the compiled
    reference (resample.c:661-775): bit for bit."""
    import ctypes as C
    info = E.plan_simulate([480], channels=1, semitones=st)[3]
    tab = E.plan_table(1, channels=1, semitones=st)
    L = O.lib()
    r = L.pvo_res_create()
    x, buf = np.zeros(256, np.float32), np.zeros(4096, np.float32)
    L.pvo_res_process(r, x.ctypes.data, 256, C.c_float(1.0 / info["pitch_scale"]), buf.ctypes.data)
    n = L.pvo_res_table(r, None, 0)
    want = np.zeros(n, np.float32)
    L.pvo_res_table(r, want.ctypes.data, n)
    num, den, fl, ov, it = C.c_uint(), C.c_uint(), C.c_int(), C.c_int(), C.c_int()
    L.pvo_res_info(r, C.byref(num), C.byref(den), C.byref(fl), C.byref(ov), C.byref(it))
    L.pvo_res_destroy(r)
    assert (info["res_num"], info["res_den"], info["res_filt_len"], info["res_oversample"], info["res_interp"]) == \
        (num.value, den.value, fl.value, ov.value, it.value)
    assert tab.shape == want.shape and np.array_equal(tab.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("N", [256, 2048, 4096, 8192])
def test_planner_window_is_the_oracles(N):
    win = E.plan_table(0, channels=1, fftsize=N)
    want, area = np.zeros(N, np.float32), np.zeros(1, np.float32)
    O.lib().pvo_hann(N, want.ctypes.data, area.ctypes.data)
    assert np.array_equal(win.view(np.uint32), want.view(np.uint32))


def test_arithmetic_setting_round_trip_and_environment_default():
    """pv_set_arithmetic / pv_get_arithmetic (include/audiomod_pv.h): process-wide, PV_ARITH_FAST by default,
    PV_ARITH_EXACT when the environment says AUDIOMOD_PV_EXACT=1 at load time; anything else is refused.  No GPU needed."""
    import subprocess
    import sys
    L = E.lib()
    prev = L.pv_get_arithmetic()
    try:
        assert L.pv_set_arithmetic(E.ARITH_EXACT) == 0 and L.pv_get_arithmetic() == E.ARITH_EXACT
        assert L.pv_set_arithmetic(E.ARITH_FAST) == 0 and L.pv_get_arithmetic() == E.ARITH_FAST
        assert L.pv_set_arithmetic(7) != 0 and L.pv_get_arithmetic() == E.ARITH_FAST
        assert E.set_arithmetic(E.ARITH_EXACT) == E.ARITH_FAST and E.get_arithmetic() == E.ARITH_EXACT
    finally:
        L.pv_set_arithmetic(prev)
    code = "import sys; sys.path.insert(0, %r); from audiomod_amd import engine as E; print(E.get_arithmetic())" % ROOT
    for env, want in (({"AUDIOMOD_PV_EXACT": "1"}, "1"), ({"AUDIOMOD_PV_EXACT": "0"}, "0")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == want, r.stdout + r.stderr
