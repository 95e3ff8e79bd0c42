"""Live cross-check of the oracle against the REAL reference binaries in oracle/_ref/ (built by
oracle/ref.mk from /root/reference).  Skipped where those binaries are absent."""
import numpy as np
import pytest

from audiomod_amd import signals
from oracle import oracle_py as O
from tests.helpers import bits_equal

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")

CASES = [
    dict(semitones=5.0, coremode=1),
    dict(semitones=-5.0, coremode=0),
    dict(semitones=2.0, coremode=2, fftsize=1024),
    dict(mode="time_stretch", time_ratio=0.75, flush=False),
    dict(mode="time_stretch", time_ratio=2.0, flush=False),
    dict(mode="gender_change", semitones=0.0),
    dict(mode="formant_pitchshift", semitones=3.0, fftsize=4096),
    dict(semitones=4.0, block=64),
    dict(semitones=4.0, block=4800),
    dict(mode="robotic"),
]


@pytest.mark.parametrize("kw", CASES, ids=[str(i) for i in range(len(CASES))])
def test_offline_matches_reference(kw):
    x = signals.voice(20000, 2, seed=321)
    want, wc = O.ref_run(x, **kw)
    got, gc, _ = O.run_offline(x, **kw)
    assert gc == wc
    assert bits_equal(got, want)


def test_realtime_matches_reference():
    x = signals.noise(20000, 2)
    want, wc = O.ref_run(x, api="rt", semitones=-7.0, block=512)
    got, gc = O.run_realtime(x, semitones=-7.0, block=512)
    assert gc == wc
    assert bits_equal(got, want)


def test_chunking_independence():
    """Output stream does not depend on call chunking."""
    x = signals.voice(20000, 2, seed=9)
    a, _, _ = O.run_offline(x, semitones=4.0, block=64)
    b, _, _ = O.run_offline(x, semitones=4.0, block=4800)
    assert bits_equal(a, b)


def test_random_configurations_bit_identical_to_the_reference():
    """A slice of tests/sweeps/fuzz_oracle_vs_ref.py (the full sweep: profiles/r01/fuzz_oracle_vs_ref_summary.txt):
    random mode / pitch / ratio / FFT size / hop / rate / channels / call size through the oracle and through the
    compiled reference, offline and real-time loops, outputs compared bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sweeps", "fuzz_oracle_vs_ref.py"), "120", "9"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "differing or failed: 0" in r.stdout
