"""ctypes access to the oracle (CPU restatement) and to the compiled real reference.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under audiomod_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpv_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_DRIVER = os.path.join(REF_DIR, "ref_driver")
REF_KAT = os.path.join(REF_DIR, "ref_kat")

MODES = {"constant": -1, "normal_pitchshift": 0, "gender_change": 1, "formant_pitchshift": 2,
         "vocoder": 3, "vocoder_chord": 4, "time_stretch": 5, "robotic": 6, "whisper": 7,
         "formant_cepstral": 8}  # 8: extension, the reference's unreachable cepstral formant shift


class Config(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("time_ratio", C.c_float),
                ("pitch_semitones", C.c_float), ("mode", C.c_int), ("coremode", C.c_int),
                ("fftsize", C.c_int), ("hopsize", C.c_int)]


class Info(C.Structure):
    _fields_ = [("fftsize", C.c_int), ("hop_in", C.c_int), ("hop_out_nominal", C.c_int),
                ("outbuf_capacity", C.c_int), ("pitch_scale", C.c_float), ("hs_ratio", C.c_float),
                ("int_ratio", C.c_int), ("resample", C.c_int), ("res_num", C.c_uint), ("res_den", C.c_uint),
                ("res_filt_len", C.c_int), ("res_oversample", C.c_int), ("res_interp", C.c_int),
                ("slices", C.c_long)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fpp = C.POINTER(C.POINTER(C.c_float))
        L.pvo_create.restype = C.c_void_p
        L.pvo_create.argtypes = [C.POINTER(Config)]
        L.pvo_destroy.argtypes = [C.c_void_p]
        L.pvo_process.argtypes = [C.c_void_p, fpp, C.c_int]
        L.pvo_available.argtypes = [C.c_void_p]
        L.pvo_retrieve.argtypes = [C.c_void_p, fpp, C.c_int]
        L.pvo_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.pvo_get_increments.restype = C.c_long
        L.pvo_get_increments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.pvo_hann.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.pvo_forward_polar.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvo_inverse_polar.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvo_formant_shift.argtypes = [C.c_int, C.c_void_p, C.c_float]
        L.pvo_princarg.restype = C.c_double
        L.pvo_princarg.argtypes = [C.c_double]
        L.pvo_res_create.restype = C.c_void_p
        L.pvo_res_destroy.argtypes = [C.c_void_p]
        L.pvo_res_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.pvo_res_info.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.pvo_res_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _pp(arrs):
    fp = C.POINTER(C.c_float)
    return (fp * len(arrs))(*[a.ctypes.data_as(fp) for a in arrs])


class OracleUndefined(RuntimeError):
    """The configuration drives the reference into undefined behaviour (pv_oracle.h PVO_UNDEFINED)."""


class Oracle:
    """Streaming handle mirroring audiomod::phasevocoder's offline + real-time drive."""

    def __init__(self, channels, mode="normal_pitchshift", semitones=0.0, time_ratio=1.0, coremode=1,
                 fftsize=2048, sample_rate=48000, hopsize=0):
        self.L = lib()
        m = MODES[mode] if isinstance(mode, str) else int(mode)
        self.cfg = Config(sample_rate, channels, time_ratio, semitones, m, coremode, fftsize, hopsize)
        self.h = self.L.pvo_create(C.byref(self.cfg))
        self.channels = channels

    def close(self):
        if self.h:
            self.L.pvo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def info(self):
        i = Info()
        self.L.pvo_get_info(self.h, C.byref(i))
        return {k: getattr(i, k) for k, _ in Info._fields_}

    def process(self, block):
        block = np.ascontiguousarray(block, dtype=np.float32)
        assert block.shape[0] == self.channels
        rows = [block[c] for c in range(self.channels)]
        got = self.L.pvo_process(self.h, _pp(rows), block.shape[1])
        if got == -2:  # PVO_UNDEFINED
            raise OracleUndefined("shift increment above the FFT size: the reference's behaviour is undefined here")
        return got

    def available(self):
        return self.L.pvo_available(self.h)

    def retrieve(self, n):
        out = np.zeros((self.channels, max(n, 1)), dtype=np.float32)
        rows = [out[c] for c in range(self.channels)]
        got = self.L.pvo_retrieve(self.h, _pp(rows), n)
        return out[:, :got]

    def increments(self):
        n = self.L.pvo_get_increments(self.h, None, None, 0)
        s = np.zeros(n, dtype=np.int32)
        p = np.zeros(n, dtype=np.int32)
        self.L.pvo_get_increments(self.h, s.ctypes.data, p.ctypes.data, n)
        return s, p


def run_offline(x, block=480, flush=True, **kw):
    """The reference CLI's offline drive loop (main/main.cc:471-510) on the oracle.
    x: float32 [channels, frames].  Returns (out [channels, n], per-call counts)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, frames = x.shape
    o = Oracle(ch, **kw)
    outs, counts = [], []
    produced = 0
    for i in range(0, frames, block):
        got = o.process(x[:, i:i + block])
        outs.append(o.retrieve(got))
        counts.append(got)
        produced += got
    if flush:
        z = np.zeros((ch, block), dtype=np.float32)
        while produced < frames:
            got = o.process(z)
            y = o.retrieve(got)
            counts.append(got)
            w = got if frames - produced > got else frames - produced
            outs.append(y[:, :w])
            produced += w
    info = o.info()
    o.close()
    return np.concatenate(outs, axis=1), counts, info


def run_realtime(x, block=480, **kw):
    """The reference's processBlock/outputReady loop (main/main.cc:561-572) on the oracle."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, frames = x.shape
    if kw.get("mode") in ("time_stretch", MODES["time_stretch"]):
        # processBlock dispatches on the mode and has no branch for NORMAL_STRETCH (phasevocoder.cc:126-152):
        # nothing is processed, the caller's buffer stays as it was and outputReady() is true
        return x.copy(), [min(block, frames - i) for i in range(0, frames, block)]
    o = Oracle(ch, **kw)
    outs, counts = [], []
    for i in range(0, frames, block):
        blk = x[:, i:i + block]
        avail = o.process(blk)
        n = blk.shape[1]
        if avail >= n:
            outs.append(o.retrieve(n))
            counts.append(n)
        else:
            counts.append(-1)
    o.close()
    out = np.concatenate(outs, axis=1) if outs else np.zeros((ch, 0), np.float32)
    return out, counts


# ---------------------------------------------------------------------------------------------
# the REAL reference (only where oracle/_ref was built, i.e. where /root/reference exists or
# the prebuilt binaries travelled with the snapshot)
# ---------------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_DRIVER) and os.path.exists(REF_KAT)


def ref_run(x, api="offline", block=480, flush=True, mode="normal_pitchshift", semitones=0.0, time_ratio=1.0,
            coremode=1, fftsize=2048, sample_rate=48000, hopsize=0, timeout=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, frames = x.shape
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    with tempfile.TemporaryDirectory() as d:
        fin, fout, fcnt = (os.path.join(d, n) for n in ("in.f32", "out.f32", "cnt.txt"))
        x.tofile(fin)
        cmd = [REF_DRIVER, api, fin, fout, fcnt, str(ch), str(frames), str(sample_rate), repr(float(time_ratio)),
               repr(float(semitones)), str(m), str(coremode), str(fftsize), str(block), "1" if flush else "0",
               str(hopsize)]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        with open(fcnt) as f:
            vals = [int(v) for v in f.read().split()]
        n = vals[0]
        out = np.fromfile(fout, dtype=np.float32).reshape(ch, n) if n else np.zeros((ch, 0), np.float32)
    return out, vals[1:]


def ref_kat(*args):
    subprocess.run([REF_KAT] + [str(a) for a in args], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
