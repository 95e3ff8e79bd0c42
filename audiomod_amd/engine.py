"""Host-side Python mirror of the reference's phase-vocoder interface, over the C ABI.

Everything here calls audiomod_amd/lib/libaudiomod_pv.so (HIP kernels for gfx950 + C++ host
engine, include/audiomod_pv.h).  There is no Python or CPU implementation of the DSP in this
package: if the library is missing or no MI355X is visible, construction fails loudly.

`PhaseVocoder` keeps the method names and call semantics of audiomod::phasevocoder
(reference include/dafx/phasevocoder.h:42-117, src/phasevocoder/phasevocoder.cc:87-183) so the
parity tests read like drives of the reference class.  `Batch` is the device-resident throughput
path (many independent streams), used by bench.py.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AUDIOMOD_PV_LIB: a diagnostic build of the library -- tools/build_variant.sh -- instead of the product's)
LIB_PATH = os.environ.get("AUDIOMOD_PV_LIB") or os.path.join(_HERE, "lib", "libaudiomod_pv.so")

MODES = {"constant": -1, "normal_pitchshift": 0, "gender_change": 1, "formant_pitchshift": 2,
         "vocoder": 3, "vocoder_chord": 4, "time_stretch": 5, "robotic": 6, "whisper": 7,
         "formant_cepstral": 8}  # 8 = extension of this engine (PV_MODE_FORMANT_CEPSTRAL)
CONSTANT, NORMAL_SHIFT, GENDER_CHANGE, FORMANT_PRESERVE = -1, 0, 1, 2
VOCODER_ROSENBERG, VOCODER_CHORD, NORMAL_STRETCH, ROBOTIC, WHISPER = 3, 4, 5, 6, 7
FORMANT_CEPSTRAL = 8
NORMAL_PV, PHASE_LOCKED, INT_RATIO = 0, 1, 2
KERNELS = ("pv_analyze_kernel", "pv_match_kernel", "pv_seq_kernel", "pv_prop_kernel", "pv_synth_kernel",
           "pv_ola_kernel", "pv_cepstral_kernel", "pv_synth_ola_kernel")


class PvError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("channels", C.c_int32), ("time_ratio", C.c_float),
                ("pitch_semitones", C.c_float), ("mode", C.c_int32), ("coremode", C.c_int32),
                ("fftsize", C.c_int32), ("hopsize", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("fftsize", C.c_int32), ("hop_in", C.c_int32), ("hop_out_nominal", C.c_int32),
                ("outbuf_capacity", C.c_int32), ("pitch_scale", C.c_float), ("hs_ratio", C.c_float),
                ("int_ratio", C.c_int32), ("resample", C.c_int32), ("res_num", C.c_uint32), ("res_den", C.c_uint32),
                ("res_filt_len", C.c_int32), ("res_oversample", C.c_int32), ("res_interp", C.c_int32),
                ("slices", C.c_int64), ("bytes_per_slice", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Load the native library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PvError(f"{LIB_PATH} not built: run `make` (or __graft_entry__.build()); there is no fallback path")
    # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process
    # cannot both own the GPU.  Loading torch first makes our DT_NEEDED libamdhip64.so.7 resolve to the copy
    # torch already mapped, so tensors and our kernels share one runtime.  Without torch (e.g. the C++
    # drop-in) the library uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    fpp = C.POINTER(C.POINTER(C.c_float))
    L.pv_strerror.restype = C.c_char_p
    L.pv_strerror.argtypes = [C.c_int]
    L.pv_last_error.restype = C.c_char_p
    L.pv_device_count.restype = C.c_int
    L.pv_kernel_name.restype = C.c_char_p
    L.pv_kernel_name.argtypes = [C.c_int]
    L.pv_plan_simulate.argtypes = [C.POINTER(Config), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int64, C.POINTER(C.c_int64), C.POINTER(Info)]
    L.pv_plan_whisper_phases.argtypes = [C.c_int64, C.c_void_p]
    L.pv_plan_table.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p, C.c_int64]
    L.pv_plan_table.restype = C.c_int64
    L.pv_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(C.c_void_p)]
    L.pv_destroy.argtypes = [C.c_void_p]
    L.pv_feed.argtypes = [C.c_void_p, fpp, C.c_int32]
    L.pv_available.argtypes = [C.c_void_p]
    L.pv_available.restype = C.c_int32
    L.pv_retrieve.argtypes = [C.c_void_p, fpp, C.c_int32]
    L.pv_retrieve.restype = C.c_int32
    L.pv_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.pv_batch_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int,
                                  C.POINTER(C.c_void_p)]
    L.pv_batch_destroy.argtypes = [C.c_void_p]
    L.pv_batch_out_frames.argtypes = [C.c_void_p]
    L.pv_batch_out_frames.restype = C.c_int64
    L.pv_batch_slices.argtypes = [C.c_void_p]
    L.pv_batch_slices.restype = C.c_int64
    L.pv_batch_launches.argtypes = [C.c_void_p]
    L.pv_batch_launches.restype = C.c_int32
    L.pv_batch_pipelined.argtypes = [C.c_void_p]
    L.pv_batch_pipelined.restype = C.c_int32
    L.pv_batch_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.pv_batch_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pv_batch_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.pv_batch_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pv_hostio_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_int32,
                                   C.c_int32, C.POINTER(C.c_void_p)]
    L.pv_hostio_destroy.argtypes = [C.c_void_p]
    L.pv_hostio_out_frames.argtypes = [C.c_void_p]
    L.pv_hostio_out_frames.restype = C.c_int64
    L.pv_hostio_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pv_debug_atan2f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L.pv_debug_polar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L.pv_debug_sqrt_sweep.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int]
    L.pv_host_alloc.argtypes = [C.c_size_t]
    L.pv_host_alloc.restype = C.c_void_p
    L.pv_host_free.argtypes = [C.c_void_p]
    L.pv_set_arithmetic.argtypes = [C.c_int]
    L.pv_get_arithmetic.restype = C.c_int
    _lib = L
    return L


ARITH_FAST, ARITH_EXACT = 0, 1


def set_arithmetic(arith):
    """Process-wide, read at engine creation (include/audiomod_pv.h pv_set_arithmetic): ARITH_FAST (default) lets the
    many-stream batch path fuse / regroup the synthesis side's arithmetic within the 1e-4 RMS contract; ARITH_EXACT keeps
    the reference's operation order everywhere.  Returns the previous setting."""
    L = lib()
    prev = L.pv_get_arithmetic()
    _check(L.pv_set_arithmetic(int(arith)), "pv_set_arithmetic")
    return prev


def get_arithmetic():
    return lib().pv_get_arithmetic()


def _check(st, what):
    if st != 0:
        L = lib()
        raise PvError(f"{what}: {L.pv_strerror(st).decode()} ({L.pv_last_error().decode()})")


def make_config(channels, mode="normal_pitchshift", semitones=0.0, time_ratio=1.0, coremode=1, fftsize=2048,
                sample_rate=48000, hopsize=0):
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    return Config(sample_rate, channels, time_ratio, semitones, m, coremode, fftsize, hopsize)


def plan_simulate(calls, max_slices=1 << 22, **kw):
    """Host planner only (works without a GPU): per-call availability and per-slice increments."""
    L = lib()
    cfg = make_config(**kw)
    n = np.ascontiguousarray(calls, dtype=np.int32)
    avail = np.zeros(len(n), np.int32)
    shift = np.zeros(max_slices, np.int32)
    phase = np.zeros(max_slices, np.int32)
    ns = C.c_int64(0)
    info = Info()
    st = L.pv_plan_simulate(C.byref(cfg), n.ctypes.data, len(n), avail.ctypes.data, shift.ctypes.data,
                            phase.ctypes.data, max_slices, C.byref(ns), C.byref(info))
    _check(st, "pv_plan_simulate")
    k = min(ns.value, max_slices)
    return avail, shift[:k], phase[:k], info.as_dict()


def plan_table(which, max_len=1 << 20, **kw):
    """The planner's window (0), Speex filter table (1) or vocoder carrier (2: first max_len samples); host only."""
    cfg = make_config(**kw)
    out = np.zeros(max_len, np.float32)
    n = lib().pv_plan_table(C.byref(cfg), which, out.ctypes.data, max_len)
    if n < 0:
        _check(int(-n), "pv_plan_table")
    return out[:min(n, max_len)].copy()


def whisper_phases(n):
    """First n phases WHISPER mode draws in a fresh reference process (host only)."""
    out = np.zeros(n, np.float32)
    _check(lib().pv_plan_whisper_phases(n, out.ctypes.data), "pv_plan_whisper_phases")
    return out


def _pp(rows):
    fp = C.POINTER(C.c_float)
    return (fp * len(rows))(*[r.ctypes.data_as(fp) for r in rows])


class PhaseVocoder:
    """audiomod::phasevocoder with the same constructor arguments and entry points."""

    def __init__(self, sampleRate, numChannels, timeratio, pitchshift, mode=NORMAL_SHIFT, coremode=PHASE_LOCKED,
                 fftsize=2048, hopsize=0, device=0):
        self.L = lib()
        self.cfg = make_config(numChannels, mode, pitchshift, timeratio, coremode, fftsize, sampleRate, hopsize)
        self.channels = numChannels
        self.mode = self.cfg.mode
        self.h = C.c_void_p()
        _check(self.L.pv_create(C.byref(self.cfg), device, C.byref(self.h)), "pv_create")
        self.num_res_ = 0
        self.outready_ = False

    def close(self):
        if getattr(self, "h", None):
            self.L.pv_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def info(self):
        i = Info()
        _check(self.L.pv_get_info(self.h, C.byref(i)), "pv_get_info")
        return i.as_dict()

    # ---- offline interface (modbase_offline)
    def processInData(self, inData):
        x = np.ascontiguousarray(inData, dtype=np.float32)
        assert x.ndim == 2 and x.shape[0] == self.channels
        rows = [x[c] for c in range(self.channels)]
        _check(self.L.pv_feed(self.h, _pp(rows), x.shape[1]), "pv_feed")
        self.num_res_ = self.L.pv_available(self.h)

    def getOutSamples(self):
        return self.num_res_

    def getOutData(self, num_out_samples):
        n = min(num_out_samples, self.num_res_)
        out = np.zeros((self.channels, max(n, 1)), np.float32)
        got = 0
        if n > 0:
            rows = [out[c] for c in range(self.channels)]
            got = self.L.pv_retrieve(self.h, _pp(rows), n)
        self.outready_ = True
        return out[:, :got]

    # ---- real-time interface (modbase)
    def processBlock(self, bufferData):
        """In place on bufferData (float32 [channels, n]); check outputReady() afterwards."""
        assert bufferData.dtype == np.float32 and bufferData.flags.c_contiguous
        n = bufferData.shape[1]
        if self.mode == NORMAL_STRETCH:  # the reference's processBlock ignores this mode (phasevocoder.cc:134-144)
            self.outready_ = True
            return
        rows = [bufferData[c] for c in range(self.channels)]
        _check(self.L.pv_feed(self.h, _pp(rows), n), "pv_feed")
        self.num_res_ = self.L.pv_available(self.h)
        if self.num_res_ >= n:
            self.L.pv_retrieve(self.h, _pp(rows), n)
            self.outready_ = True
        else:
            self.outready_ = False

    def outputReady(self):
        return self.outready_


def run_offline(x, block=480, flush=True, device=0, **kw):
    """The reference CLI's offline loop (main/main.cc:471-510) on the GPU streaming engine."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, frames = x.shape
    cfg = make_config(ch, **kw)
    pv = PhaseVocoder(cfg.sample_rate, ch, cfg.time_ratio, cfg.pitch_semitones, cfg.mode, cfg.coremode, cfg.fftsize,
                      cfg.hopsize, device)
    outs, counts, produced = [], [], 0
    for i in range(0, frames, block):
        pv.processInData(x[:, i:i + block])
        got = pv.getOutSamples()
        outs.append(pv.getOutData(got))
        counts.append(got)
        produced += got
    if flush:
        z = np.zeros((ch, block), np.float32)
        while produced < frames:
            pv.processInData(z)
            got = pv.getOutSamples()
            y = pv.getOutData(got)
            counts.append(got)
            w = got if frames - produced > got else frames - produced
            outs.append(y[:, :w])
            produced += w
    pv.close()
    return np.concatenate(outs, axis=1), counts


def run_realtime(x, block=480, device=0, **kw):
    """The reference's processBlock/outputReady loop (main/main.cc:561-572)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ch, frames = x.shape
    cfg = make_config(ch, **kw)
    pv = PhaseVocoder(cfg.sample_rate, ch, cfg.time_ratio, cfg.pitch_semitones, cfg.mode, cfg.coremode, cfg.fftsize,
                      cfg.hopsize, device)
    outs, counts = [], []
    for i in range(0, frames, block):
        blk = np.ascontiguousarray(x[:, i:i + block]).copy()
        pv.processBlock(blk)
        if pv.outputReady():
            outs.append(blk)
            counts.append(blk.shape[1])
        else:
            counts.append(-1)
    pv.close()
    out = np.concatenate(outs, axis=1) if outs else np.zeros((ch, 0), np.float32)
    return out, counts


class Batch:
    """nstreams independent streams, device-resident in and out (torch CUDA tensors)."""

    def __init__(self, nstreams, frames, channels=2, block=480, flush=True, device=0, **kw):
        self.L = lib()
        self.cfg = make_config(channels, **kw)
        self.nstreams, self.frames, self.channels, self.device = nstreams, frames, channels, device
        self.h = C.c_void_p()
        _check(self.L.pv_batch_create(C.byref(self.cfg), nstreams, frames, block, 1 if flush else 0, device,
                                      C.byref(self.h)), "pv_batch_create")
        self.out_frames = self.L.pv_batch_out_frames(self.h)
        self.slices = self.L.pv_batch_slices(self.h)
        self.launches = self.L.pv_batch_launches(self.h)
        self.pipelined = bool(self.L.pv_batch_pipelined(self.h))  # chain kernel on a second stream (overlapped)

    def close(self):
        if getattr(self, "h", None):
            self.L.pv_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def info(self):
        i = Info()
        _check(self.L.pv_batch_get_info(self.h, C.byref(i)), "pv_batch_get_info")
        return i.as_dict()

    def alloc_out(self):
        import torch
        return torch.empty((self.nstreams, self.channels, self.out_frames), dtype=torch.float32,
                           device=f"cuda:{self.device}")

    def run(self, d_in, d_out=None, stream=None):
        """d_in: torch float32 CUDA tensor [nstreams, channels, frames], contiguous.  Asynchronous."""
        import torch
        assert d_in.is_cuda and d_in.dtype == torch.float32 and d_in.is_contiguous()
        assert tuple(d_in.shape) == (self.nstreams, self.channels, self.frames)
        if d_out is None:
            d_out = self.alloc_out()
        assert d_out.is_cuda and d_out.is_contiguous() and tuple(d_out.shape) == (self.nstreams, self.channels,
                                                                                 self.out_frames)
        s = stream if stream is not None else torch.cuda.current_stream(d_in.device)
        _check(self.L.pv_batch_run(self.h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                   C.c_void_p(s.cuda_stream)), "pv_batch_run")
        return d_out

    def enable_timing(self, every=1):
        """every = 0/False: off; n: HIP events around the kernels of every n-th chunk."""
        _check(self.L.pv_batch_enable_timing(self.h, int(every)), "pv_batch_enable_timing")

    def kernel_times(self):
        """{kernel: (total_ms, launches)} since enable_timing; synchronise the stream first."""
        ms = (C.c_double * len(KERNELS))()
        n = (C.c_int64 * len(KERNELS))()
        _check(self.L.pv_batch_kernel_times(self.h, ms, n), "pv_batch_kernel_times")
        return {KERNELS[k]: (ms[k], n[k]) for k in range(len(KERNELS))}


class HostIO:
    """nstreams independent streams whose input and output live in HOST memory (what the reference's callers hold:
    main/main.cc:152-162,484-491), float32 or int16 on the wire; groups of streams are staged through the GPU with
    copy-in, kernels and copy-out overlapped (include/audiomod_pv.h pv_hostio_*)."""

    def __init__(self, nstreams, frames, channels=2, block=480, flush=True, device=0, streams_per_group=16,
                 wire="f32", **kw):
        self.L = lib()
        self.cfg = make_config(channels, **kw)
        self.nstreams, self.frames, self.channels = nstreams, frames, channels
        self.wire = {"f32": 0, "i16": 1}[wire]
        self.dtype = np.float32 if self.wire == 0 else np.int16
        self.h = C.c_void_p()
        _check(self.L.pv_hostio_create(C.byref(self.cfg), nstreams, frames, block, 1 if flush else 0, device,
                                       streams_per_group, self.wire, C.byref(self.h)), "pv_hostio_create")
        self.out_frames = self.L.pv_hostio_out_frames(self.h)
        self._pinned = []

    def pinned(self, shape):
        """numpy array of this job's wire type in page-locked host memory (freed with the object)"""
        n = int(np.prod(shape)) * np.dtype(self.dtype).itemsize
        p = self.L.pv_host_alloc(max(n, 1))
        if not p:
            raise PvError("pv_host_alloc failed")
        self._pinned.append(p)
        buf = (C.c_char * max(n, 1)).from_address(p)
        return np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def run(self, host_in, host_out=None):
        assert host_in.dtype == self.dtype and host_in.flags.c_contiguous
        assert tuple(host_in.shape) == (self.nstreams, self.channels, self.frames)
        if host_out is None:
            host_out = self.pinned((self.nstreams, self.channels, self.out_frames))
        assert host_out.dtype == self.dtype and host_out.flags.c_contiguous
        _check(self.L.pv_hostio_run(self.h, host_in.ctypes.data, host_out.ctypes.data), "pv_hostio_run")
        return host_out

    def close(self):
        if getattr(self, "h", None):
            self.L.pv_hostio_destroy(self.h)
            self.h = None
        for p in getattr(self, "_pinned", []):
            self.L.pv_host_free(p)
        self._pinned = []

    def __del__(self):
        self.close()
