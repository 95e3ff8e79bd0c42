#!/usr/bin/env python3
"""bench.py -- throughput of the phase-vocoder hot path on MI355X (contract: see the task statement).

A "step" is one pass of the whole hot path (analysis FFT -> phase propagation -> synthesis FFT ->
overlap-add + resample) over one batch of synthetic input that is already resident in HBM:
STREAMS independent 48 kHz stereo streams of SECONDS seconds each per GPU, BASELINE.json
configs[1] (normal_pitchshift +4 semitones, phase-locked, fft 2048).  Streams shard one set per
GPU with no data-path collective (weak scaling: per-GPU work is fixed).

`--gpus N` with N > 1: unless already started under torch.distributed.run (WORLD_SIZE set), this process
launches N ranks of itself through `python -m torch.distributed.run` BEFORE importing torch or touching HIP,
relays their output and exits with their code.  Ranks exchange nothing on the data path; the timing barrier and
the MAX-over-ranks reduction go through gloo on CPU tensors (no RCCL on this path).

After the timed region the output of the last step is checked (`"verified"` in the line): sampled streams
against the oracle (RMS <= 1e-4), duplicate-input streams bit-identical, and an exact checksum of the whole
batch that must be the same after every step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


CONFIGS = {  # BASELINE.json configs[1..3]; configs[4] is cfg2 at 128 streams per GPU x 8 GPUs (the default workload)
    "cfg2": ("configs[1]: normal_pitchshift +4 st, stereo 48 kHz, fft=2048",
             dict(mode="normal_pitchshift", semitones=4.0, fftsize=2048), True),
    "cfg3": ("configs[2]: time_stretch ratio=1.5, stereo 48 kHz, fft=4096",
             dict(mode="time_stretch", time_ratio=1.5, fftsize=4096), False),
    "cfg4_formant+7": ("configs[3]: formant_pitchshift +7 st, stereo 48 kHz, fft=2048",
                       dict(mode="formant_pitchshift", semitones=7.0, fftsize=2048), True),
    "cfg4_formant-7": ("configs[3]: formant_pitchshift -7 st, stereo 48 kHz, fft=2048",
                       dict(mode="formant_pitchshift", semitones=-7.0, fftsize=2048), True),
    "cfg4_gender+7": ("configs[3]: gender_change +7 st, stereo 48 kHz, fft=2048",
                      dict(mode="gender_change", semitones=7.0, fftsize=2048), True),
    "cfg4_gender-7": ("configs[3]: gender_change -7 st, stereo 48 kHz, fft=2048",
                      dict(mode="gender_change", semitones=-7.0, fftsize=2048), True),
}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the kernel and engine sources: counter-derived figures in profiles/*.json carry
    the hash of the code they were measured on, and bench.py refuses to quote them for any other code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "audiomod_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(seconds=480, all_cores_seconds=240):
    """The real reference (oracle/_ref/ref_driver, kind 'reference') or, where absent, the oracle port,
    timed on ONE host core on one stereo stream of the same workload (the reference is single-threaded);
    then, as the fair whole-host figure, one independent stream per available core run side by side."""
    from audiomod_amd import signals
    from oracle import oracle_py as O
    frames = seconds * 48000
    x = np.tile(signals.voice(10 * 48000, 2), (1, (seconds + 9) // 10))[:, :frames]
    kw = dict(mode="normal_pitchshift", semitones=4.0, coremode=1, fftsize=2048)
    res = {}
    if O.have_ref():
        kind = "reference"
        with tempfile.TemporaryDirectory() as d:
            fin = os.path.join(d, "in.f32")
            x.tofile(fin)

            def cmd(n_frames, tag):
                return [O.REF_DRIVER, "offline", fin, os.path.join(d, f"out{tag}.f32"), os.path.join(d, f"cnt{tag}.txt"),
                        "2", str(n_frames), "48000", "1.0", "4.0", "0", "1", "2048", "480", "1"]
            t0 = time.perf_counter()
            subprocess.run(cmd(frames, ""), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            # BASELINE.json configs[0]: the MONO run of the CPU reference (main/main.cc:216-227), same settings
            fmono = os.path.join(d, "mono.f32")
            x[:1].tofile(fmono)
            mono_s = min(seconds, 240)
            mcmd = [O.REF_DRIVER, "offline", fmono, os.path.join(d, "outm.f32"), os.path.join(d, "cntm.txt"),
                    "1", str(mono_s * 48000), "48000", "1.0", "4.0", "0", "1", "2048", "480", "1"]
            t0 = time.perf_counter()
            mono_ok = subprocess.run(mcmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0
            dtm = time.perf_counter() - t0
            if mono_ok:
                res["configs0_mono"] = {"value": round(mono_s * 48000 / dtm / 1e6, 4), "unit": "Msamples/s",
                                        "x_realtime": round(mono_s / dtm, 2), "cores": 1, "kind": "reference",
                                        "sample": f"configs[0]: 1 mono stream x {mono_s} s, +4 st, fft 2048, "
                                                  f"phase-locked, block 480, wall {dtm:.2f} s"}
            ncpu = usable_cores()
            if ncpu > 1:
                fa = min(all_cores_seconds, seconds) * 48000
                t0 = time.perf_counter()
                ps = [subprocess.Popen(cmd(fa, i), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                      for i in range(ncpu)]
                ok = all(p.wait() == 0 for p in ps)
                dta = time.perf_counter() - t0
                if ok:
                    # (also as flat fields, for readers that keep only scalars of this object)
                    res["all_cores_value"] = round(ncpu * fa * 2 / dta / 1e6, 3)
                    res["all_cores_count"] = ncpu
                    res["all_cores"] = {"cores": ncpu, "value": round(ncpu * fa * 2 / dta / 1e6, 3),
                                        "x_realtime": round(ncpu * fa / 48000 / dta, 1),
                                        "sample": f"{ncpu} independent stereo streams x {fa // 48000} s side by side, "
                                                  f"wall {dta:.2f} s"}
    else:
        kind = "port"
        t0 = time.perf_counter()
        O.run_offline(x, **kw)
        dt = time.perf_counter() - t0
    fast = native_port_speed(min(seconds, 240))
    if fast:
        res["native_port"] = fast
    res.update({"value": round(frames * 2 / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": kind,
                "x_realtime": round(seconds / dt, 2),
                "sample": f"1 stereo stream x {seconds} s, same config, block 480, wall {dt:.2f} s"})
    return res


def native_port_speed(seconds):
    """SURVEY 8(d)'s fair-speed figure: this repository's own CPU port (oracle/pv_oracle.c) rebuilt for the host it
    runs on (-O3 -march=native: vector units and FMA allowed, so its output is no longer bit-identical -- a speed
    figure only), one stereo stream on one core.  None when there is no compiler."""
    import shutil
    if shutil.which("gcc") is None:
        return None
    try:
        with tempfile.TemporaryDirectory() as d:
            so = os.path.join(d, "libpv_oracle_native.so")
            subprocess.run(["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-shared",
                            os.path.join(ROOT, "oracle", "pv_oracle.c"), "-o", so, "-lm"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            code = ("import sys, time, numpy as np\n"
                    f"sys.path.insert(0, {ROOT!r})\n"
                    "from oracle import oracle_py as O\n"
                    "from audiomod_amd import signals\n"
                    f"O.LIB_PATH = {so!r}\n"
                    f"x = np.tile(signals.voice(10 * 48000, 2), (1, {(seconds + 9) // 10}))[:, :{seconds * 48000}]\n"
                    "t0 = time.perf_counter()\n"
                    "O.run_offline(x, mode='normal_pitchshift', semitones=4.0, coremode=1, fftsize=2048)\n"
                    "print(time.perf_counter() - t0)\n")
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
            dt = float(r.stdout.strip().splitlines()[-1])
        return {"value": round(seconds * 48000 * 2 / dt / 1e6, 4), "x_realtime": round(seconds / dt, 2), "cores": 1,
                "kind": "port", "flags": "-O3 -march=native",
                "sample": f"1 stereo stream x {seconds} s, wall {dt:.2f} s (480-frame calls from Python)"}
    except Exception:  # a figure for context only: never let it break the bench line
        return None


def usable_cores(cap=16):
    """Cores this job may really use: the affinity mask, cut down by the cgroup CPU quota when there is one and
    by `cap` (a one-GPU box's CPU share is 16 cores whatever the host has)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def copy_ceiling_gbps(torch, device, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (bytes read + bytes written per second): the practical HBM ceiling next to
    the 8 TB/s spec."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run as a CHILD process (this process has not imported
    torch or touched HIP yet, and never replaces itself: a process that has initialised the GPU must not exec)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def verify_outputs(torch, batch, d_in, out, dup_pairs, sample_streams, kw, flush, tol=1e-4):
    """Check what the timed steps produced (runs after the timed region, on this rank's batch): sampled streams
    against the oracle on the same input, duplicate-input streams bit for bit."""
    from oracle import oracle_py as O
    res = {"streams_checked": [int(s) for s in sample_streams], "tolerance_rms": tol}
    worst = 0.0
    for s in sample_streams:
        x = d_in[s].cpu().numpy()
        want, _, _ = O.run_offline(x, block=480, flush=flush, **kw)
        got = out[s].cpu().numpy()
        if got.shape != want.shape:
            worst = float("inf")
            continue
        worst = max(worst, float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))))
    res["max_rms_vs_oracle"] = worst
    res["duplicate_streams_bit_identical"] = bool(all(torch.equal(out[a].view(torch.int32), out[b].view(torch.int32))
                                                      for a, b in dup_pairs))
    res["duplicate_pairs"] = [[int(a), int(b)] for a, b in dup_pairs]
    res["ok"] = bool(worst <= tol and res["duplicate_streams_bit_identical"])
    return res


def host_io_leg(torch, E, args, kw, flush, device_index, d_in, d_out_ref, steps=2):
    """The same job with the staging included (never `value`): inputs and outputs in page-locked host memory, groups
    of streams triple-buffered through the GPU (H2D / kernels / D2H overlapped), float32 and int16 on the wire."""
    frames = args.seconds * 48000
    res = {}
    x_host = d_in.cpu().numpy()
    for wire in ("f32", "i16"):
        h = E.HostIO(args.streams, frames, channels=2, block=480, flush=flush, device=device_index,
                     streams_per_group=args.host_io_group, wire=wire, **kw)
        hin = h.pinned((args.streams, 2, frames))
        if wire == "f32":
            hin[...] = x_host
        else:
            hin[...] = np.round(x_host * 32768.0).astype(np.int16)  # the inputs sit on the int16 grid: exact
        hout = h.pinned((args.streams, 2, h.out_frames))
        h.run(hin, hout)  # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            h.run(hin, hout)
        dt = (time.perf_counter() - t0) / steps
        ref = d_out_ref.cpu().numpy()
        if wire == "f32":
            same = bool(np.array_equal(hout.view(np.uint32), ref.view(np.uint32)))
        else:
            v = np.clip(ref * np.float32(32768.0), -32768.0, 32767.0)
            same = bool(np.array_equal(hout, np.trunc(v).astype(np.int16)))
        bytes_each_way = args.streams * 2 * frames * hin.itemsize
        res[wire] = {"Msamples_s": round(args.streams * 2 * frames / dt / 1e6, 1),
                     "x_realtime": round(args.streams * args.seconds / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                     "pcie_GBps_each_way": round(bytes_each_way / dt / 1e9, 2),
                     "equals_device_resident_output": same}
        h.close()
    res["streams_per_group"] = args.host_io_group
    res["note"] = ("host buffers page-locked; three groups in flight (copy-in / kernels / copy-out on three HIP "
                   "streams); int16 converted on the device as the reference's WAV reader / writer do")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=128, help="stereo streams per GPU (cfg5: 1024 / 8)")
    ap.add_argument("--seconds", type=int, default=60, help="audio seconds per stream per step (BASELINE.md: 60)")
    ap.add_argument("--coremode", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS),
                    help="BASELINE config to run (default cfg2 = configs[1], the one the metric is quoted on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing check of the output")
    ap.add_argument("--dist-backend", default="gloo",
                    help="gloo (default: barrier + MAX reduce on CPU tensors, no RCCL on this path) or nccl")
    ap.add_argument("--force-device", type=int, default=-1, help="use this device on every rank (rehearsals only)")
    ap.add_argument("--groups", type=int, default=1,
                    help="split the streams into this many batches run concurrently on separate HIP streams")
    ap.add_argument("--sample-every", type=int, default=8, help="instrument every n-th chunk with HIP events")
    ap.add_argument("--host-io", action="store_true",
                    help="also measure the host-staged path (page-locked host in / out, H2D / kernels / D2H overlapped)")
    ap.add_argument("--host-io-group", type=int, default=32, help="streams per staged group of the --host-io leg")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))  # nothing below has run: no torch import, no HIP call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: started with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a "
                         f"line whose n_gpus would not match the request")

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    if args.force_device >= 0:  # rehearsal of the N > 1 path on a one-GPU box
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.dist_backend)
    red_device = device if args.dist_backend == "nccl" else None

    from audiomod_amd import engine as E
    from audiomod_amd import signals
    from audiomod_amd.sharding import max_over_ranks, sum_over_ranks
    frames = args.seconds * 48000
    cfg_name, cfg_kw, cfg_flush = CONFIGS[args.config]
    kw = dict(coremode=args.coremode, **cfg_kw)
    G = max(1, args.groups)
    assert args.streams % G == 0, "--streams must be a multiple of --groups"
    # two streams repeat the input of two others (far apart in the grid): equal inputs must give equal bits
    S = args.streams
    dup_pairs = [(S - 1, 0), (S // 2, 1)] if S >= 8 else []
    d_in = signals.synthetic_batch(torch, S, frames, device, rank, duplicates=dup_pairs)
    per = S // G
    batches = [E.Batch(per, frames, channels=2, block=480, flush=cfg_flush, device=local_rank, **kw) for _ in range(G)]
    ins = [d_in[g * per:(g + 1) * per] for g in range(G)]
    out_all = torch.empty((S, 2, batches[0].out_frames), dtype=torch.float32, device=device)
    outs = [out_all[g * per:(g + 1) * per] for g in range(G)]
    hip_streams = [torch.cuda.current_stream(device)] + [torch.cuda.Stream(device) for _ in range(G - 1)]
    batch = batches[0]
    info = batch.info()

    def step():
        for b, x, y, hs in zip(batches, ins, outs, hip_streams):
            b.run(x, y, stream=hs)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    warm_sum = None
    if not args.no_verify and args.warmup > 0:
        warm_sum = signals.batch_checksum(torch, out_all)[0]
        barrier()
    # HIP events around the kernels of every 8th chunk of group 0, inside the timed region (an event record
    # costs stream time: instrumenting every launch slows the run by ~10 %, every 8th by ~1 %)
    batch.enable_timing(args.sample_every)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ktimes = batch.kernel_times()
    batch.enable_timing(0)
    dt = max_over_ranks(dt, dist, red_device)
    ranks_reporting = int(round(sum_over_ranks(1.0, dist, red_device)))

    verified = None
    if not args.no_verify:
        digest, _ = signals.batch_checksum(torch, out_all)
        picks = sorted({0, S - 1} | ({S // 2, S // 3} if rank == 0 and S >= 8 else set()))
        verified = verify_outputs(torch, batch, d_in, out_all, dup_pairs, picks, kw, cfg_flush)
        verified["batch_checksum_sha256"] = digest
        verified["same_checksum_after_warmup_and_timed_steps"] = (warm_sum == digest) if warm_sum else None
        if warm_sum is not None and warm_sum != digest:
            verified["ok"] = False
        all_ok = sum_over_ranks(1.0 if verified["ok"] else 0.0, dist, red_device)
        verified["ranks_ok"] = int(round(all_ok))
        verified["ok"] = bool(verified["ok"] and verified["ranks_ok"] == world)
        verified["how"] = ("after the timed region, on the last step's output: sampled streams vs the oracle "
                           "(CPU restatement pinned on the compiled reference), duplicate inputs bit for bit, "
                           "exact integer checksum of the whole batch equal after warm-up and after the timed steps")

    host_io = None
    if args.host_io:
        host_io = host_io_leg(torch, E, args, kw, cfg_flush, local_rank, d_in, out_all)

    if rank == 0:
        total_streams = args.streams * world
        ch_samples = total_streams * frames * 2 * args.steps
        value = ch_samples / dt / 1e6
        xrt_gpu = args.streams * args.seconds * args.steps / dt
        slices_per_step_gpu = batch.slices * 2 * args.streams
        slices_per_launch = batch.slices * 2 * per / batch.launches  # one launch = one chunk of one group
        N, H, s, h = info["fftsize"], info["fftsize"] // 2 + 1, info["hop_out_nominal"], info["hop_in"]
        rs = 1 if info["resample"] else 0  # without resampling SURVEY 8(d) drops the s + h terms
        # Algorithmic bytes per slice of each pipeline stage; they sum to SURVEY 8(d)'s
        # B_slice = 4*(3N+7H+2s+h).  The phase stage runs as two kernels (match + seq) in phase-locked
        # mode and as one (prop) in coremode 0; a stage's time is the sum over its kernels.  The fused
        # synthesis + overlap-add kernel carries the bytes of both of its stages.
        stages = {
            "analysis": (4 * (N + 2 * H), ["pv_analyze_kernel"]),
            "phase": (4 * (3 * H), ["pv_match_kernel", "pv_seq_kernel", "pv_prop_kernel"]),
            "synthesis": (4 * (2 * H + N), ["pv_synth_kernel"]),
            "ola_resample": (4 * (N + s + rs * (s + h)), ["pv_ola_kernel"]),
        }
        assert sum(b for b, _ in stages.values()) == info["bytes_per_slice"]
        per_kernel, per_stage = {}, {}
        for k, (ms, n) in ktimes.items():
            if n:
                per_kernel[k] = {"avg_ms": round(ms / n, 4), "samples": int(n)}
        if "pv_synth_ola_kernel" in per_kernel:
            # fused path: synthesis + overlap-add in one kernel (frames never leave LDS); where the configuration
            # resamples, pv_ola_kernel is the resampling kernel that follows it.  The fused kernel carries the
            # synthesis stage's bytes and the overlap-add's N + s; the resampler its s + h.
            stages["synthesis"] = (4 * (2 * H + N) + 4 * (N + s), ["pv_synth_ola_kernel"])
            if rs:
                stages["ola_resample"] = (4 * (s + h), ["pv_ola_kernel"])
            else:
                del stages["ola_resample"]
        # Counter-derived HBM bytes per launch (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
        # this workload, tools/profile_round.sh): quoted only when they were measured on THIS code (source hash), this
        # geometry and this arithmetic setting -- a stale profile silently describing other kernels is worse than null.
        traffic_db, traffic_note = {}, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        src_hash = kernel_source_hash()
        arith = "exact" if E.get_arithmetic() == E.ARITH_EXACT else "fast"
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("source_sha16") == src_hash and tj.get("arithmetic", "exact") == arith:
                traffic_db = tj.get("bytes_per_launch", {})
            else:
                traffic_note = (f"profiles/traffic.json was measured on other code or settings (its source hash "
                                f"{tj.get('source_sha16')}, arithmetic {tj.get('arithmetic')}; this run {src_hash}, {arith})")
        for name, (nbytes, ks) in stages.items():
            live = [k for k in ks if k in per_kernel]
            if not live:
                continue
            avg_ms = sum(per_kernel[k]["avg_ms"] for k in live)
            per_stage[name] = {"kernels": live, "avg_ms": round(avg_ms, 4), "bytes_per_slice": nbytes,
                               "GBps": round(nbytes * slices_per_launch / (avg_ms * 1e-3) / 1e9, 1)}
        # With the pipelined batch path the phase stage's sequential kernel runs on a second HIP stream beside the others:
        # its duration is not on the critical path, so the roofline entry is the longest kernel of the MAIN stream
        # (a stage's GB/s is still computed over all of its kernels' durations).
        overlapped = [k for k in ("pv_seq_kernel", "pv_prop_kernel") if batch.pipelined and k in per_kernel]
        main = {k: v for k, v in per_kernel.items() if k not in overlapped}
        if not main:  # --sample-every 0 (counter passes: no events in the stream): nothing per kernel to report
            main = {"none": {"avg_ms": float("nan")}}
            stages["none"] = (0, ["none"])
            per_stage["none"] = {"GBps": 0.0}
        dom = max(main, key=lambda k: main[k]["avg_ms"])
        dom_stage = next(n for n, (_, ks) in stages.items() if dom in ks and n in per_stage)
        dom_bytes = stages[dom_stage][0]
        achieved = round(dom_bytes * slices_per_launch / (main[dom]["avg_ms"] * 1e-3) / 1e9, 1) \
            if dom_stage != "phase" or not overlapped else per_stage[dom_stage]["GBps"]
        if dom == "none":
            achieved = 0.0
        std_geometry = G == 1 and args.streams == 128 and args.config == "cfg2" and args.seconds == 60
        traffic = traffic_db.get(dom) if std_geometry else None
        hbm_real = None
        if traffic:
            hbm_real = traffic / (main[dom]["avg_ms"] * 1e-3) / 1e9  # the kernel's real HBM rate (counter bytes / live duration)
        # Beside the contract's HBM figure: how close the same kernel runs to the chip's vector-instruction issue rate,
        # which is what bounds this path.  Instruction counts per slice and the mean issue cost
        # of the kernel's instruction mix are a profile's (profiles/valu.json: PMC pass + tools/pk_probe.hip's price
        # list), the duration is this run's.
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vpath) and std_geometry and args.coremode == 1:
            vdb = json.load(open(vpath))
            vk = vdb.get("kernels", {}).get(dom) if (vdb.get("source_sha16") == src_hash and
                                                     vdb.get("arithmetic", "exact") == arith) else None
            if vk:
                rate = vk["valu_insts_per_slice"] * slices_per_launch * vk["mean_cost"] / main[dom]["avg_ms"] / 1e6
                valu = {"kernel": dom, "valu_insts_per_slice": vk["valu_insts_per_slice"], "mean_issue_cost": vk["mean_cost"],
                        "weighted_M_insts_per_ms": round(rate, 1), "chip_peak_M_per_ms": vdb["chip_peak_M_per_ms"],
                        "frac": round(rate / vdb["chip_peak_M_per_ms"], 3),
                        "source": "profiles/valu.json (SQ_INSTS_VALU pass, static mix priced by tools/pk_probe.hip)"}
        pipeline_gbps = info["bytes_per_slice"] * slices_per_step_gpu * args.steps / dt / 1e9
        copy_gbps = copy_ceiling_gbps(torch, device)
        # What binds the dominant kernel.  `achieved` is the contract's figure (algorithmic bytes / duration); it says how
        # fast the work is done, not that HBM is busy.  When the kernel moves fewer real bytes than that (hbm_real_frac
        # below frac) and runs closer to the chip's vector-issue rate than to its real HBM rate, the bound is named so.
        bound = "hbm"
        if valu is not None and hbm_real is not None and valu["frac"] > hbm_real / HBM_PEAK_GBPS:
            bound = "valu_issue"
        line = {
            "metric": "Msamples/s (48 kHz stereo) phase-vocoder pitch-shift; x real-time per GPU",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": ranks_reporting, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "x_realtime_per_gpu": round(xrt_gpu, 1),
            "config": {"workload": cfg_name + (", phase-locked" if args.coremode == 1 else f", coremode {args.coremode}"),
                       "arithmetic": arith + (" (synthesis side free within the 1e-4 RMS contract; analysis and phase "
                                              "propagation bit-exact)" if arith == "fast" else
                                              " (reference operation order everywhere)"),
                       "streams_per_gpu": args.streams, "seconds_per_stream": args.seconds, "channels": 2,
                       "block": 480, "hop_in": h, "slices_per_channel": int(batch.slices),
                       "concurrent_stream_groups": G, "launches_per_step": int(batch.launches) * G,
                       "parallelism": f"stream-sharded x{world}, no collective",
                       "rank_sync": f"{args.dist_backend} (barrier + MAX of the wall time only)" if world > 1 else "none"},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "hbm_real_GBps": round(hbm_real, 1) if hbm_real else None,
                         "hbm_real_frac": round(hbm_real / HBM_PEAK_GBPS, 4) if hbm_real else None,
                         "traffic_note": traffic_note, "source_sha16": src_hash,
                         "copy_ceiling_GBps": round(copy_gbps, 1), "frac_of_copy_ceiling": round(achieved / copy_gbps, 4),
                         "stage": dom_stage, "slices_per_launch": round(slices_per_launch, 1),
                         "pipeline_GBps": round(pipeline_gbps, 1),
                         "pipeline_frac": round(pipeline_gbps / HBM_PEAK_GBPS, 4),
                         "overlapped_on_second_stream": overlapped,
                         "valu_issue": valu,
                         "per_stage": per_stage,
                         "per_kernel": per_kernel},
        }
        if verified is not None:
            line["verified"] = verified
        if host_io is not None:
            line["host_io"] = host_io
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if verified is not None and not verified["ok"]:
        raise SystemExit("bench.py: the timed output failed its check (see \"verified\" in the line above)")


if __name__ == "__main__":
    main()
