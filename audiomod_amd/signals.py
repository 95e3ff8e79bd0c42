"""Deterministic synthetic test/bench signals.

48 kHz float32 in [-1, 1) on the int16 grid (what the reference's WAV reader hands the
effect: int16 * 1/32768, /root/reference/main/wavfile.cc:733-755).  Per channel: 19 harmonics
(1/k amplitude) of f0 with 5 Hz / 0.2 % vibrato and 0.7 Hz tremolo plus Gaussian noise
(sigma 0.02), normalised to peak 0.25.  Fixed seed -> bit-reproducible on any host.
"""
import numpy as np

F0 = (220.0, 277.0)


def voice(frames, channels=2, seed=1234, sample_rate=48000, stream=0, dtype=np.float32):
    rng = np.random.default_rng(seed + stream)
    t = np.arange(frames, dtype=np.float64) / sample_rate
    out = np.empty((channels, frames), dtype=np.float64)
    jitter = 1.0 + 0.03 * ((stream * 0.6180339887) % 1.0 - 0.5) if stream else 1.0
    for c in range(channels):
        f0 = F0[c % 2] * jitter * (1.0 + 0.5 * (c // 2))
        # phase of a vibrato'd carrier: integral of f0 * (1 + 0.002 sin(2 pi 5 t))
        ph = 2 * np.pi * f0 * (t - 0.002 / (2 * np.pi * 5.0) * (np.cos(2 * np.pi * 5.0 * t) - 1.0))
        x = np.zeros(frames, dtype=np.float64)
        for k in range(1, 20):
            if k * f0 < 0.45 * sample_rate:
                x += np.sin(k * ph + 0.37 * k * (c + 1)) / k
        x *= 1.0 + 0.3 * np.sin(2 * np.pi * 0.7 * t + c)
        x += rng.normal(0.0, 0.02, frames) * np.max(np.abs(x)) / 0.25 * 0.25
        x *= 0.25 / np.max(np.abs(x))
        out[c] = x
    q = np.clip(np.round(out * 32768.0), -32768, 32767) / 32768.0
    return q.astype(dtype)


def sweep(frames, channels=2, sample_rate=48000):
    t = np.arange(frames, dtype=np.float64) / sample_rate
    dur = frames / sample_rate
    out = np.empty((channels, frames))
    for c in range(channels):
        f_lo, f_hi = 80.0 * (c + 1), 12000.0 / (c + 1)
        k = (f_hi / f_lo) ** (1.0 / dur)
        ph = 2 * np.pi * f_lo * (k ** t - 1.0) / np.log(k)
        out[c] = 0.3 * np.sin(ph)
    return (np.round(out * 32768.0) / 32768.0).astype(np.float32)


def noise(frames, channels=2, seed=99):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.4, 0.4, (channels, frames))
    return (np.round(x * 32768.0) / 32768.0).astype(np.float32)


def silence_burst(frames, channels=2, seed=7, sample_rate=48000):
    """exact digital silence, then a burst, then silence again (exercises the empty-peak branch)."""
    x = np.zeros((channels, frames), dtype=np.float32)
    a, b = frames // 3, 2 * frames // 3
    x[:, a:b] = voice(b - a, channels, seed, sample_rate)
    return x


def dual_mono(frames, seed=1234, sample_rate=48000):
    """L == R: pins the reference's cross-channel state sharing."""
    m = voice(frames, 1, seed, sample_rate)
    return np.concatenate([m, m], axis=0)


def synthetic_batch(torch, streams, frames, device, rank=0, channels=2, duplicates=()):
    """[streams, channels, frames] float32 on `device`, on the int16 grid: a few host-synthesised voices varied per
    stream on the GPU (rolled and scaled), so a batch of hundreds of minute-long streams costs seconds to make.
    `duplicates`: pairs (dst, src) -- stream dst becomes a copy of stream src (independent streams holding the
    same input must produce the same bits)."""
    base_n = min(frames, 10 * 48000)
    nb = 4
    base = np.stack([voice(base_n, channels, stream=rank * nb + i) for i in range(nb)])
    b = torch.from_numpy(base).to(device)
    reps = (frames + base_n - 1) // base_n
    x = torch.empty((streams, channels, frames), dtype=torch.float32, device=device)
    for s in range(streams):
        v = b[s % nb].repeat(1, reps)[:, :frames]
        v = torch.roll(v, shifts=(s // nb) * 4099, dims=1) * (1.0 - 0.01 * (s % 7))
        x[s] = torch.round(v * 32768.0) / 32768.0
    for dst, src in duplicates:
        x[dst] = x[src]
    return x


def batch_checksum(torch, y, group=16):
    """Exact checksum of a float32 device tensor [streams, ...]: per stream the wrapping int64 sum of the samples'
    bit patterns, each weighted by (position mod 251) + 1, then SHA-256 over the per-stream sums ("a checksum of
    checksums").  Integer arithmetic only, so it does not depend on reduction order; computed on the device, a
    few streams at a time.  Returns (hex digest, int64 tensor of per-stream sums on the host)."""
    import hashlib
    flat = y.contiguous().view(torch.int32).reshape(y.shape[0], -1)
    w = (torch.arange(flat.shape[1], device=y.device, dtype=torch.int64) % 251) + 1
    sums = torch.empty(flat.shape[0], dtype=torch.int64, device=y.device)
    for s0 in range(0, flat.shape[0], group):
        sums[s0:s0 + group] = (flat[s0:s0 + group].to(torch.int64) * w).sum(dim=1)
    sums = sums.cpu()
    return hashlib.sha256(sums.numpy().tobytes()).hexdigest(), sums
