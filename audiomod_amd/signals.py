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
