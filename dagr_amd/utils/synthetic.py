"""Synthetic event streams for tests and bench.

Output contract = the reference dataset's (``src/dagr/data/utils.py:6-19``, ``dsec_data.py:141-147``):
x,y int16, t int32 (us, shifted so that the last event sits at ``time_window``), p int8 in {-1,+1}.
"""
import numpy as np


def _finish(x, y, t, rng, time_window):
    order = np.argsort(t, kind="stable")
    x, y, t = x[order], y[order], t[order]
    if len(t) > 0:
        t = time_window + t - t[-1]  # dsec_data.py:145
    p = (2 * rng.integers(0, 2, len(t)) - 1).astype(np.int8)  # dsec_data.py:146
    return x.astype(np.int16), y.astype(np.int16), t.astype(np.int32), p


def uniform_window(n, width, height, seed, window_us=50000, time_window=1000000):
    """S-uniform: x,y uniform over the sensor, t uniform over a 50 ms window."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.integers(0, width, n)
    y = rng.integers(0, height, n)
    t = rng.integers(0, window_us, n)
    return _finish(x, y, t, rng, time_window)


def edges_window(n, width, height, seed, window_us=50000, time_window=1000000, n_lines=20, noise=0.1):
    """S-edges: moving line segments (<= 2 px/ms) with N(0,1.5^2) px jitter + uniform noise.
    Saturates K=16 and stresses the per-pixel FIFO depth."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_noise = int(noise * n)
    n_sig = n - n_noise
    line = rng.integers(0, n_lines, n_sig)
    x0 = rng.uniform(0, width, n_lines)
    y0 = rng.uniform(0, height, n_lines)
    ang = rng.uniform(0, np.pi, n_lines)
    length = rng.uniform(0.1, 0.4, n_lines) * width
    vx = rng.uniform(-2, 2, n_lines) / 1000.0  # px/us
    vy = rng.uniform(-2, 2, n_lines) / 1000.0
    t = rng.integers(0, window_us, n_sig)
    s = rng.uniform(-0.5, 0.5, n_sig)
    xs = x0[line] + s * length[line] * np.cos(ang[line]) + vx[line] * t + rng.normal(0, 1.5, n_sig)
    ys = y0[line] + s * length[line] * np.sin(ang[line]) + vy[line] * t + rng.normal(0, 1.5, n_sig)
    xs = np.clip(np.rint(xs), 0, width - 1).astype(np.int64)
    ys = np.clip(np.rint(ys), 0, height - 1).astype(np.int64)
    xn = rng.integers(0, width, n_noise)
    yn = rng.integers(0, height, n_noise)
    tn = rng.integers(0, window_us, n_noise)
    return _finish(np.concatenate([xs, xn]), np.concatenate([ys, yn]), np.concatenate([t, tn]), rng, time_window)


def batch_windows(gen, n_per_sample, batch_size, width, height, seed, **kw):
    """Concatenate ``batch_size`` independent windows like PyG ``Batch`` collation does:
    returns x,y,t,p and ``batch`` (int64 sample index, sorted)."""
    xs, ys, ts, ps, bs = [], [], [], [], []
    for b in range(batch_size):
        x, y, t, p = gen(n_per_sample, width, height, seed + b, **kw)
        xs.append(x); ys.append(y); ts.append(t); ps.append(p)
        bs.append(np.full(len(x), b, np.int64))
    return (np.concatenate(xs), np.concatenate(ys), np.concatenate(ts), np.concatenate(ps), np.concatenate(bs))


def format_data_np(x, y, t, width, height, time_window=1000000):
    """numpy twin of ``format_data`` (src/dagr/utils/buffers.py:33-44): fp32 true division."""
    pos = np.stack([x.astype(np.float32) / np.float32(width), y.astype(np.float32) / np.float32(height),
                    t.astype(np.float32) / np.float32(time_window)], axis=-1)
    return np.ascontiguousarray(pos.astype(np.float32))
