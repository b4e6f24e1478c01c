"""Synthetic moving-glyph sequences with the tensor layout and value ranges of the reference's
moving multi-MNIST (MNIST itself is not available offline).

Generator semantics restated from the reference:
  * static frame: ``n_obj ~ U{n_min..n_max}`` glyphs per sequence, tight-cropped <= 28x28 templates,
    initial top-left corner uniform such that the glyph lies fully inside the canvas, retried a few
    times to avoid overlap (sqair/data/data.py:64-186);
  * motion: position/velocity/acceleration state, vel ~ U[-10,10], acc ~ U[-3,3], one forward step
    at t=0, acceleration noise N(0, 0.01), clipping to the bounds and *bouncing* at the position
    bounds [0, canvas - overlap*28] of the top-left corner (sqair/data/trajectory.py:54-143,
    sqair/data/create_seq_mnist.py:43-56) — glyphs may slide partly off the bottom/right edge;
  * rendering: integer-rounded positions, glyphs clipped to the canvas and max-blended
    (sqair/data/template.py:69-104), the whole set rescaled to uint8 by its global min/max
    (template.py:38-42) and fed as float32 / 255 (sqair/data/data.py:199).

Procedural stroke glyphs stand in for the MNIST digits.  Returns the dict the reference pickles:
imgs uint8 [T,N,H,W], nums [T,N,n_max+1] prefix-ones, coords [T,N,n_max,4] (y,x,h,w).
"""
from __future__ import annotations

import numpy as np


def make_glyph(rng, size=28):
    """A random 'digit-like' glyph: 2-4 thick polyline strokes, blurred, tight-cropped, in [0,1]."""
    hi = 4 * size
    img = np.zeros((hi, hi), dtype=np.float64)
    n_strokes = int(rng.integers(2, 5))
    yy, xx = np.mgrid[0:hi, 0:hi]
    for _ in range(n_strokes):
        pts = rng.uniform(0.2 * hi, 0.8 * hi, size=(int(rng.integers(2, 4)), 2))
        for a, b in zip(pts[:-1], pts[1:]):
            d = b - a
            L2 = float(d @ d) + 1e-9
            tpar = np.clip(((yy - a[0]) * d[0] + (xx - a[1]) * d[1]) / L2, 0.0, 1.0)
            dist2 = (yy - (a[0] + tpar * d[0])) ** 2 + (xx - (a[1] + tpar * d[1])) ** 2
            img = np.maximum(img, np.exp(-dist2 / (2.0 * (0.035 * hi) ** 2)))
    img = img.reshape(size, 4, size, 4).mean((1, 3))
    img[img < 0.15] = 0.0
    ys, xs = np.nonzero(img)
    img = img[ys.min():ys.max() + 1, xs.min():xs.max() + 1]
    return img / img.max()


def _trajectory(rng, init_pos, T, bounds, noise_std=0.01, max_speed=10.0, max_acc=3.0):
    """NoisyAccelerationTrajectory(bounce=True) for one object (trajectory.py:109-143)."""
    lo = np.array([bounds[0][0], bounds[1][0], -max_speed, -max_speed, -max_acc, -max_acc], dtype=np.float64)
    hi = np.array([bounds[0][1], bounds[1][1], max_speed, max_speed, max_acc, max_acc], dtype=np.float64)
    state = lo + rng.uniform(size=6) * (hi - lo)

    def fwd(state):
        pos, vel, acc = state[0:2].copy(), state[2:4].copy(), state[4:6].copy()
        pos += vel
        vel += acc
        acc += rng.normal(0.0, noise_std, size=2)
        for dd in range(2):
            if pos[dd] < lo[dd]:
                pos[dd] = 2 * lo[dd] - pos[dd]; vel[dd] *= -1; acc[dd] *= -1
            elif pos[dd] > hi[dd]:
                pos[dd] = 2 * hi[dd] - pos[dd]; vel[dd] *= -1; acc[dd] *= -1
        return np.clip(np.concatenate([pos, vel, acc]), lo, hi)

    state = fwd(state)
    state[0:2] = init_pos  # create(init_from=...) overrides the first position
    out = np.empty((T, 2))
    out[0] = init_pos
    for t in range(1, T):
        state = fwd(state)
        out[t] = state[0:2]
    return out


def make_sequences(n_seq, T=10, canvas=(50, 50), n_objects=(0, 2), obj_size=28, seed=1234, overlap=0.0,
                   n_glyphs=32):
    rng = np.random.default_rng(seed)
    H, W = canvas
    n_min, n_max = n_objects
    glyphs = [make_glyph(rng, obj_size) for _ in range(n_glyphs)]
    imgs = np.zeros((T, n_seq, H, W), dtype=np.float64)
    nums = np.zeros((T, n_seq, n_max + 1), dtype=np.float32)
    coords = np.zeros((T, n_seq, n_max, 4), dtype=np.float32)
    bounds = [[-overlap * obj_size, H - overlap * obj_size], [-overlap * obj_size, W - overlap * obj_size]]
    for i in range(n_seq):
        n = int(rng.integers(n_min, n_max + 1))
        nums[:, i, :n] = 1.0
        placed = []
        for o in range(n):
            g = glyphs[int(rng.integers(len(glyphs)))]
            gh, gw = g.shape
            for _ in range(5):  # data.py:140-149: retry to avoid overlap
                y0 = rng.uniform(0, H - gh)
                x0 = rng.uniform(0, W - gw)
                if all(y0 + gh <= py or py + ph <= y0 or x0 + gw <= px or px + pw <= x0 for py, px, ph, pw in placed):
                    break
            placed.append((y0, x0, gh, gw))
            tj = _trajectory(rng, np.array([y0, x0]), T, bounds)
            for t in range(T):
                y, x = int(np.round(tj[t, 0])), int(np.round(tj[t, 1]))
                ys, ye, xs, xe = max(y, 0), min(y + gh, H), max(x, 0), min(x + gw, W)
                if ye > ys and xe > xs:
                    patch = g[ys - y:ye - y, xs - x:xe - x]
                    imgs[t, i, ys:ye, xs:xe] = np.maximum(imgs[t, i, ys:ye, xs:xe], patch)
                coords[t, i, o] = (tj[t, 0], tj[t, 1], gh, gw)
    lo, hi = imgs.min(), max(imgs.max(), 1e-9)
    u8 = np.round((imgs - lo) / (hi - lo) * 255.0).astype(np.uint8)
    return dict(imgs=u8, nums=nums, coords=coords)


def to_float(imgs_u8):
    """uint8 -> float32 in [0,1] (data.py:199)."""
    return imgs_u8.astype(np.float32) / 255.0


def config_inputs(cfg_id, B=None):
    """The five BASELINE.json configurations as concrete inputs.
    Returns (flags overrides, obs float32 [T,B,H,W], nums, coords)."""
    table = {
        1: dict(T=3, B=4, K=1, N=3, n_max=2, hw=(50, 50), obj=28),
        2: dict(T=10, B=32, K=5, N=4, n_max=2, hw=(50, 50), obj=28),
        3: dict(T=10, B=32, K=5, N=4, n_max=2, hw=(50, 50), obj=28),   # per-GPU shard of the global 256
        4: dict(T=10, B=64, K=5, N=6, n_max=4, hw=(50, 50), obj=28),
        5: dict(T=10, B=32, K=5, N=4, n_max=2, hw=(128, 128), obj=72),
    }
    c = table[int(cfg_id)]
    nb = int(B) if B is not None else c["B"]
    d = make_sequences(nb, T=c["T"], canvas=c["hw"], n_objects=(0, c["n_max"]), obj_size=c["obj"],
                       seed=1234 + int(cfg_id))
    overrides = dict(k_particles=c["K"], n_steps_per_image=c["N"])
    return overrides, to_float(d["imgs"]), d["nums"], d["coords"]
