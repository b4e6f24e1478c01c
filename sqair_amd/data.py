"""Moving-glyph sequence generator: the reference's moving multi-MNIST pipeline restated step by step, with procedural
glyphs standing in for the MNIST digits (MNIST is not available offline).  SURVEY.md 8(f) rank 2.

Reference pipeline (sqair/data/create_seq_mnist.py:89-131) and where each stage lives here:

  * static frame — ``create_mnist`` (sqair/data/data.py:64-186): ``nums ~ randint(n_min, n_max + 1)`` objects per sample;
    each template is tight-cropped to its non-zero bounding box (``template_dimensions``, data.py:47-60);
    its top-left corner is ``round(rand(2) * (canvas - size))`` (``make_coord`` with ``fraction_outside_canvas = 0``),
    re-drawn while the box overlaps an occupied region, at most 5 draws in total per SAMPLE, after which the whole sample
    is started again                                                   -> ``place_templates``
  * motion — ``NoisyAccelerationTrajectory(noise_std=.01, n_dim=2, pos_bounds, max_speed=10, max_acc=3, bounce=True)``
    (sqair/data/trajectory.py:109-143; base class :28-106) with the top-left corner bounded to
    ``[-overlap * template, canvas - overlap * template]`` = ``[0, canvas]`` for ``overlap = 0``
    (create_seq_mnist.py:43-56, :98)                                   -> ``NoisyAccelerationTrajectory``, ``position_bounds``
  * rendering — ``TemplateDataset._blend`` (sqair/data/template.py:69-104): positions rounded to integers, templates
    clipped to the canvas and merged with ``np.maximum``              -> ``blend``, ``_visible_span``
  * ``convert_img_dtype`` (template.py:38-42): ``(imgs - min) / (max / 255)`` then a TRUNCATING cast to uint8
                                                                       -> ``convert_img_dtype``
  * fed as ``float32 / 255`` (data.py:199)                             -> ``to_float``

Random numbers: the reference draws from numpy's legacy global generator; every function here takes a
``numpy.random.RandomState`` and consumes it in the reference's call order, so that a trajectory / blend computed here is
bit-identical to the reference's for the same seed (tests/test_generator.py, fixtures produced by running the reference's
own trajectory.py / template.py: tests/golden/make_generator_golden.py).

Returns the dict the reference pickles: imgs uint8 [T,N,H,W], nums prefix-ones, coords [T,N,n_max,4] (y, x, h, w).
"""
from __future__ import annotations

import numpy as np


# ---------------------------------------------------------------------------------------------------- motion
def position_bounds(canvas_size, template_size, overlap=0.0):
    """Bounds of a template's TOP-LEFT corner (create_seq_mnist.py:43-50): ``[-overlap * size, canvas - overlap * size]``
    per axis — with the shipped ``overlap = 0`` that is ``[0, canvas]``: templates may slide off the bottom / right edge
    completely and bounce back."""
    ts = np.asarray(template_size, dtype=np.float64)
    allowed = np.asarray(canvas_size, dtype=np.float64) - overlap * ts
    return [[-overlap * ts[0], allowed[0]], [-overlap * ts[1], allowed[1]]]


class NoisyAccelerationTrajectory(object):
    """State = [position(2), velocity(2), acceleration(2)] per trajectory, all trajectories advanced together
    (trajectory.py:109-143).  One step: ``pos += vel; vel += acc; acc += N(0, noise_std)``; with ``bounce`` a position
    outside its bounds is reflected about the bound it crossed and velocity and acceleration of that axis change sign;
    then the WHOLE state is clipped to its bounds (trajectory.py:75-80) — positions to ``pos_bounds``, velocities to
    ``+-max_speed``, accelerations to ``+-max_acc``."""

    def __init__(self, noise_std, pos_bounds, max_speed, max_acc, bounce=False, n_dim=2):
        self.noise_std, self.bounce, self.n_dim = noise_std, bool(bounce), int(n_dim)
        self.bounds = np.asarray(list(pos_bounds) + [[-max_speed, max_speed]] * n_dim + [[-max_acc, max_acc]] * n_dim)
        assert self.bounds.shape == (3 * n_dim, 2)

    def _advance(self, state, rng):
        d = self.n_dim
        acc_noise = rng.normal(0, self.noise_std, size=(state.shape[0], d))
        pos, vel, acc = state[:, :d].copy(), state[:, d:2 * d].copy(), state[:, 2 * d:].copy()
        pos += vel
        vel += acc
        acc += acc_noise
        if self.bounce:
            for k in range(d):
                too_small = pos[:, k] < self.bounds[k, 0]
                too_big = pos[:, k] > self.bounds[k, 1]
                pos[too_small, k] = 2 * self.bounds[k, 0] - pos[too_small, k]
                pos[too_big, k] = 2 * self.bounds[k, 1] - pos[too_big, k]
                flip = np.logical_or(too_small, too_big)
                vel[flip, k] *= -1
                acc[flip, k] *= -1
        return np.concatenate((pos, vel, acc), -1)

    def forward(self, state, rng):
        """-> (points [n, n_dim], new state [n, 3 n_dim]) (trajectory.py:78-80)."""
        state = np.clip(self._advance(state, rng), self.bounds[:, 0], self.bounds[:, 1])
        return state[:, :self.n_dim].copy(), state

    def init(self, n, rng):
        """Uniform state inside the bounds, propagated forward once (trajectory.py:54-73)."""
        state = rng.uniform(size=(n, 3 * self.n_dim))
        state = self.bounds[None, :, 0] + state * (self.bounds[None, :, 1] - self.bounds[None, :, 0])
        return self.forward(state, rng)

    def create(self, n_timesteps, n, rng, init_from=None):
        """[T, n, n_dim] float32 (trajectory.py:82-106): the first point — and the position part of the state — is
        overridden by ``init_from`` (the static frame's coordinates), velocity and acceleration keep their random
        initialisation."""
        tjs = np.empty((n_timesteps, n, self.n_dim), dtype=np.float32)
        tjs[0], state = self.init(n, rng)
        if init_from is not None:
            tjs[0] = init_from
            state[:, :self.n_dim] = np.asarray(init_from).copy()
        for t in range(1, n_timesteps):
            tjs[t], state = self.forward(state, rng)
        return tjs


# ---------------------------------------------------------------------------------------------------- rendering
def _visible_span(first, extent, limit):
    """Intersection of the interval [first, first + extent) with a canvas axis [0, limit): returns (canvas_lo, canvas_hi,
    template_lo) — the canvas slice that is covered and the template index it starts from.  An object entirely off the canvas
    gives an empty slice.  (Same result as the reference's clip / slice arithmetic, template.py:31-35, :85-89, expressed as an
    interval intersection; pinned bit for bit by tests/golden/generator_ref.npz.)"""
    lo = min(max(first, 0), limit)
    hi = max(min(max(first + extent, 0), limit), lo)
    return lo, hi, lo - first


def blend(canvas, template, pos):
    """Max-blends ``template`` into ``canvas`` (in place) with its top-left corner at ``round(pos)``; parts outside the
    canvas are cut off (template.py:69-104)."""
    corner = np.round(pos).astype(np.int64)
    rows = _visible_span(int(corner[0]), template.shape[0], canvas.shape[0])
    cols = _visible_span(int(corner[1]), template.shape[1], canvas.shape[1])
    (r0, r1, tr), (c0, c1, tc) = rows, cols
    if r1 <= r0 or c1 <= c0:
        return
    patch = canvas[r0:r1, c0:c1]
    np.maximum(patch, template[tr:tr + (r1 - r0), tc:tc + (c1 - c0)], out=patch)


def convert_img_dtype(imgs, dtype=np.uint8):
    """Whole-dataset rescale to uint8 (template.py:38-42).  Quirks kept on purpose: the offset is the global minimum but the
    scale is ``max / 255`` (not the range), and the float -> uint8 cast truncates instead of rounding."""
    if dtype != np.uint8:
        return imgs
    shifted = imgs - imgs.min()
    return (shifted / (imgs.max() / 255.)).astype(np.uint8)


def to_float(imgs_u8):
    """uint8 -> float32 in [0,1] (data.py:199)."""
    return imgs_u8.astype(np.float32) / 255.0


# ---------------------------------------------------------------------------------------------------- static frame
def template_dimensions(template):
    """Tight bounding box of the non-zero pixels of a template as ((y, x), (height, width)) (data.py:47-60).  Like the
    reference, the size along an axis is the NUMBER of non-empty rows / columns (not last - first + 1) and the start is the
    last non-empty index minus that count plus one — identical for the solid glyphs used here and for MNIST digits."""
    out = []
    for axis in (1, 0):
        occupied = np.flatnonzero(template.sum(axis) > 0.)
        count = int(occupied.size)
        last = int(occupied[-1]) if count else 0
        out.append((last - count + 1, count))
    (y, h), (x, w) = out
    return (y, x), (h, w)


def make_coord(size, canvas_size, rng, fraction_outside_canvas=0.0):
    """Random top-left corner for a template of ``size`` (data.py:98-115): uniform over the positions that keep at most
    ``fraction_outside_canvas`` of the template off the canvas, rounded to whole pixels.  Consumes ``rng.rand(2)``."""
    f = float(fraction_outside_canvas)
    extent = np.asarray(size, dtype=np.float64)
    free = np.asarray(canvas_size, dtype=np.float64) - (1.0 - 2.0 * f) * extent
    return np.round(rng.rand(2) * free - f * extent).astype(np.int32)


def place_templates(templates, canvas_size, rng, with_overlap=False, n_tries=5):
    """Positions for the templates of ONE sample (data.py:127-170): a position is re-drawn while its box touches an occupied
    pixel; the try counter is shared by the objects of the sample and when it reaches ``n_tries`` the sample is abandoned
    (None is returned and the caller starts it again, data.py:172-175)."""
    occupancy = np.zeros(canvas_size, dtype=bool)
    tries, out = 0, []
    for tpl in templates:
        size = np.asarray(tpl.shape[:2])
        pos = make_coord(size, canvas_size, rng)
        if not with_overlap:
            tp = np.maximum(pos, 0)
            while occupancy[tp[0]:tp[0] + size[0], tp[1]:tp[1] + size[1]].any() and tries < n_tries:
                pos = make_coord(size, canvas_size, rng)
                tp = np.maximum(pos, 0)
                tries += 1
            if tries == n_tries:
                return None
        tp = np.maximum(pos, 0)
        occupancy[tp[0]:tp[0] + size[0], tp[1]:tp[1] + size[1]] = True
        out.append(pos)
    return out


# ---------------------------------------------------------------------------------------------------- glyphs (MNIST stand-in)
def make_glyph(rng, size=28):
    """A digit-like template with MNIST's statistics: strokes ~2.5 px wide on a ``size`` x ``size`` field with a saturated
    core (most stroke pixels at 1.0, soft one-pixel edges), confined to the central 20/28 of the field like MNIST's
    size-normalised digits, then tight-cropped to its bounding box (data.py:141-143).  Values in [0, 1]."""
    hi = 4 * size
    img = np.zeros((hi, hi), dtype=np.float64)
    yy, xx = np.mgrid[0:hi, 0:hi]
    lo_f, hi_f = 5.5 / 28.0 * hi, 22.5 / 28.0 * hi
    s_hi = 4.2 * size / 28.0   # profile scale in hi-res pixels: saturated core ~2 px wide, soft edges out to ~3 px
    for _ in range(int(rng.randint(2, 5))):
        pts = rng.uniform(lo_f, hi_f, size=(int(rng.randint(2, 4)), 2))
        for a, b in zip(pts[:-1], pts[1:]):
            d = b - a
            L2 = float(d @ d) + 1e-9
            tpar = np.clip(((yy - a[0]) * d[0] + (xx - a[1]) * d[1]) / L2, 0.0, 1.0)
            dist2 = (yy - (a[0] + tpar * d[0])) ** 2 + (xx - (a[1] + tpar * d[1])) ** 2
            img = np.maximum(img, np.exp(-dist2 / (2.0 * s_hi ** 2)))
    img = np.clip((img - 0.35) / 0.25, 0.0, 1.0)          # plateau at 1 inside the stroke, soft edge
    img = img.reshape(size, 4, size, 4).mean((1, 3))
    img[img < 0.1] = 0.0
    (y0, x0), (h, w) = template_dimensions(img)
    img = img[y0:y0 + h, x0:x0 + w]
    return img / img.max()


# ---------------------------------------------------------------------------------------------------- the pipeline
def make_sequences(n_seq, T=10, canvas=(50, 50), n_objects=(0, 2), obj_size=28, seed=1234, overlap=0.0, n_glyphs=32):
    """``create_seq_mnist.py``'s main block for ``n_seq`` samples: static frames, trajectories for all objects of all samples
    from ONE trajectory object (its state is [n_objects_total, 6]), rendering, uint8 conversion, ``fix_data``'s coordinate
    array (create_seq_mnist.py:64-86)."""
    rng = np.random.RandomState(seed)
    H, W = canvas
    n_min, n_max = sorted(n_objects)
    glyphs = [make_glyph(rng, obj_size) for _ in range(n_glyphs)]
    nums = rng.randint(n_min, n_max + 1, size=n_seq)
    templates, coords0 = [], []
    i = 0
    while i < n_seq:
        tpls = [glyphs[j] for j in rng.choice(len(glyphs), int(nums[i]), replace=False)] if nums[i] > 0 else []
        pos = place_templates(tpls, (H, W), rng)
        if pos is None:
            continue
        templates.append(tpls)
        coords0.append(pos)
        i += 1
    flat = np.asarray([p for ps in coords0 for p in ps], dtype=np.float64).reshape(-1, 2)
    traj = NoisyAccelerationTrajectory(noise_std=.01, pos_bounds=position_bounds((H, W), (obj_size, obj_size), overlap),
                                       max_speed=10, max_acc=3, bounce=True)
    tjs = traj.create(T, len(flat), rng, init_from=flat) if len(flat) else np.zeros((T, 0, 2), dtype=np.float32)
    imgs = np.zeros((T, n_seq, H, W), dtype=np.float32)
    coords = np.zeros((T, n_seq, n_max, 4), dtype=np.float32)
    k = 0
    for i in range(n_seq):
        for o, tpl in enumerate(templates[i]):
            for t in range(T):
                blend(imgs[t, i], tpl, tjs[t, k])
            coords[:, i, o, :2] = tjs[:, k]
            coords[:, i, o, 2:] = tpl.shape
            k += 1
    nums_exp = np.zeros((T, n_seq, n_max + 1), dtype=np.float32)     # expand_nums (data.py:178-182), tiled over time
    for i, n in enumerate(nums):
        nums_exp[:, i, :n] = 1.0
    if imgs.max() <= 0:
        u8 = np.zeros(imgs.shape, dtype=np.uint8)
    else:
        u8 = convert_img_dtype(imgs, np.uint8)
    return dict(imgs=u8, nums=nums_exp, coords=coords)


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
