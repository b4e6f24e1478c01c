"""Dataset generator against the reference's own code: tests/golden/generator_ref.npz holds
trajectories, blended canvases and uint8 conversions produced by EXECUTING sqair/data/trajectory.py and the NumPy functions
of sqair/data/template.py (tests/golden/make_generator_golden.py, build container only); the restatement in
sqair_amd/data.py has to reproduce them bit for bit.  Plus known answers worked out by hand from the reference's formulas
(bounce reflection and clipping trajectory.py:109-143, max blend and edge clipping template.py:69-104, top-left bounds with
overlap 0 create_seq_mnist.py:43-56,98, placement data.py:98-170).  No GPU."""
import os

import numpy as np
import pytest

from sqair_amd import data as D

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "generator_ref.npz"))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_trajectories_are_bit_identical_to_the_reference(i):
    seed, n, T, noise, speed, acc = G["traj%d_params" % i]
    bounds = G["traj%d_bounds" % i]
    rng = np.random.RandomState(int(seed))
    init = rng.uniform(size=(int(n), 2)) * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]   # the fixture's init draw
    assert np.array_equal(init, G["traj%d_init" % i])
    tr = D.NoisyAccelerationTrajectory(noise_std=noise, pos_bounds=bounds.tolist(), max_speed=speed, max_acc=acc, bounce=True)
    tjs = tr.create(int(T), int(n), rng, init_from=init)
    want = G["traj%d_out" % i]
    assert tjs.dtype == want.dtype == np.float32 and tjs.shape == want.shape
    assert np.array_equal(tjs, want)
    assert (want >= bounds[:, 0] - 1e-6).all() and (want <= bounds[:, 1] + 1e-6).all()
    assert np.array_equal(want[0], init.astype(np.float32))                                # init_from overrides frame 0


def test_one_step_bounce_and_clip_known_answer():
    """Worked by hand from trajectory.py:116-143 + :75-80 (noise_std = 0) and checked against the reference's output."""
    tr = D.NoisyAccelerationTrajectory(noise_std=0.0, pos_bounds=[[0.0, 50.0], [0.0, 50.0]], max_speed=10, max_acc=3, bounce=True)
    state = G["step_state_in"]
    pts, new = tr.forward(state.copy(), np.random.RandomState(0))
    assert np.array_equal(pts, G["step_points"]) and np.array_equal(new, G["step_state_out"])
    # row 0: y 48 + 9 = 57 > 50 -> 2*50 - 57 = 43, vy = -(9 + 2) = -11 -> clipped to -10, ay = -2;
    #        x 1 - 4 = -3 < 0 -> 3, vx = -(-4 - 2.5) = 6.5, ax = 2.5
    assert np.allclose(new[0], [43.0, 3.0, -10.0, 6.5, -2.0, 2.5])
    # row 1: no bounce; vy 9.5 + 2.9 = 12.4 -> clipped to max_speed
    assert np.allclose(new[1], [19.5, 10.0, 10.0, 0.0, 2.9, 0.0])
    # row 2: y 0 - 70 = -70 -> reflected to 70, still outside -> the state clip puts it on the bound; x likewise
    assert np.allclose(new[2, :2], [50.0, 0.0]) and np.allclose(new[2, 2:4], [10.0, -10.0])


def test_position_bounds_of_the_shipped_configuration():
    # create_seq_mnist.py:98 overlap = 0., canvas 50, template 28: the TOP-LEFT corner roams [0, 50] on both axes
    assert D.position_bounds((50, 50), (28, 28), 0.0) == [[0.0, 50.0], [0.0, 50.0]]
    assert D.position_bounds((50, 50), (28, 28), 0.5) == [[-14.0, 36.0], [-14.0, 36.0]]


def test_blending_is_bit_identical_to_the_reference():
    tpls = [G["tpl%d" % i] for i in range(3)]
    pos = G["blend_positions"]
    k = 0
    for tpl in tpls:
        for p in pos:
            c = np.zeros((50, 50), dtype=np.float32)
            D.blend(c, tpl, p)
            assert np.array_equal(c, G["blend_single"][k]), (tpl.shape, p)
            k += 1
    allc = np.zeros((50, 50), dtype=np.float32)
    for tpl in tpls:
        for p in pos[:5]:
            D.blend(allc, tpl, p)
    assert np.array_equal(allc, G["blend_all"])
    # the reference's template-slice arithmetic (recorded input / output pairs of its constrain_dims) selects the same template
    # rows as the interval intersection used here
    for (a, b, dim), (ai, bi) in zip(G["constrain_dims_in"], G["constrain_dims_out"]):
        lo, hi, t0 = D._visible_span(int(a), int(b - a), int(dim))
        assert list(range(t0, t0 + hi - lo)) == list(range(int(ai), int(bi))), (a, b, dim)


def test_blend_known_answers():
    """template.py:69-104 by hand: max (not sum, not overwrite); rounding to the nearest pixel; parts outside are cut."""
    c = np.zeros((6, 6), dtype=np.float32)
    t = np.array([[1.0, 2.0], [3.0, 4.0]])
    D.blend(c, t, np.array([1.4, 2.6]))                       # -> row 1, column 3
    assert c[1, 3] == 1 and c[1, 4] == 2 and c[2, 3] == 3 and c[2, 4] == 4 and c.sum() == 10
    D.blend(c, np.full((2, 2), 2.5), np.array([1.0, 3.0]))    # overlap: element-wise maximum
    assert np.array_equal(c[1:3, 3:5], [[2.5, 2.5], [3.0, 4.0]])
    e = np.zeros((6, 6), dtype=np.float32)
    D.blend(e, t, np.array([-1.0, 5.0]))                      # only the bottom-left element lands on the canvas
    assert e[0, 5] == 3 and e.sum() == 3
    D.blend(e, t, np.array([6.0, 0.0]))                       # top-left at the bound canvas = 6: fully outside
    D.blend(e, t, np.array([-2.0, -2.0]))
    assert e.sum() == 3
    assert D._visible_span(-5, 28, 50) == (0, 23, 5) and D._visible_span(40, 28, 50) == (40, 50, 0)
    lo, hi, _ = D._visible_span(55, 28, 50)
    assert hi == lo


def test_uint8_conversion_matches_the_reference():
    assert np.array_equal(D.convert_img_dtype(G["u8_in"].copy(), np.uint8), G["u8_out"])
    assert np.array_equal(D.convert_img_dtype(G["u8_shifted_in"].copy(), np.uint8), G["u8_shifted_out"])
    # template.py:40: divides by max / 255, truncating cast: [0, .5, 1] -> [0, 127, 255 or 254]
    got = D.convert_img_dtype(np.array([0.0, 0.5, 1.0], dtype=np.float32), np.uint8)
    assert got[0] == 0 and got[1] == 127 and got[2] in (254, 255)


def test_placement_known_answers():
    """data.py:98-115 with fraction_outside_canvas = 0: corner = round(rand(2) * (canvas - size)) — always fully inside;
    data.py:145-153: at most n_tries re-draws per SAMPLE, then the sample is abandoned."""
    rng = np.random.RandomState(0)
    for _ in range(200):
        p = D.make_coord((20, 17), (50, 50), rng)
        assert (p >= 0).all() and p[0] <= 30 and p[1] <= 33
    u = np.random.RandomState(7).rand(2)
    assert np.array_equal(D.make_coord((28, 28), (50, 50), np.random.RandomState(7)), np.round(u * 22).astype(np.int32))
    big = [np.ones((40, 40)), np.ones((40, 40))]              # two 40x40 boxes cannot avoid each other on 50x50
    assert D.place_templates(big, (50, 50), np.random.RandomState(1)) is None
    pos = D.place_templates([np.ones((10, 10)), np.ones((10, 10))], (50, 50), np.random.RandomState(1))
    (y0, x0), (y1, x1) = pos
    assert y0 + 10 <= y1 or y1 + 10 <= y0 or x0 + 10 <= x1 or x1 + 10 <= x0
    assert D.template_dimensions(np.pad(np.ones((3, 5)), ((2, 4), (6, 1)))) == ((2, 6), (3, 5))


def test_pipeline_layout_and_statistics():
    d = D.make_sequences(24, T=10, canvas=(50, 50), n_objects=(0, 2), seed=3)
    assert d["imgs"].dtype == np.uint8 and d["imgs"].shape == (10, 24, 50, 50)
    assert d["nums"].shape == (10, 24, 3) and d["coords"].shape == (10, 24, 2, 4)
    n = d["nums"][0].sum(-1)
    assert set(np.unique(n)) <= {0.0, 1.0, 2.0} and np.array_equal(d["nums"][0], d["nums"][9])
    assert np.array_equal(d["nums"][0], (np.arange(3)[None] < n[:, None]).astype(np.float32))      # prefix ones
    # first frame: every object fully inside, boxes (y, x, h, w); later frames: top-left corner within [0, 50]
    for i in range(24):
        for o in range(int(n[i])):
            y, x, h, w = d["coords"][0, i, o]
            assert y >= 0 and x >= 0 and y + h <= 50 and x + w <= 50 and h <= 28 and w <= 28
        for o in range(int(n[i]), 2):
            assert not d["coords"][:, i, o].any()
    assert d["coords"][..., :2].min() >= 0 and d["coords"][..., :2].max() <= 50
    empty = n == 0
    assert not d["imgs"][:, empty].any() and d["imgs"][0][~empty].any()
    x = D.to_float(d["imgs"])
    assert x.dtype == np.float32 and 0.9 < x.max() <= 1.0
    assert np.array_equal(D.make_sequences(24, T=10, seed=3)["imgs"], d["imgs"])                   # deterministic
