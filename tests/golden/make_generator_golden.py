"""Golden vectors for the dataset generator produced by RUNNING the reference's own code.

Build container only (reads /root/reference; only the .npz travels):

    python tests/golden/make_generator_golden.py

* sqair/data/trajectory.py has no TF dependency: its classes are executed as they are (text up to the `__main__` block,
  with `xrange` bound to `range` — the only Python-2 name it uses), driven by numpy's legacy global RandomState exactly
  like `create_seq_mnist.py` drives them.  Recorded: seeds / bounds / init positions in, trajectories out (bounce
  reflection, velocity / acceleration clipping, first-position override).
* sqair/data/template.py imports TensorFlow and scipy.misc at module level for code that is not on this path; the pure
  NumPy functions `constrain_dims`, `convert_img_dtype` and `TemplateDataset._blend / _blend_slice` are executed from
  their own source text (ast-extracted, nothing rewritten).  Recorded: templates + positions in (inside, straddling every
  edge, fully outside), canvases and the uint8 conversion out.
The fixture holds arrays only — no reference source text.
"""
import ast
import os

import numpy as np

REF = os.environ.get("SQAIR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_trajectory_module():
    src = open(os.path.join(REF, "sqair", "data", "trajectory.py")).read()
    src = src[:src.index("if __name__ == '__main__':")]
    ns = {"xrange": range, "__name__": "ref_trajectory"}
    exec(compile(src, "trajectory.py", "exec"), ns)
    return ns


def load_template_functions():
    src = open(os.path.join(REF, "sqair", "data", "template.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in ("constrain_dims", "convert_img_dtype"))
            or (isinstance(n, ast.ClassDef) and n.name == "TemplateDataset")]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"np": np, "xrange": range, "__name__": "ref_template"}
    exec(compile(mod, "template.py", "exec"), ns)
    return ns


def main():
    out = {}
    T = load_trajectory_module()
    # (1) the reference's own configuration (create_seq_mnist.py:43-56, :98: canvas 50, template 28, overlap 0)
    cases = [
        dict(seed=1, n=64, T=10, bounds=[[0.0, 50.0], [0.0, 50.0]], noise=0.01, speed=10.0, acc=3.0),
        dict(seed=2, n=16, T=30, bounds=[[-14.0, 36.0], [-14.0, 36.0]], noise=0.01, speed=10.0, acc=3.0),   # overlap .5
        dict(seed=3, n=8, T=12, bounds=[[0.0, 20.0], [0.0, 60.0]], noise=1.0, speed=5.0, acc=5.0),          # its __main__ demo
    ]
    for i, c in enumerate(cases):
        np.random.seed(c["seed"])
        init = np.random.uniform(size=(c["n"], 2)) * (np.asarray(c["bounds"])[:, 1] - np.asarray(c["bounds"])[:, 0]) \
            + np.asarray(c["bounds"])[:, 0]
        tr = T["NoisyAccelerationTrajectory"](noise_std=c["noise"], n_dim=2, pos_bounds=c["bounds"], max_speed=c["speed"],
                                              max_acc=c["acc"], bounce=True)
        tjs = tr.create(c["T"], c["n"], init_from=init)
        out["traj%d_params" % i] = np.asarray([c["seed"], c["n"], c["T"], c["noise"], c["speed"], c["acc"]], dtype=np.float64)
        out["traj%d_bounds" % i] = np.asarray(c["bounds"], dtype=np.float64)
        out["traj%d_init" % i] = init
        out["traj%d_out" % i] = np.asarray(tjs)
    # (2) one hand-made forward step through the bounce / clip logic
    tr = T["NoisyAccelerationTrajectory"](noise_std=0.0, n_dim=2, pos_bounds=[[0.0, 50.0], [0.0, 50.0]], max_speed=10, max_acc=3,
                                          bounce=True)
    state = np.asarray([[48.0, 1.0, 9.0, -4.0, 2.0, -2.5],      # crosses the upper y bound and the lower x bound
                        [10.0, 10.0, 9.5, 0.0, 2.9, 0.0],       # velocity clipped at max_speed
                        [0.0, 50.0, -70.0, 70.0, 0.0, 0.0]])    # reflection lands outside again -> position clipped
    np.random.seed(0)
    pts, new_state = tr.forward(state.copy())
    out["step_state_in"], out["step_points"], out["step_state_out"] = state, pts, new_state
    # (3) template blending
    F = load_template_functions()
    rng = np.random.RandomState(5)
    td = F["TemplateDataset"]((50, 50), 1)
    templates = [rng.uniform(0.1, 1.0, size=s) for s in ((20, 17), (28, 28), (5, 31))]
    positions = np.asarray([[10.2, 12.7], [-6.4, 3.0], [3.0, -9.5], [40.0, 41.6], [47.5, -3.2], [-30.0, 5.0], [5.0, 55.0],
                            [49.6, 49.6], [0.0, 0.0], [22.0, 33.0]])
    canv = np.zeros((len(positions) * len(templates), 50, 50), dtype=np.float32)
    k = 0
    for tpl in templates:
        for pos in positions:
            td._blend(canv[k], tpl, pos)
            k += 1
    allc = np.zeros((50, 50), dtype=np.float32)      # all of them max-blended into one canvas
    for tpl in templates:
        for pos in positions[:5]:
            td._blend(allc, tpl, pos)
    for i, tpl in enumerate(templates):
        out["tpl%d" % i] = tpl
    out["blend_positions"], out["blend_single"], out["blend_all"] = positions, canv, allc
    out["constrain_dims_in"] = np.asarray([[0, 28, 50], [-5, 23, 50], [40, 68, 50], [-30, -2, 50], [55, 83, 50], [49, 50, 50]])
    out["constrain_dims_out"] = np.asarray([F["constrain_dims"](*r) for r in out["constrain_dims_in"]])
    stack = np.stack([allc, canv[0] * 0.5, canv[3]])[None].astype(np.float32)
    out["u8_in"], out["u8_out"] = stack, F["convert_img_dtype"](stack.copy(), np.uint8)
    shifted = stack + 0.25                                   # min > 0: shows that it divides by max, not by (max - min)
    out["u8_shifted_in"], out["u8_shifted_out"] = shifted, F["convert_img_dtype"](shifted.copy(), np.uint8)
    path = os.path.join(HERE, "generator_ref.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
