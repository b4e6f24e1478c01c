import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sqair_amd.data import make_sequences, to_float
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from tests.hip_util import draw_noise, params32
B, K, N, T, hw = 8, 2, 3, 2, (50, 50)
F = make_flags(k_particles=K, n_steps_per_image=N)
d = make_sequences(B, T=T, canvas=hw, seed=13)
obs = to_float(d["imgs"])
P = params32(F, hw, 3, 0.05, obs.mean((0, 1)))
noise = draw_noise(np.random.default_rng(7), T, B * K, N, 55)
res = []
for fusion in (0, 1):
    core = SqairCore(F, hw, options={"what_fusion": fusion})
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"])
    m.run(noise=noise)
    torch.cuda.synchronize()
    # raw records of frame 0 from the workspace are not exposed; compare outputs + presence masks
    res.append({k: v.detach().cpu().numpy().copy() for k, v in core.out.items()})
a, b = res
for k in ("what", "what_loc", "what_scale", "where", "presence", "obj_id"):
    dd = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
    print(k, "max diff", dd.max(), "n diff", int((dd > 0).sum()), "of", dd.size)
w = np.abs(a["what"] - b["what"]) > 0
print("what differs where what_loc equal:", int((w & (a["what_loc"] == b["what_loc"]) & (a["what_scale"] == b["what_scale"])).sum()), "of", int(w.sum()))
# frame 0 only (before any feedback): which slots
print("frame 0 diffs per slot:", (np.abs(a["what"][0] - b["what"][0]) > 0).sum((0, 2)), "ids", np.unique(a["obj_id"][0]))
print("---- T=1, N=1")
F = make_flags(k_particles=1, n_steps_per_image=1)
d = make_sequences(16, T=1, canvas=hw, seed=13)
obs = to_float(d["imgs"])
P = params32(F, hw, 3, 0.05, obs.mean((0, 1)))
noise = draw_noise(np.random.default_rng(7), 1, 16, 1, 55)
res = []
for fusion in (0, 1):
    core = SqairCore(F, hw, options={"what_fusion": fusion})
    core.set_params(P)
    m = Model(obs, None, core, 1, presence=d["nums"])
    m.run(noise=noise)
    torch.cuda.synchronize()
    res.append({k: v.detach().cpu().numpy().copy() for k, v in core.out.items()})
a, b = res
print("presence", a["presence"].reshape(-1))
for k in ("what_loc", "what_scale", "what"):
    dd = np.abs(a[k] - b[k])[0, :, 0]
    print(k, "rows with diffs", np.flatnonzero(dd.max(1) > 0), "cols", np.flatnonzero(dd.max(0) > 0), "max", dd.max())
    r = int(np.argmax(dd.max(1)))
    print("  row", r, "a", a[k][0, r, 0, :6], "b", b[k][0, r, 0, :6])
for k in sorted(a):
    dd = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
    if np.nanmax(dd) > 0:
        print("differs:", k, float(np.nanmax(dd)))
print("equal:", [k for k in sorted(a) if not np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))) > 0][:40])
