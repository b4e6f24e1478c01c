import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from sqair_amd.data import config_inputs, make_sequences, to_float
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params


def params32(F, hw, seed, jitter, mean_img=None):
    return {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=seed, mean_img=mean_img, jitter=jitter).items()}
small = len(sys.argv) > 1 and sys.argv[1] == "small"
if small:
    K, N, T, B, hw = 3, 3, 3, 8, (50, 50)
    F = make_flags(k_particles=K, n_steps_per_image=N)
    obs = to_float(make_sequences(B, T=T, canvas=hw, seed=3)["imgs"])
else:
    ov, obs, _, _ = config_inputs(2)
    F = make_flags(**ov)
    hw = obs.shape[2:4]
core = SqairCore(F, hw)
core.set_params(params32(F, hw, 1, 0.02, obs.mean((0, 1))))
Model(obs, None, core, int(F.k_particles), outputs="all")
with core.on_stream():
    core.draw_noise(torch.Generator(device="cuda").manual_seed(0))
    core.forward(use_graph=False)
    core.stream.synchronize()
    ref = {k: v.clone() for k, v in core.out.items()}
    lw = core.log_weights.clone()
    for v in core.out.values():
        v.zero_()
    core.forward(persistent=True)
    core.stream.synchronize()
    print("status", core.persistent_status())
    worst = 0.0
    for k, v in core.out.items():
        e = float((v - ref[k]).abs().max()); sc = float(ref[k].abs().max())
        rel = e / max(sc, 1e-6)
        worst = max(worst, rel)
        if rel > 1e-5:
            print("  DIFF", k, e, sc)
    print("worst rel diff", worst, "presence equal", bool(torch.equal(core.out["presence"], ref["presence"])), "logw diff", float((core.log_weights - lw).abs().max()))
    for name, fn in (("persistent", lambda: core.forward(persistent=True)), ("graph", lambda: core.forward(use_graph=True))):
        for _ in range(3):
            fn()
        core.stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        core.stream.synchronize()
        print(name, "%.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    print("status", core.persistent_status())
