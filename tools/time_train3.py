import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from sqair_amd.data import config_inputs
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params


def params32(F, hw, seed, jitter, mean_img=None):
    return {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=seed, mean_img=mean_img, jitter=jitter).items()}
ov, obs, _, _ = config_inputs(2)
F = make_flags(**ov)
hw = obs.shape[2:4]
core = SqairCore(F, hw)
core.set_params(params32(F, hw, 1, 0.02, obs.mean((0, 1))))
Model(obs, None, core, int(F.k_particles), outputs="minimal")
core.draw_noise(torch.Generator(device="cuda").manual_seed(0))
torch.cuda.synchronize()
with torch.cuda.stream(core.stream):   # everything on the core's own stream: the null stream stays idle
    for name, fn in (("grad_step", lambda: core.grad_step(use_graph=True)), ("forward", lambda: core.forward(use_graph=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print("same-stream %s: min %.3f ms" % (name, min(ts)))
