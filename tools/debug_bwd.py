import numpy as np, torch, sys
sys.path.insert(0, ".")
from sqair_amd.flags import make_flags
from sqair_amd.data import make_sequences, to_float
from sqair_amd.model import Model, SqairCore
from tests.hip_util import draw_noise, params32
K, N, T, B, hw = 3, 3, 3, 4, (50, 50)
F = make_flags(k_particles=K, n_steps_per_image=N)
obs = to_float(make_sequences(B, T=T, canvas=hw, seed=3)["imgs"])
core = SqairCore(F, hw)
core.set_params(params32(F, hw, 4, 0.05, obs.mean((0, 1))))
Model(obs, None, core, K, outputs="minimal")
core.noise.copy_(torch.as_tensor(draw_noise(np.random.default_rng(0), T, B * K, N, 55)).reshape(core.noise.shape))
def diff(a, b, tag):
    d = (a - b).abs()
    print(tag, "max diff", float(d.max()), "scale", float(a.abs().max()))
    bad = []
    for name, (o, shape) in core.offsets.items():
        n = int(np.prod(shape)) if len(shape) else 1
        e = float(d[o:o+n].max())
        if e > 1e-4 * float(a.abs().max()):
            bad.append((name, e))
    print("   ", len(bad), "params differ; first:", bad[:6])
g0 = core.grad_step(use_graph=False).clone(); torch.cuda.synchronize(); print("scalars", core.scalars[:4].tolist())
g1 = core.grad_step(use_graph=False).clone(); torch.cuda.synchronize(); print("scalars", core.scalars[:4].tolist())
diff(g0, g1, "eager vs eager")
g2 = core.grad_step(use_graph=True).clone(); torch.cuda.synchronize(); print("scalars", core.scalars[:4].tolist(), core.train_graph_nodes)
diff(g0, g2, "eager vs graph#1")
g3 = core.grad_step(use_graph=True).clone(); torch.cuda.synchronize()
diff(g0, g3, "eager vs graph#2")
for i in range(4):
    g3 = core.grad_step(use_graph=True).clone(); torch.cuda.synchronize()
    diff(g0, g3, "eager vs graph#%d" % (3 + i))
