import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import sqair_oracle as O
from sqair_amd.flags import make_flags
from sqair_amd.data import make_sequences, to_float
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import flatten_params, unflatten_params
from sqair_amd.train import Trainer, learning_rate, rmsprop_reference
from tests.hip_util import draw_noise, params32
K, N, T, B, hw = 3, 3, 3, 3, (50, 50)
F = make_flags(k_particles=K, n_steps_per_image=N, learning_rate=1e-3, train_itr=100)
obs = to_float(make_sequences(B, T=T, canvas=hw, n_objects=(1, 2), obj_size=24, seed=11)["imgs"])
P = params32(F, hw, 4, 0.05, obs.mean((0, 1)))
core = SqairCore(F, hw)
core.set_params(P)
m = Model(obs, None, core, K, outputs=["log_weights_per_timestep", "discrete_log_prob", "prop_pres", "disc_pres"])
tr = Trainer(m, F, use_graph=(len(sys.argv) < 2))
theta = flatten_params(P, core.spec).astype(np.float64)
ms, mom = np.ones_like(theta), np.zeros_like(theta)
for it in range(3):
    noise = draw_noise(np.random.default_rng(200 + it), T, B * K, N, 55)
    theta = core.flat.cpu().numpy().astype(np.float64)
    orc = O.SqairOracle(unflatten_params(theta.astype(np.float32), core.spec), O.make_cfg(F, hw), torch.float64, requires_grad=True)
    ref = orc.model(obs, noise)
    tgt = orc.make_target(ref); tgt.backward()
    before = core.flat.cpu().numpy().astype(np.float64)
    print("param diff before step", it, np.abs(before - theta).max())
    g_hip = tr.step(noise=noise).cpu().numpy().astype(np.float64)
    torch.cuda.synchronize()
    same = np.array_equal(core.out["prop_pres"].cpu().numpy(), ref.prop_pres.detach().numpy()) and np.array_equal(core.out["disc_pres"].cpu().numpy(), ref.disc_pres.detach().numpy())
    g = flatten_params({k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in orc.P.items()}, core.spec).astype(np.float64)
    print("step", it, "presence same", same, "target", float(tgt), float(core.scalars[2]), "grad err", np.abs(g - g_hip).max(), "scale", np.abs(g).max())
    new, ms, mom = rmsprop_reference(theta, g, ms, mom, learning_rate(F, it))
    got = core.flat.cpu().numpy().astype(np.float64)
    print("   delta err", np.abs((got - before) - (new - theta)).max(), "delta scale", np.abs(new - theta).max())
    theta = new
print("---- pack consistency")
a = core.packed.clone(); torch.cuda.synchronize()
core.pack(); torch.cuda.synchronize()
print("packed changed by an extra pack():", float((core.packed - a).abs().max()))
# eager forward at current params vs graph replay
noise = draw_noise(np.random.default_rng(300), T, B * K, N, 55)
core.noise.copy_(torch.as_tensor(noise).reshape(core.noise.shape))
ge = core.grad_step(use_graph=False).clone(); torch.cuda.synchronize(); se = core.scalars[:3].tolist()
gg = core.grad_step(use_graph=True).clone(); torch.cuda.synchronize(); sg = core.scalars[:3].tolist()
print("eager", se, "graph", sg, "grad diff", float((ge - gg).abs().max()))
orc = O.SqairOracle(unflatten_params(core.flat.cpu().numpy(), core.spec), O.make_cfg(F, hw), torch.float64, requires_grad=True)
ref = orc.model(obs, noise); tgt = orc.make_target(ref); tgt.backward()
g = flatten_params({k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in orc.P.items()}, core.spec)
print("oracle target", float(tgt), "grad err eager", np.abs(g - ge.cpu().numpy()).max(), "scale", np.abs(g).max())
print("presence same", np.array_equal(core.out["prop_pres"].cpu().numpy(), ref.prop_pres.detach().numpy()), np.array_equal(core.out["disc_pres"].cpu().numpy(), ref.disc_pres.detach().numpy()))
