"""Bounded experiment on the "zero objects" collapse of round 1's training runs.

Trains with the reference's recipe (scripts/train_multi_mnist.sh: seq_len 3, K = 5, batch 32, RMSProp(momentum .9), lr 1e-5;
n_steps_per_image 3 as shipped) on the restated generator's MNIST-like glyphs, mean image as scripts/experiment.py:109-118
computes it, and logs every `every` iterations: inferred objects per frame, ELBO / data_ll / KL per frame, and the gradient of the
two step-predictor output biases (disc.steps.l1.b, prop.steps.l1.b: d target / d logit summed over rows) split into its two
sources -- the VIMCO score-function term (signal x d log q(presence)) and the pathwise IWAE term (importance weight x d log w) --
by re-running the backward pass on the same tape with one of the two seeds zeroed.
    python tools/collapse_probe.py [steps] [every] [variant] > gpurun_out/r02_collapse_<variant>.json
variants: ref (as above) | noscore (score-function seed zeroed in the UPDATE: shows what the pathwise terms alone do) | any
comma-separated flag overrides of the reference's own flags, e.g. "opt=adam,learning_rate=1e-4,disc_step_bias=5"
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from sqair_amd import _capi
from sqair_amd.data import make_sequences, to_float
from sqair_amd.dataio import MinibatchFeed
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params
from sqair_amd.train import Optimizer, learning_rate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
variant = sys.argv[3] if len(sys.argv) > 3 else "ref"
T, B, K, N, hw = 3, 32, 5, 3, (50, 50)
over = dict(kv.split("=") for kv in variant.split(",") if "=" in kv)
F = make_flags(**dict(dict(k_particles=K, n_steps_per_image=N, learning_rate=1e-5, train_itr=2000000), **over))
lr = float(F.learning_rate)
train = make_sequences(4096, T=10, canvas=hw, n_objects=(0, 2), seed=1)
feed = MinibatchFeed(dict(imgs=to_float(train["imgs"]), nums=train["nums"], coords=train["coords"]), B, shuffle=True, seed=0, seq_len=T,
                     stage_itr=10 ** 9)
mean_img = to_float(train["imgs"]).mean((0, 1))
core = SqairCore(F, hw)
core.set_params({k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=mean_img).items()})
model = Model(to_float(train["imgs"][:T, :B]), None, core, K, presence=train["nums"][:T, :B], outputs="minimal")
opt = Optimizer(core, F.opt)
gen = torch.Generator(device="cuda").manual_seed(0)
names = list(core.mean_names)
off = {n: core.offsets[n][0] for n in ("disc.steps.l1.b", "prop.steps.l1.b", "dec.l2.b", "dec.output_scale")}


def grads_split():
    """(total, score-only, pathwise-only) flat gradients on the CURRENT tape."""
    iw, sig = core.importance_weights.clone(), core.vimco_signal.clone()
    g_tot = core.backward().clone()
    core.importance_weights.zero_()
    g_score = core.backward().clone()
    core.importance_weights.copy_(iw)
    core.vimco_signal.zero_()
    g_path = core.backward().clone()
    core.vimco_signal.copy_(sig)
    return g_tot, g_score, g_path


log, t0 = [], time.perf_counter()
with core.on_stream():
    for it in range(steps + 1):
        batch = feed.next(it)
        core.obs.copy_(torch.as_tensor(batch["imgs"][:T]))
        core.draw_noise(gen)
        core.forward(train=True)
        if it % every == 0:
            g_tot, g_score, g_path = grads_split()
            rec = dict(step=it, elbo_iwae_per_frame=float(core.scalars[1]) / T, elbo_vae_per_frame=float(core.scalars[0]) / T,
                       **{n.replace("_per_sample", ""): float(core.iw_means[i]) for i, n in enumerate(names)},
                       true_objects_per_frame=float(batch["nums"][0].sum(-1).mean()))
            for n, o in off.items():
                rec["g[%s]" % n] = dict(total=float(g_tot[o]), score=float(g_score[o]), pathwise=float(g_path[o]))
            rec["|g_score|/|g_path|"] = float(g_score.norm() / g_path.norm().clamp_min(1e-30))
            rec["disc.steps.l1.b"] = float(core.flat[off["disc.steps.l1.b"]])
            rec["prop.steps.l1.b"] = float(core.flat[off["prop.steps.l1.b"]])
            rec["seconds"] = time.perf_counter() - t0
            log.append(rec)
            print(json.dumps(rec), file=sys.stderr)
            g = g_path if variant == "noscore" else g_tot
        else:
            if variant == "noscore":
                core.vimco_signal.zero_()
            g = core.backward()
        if it < steps:
            opt.apply_gradients(g, learning_rate(F, it))
core.stream.synchronize()
print(json.dumps(dict(variant=variant, config=dict(T=T, B=B, K=K, N=N, lr=lr, steps=steps, build_id=_capi.build_id()), curve=log)))
