"""End-to-end training run on synthetic moving glyphs with the reference's driver semantics (scripts/experiment.py:126-185):
minibatches drawn with replacement (data.py:203-216), RMSProp(momentum .9) with the piecewise-constant learning rate,
VIMCO target, K = 5 particles; logs the training ELBO per frame and the validation ELBO on held-out sequences.
    python tools/train_demo.py [steps] [lr] [train_itr] [seq_len] [stage_itr] [flag=value ...] > profiles/rNN_train_curve.json
(trailing flag=value pairs override any of the reference's flags, e.g. disc_step_bias=5 n_steps_per_image=3 opt=adam)
The reference's own recipe (scripts/train_multi_mnist.sh) is seq_len 3, stage_itr 100000 (sequence-length curriculum), 1 M iterations.
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from sqair_amd.data import make_sequences, to_float
from sqair_amd.dataio import MinibatchFeed
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params
from sqair_amd.train import Trainer

over = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
n_train = int(over.pop("n_train", 2048))   # n_train=16384: a training set the model cannot memorise
hw_over = over.pop("hw", None)              # hw=128x128: BASELINE configs[4]'s frames (the row-wave canvas kernels)
slot_chain = int(over.pop("slot_chain", 0))  # slot_chain=1: the library's in-launch slot chain (sqair_set_option) for every pass
batch_over = int(over.pop("batch", 0))       # batch=128: sequences per step (the large-row dense kernels; default 32)
sys.argv = [a for a in sys.argv if "=" not in a]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
T, B, K, N, hw = 10, batch_over or 32, 5, 4, (50, 50)
if hw_over:
    hw = tuple(int(v) for v in hw_over.split("x"))
train_itr = int(sys.argv[3]) if len(sys.argv) > 3 else steps   # the piecewise-constant schedule is relative to train_itr
seq_len = int(sys.argv[4]) if len(sys.argv) > 4 else 0
stage_itr = int(sys.argv[5]) if len(sys.argv) > 5 else 0
N = int(over.get("n_steps_per_image", N))
F = make_flags(**dict(dict(k_particles=K, n_steps_per_image=N, learning_rate=lr, train_itr=train_itr, seq_len=seq_len, stage_itr=stage_itr), **over))
train = make_sequences(n_train, T=T, canvas=hw, n_objects=(0, 2), seed=1)
valid = make_sequences(256, T=T, canvas=hw, n_objects=(0, 2), seed=2)
feed = MinibatchFeed(dict(imgs=to_float(train["imgs"]), nums=train["nums"], coords=train["coords"]), B, shuffle=True, seed=0,
                     seq_len=seq_len, stage_itr=stage_itr)
vfeed = MinibatchFeed(dict(imgs=to_float(valid["imgs"]), nums=valid["nums"], coords=valid["coords"]), B, shuffle=False)
mean_img = to_float(train["imgs"]).mean((0, 1))
core = SqairCore(F, hw, options={"slot_chain": 1} if slot_chain else None)
core.set_params({k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=mean_img).items()})
model = Model(to_float(train["imgs"][:(seq_len if seq_len and stage_itr else T), :B]), None, core, K, outputs="minimal")
trainer = Trainer(model, F)
gen = torch.Generator(device="cuda").manual_seed(0)


def validate():
    """validation ELBO per frame and the importance-weighted mean number of inferred objects per frame (model.py:107-110)"""
    tot, n, steps_tot = 0.0, 0, 0.0
    i_ns = list(core.mean_names).index("num_steps_per_sample")
    Tc = core.T   # the curriculum's current sequence length (validation sequences are truncated to it)
    with core.on_stream():
        for _ in range(256 // B):
            core.obs.copy_(torch.as_tensor(vfeed.next()["imgs"][:Tc]))
            core.draw_noise(gen)
            core.forward(use_graph=True)
            tot += float(core.scalars[1])
            steps_tot += float(core.iw_means[i_ns])
            n += 1
    return tot / n / Tc, steps_tot / n


log = []
t0 = time.perf_counter()
run = 0.0
for it in range(steps + 1):
    if it % max(1, steps // 30) == 0:
        v_elbo, v_steps = validate()
        rec = dict(step=it, valid_elbo_iwae_per_frame=v_elbo, valid_num_steps_per_frame=v_steps,
                   train_elbo_iwae_per_frame_running=run, seq_len=core.T, seconds=time.perf_counter() - t0)
        log.append(rec)
        print(rec, file=sys.stderr)
    if it == steps:
        break
    batch = feed.next(it)
    trainer.step(obs=batch["imgs"], generator=gen)
    if it % 10 == 0:
        with core.on_stream():
            e = float(core.scalars[1]) / core.T
        run = e if it == 0 else 0.9 * run + 0.1 * e
if slot_chain:   # every chain launch of the last training and validation passes completed
    core.check_chain(train=True)
    core.check_chain(train=False)
print(json.dumps(dict(config=dict(slot_chain=slot_chain, T=T, B=B, K=K, N=N, hw=hw, steps=steps, train_itr=train_itr, learning_rate=lr, opt=str(F.opt), flag_overrides=over,
                                  schedule=F.schedule, data="%d synthetic 2-glyph sequences, 256 held out" % n_train,
                                  true_objects_per_frame=float(valid["nums"].sum(-1).mean())),
                      upper_bound_per_frame=hw[0] * hw[1] * (-np.log(0.3) - 0.5 * np.log(2 * np.pi)), curve=log)))
