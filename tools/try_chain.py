"""Layer chains (sqair_enable_chains) against the launch-per-layer path at BASELINE configs[1]: bit-identity of all outputs,
status word, time per pass and per gradient evaluation, device-clock phase stamps of the chain launches.
    python tools/try_chain.py
"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from sqair_amd.data import config_inputs
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params
ov, obs, nums, _ = config_inputs(2)
F = make_flags(**ov); hw=(50,50)
P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
res = {}
for chains in (False, True):
    core = SqairCore(F, hw, chains=chains)
    torch.cuda.set_stream(core.stream)
    core.set_params(P)
    m = Model(obs, None, core, 5, presence=nums, outputs="all")
    core.draw_noise(seed=1, step=0, global_batch=32, b0=0)
    core.forward(use_graph=False)
    torch.cuda.synchronize()
    st = core.chain_status()
    out = {k: v.clone() for k, v in core.out.items()}
    core.forward(use_graph=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): core.forward(use_graph=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    print("chains", chains, "status", st, "nodes", core.graph_nodes(), "ms/pass %.3f" % ms, "elbo", float(core.scalars[1]))
    res[chains] = out
    # training step
    g = core.grad_step(use_graph=True).clone(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): core.grad_step(use_graph=True)
    torch.cuda.synchronize()
    print("   grad_step ms %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), "train status", core.chain_status(train=True), "nodes", core.train_graph_nodes)
    res[("g", chains)] = g
bad = [k for k in res[False] if not torch.equal(res[False][k], res[True][k])]
print("outputs not bit-identical:", bad)
print("grad max abs diff", float((res[("g", False)] - res[("g", True)]).abs().max()), "of", float(res[("g", False)].abs().max()))

# per-launch device-clock durations (100 MHz ticks) of the dense-layer launches, chains on
import os, csv, collections
os.environ["SQAIR_PROF_DUMP"] = "/tmp/chain_prof.csv"
core.profile_linear()
rows = list(csv.reader(open("/tmp/chain_prof.csv")))
dur = collections.defaultdict(list)
for r in rows[1:] if not rows[0][0].lstrip("-").isdigit() else rows:
    lid, m, t0, t1 = int(r[0]), int(r[1]), int(r[2]), int(r[3])
    dur[(lid, m)].append((t1 - t0) * 0.01)
    if lid >= 1000 and len(r) >= 7:
        dur[(lid + 0.1, m)].append(int(r[4]) * 0.01); dur[(lid + 0.2, m)].append(int(r[5]) * 0.01); dur[(lid + 0.3, m)].append(int(r[6]) * 0.01)
for k in sorted(dur):
    v = dur[k]
    print("layer %7.1f M %5d  n %4d  mean %.2f us  min %.2f" % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
