"""Trains with the reference recipe (tools/train_demo.py's setting) and checks the gradient after EVERY step: at the first
non-finite value prints which parameters are affected and the state of the forward pass that produced it.
    python tests/nan_probe.py [steps] [flag=value ...]"""
import sys

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
args = [a for a in sys.argv[1:] if "=" not in a]
steps = int(args[0]) if args else 8000
T, B, K, hw = int(over.get("T", 10)), 32, 5, (50, 50)
N = int(over.get("n_steps_per_image", 3))
over_f = {k: v for k, v in over.items() if k not in ("seed", "T", "spike")}
F = make_flags(**dict(dict(k_particles=K, n_steps_per_image=N, learning_rate=1e-5, train_itr=2000000, disc_step_bias=5), **over_f))
train = make_sequences(2048, T=T, canvas=hw, n_objects=(0, 2), seed=1)
feed = MinibatchFeed(dict(imgs=to_float(train["imgs"]), nums=train["nums"], coords=train["coords"]), B, shuffle=True, seed=0)
mean_img = to_float(train["imgs"]).mean((0, 1))
core = SqairCore(F, hw)
core.set_params({k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=mean_img).items()})
names = ["log_weights_per_timestep", "discrete_log_prob", "presence", "what_scale", "where_scale", "what", "where", "what_loc", "where_loc",
         "presence_logit", "presence_prob", "data_ll_per_sample", "log_q_z_given_x_per_sample", "log_p_z_per_sample"]
model = Model(to_float(train["imgs"][:T, :B]), None, core, K, outputs=names)
trainer = Trainer(model, F)
gen = torch.Generator(device="cuda").manual_seed(int(over.get("seed", 0)))
spec = core.spec
hist = []
ch_off = 0
for entry in spec:
    if entry[0] == "prop.cholesky_scale":
        break
    ch_off += int(np.prod(entry[1]))
spike = float(over.get("spike", 0))   # spike=1e4: at the first finite gradient above it, recompute that gradient with the fp64 oracle
for it in range(steps):
    batch = feed.next(it)
    if spike:
        with core.on_stream():
            flat_before = core.flat.clone()
    g = trainer.step(obs=batch["imgs"][:T], generator=gen)
    with core.on_stream():
        gmax = float(g.abs().max())
        fin = bool(torch.isfinite(g).all()) and bool(torch.isfinite(core.flat).all())
        elbo = float(core.scalars[1]) / core.T
    hist.append((it, gmax, elbo))
    if spike and fin and gmax > spike:
        from oracle import sqair_oracle as O
        from sqair_amd.params import unflatten_params
        print("step %d: max|grad| %.3e -- recomputing with the fp64 oracle (same parameters, frames, noise)" % (it, gmax), flush=True)
        with core.on_stream():
            P0 = unflatten_params(flat_before.cpu().numpy(), spec)
            noise = core.noise.cpu().numpy().reshape(T, B * K, 2, N, -1)
            obs = core.obs.cpu().numpy().reshape(T, B, hw[0], hw[1])
            got = g.cpu().numpy().copy()
        orc = O.SqairOracle({k: np.asarray(v, dtype=np.float64) for k, v in P0.items()}, O.make_cfg(F, hw), torch.float64, requires_grad=True)
        ref = orc.model(obs, noise)
        orc.make_target(ref).backward()
        orc32 = O.SqairOracle({k: np.asarray(v, dtype=np.float32) for k, v in P0.items()}, O.make_cfg(F, hw), torch.float32, requires_grad=True)
        ref32 = orc32.model(obs, noise)
        orc32.make_target(ref32).backward()
        pres_same = np.array_equal(core.out["presence"].cpu().numpy().reshape(-1), ref.outputs["presence"].detach().numpy().reshape(-1))
        print("presence decisions identical: %s" % pres_same)
        lw_h = core.log_weights.cpu().numpy().reshape(B, K); lw_o = ref.log_weights.detach().numpy().reshape(B, K)
        iw_h = core.importance_weights.cpu().numpy().reshape(B, K); iw_o = ref.importance_weights.numpy().reshape(B, K)
        b_ = int(np.abs(lw_h - lw_o).max(1).argmax())
        print("log-weights: max |HIP - oracle| %.4g (rel %.3g); worst sequence %d: HIP %s oracle %s" % (
            np.abs(lw_h - lw_o).max(), np.abs(lw_h - lw_o).max() / np.abs(lw_o).max(), b_, np.array2string(lw_h[b_], precision=2),
            np.array2string(lw_o[b_], precision=2)))
        print("importance weights: max |diff| %.4g; that sequence HIP %s oracle %s" % (np.abs(iw_h - iw_o).max(), np.array2string(iw_h[b_], precision=4),
                                                                                      np.array2string(iw_o[b_], precision=4)))
        lwt_h = core.out["log_weights_per_timestep"].cpu().numpy().reshape(T, B, K); lwt_o = ref.outputs["log_weights_per_timestep"].detach().numpy().reshape(T, B, K)
        t_, bb, kk = np.unravel_index(np.abs(lwt_h - lwt_o).argmax(), lwt_h.shape)
        print("per-frame log-weights: max |diff| %.4g at frame %d seq %d particle %d: HIP %.6g oracle %.6g; most negative oracle %.6g" % (
            np.abs(lwt_h - lwt_o).max(), t_, bb, kk, lwt_h[t_, bb, kk], lwt_o[t_, bb, kk], lwt_o.min()))
        for nm in ("data_ll_per_sample", "log_q_z_given_x_per_sample", "log_p_z_per_sample"):
            h_ = core.out[nm].cpu().numpy().reshape(T, B, K)[t_, bb, kk]; o_ = ref.outputs[nm].detach().numpy().reshape(T, B, K)[t_, bb, kk]
            print("   %-28s HIP %.6f oracle %.6f" % (nm, h_, o_))
        r_ = bb * K + kk
        for nm in ("where", "where_scale", "where_loc", "what_scale", "presence"):
            print("   %-12s HIP %s" % (nm, np.array2string(core.out[nm].cpu().numpy().reshape(T, B * K, N, -1)[t_, r_].reshape(-1), precision=4)))
            print("   %-12s orc %s" % ("", np.array2string(ref.outputs[nm].detach().numpy().reshape(T, B * K, N, -1)[t_, r_].reshape(-1), precision=4)))
        off = 0
        rows = []
        for entry in spec:
            n = int(np.prod(entry[1]))
            w = orc.P[entry[0]].grad
            w = np.zeros(n) if w is None else w.numpy().reshape(-1)
            a = got[off:off + n]
            w32 = orc32.P[entry[0]].grad
            w32 = np.zeros(n) if w32 is None else w32.numpy().reshape(-1).astype(np.float64)
            rows.append((float(np.abs(a).max()), float(np.abs(w).max()), float(np.abs(a - w).max()), entry[0], float(np.abs(w32).max()), float(np.abs(w32 - w).max())))
            off += n
        print("presence of the fp32 oracle identical to fp64: %s" % np.array_equal(ref32.outputs["presence"].detach().numpy(), ref.outputs["presence"].detach().numpy()))
        for ga, gw, err, name, g32, e32 in sorted(rows, reverse=True)[:12]:
            print("  %-30s |grad| HIP %.3e  oracle fp64 %.3e (diff %.3e)  oracle fp32 %.3e (diff to fp64 %.3e)" % (name, ga, gw, err, g32, e32))
        break
    if gmax > 2e3 and fin:
        with core.on_stream():
            i = int(g.abs().argmax())
        off = 0
        for entry in spec:
            n = int(np.prod(entry[1]))
            if off <= i < off + n:
                with core.on_stream():
                    ch = core.flat[ch_off:ch_off + 10].cpu().numpy()
                    ws = float(core.out["where_scale"].min())
                # fill_triangular(v) for n = 4: diagonal entries are v[4], v[9], v[0]?? -> print all ten, the diagonal of T is (v[4+0], v[4+5]...) see oracle
                print("step %d  max|grad| %.3e at %s[%d]  elbo/frame %.2f  min where_scale %.4f  cholesky_scale %s" % (
                    it, gmax, entry[0], i - off, elbo, ws, np.array2string(ch, precision=3)), flush=True)
                break
            off += n
    if not fin or it % 500 == 0:
        print("step %d  max|grad| %.3e  elbo/frame %.2f" % (it, gmax, elbo), flush=True)
    if not fin:
        print("last 8 steps:", hist[-8:])
        gg = g.cpu().numpy(); fl = core.flat.cpu().numpy()
        off = 0
        for entry in spec:
            name, shape = entry[0], entry[1]
            n = int(np.prod(shape))
            a, p = gg[off:off + n], fl[off:off + n]
            if not np.isfinite(a).all() or not np.isfinite(p).all():
                print("  non-finite: %-40s grad nan %d inf %d | param nan %d" % (name, np.isnan(a).sum(), np.isinf(a).sum(), np.isnan(p).sum()))
            off += n
        for k in names:
            v = core.out[k].cpu().numpy()
            print("  out %-28s finite %s  min %.4g max %.4g" % (k, np.isfinite(v).all(), np.nanmin(v), np.nanmax(v)))
        break
else:
    print("no non-finite value in %d steps; max|grad| over the run %.3e" % (steps, max(h[1] for h in hist)))
