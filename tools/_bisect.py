import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sqair_amd.data import make_sequences, to_float
from sqair_amd.flags import make_flags
from sqair_amd.model import Model, SqairCore
from sqair_amd.params import init_params
mode = sys.argv[1]
torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
print("seed", torch.initial_seed())
B, K, N, T = 32, 5, 4, 10
F = make_flags(k_particles=K, n_steps_per_image=N)
hw = (50, 50)
d = make_sequences(B, T=T, canvas=hw, seed=3)
obs = to_float(d["imgs"])
P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
rng = np.random.default_rng(0)
noise = rng.standard_normal((T, B * K, 2, N, 55)).astype(np.float32)
noise[..., -1] = rng.uniform(size=noise.shape[:-1])
import sqair_amd.csrc.build as B_
core = SqairCore(F, hw, lib_path=(B_.OUT_KNOBS if os.environ.get("KNOBS") else None), options={"slot_chain": int(os.environ.get("SC", "1"))})
core.set_params(P)
m = Model(obs, None, core, K, presence=d["nums"])
if mode == "fwd_only":
    with core.on_stream():
        core.draw_noise(None)
        rc = core.lib.sqair_forward(*core._args(0))
        print("rc", rc, flush=True)
elif mode == "fwd_elbo":
    with core.on_stream():
        core.draw_noise(None)
        core.forward(use_graph=False)
    torch.cuda.synchronize()
    print("forward+elbo done", flush=True)
elif mode == "eager_noise":
    m.run(noise=noise, use_graph=False)
elif mode == "graph_noise":
    m.run(noise=noise, use_graph=True)
elif mode == "eager_torch":
    m.run(use_graph=False)
elif mode == "graph_torch":
    m.run(use_graph=True)
import ctypes as C
core.lib.sqair_chain_status.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
print("chain status", core.lib.sqair_chain_status(core.handle, core.workspace.data_ptr(), T, B, 0, core._stream()), flush=True)
torch.cuda.synchronize()
st = core.workspace  # status words live in the ctl blocks at the end of the workspace
print(mode, "ok, elbo", float(m.elbo_iwae), flush=True)
