"""What clock does the latency-bound pass run at?  Replays the forward graph of BASELINE configs[1] for a few seconds at a time
while sampling the shader / fabric / memory clocks and the power draw (`rocm-smi`), under the driver's default performance level
and -- when the box lets root set it -- under `--setperflevel high` and `--setperfdeterminism <MHz>`; prints the step time of
each setting.  A measurement tool: the library and bench.py never touch the performance level.
    python tools/clock_probe.py [seconds_per_setting]"""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd.data import make_sequences, to_float  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import Model, SqairCore  # noqa: E402
from sqair_amd.params import init_params  # noqa: E402


def smi(*args):
    try:
        return subprocess.run(["rocm-smi", *args], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return "rocm-smi {}: {}".format(" ".join(args), e)


def sample(stop, out):
    while not stop.is_set():
        txt = smi("--showclocks", "--showpower", "--showperflevel")
        keep = [l.strip() for l in txt.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "Performance Level"))]
        out.append(" | ".join(keep))
        time.sleep(0.5)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    B, K, N, T, hw = 32, 5, 4, 10, (50, 50)
    F = make_flags(k_particles=K, n_steps_per_image=N)
    d = make_sequences(B, T=T, canvas=hw, seed=3)
    obs = to_float(d["imgs"])
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    core = SqairCore(F, hw)
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"])
    m.run(use_graph=True)
    torch.cuda.synchronize()

    def timed(label):
        stop, rows = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, rows))
        with core.on_stream():
            for _ in range(20):
                core.forward(use_graph=True)
            core.stream.synchronize()
            th.start()
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < secs:
                for _ in range(100):
                    core.forward(use_graph=True)
                core.stream.synchronize()
                reps += 100
            ms = (time.perf_counter() - t0) * 1e3 / reps
        stop.set()
        th.join()
        print("{:34s} {:.4f} ms per forward pass ({} replays)".format(label, ms, reps), flush=True)
        for r in rows[:: max(1, len(rows) // 4)]:
            print("      " + r, flush=True)
        return ms

    print(smi("--showclocks", "--showperflevel", "--showpower", "--showsclkrange"), flush=True)
    base = timed("default performance level")
    for label, args in (("--setperflevel high", ["--setperflevel", "high"]),
                        ("--setperfdeterminism 2400", ["--setperfdeterminism", "2400"]),
                        ("--setperfdeterminism 2100", ["--setperfdeterminism", "2100"])):
        print(smi(*args).strip()[-300:], flush=True)
        t = timed(label)
        print("      {:+.1f} % against the default level".format(100.0 * (t / base - 1.0)), flush=True)
    print(smi("--resetperfdeterminism").strip()[-200:])
    print(smi("--setperflevel", "auto").strip()[-200:])
    timed("back at auto")
    return 0


if __name__ == "__main__":
    sys.exit(main())
