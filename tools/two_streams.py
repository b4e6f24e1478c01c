"""How much throughput do INDEPENDENT passes in flight buy at BASELINE configs[1] (32 sequences each)?  n cores, each with its own
launch stream and graph, replayed concurrently: python tools/two_streams.py [n_max]."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sqair_amd.data import config_inputs  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import Model, SqairCore  # noqa: E402
from sqair_amd.params import init_params  # noqa: E402


def main():
    n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ov, obs, nums, _ = config_inputs(2)
    F = make_flags(**ov)
    hw = tuple(int(v) for v in obs.shape[2:])
    T, B = int(obs.shape[0]), int(obs.shape[1])
    P = {k: torch.as_tensor(v) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.02).items()}
    cores = []
    for i in range(n_max):
        core = SqairCore(F, hw)
        core.set_params(P)
        Model(obs, None, core, int(F.k_particles), presence=nums, outputs="minimal")
        with core.on_stream():
            core.draw_noise(torch.Generator(device="cuda").manual_seed(i))
            core.forward(use_graph=True)   # captures the pass as a graph on the core's own stream
        cores.append(core)
    torch.cuda.synchronize()
    results = []
    for n in range(1, n_max + 1):
        reps = 30
        def launch(c):  # graph replay on the core's stream, no joins with other streams
            assert c.lib.sqair_graph_launch(c.handle, c._stream()) == 0
        for _ in range(3):
            for c in cores[:n]:
                launch(c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for c in cores[:n]:
                launch(c)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        print("%d independent passes in flight: %.3f ms per round of %d x %d frames = %.1f k frames/s" % (n, el * 1e3, n, B * T, n * B * T / el / 1e3))
        results.append(dict(passes_in_flight=n, ms_per_round=el * 1e3, frames_per_s=n * B * T / el))
    import json
    print(json.dumps(dict(what="n SqairCore handles (same parameters, different noise), each replaying its forward graph on its own "
                               "stream, BASELINE configs[1] (32 sequences x T=10 x K=5 each)", results=results)))


if __name__ == "__main__":
    main()
