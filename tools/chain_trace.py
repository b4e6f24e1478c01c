"""Per-item trace of the in-launch slot chain (knob build): where a hop's time goes.
    python tools/chain_trace.py [B] [K] [N] [T]      (on the GPU box)
One record per item (its own slot per launch, workgroup and item ordinal: no atomics in the loop): op, row tile, sub item, XCD,
rank, kind and seven stamps of the 100 MHz device clock -- t0 item start, t1 operand polls issued, t2 descriptor / weights /
epilogue operands prepared, t3 polls succeeded, t4 MFMAs done and partial sums in LDS, t5 past the barrier, t6 item end (the
stamps 1-5 exist for dense items).  Printed per op of the propagation and the discovery launch of frame 1: the hop (last end of
the op minus last end of the previous op on the same XCD) and the item time for XCDs serving one / two row tiles, and the mean
phase lengths of the dense items."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sqair_amd import _capi  # noqa: E402
from sqair_amd.csrc import build as B_  # noqa: E402
from sqair_amd.data import make_sequences, to_float  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import Model, SqairCore  # noqa: E402
from sqair_amd.params import init_params  # noqa: E402

KIND = {0: "dense", 1: "crop", 2: "rnn+tail", 3: "tail"}


def main():
    a = [int(x) for x in sys.argv[1:]]
    B, K, N, T = (a + [32, 5, 4, 10][len(a):])[:4]
    F = make_flags(k_particles=K, n_steps_per_image=N)
    hw = (50, 50)
    d = make_sequences(B, T=T, canvas=hw, seed=3)
    obs = to_float(d["imgs"])
    P = {k: np.asarray(v, dtype=np.float32) for k, v in init_params(F, hw, seed=0, mean_img=obs.mean((0, 1)), jitter=0.05).items()}
    product = os.environ.get("CHAIN_TRACE_LIB") == "product"
    core = SqairCore(F, hw, lib_path=None if product else B_.OUT_KNOBS, options={"slot_chain": 1})
    core.set_params(P)
    m = Model(obs, None, core, K, presence=d["nums"])
    m.run(use_graph=True)
    torch.cuda.synchronize()
    print("first run done", flush=True)
    if product:
        for _ in range(3):
            m.run(use_graph=True)
        torch.cuda.synchronize()
        print("product library: no trace")
        return
    lib = core.lib
    lib.sqair_chain_trace.argtypes = [C.c_void_p, C.c_uint]
    n_launch = 2 * T
    cap = n_launch * 256 * 96
    buf = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        m.run(use_graph=True)
    assert lib.sqair_chain_trace(C.c_void_p(buf.data_ptr()), cap) == 0
    m.run(use_graph=True)
    torch.cuda.synchronize()
    lib.sqair_chain_trace(None, 0)
    rec = buf.cpu().numpy().reshape(n_launch, 256, 96, 8)
    for li in (2, 3):   # frame 1: propagation, discovery
        r = rec[li].reshape(-1, 8)
        r = r[r[:, 1] != 0]
        w0 = r[:, 0].astype(np.uint64)
        op = (w0 & np.uint64(0xffff)).astype(int)
        tile = ((w0 >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
        xcc = ((w0 >> np.uint64(48)) & np.uint64(0xf)).astype(int)
        kind = ((w0 >> np.uint64(60)) & np.uint64(0xf)).astype(int)
        t = r[:, 1:8].astype(np.float64) * 0.01   # us: t0 start, t1 weights issued, t2 addresses done, t3 polled, t4 mfma + red written, t5 barrier, t6 end
        base = t[:, 0].min()
        xs = sorted(set(xcc.tolist()))
        ntile = {xx: len(set(tile[xcc == xx].tolist())) for xx in xs}
        print("launch {}: {} items, {:.1f} us from first item start to last end; row tiles per XCD {}".format(
            li, len(op), t[:, 6].max() - base, ntile))
        print("  op kind items | XCDs with 1 row tile: hop, item | XCDs with 2: hop, item || dense phases (all): desc+w issue, addr, poll, mfma, barrier, epilogue")
        prev = {xx: None for xx in xs}
        for k in sorted(set(op.tolist())):
            sel = op == k
            line = "  {:3d} {:8s} {:4d} |".format(k, KIND[int(kind[sel][0])], int(sel.sum()))
            for c in (1, 2):
                hop, item = [], []
                for xx in xs:
                    if ntile[xx] != c:
                        continue
                    s2 = sel & (xcc == xx)
                    if not s2.any():
                        continue
                    end = t[s2, 6].max()
                    if prev[xx] is not None:
                        hop.append(end - prev[xx])
                    prev[xx] = end
                    item.append((t[s2, 6] - t[s2, 0]).mean())
                line += " hop {:5.2f} item {:5.2f} |".format(float(np.mean(hop)) if hop else 0.0, float(np.mean(item)) if item else 0.0)
            if int(kind[sel][0]) == 0:
                d = t[sel]
                line += "| {:4.2f} {:4.2f} {:4.2f} {:4.2f} {:4.2f} {:4.2f}".format(*[float((d[:, i + 1] - d[:, i]).mean()) for i in range(6)])
            print(line)


if __name__ == "__main__":
    main()
