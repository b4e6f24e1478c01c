"""Wave lifetimes inside ONE dense launch, from the timeline build's per-wave stamps: python tools/wave_lifetimes.py M,K,N[,act] ...
Prints, per kernel launched by sqair_linear_test: waves, span of the launch, spread of the wave starts, and the distribution of
wave lifetimes (end - start) -- tells a latency-bound launch (lifetimes ~ span) from a throughput-bound one (lifetimes << span)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sqair_amd import _capi  # noqa: E402
from sqair_amd.flags import make_flags  # noqa: E402
from sqair_amd.model import make_config  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(640, 362, 1152, 0)]
    lib = _capi.lib(_capi.TIMELINE_LIB_PATH)
    h = C.c_void_p()
    cfg = make_config(make_flags(), (50, 50))
    assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = torch.zeros((256 << 20) // 8, dtype=torch.int64, device="cuda")
    for shp in shapes:
        M, K, N = shp[:3]
        act = shp[3] if len(shp) > 3 else 0
        x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / np.sqrt(K); b = torch.randn(N, device="cuda")
        y = torch.zeros(M, N, device="cuda")
        nt, kc = (N + 15) // 16, (K + 15) // 16
        scratch = torch.zeros(2 * nt * kc * 256 + 2 * nt * 16 + 256 + M * ((K + 3) // 4 * 4) + 64, device="cuda")
        args = (h, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, act, scratch.data_ptr(), scratch.numel() * 4, s)
        for _ in range(3):
            assert lib.sqair_linear_test(*args) == 0
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.sqair_timeline_begin(h, buf.data_ptr(), buf.numel() * 8) == 0
        assert lib.sqair_linear_test(*args) == 0
        torch.cuda.synchronize()
        n = lib.sqair_timeline_end(h)
        raw = buf.cpu().numpy().view(np.uint64)
        name, off, waves, wgs = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
        for i in range(n):
            lib.sqair_timeline_record(h, i, C.byref(name), C.byref(off), C.byref(waves), C.byref(wgs))
            p = raw[off.value:off.value + 2 * waves.value].reshape(-1, 2)
            p = p[p[:, 0] != 0].astype(np.int64)
            if len(p) == 0 or not name.value.decode().strip("() ").startswith("k_linear"):
                continue
            t0 = p[:, 0].min()
            life = (p[:, 1] - p[:, 0]) * 0.01
            st = (p[:, 0] - t0) * 0.01
            print("%dx%dx%d %-34s wgs %5d waves %6d | span %6.2f us | starts: median %5.2f max %5.2f | lifetime: min %5.2f median %5.2f p90 %5.2f max %5.2f" % (
                M, K, N, name.value.decode().strip("() ")[:34], wgs.value, len(p), (p[:, 1].max() - t0) * 0.01, np.median(st), st.max(),
                life.min(), np.median(life), np.percentile(life, 90), life.max()))
    lib.sqair_destroy(h)


if __name__ == "__main__":
    main()
