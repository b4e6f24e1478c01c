"""SHA-1 of the outputs of dense layers on the knob build: python tools/big33_check.py [M,K,N,act ...] -- run it under different
knob settings (SQAIR_BIG_SHAPE, SQAIR_KL_TN2_TILES, SQAIR_MT_ROWS ...): variants that sum in the same order print the same digests."""
import ctypes as C, os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sqair_amd import _capi
from sqair_amd.flags import make_flags
from sqair_amd.model import make_config
lib = _capi.lib("tools/bin/libsqair_hip_knobs.so", allow_stale=True)
h = C.c_void_p(); cfg = make_config(make_flags(), (50, 50)); assert lib.sqair_create(C.byref(cfg), C.byref(h)) == 0
SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1920, 362, 1152, 0), (1900, 311, 1100, 1), (5120, 256, 256, 1)]
for M, K, N, act in SHAPES:
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal((M, K)).astype(np.float32)).cuda(); w = torch.tensor((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)).cuda()
    b = torch.tensor(rng.standard_normal(N).astype(np.float32)).cuda(); y = torch.zeros(M, N, device="cuda")
    scratch = torch.empty(4 * ((K + 15) // 16) * ((N + 15) // 16) * 256 + 8192 + M * (K + 4), dtype=torch.float32, device="cuda")
    rc = lib.sqair_linear_test(h, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, act, scratch.data_ptr(), scratch.numel() * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); assert rc == 0
    print(M, K, N, act, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
