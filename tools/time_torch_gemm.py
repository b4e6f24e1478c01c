"""Reference point only (not a product path): what the vendor fp32 GEMM reaches on the same shapes."""
import torch
SHAPES = [(640, 362, 1152), (640, 312, 768), (640, 400, 256), (6400, 56, 256), (6400, 256, 256), (6400, 256, 400), (1280, 256, 256),
          (5120, 362, 1152), (51200, 256, 256), (8192, 8192, 8192)]
torch.backends.cuda.matmul.allow_tf32 = False
for M, K, N in SHAPES:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda"); y = torch.empty(M, N, device="cuda")
    for _ in range(3): torch.matmul(x, w, out=y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50 if M * K * N < 1e11 else 3
    a.record()
    for _ in range(reps): torch.matmul(x, w, out=y)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print("M=%-6d K=%-5d N=%-5d : %9.2f us  %6.1f TFLOP/s" % (M, K, N, us, 2.0 * M * K * N / us / 1e6))
