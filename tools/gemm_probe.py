"""time the three products of one MLP layer (fp32 MFMA GEMMs) alone: forward, dX, dW at M x K -> N; TFLOP/s each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rechorus_amd import engine as eng
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
dY = torch.randn(M, N, device=dev)
def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
fl = 2.0 * M * N * K
ms = timed(lambda: eng.linear_fwd(X, W, b, relu=True))
print(f"fwd   {ms * 1000:8.1f} us {fl / ms / 1e9:6.1f} TFLOP/s", flush=True)
Y = eng.linear_fwd(X, W, b, relu=True)
ms2 = timed(lambda: eng.linear_bwd(X, W, None, dY, need_dx=True))
print(f"bwd (dX + dW, no mask) {ms2 * 1000:8.1f} us {2 * fl / ms2 / 1e9:6.1f} TFLOP/s", flush=True)
ms3 = timed(lambda: eng.linear_bwd(X, W, None, dY, need_dx=False))
print(f"dW    {ms3 * 1000:8.1f} us {fl / ms3 / 1e9:6.1f} TFLOP/s   dX {1000 * (ms2 - ms3):8.1f} us {fl / (ms2 - ms3) / 1e9:6.1f} TFLOP/s", flush=True)
