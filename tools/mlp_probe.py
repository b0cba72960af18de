import torch, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rechorus_amd import engine as eng
dev = torch.device("cuda:0")
for (M, N, K) in ((1024, 512, 512), (1024, 64, 512)):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for it in range(3):
        Y = eng.linear_fwd(X, W, b, relu=True)
        torch.cuda.synchronize()
    # busy loop: 200 back-to-back launches then one timed
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(50):
        Y = eng.linear_fwd(X, W, b, relu=True)
    e1.record(); torch.cuda.synchronize()
    print("avg us per call back-to-back", M, N, K, e0.elapsed_time(e1) * 1000 / 50)
