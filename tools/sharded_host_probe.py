"""host time (enqueue only, cProfile) against GPU time of ShardedNeumf.step for one rank alone through the exchange path"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd.sharded import ShardedNeumf
B, K, d, hidden, n_users, n_items = 65536, 4, 128, 64, 1_250_001, 12_500_001
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(0)
batches = [(torch.randint(1, n_users, (B,), device=dev, generator=g), torch.randint(1, n_items, (B, 1 + K), device=dev, generator=g)) for _ in range(4)]
m = ShardedNeumf(n_users, n_items, d, hidden, opt="SGD", lr=0.01, device=dev, micro_batches=4, force_exchange=True)
for w in range(8):
    m.step(*batches[w % 4], next_batch=batches[(w + 1) % 4])
torch.cuda.synchronize()
host = []
t00 = time.perf_counter()
for k in range(40):
    t0 = time.perf_counter()
    m.step(*batches[k % 4], next_batch=batches[(k + 1) % 4])
    host.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t00
torch.cuda.synchronize()
t_all = time.perf_counter() - t00
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    m.step(*batches[k % 4], next_batch=batches[(k + 1) % 4])
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
print("host ms/step median %.3f  all enqueued after %.1f ms, GPU done after %.1f ms (%.3f ms/step)" % (1e3 * sorted(host)[20], 1e3 * t_enq, 1e3 * t_all, 1e3 * t_all / 40))
dist.destroy_process_group()
