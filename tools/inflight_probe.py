import sys, time, threading
sys.path.insert(0, "/root/repo")
import torch
torch.zeros(1, device="cuda")
from chromap_amd import ChromapGPU, Stats
N = 4_000_000
g0 = ChromapGPU(synthetic=(3_100_000_000, 24, 12345), preset="atac")
gs = [g0] + [ChromapGPU(shared_from=g0) for _ in range(int(sys.argv[1]) - 1)]
for i, g in enumerate(gs):
    g.generate_resident(N, read_length=50, frag_min=30, frag_max=600, sub_rate=0.01, seed=1000 + i)
    g.map_resident(Stats())
K = 6
def work(g):
    st = Stats()
    for _ in range(K):
        g.map_resident(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(g,)) for g in gs]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(len(gs), "ctx:", round(N * K * len(gs) / dt / 1e6, 1), "M pairs/s", round(dt / K * 1e3, 2), "ms per round")
