"""Host time of one per-rank step of a G-GPU strong-scaling job (emulated, tools/emulate.py):
cProfile of fit() + batched training_XTX_XTY over the rank's folds at the C3 shape.
    python tools/profile_rank_host.py [G]"""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import Partitioner
from cvmatrix_amd.distributed import shard_folds
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emulate import EmulatedRank, others_share

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, K, M, P = 100000, 512, 16, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device="cuda", generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device="cuda", generator=g)
w = torch.rand((N,), dtype=torch.float64, device="cuda", generator=g)
labels = np.arange(N) % P
keys, rows, local = shard_folds(labels, G, 0)
sel = torch.from_numpy(rows).to(dev)
Xd, Yd, wd = X[sel].contiguous(), Y[sel].contiguous(), w[sel].contiguous()
flags = (True,) * 4
oth = others_share(flags, np.float64, dev, "row_sharded", (X, Y, w), (Xd, Yd, wd))
m = EmulatedRank(*flags, copy=False, lazy_fit=True, device=dev, emu_world=G, emu_rank=0, others=oth)
m.fit(Xd, Yd, wd)
part = Partitioner(local)
b = m.prepare_folds([part.get_validation_indices(k) for k in keys])
def step():
    m.fit(Xd, Yd, wd)
    return m.training_XTX_XTY_batched(b)
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("G=%d: issue %.1f us per step, with the device %.1f us per step" % (G, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
