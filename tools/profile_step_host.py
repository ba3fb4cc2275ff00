"""Host time of one bench step (fit + batched training_XTX_XTY) on the path the multi-GPU run takes
(sweep fit, exchange hook, fold stage: two library calls), one process: cProfile."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import Partitioner
from cvmatrix_amd.distributed import ShardedCVMatrix


class TwoCalls(ShardedCVMatrix):
    def _exchanges_globals(self):
        return True


N, K, M, P = 100000, 512, 16, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device="cuda", generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device="cuda", generator=g)
w = torch.rand((N,), dtype=torch.float64, device="cuda", generator=g)
for cls in (ShardedCVMatrix, TwoCalls):
    m = cls(copy=False, lazy_fit=True)
    m.fit(X, Y, w)
    b = m.prepare_folds(Partitioner(np.arange(N) % P))
    def step():
        m.fit(X, Y, w)
        return m.training_XTX_XTY_batched(b)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host %.1f us per step, with the device %.1f us per step" % (cls.__name__, (t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
