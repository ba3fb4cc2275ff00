"""Where does the host time of a one-fold call go?  cProfile over a leave-one-out style loop
(3000 calls of training_XTX_XTY with a one-row fold) and over the 10-fold loop served from a sweep."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner
rng = np.random.default_rng(0)
N, K, M = 100000, 500, 10
X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
m = CVMatrix()
m.fit(X, Y, w)
p = Partitioner(np.arange(N))
keys = list(p.folds_dict)[:3000]
for k in keys[:50]: m.training_XTX_XTY(p.get_validation_indices(k))
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in keys: m.training_XTX_XTY(p.get_validation_indices(k))
torch.cuda.synchronize()
print("one-row folds, one call each: %.1f us per call" % ((time.perf_counter() - t0) / len(keys) * 1e6))
pr = cProfile.Profile(); pr.enable()
for k in keys: m.training_XTX_XTY(p.get_validation_indices(k))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

# ---- the reference's 10-fold loop at C3 (fit + one call per fold), served from one sweep
N, K, M, P = 100000, 512, 16, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device="cuda", generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device="cuda", generator=g)
w = torch.rand((N,), dtype=torch.float64, device="cuda", generator=g)
p = Partitioner(np.arange(N) % P)
vs = [p.get_validation_indices(k) for k in p.folds_dict]
m = CVMatrix(copy=False, lazy_fit=True)
def step():
    m.fit(X, Y, w)
    return [m.training_XTX_XTY(v) for v in vs]
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("C3 loop: host %.1f us per step, with the device %.1f us per step" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
