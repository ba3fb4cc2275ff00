"""Where the reference protocol's wall time goes at C3 (ctor + Partitioner + fit from host arrays + one call per
fold): phase timers around the same calls bench.py's `reference_protocol` makes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner

N, K, M, P = 100000, 512, 16, 10
rng = np.random.default_rng(42)
X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
sync = torch.cuda.synchronize


def once(detail):
    t = [time.perf_counter()]
    m = CVMatrix(copy=True); t.append(time.perf_counter())
    p = Partitioner(np.arange(N) % P); t.append(time.perf_counter())
    m.fit(X, Y, w); t.append(time.perf_counter())
    if detail: sync(); t.append(time.perf_counter())
    r = []
    for i, f in enumerate(p.folds_dict):
        r.append(m.training_XTX_XTY(p.get_validation_indices(f)))
        if i == 0:
            t.append(time.perf_counter())
    t.append(time.perf_counter())
    sync(); t.append(time.perf_counter())
    return np.diff(t) * 1e3


once(False); once(False)
for detail in (False, True):
    a = np.median([once(detail) for _ in range(5)], axis=0)
    names = ["ctor", "Partitioner", "fit (uploads inside)"] + (["sync after fit"] if detail else []) + ["first fold call (grouping, sweep launch)", "other 9 calls", "final sync"]
    print("detail" if detail else "as the protocol runs", "total %.2f ms" % a.sum())
    for n, v in zip(names, a): print("   %-42s %6.2f ms" % (n, v))
