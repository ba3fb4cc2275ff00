"""Total cross-validation wall time by the reference's own benchmark protocol, on this
implementation (protocol of benchmarks/benchmark.py:101-158, 293-308).

One timed run = constructor + Partitioner + fit (host NumPy arrays in, so the host->device copy
of X, Y, weights is inside) + training_XTX_XTY for every fold, with timeit(number=1), for
P-fold CV with folds arange(N) % P.  Two call styles:
  loop     one training_XTX_XTY(validation_indices) call per fold (the reference's NumPy style)
  batched  training_XTX_XTY_batched over chunks of `--batch` folds (its jax.vmap style)
Results stay on the device, as in the reference's JAX timing.  Output: a CSV with the
reference's columns (benchmarks/benchmark.py:33-49), so its plotting script can read it.

  python tools/benchmark_protocol.py --ps 3,5,10,100,1000,10000,100000 --csv out.csv
"""
import argparse
import os
import sys
from itertools import product
from timeit import timeit

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cvmatrix_amd  # noqa: E402
from cvmatrix_amd import CVMatrix, Partitioner  # noqa: E402

HEADER = "model,weights,P,N,K,M,center_X,center_Y,scale_X,scale_Y,time,version\n"


def run_cv(style, cv_splits, flags, X, Y, weights, batch):
    model = CVMatrix(*flags, dtype=X.dtype, copy=True, backend="hip")
    p = Partitioner(folds=cv_splits)
    model.fit(X, Y, weights)
    if style == "loop":
        for fold in p.folds_dict:
            model.training_XTX_XTY(p.get_validation_indices(fold))
    else:
        keys = list(p.folds_dict)
        for s in range(0, len(keys), batch):
            chunk = [p.folds_dict[k] for k in keys[s:s + batch]]
            out = model.training_XTX_XTY_batched(chunk)
            del out
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--k", type=int, default=500)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--ps", default="3,5,10,100,1000,10000,100000")
    ap.add_argument("--styles", default="loop,batched")
    ap.add_argument("--configs", default="plot", choices=["plot", "all"])
    ap.add_argument("--weights", default="True,False")
    ap.add_argument("--batch", type=int, default=2000, help="folds per batched call")
    ap.add_argument("--max-loop-p", type=int, default=100000)
    ap.add_argument("--csv", default="benchmark_results_hip.csv")
    args = ap.parse_args()

    rng = np.random.default_rng(seed=42)
    N, K, M = args.n, args.k, args.m
    X = rng.random((N, K), dtype=np.float64)
    Y = rng.random((N, M), dtype=np.float64)
    weights = rng.random((N,), dtype=np.float64)
    cv_splits = np.arange(N)
    if args.configs == "plot":
        configs = [(False,) * 4, (True, True, False, False), (True,) * 4]
    else:
        configs = list(product([True, False], repeat=4))
    use_w = [s == "True" for s in args.weights.split(",")]
    ps = [int(p) for p in args.ps.split(",")]
    # warm-up: load the library, create the context, touch the allocator
    run_cv("batched", cv_splits % 10, (True,) * 4, X, Y, weights, args.batch)
    if not os.path.exists(args.csv):
        with open(args.csv, "w") as f:
            f.write(HEADER)
    for w_, flags, P in product(use_w, configs, ps):
        for style in args.styles.split(","):
            if style == "loop" and P > args.max_loop_p:
                continue
            t = timeit(lambda: run_cv(style, cv_splits % P, flags, X, Y, weights if w_ else None,
                                      args.batch), number=1)
            name = "cvmatrix_amd-hip" + ("" if style == "loop" else "-batched")
            print(f"{name:26s} weights={w_!s:5s} P={P:6d} flags={flags}: {t:8.4f} s  {P / t:12.1f} folds/s",
                  flush=True)
            with open(args.csv, "a") as f:
                f.write(f"{name},{w_},{P},{N},{K},{M},{flags[0]},{flags[1]},{flags[2]},{flags[3]},"
                        f"{t},{cvmatrix_amd.__version__}\n")


if __name__ == "__main__":
    main()
