#!/usr/bin/env python3
"""bench.py -- folds/sec of the cvmatrix hot path on MI355X.

One "step" = one full cross-validation pass over device-resident inputs, the quantity the
reference's benchmark times (benchmarks/benchmark.py:101-158): CVMatrix.fit() (full-data
Gram + column statistics) followed by training_XTX_XTY for every fold (one batched call).
`value` is the package's default path: fit() is lazy and, the folds partitioning the rows,
one sweep of the Gram kernel yields the full-data matrices and every fold's matrices
(DESIGN.md 4.5); the eager two-stage path is timed next to it (two_stage_*, fit_ms,
fold_stage_ms).
Workload = BASELINE.json configs[2] ("C3"): N=100000, K=512, M=16, 10 folds
(folds = arange(N) % P), weighted, center+scale X and Y, float64, inputs from
default_rng(42).random exactly as benchmarks/benchmark.py:223-233.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): weak scaling.  Every rank
owns its own N rows and the P folds made of them; the fit stage runs on the local rows and
ONE RCCL all-reduce of [G | H | column stats] (2.2 MB) makes the full-data matrices of the
world*N-row data set; the fold stage then needs no communication.  value = folds of all
ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings)."""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, K, M, P, weighted, flags, dtype)
    "C2": (100000, 512, 16, 10, False, (False,) * 4, np.float64),
    "C3": (100000, 512, 16, 10, True, (True,) * 4, np.float64),
    "C4": (1000000, 1024, 32, 64, True, (True,) * 4, np.float64),
    "C5": (200000, 4096, 1, 20, True, (True,) * 4, np.float32),
}
PEAK_TFLOPS = {np.float64: 78.6, np.float32: 157.3}  # MI355X_MICROARCH.md (MFMA = vector peak)
PEAK_HBM_GBS = 8000.0


def synth(N, K, M, dtype, seed):
    """benchmarks/benchmark.py:223-233 (draw order X, Y, weights)."""
    rng = np.random.default_rng(seed=seed)
    X = rng.random((N, K), dtype=dtype)
    Y = rng.random((N, M), dtype=dtype)
    w = rng.random((N,), dtype=dtype)
    return X, Y, w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50,
                    help="untimed steps; the GPU needs ~30 ms of work to reach its steady clocks")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="run only warmup + the timed steps (no split timers, two-stage, per-fold-call, "
                         "supplementary or CPU legs): the command profiled under "
                         "rocprofv3 for profiles/, so that every launch in the trace is a "
                         "launch of the timed region")
    ap.add_argument("--rows", type=int, default=0, help="override N per GPU (debug)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
    # one rank per GPU.  (CVM_DIST_BACKEND=gloo lets several ranks share one GPU: used only
    # to exercise the N>1 code path on a 1-GPU box.)
    backend = os.environ.get("CVM_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cvmatrix_amd import Partitioner, _lib
    from cvmatrix_amd.distributed import ShardedCVMatrix

    lib = _lib.load()
    N, K, M, P, weighted, flags, dtype = WORKLOADS[args.workload]
    if args.rows:
        N = args.rows
    tdt = torch.float64 if dtype is np.float64 else torch.float32

    X, Y, w = synth(N, K, M, dtype, 42 + rank)
    folds = np.arange(N) % P
    Xd = torch.from_numpy(X).to(dev)
    Yd = torch.from_numpy(Y).to(dev)
    wd = torch.from_numpy(w).to(dev) if weighted else None

    # `model`: the default behaviour of the package (lazy_fit): fit() + a batched call whose folds
    # partition the rows is served by ONE sweep of the Gram kernel (full-data matrices = sum of
    # the folds' validation matrices).  `eager`: the two-stage path (fit kernel, then fold update).
    model = ShardedCVMatrix(*flags, ddof=1, dtype=dtype, copy=False, device=dev,
                            mode="row_sharded", lazy_fit=True)
    eager = ShardedCVMatrix(*flags, ddof=1, dtype=dtype, copy=False, device=dev,
                            mode="row_sharded", lazy_fit=False)
    model.fit(Xd, Yd, wd)
    eager.fit(Xd, Yd, wd)
    batch = model.prepare_folds(Partitioner(folds))

    def step():
        model.fit(Xd, Yd, wd)
        return model.training_XTX_XTY_batched(batch)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU needs ~30 ms of work to reach its steady clocks: bring it there whatever --warmup says
    # (a fixed number of steps, the same on every rank: the step contains a collective when N > 1)
    for _ in range(75 if args.workload in ("C2", "C3") else 3):
        out = step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    fence()
    lib.cvm_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    t1 = time.perf_counter()
    ms_fit, ms_fold = C.c_double(), C.c_double()
    n_fit, n_fold = C.c_int64(), C.c_int64()
    lib.cvm_timing_read(C.byref(ms_fit), C.byref(n_fit), C.byref(ms_fold), C.byref(n_fold))
    lib.cvm_timing_enable(0)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # split timers (outside the timed region): fit alone, fold stage alone
    def timed(fn, reps=10):
        fence()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - a) / reps * 1e3

    ho = args.headline_only

    def eager_step():
        eager.fit(Xd, Yd, wd)
        return eager.training_XTX_XTY_batched(batch)

    fit_ms = fold_ms = two_ms = float("nan")
    ms_fit2, ms_fold2 = C.c_double(), C.c_double()
    n_fit2, n_fold2 = C.c_int64(), C.c_int64()
    eager_out = None
    if not ho:
        for _ in range(5):
            eager_out = eager_step()
        lib.cvm_timing_enable(1)
        two_ms = timed(eager_step, reps=20)
        lib.cvm_timing_read(C.byref(ms_fit2), C.byref(n_fit2), C.byref(ms_fold2), C.byref(n_fold2))
        lib.cvm_timing_enable(0)
        if world > 1:
            t = torch.tensor([two_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            two_ms = float(t.item())
        fit_ms = timed(lambda: eager.fit(Xd, Yd, wd))
        fold_ms = timed(lambda: eager.training_XTX_XTY_batched(batch))

    # the reference's NumPy call pattern, fold by fold (benchmarks/benchmark.py:153-158):
    # fit, then one training_XTX_XTY(validation_indices) call per fold with host index arrays
    part = Partitioner(folds)
    fold_idx = [part.get_validation_indices(f) for f in part.folds_dict]

    def loop_step():
        eager.fit(Xd, Yd, wd)
        return [eager.training_XTX_XTY(v) for v in fold_idx]

    loop_ms = float("nan")
    if not ho:
        loop_step()
        loop_ms = timed(loop_step, reps=5)

    result = None
    if rank == 0:
        total_folds = P * world * args.steps
        value = total_folds / elapsed
        n_val = np.diff(batch.host_offsets).astype(np.float64)
        # algorithmic flops of one fold-stage Gram launch.  SURVEY.md 8(d) gives two
        # conventions; `achieved` uses the smaller, symmetric one (what has to be computed:
        # upper triangle of XTX + XTY), the dense one (what the reference's dgemm executes,
        # F = 2 n K (K+M)) is reported next to it.
        f_tri = float((n_val * (K * (K + 1) + 2.0 * K * M)).sum())
        f_dense = float((2.0 * n_val * K * (K + M)).sum())
        es = np.dtype(dtype).itemsize
        b_alg = float((es * n_val * (K + M + 1) + 8 * n_val).sum() + 2.0 * es * K * (K + M) * P)
        gram_ms = ms_fold.value / max(n_fold.value, 1)
        fit_gram_ms = ms_fit2.value / max(n_fit2.value, 1) if n_fit2.value else float("nan")
        two_gram_ms = ms_fold2.value / max(n_fold2.value, 1) if n_fold2.value else float("nan")
        peak = PEAK_TFLOPS[dtype]
        achieved = f_tri / (gram_ms * 1e-3) / 1e12
        info = (C.c_int64 * 8)()
        fl = 0x3F
        lib.cvm_plan_fold(P, int(n_val.max()), K, M, _lib.CVM_F64 if es == 8 else _lib.CVM_F32,
                          fl, C.c_size_t(1 << 40), info)
        executed = float(info[5]) * 2048.0 * float(np.ceil(n_val / 4.0).sum())
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("fold_gram_bytes_per_launch")
        roofline = {
            "kernel": "wgram4_kernel<T,WEIGHTED,GATHER,FUSED> (gather + weighted Gram of all folds, 1 launch/step; "
                      "in the default lazy-fit path the same launch also yields the full-data matrices)",
            "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "flops_per_launch": f_tri, "flops_convention": "n*K*(K+1) + 2*n*K*M per fold (symmetric)",
            "achieved_dense_convention": round(f_dense / (gram_ms * 1e-3) / 1e12, 3),
            "mfma_executed_tflops": round(executed / (gram_ms * 1e-3) / 1e12, 3),
            "avg_launch_ms": round(gram_ms, 4), "launches_timed": int(n_fold.value),
            "two_stage_fit_gram_avg_launch_ms": round(fit_gram_ms, 4),
            "two_stage_fit_gram_achieved": round((N * (K * (K + 1) + 2.0 * K * M)) / (fit_gram_ms * 1e-3) / 1e12, 3),
            "two_stage_fold_gram_avg_launch_ms": round(two_gram_ms, 4),
            "algorithmic_hbm_bytes_per_launch": b_alg,
            "hbm_frac_if_bytes_bound": round(b_alg / (gram_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
        }
        # parity gate in the same run (C2/C3 only: digests of the reference at these inputs)
        parity = "not checked"
        if world == 1 and not args.rows and args.workload in ("C2", "C3"):
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import parity_cases as pc
                from conftest import load_npz

                z = load_npz("g6_digest.npz")
                for res in (out,) if eager_out is None else (out, eager_out):
                    (bx, by), bst = res
                    for f in (0, 4, 9):
                        st = tuple(None if s is None else s[f] for s in bst)
                        pc.check_digest(z, args.workload.lower(), f, bx[f], by[f], st, 1e-10)
                parity = ("ok: folds 0,4,9 within 1e-10 norm-wise of the reference digests "
                          "(lazy one-sweep and two-stage)")
            except AssertionError as e:  # pragma: no cover
                parity = f"FAILED: {e}"
        # supplementary, HBM-bound regime (BASELINE.md section 2 "C5-hbm"): K=4096, M=1,
        # float32, folds of 16 rows -> the direct small-fold kernels; and leave-one-out at
        # the reference's published shape (N=1e5, K=500, M=10; benchmarks/README.md:11-20)
        supp = None
        if world == 1 and not args.rows and not ho:
            supp = {}
            for name, (n_, k_, m_, nv_, nf_, dt_) in {
                "C5-hbm (K=4096,M=1,f32,n_val=16)": (20000, 4096, 1, 16, 48, np.float32),
                "LOOCV (K=500,M=10,f64,n_val=1)": (100000, 500, 10, 1, 2000, np.float64),
            }.items():
                tt = torch.float64 if dt_ is np.float64 else torch.float32
                gen = torch.Generator(device=dev); gen.manual_seed(1)
                Xs = torch.rand((n_, k_), dtype=tt, device=dev, generator=gen)
                Ys = torch.rand((n_, m_), dtype=tt, device=dev, generator=gen)
                ws_ = torch.rand((n_,), dtype=tt, device=dev, generator=gen)
                ms_ = ShardedCVMatrix(dtype=dt_, copy=False, device=dev)
                ms_.fit(Xs, Ys, ws_)
                bs_ = ms_.prepare_folds([np.arange(i * nv_, (i + 1) * nv_) for i in range(nf_)])
                o_ = ms_.training_XTX_XTY_batched(bs_); del o_
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tl = []
                for _ in range(5):
                    e0.record(); o_ = ms_.training_XTX_XTY_batched(bs_); e1.record()
                    torch.cuda.synchronize(); tl.append(e0.elapsed_time(e1)); del o_
                ms1 = float(np.median(tl))
                sz = np.dtype(dt_).itemsize
                bts = nf_ * (sz * nv_ * (k_ + m_ + 1) + 8 * nv_ + 2 * sz * k_ * (k_ + m_))
                supp[name] = {"folds": nf_, "ms": round(ms1, 4), "folds_per_s": round(nf_ / ms1 * 1e3, 1),
                              "roofline": {"bound": "hbm", "achieved": round(bts / ms1 / 1e6, 1),
                                           "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": round(bts / ms1 / 1e6 / PEAK_HBM_GBS, 4),
                                           "bytes_per_fold": "s*n*(K+M+1) + 8n + 2*s*K*(K+M)"}}
                del Xs, Ys, ws_, ms_, bs_
        # statistics only (training_statistics): the column-statistics kernel
        # streams the validation rows once -> HBM-bound
        if supp is not None:
            # (on the eager object: after a sweep the lazy one derives the statistics from the
            #  partials it still holds, without touching the rows)
            eager.training_statistics_batched(batch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tl = []
            for _ in range(10):
                e0.record(); eager.training_statistics_batched(batch); e1.record()
                torch.cuda.synchronize(); tl.append(e0.elapsed_time(e1))
            ms1 = float(np.median(tl))
            bts = float((es * n_val * (K + M + 1) + 8 * n_val).sum())
            supp[f"training_statistics ({args.workload})"] = {
                "folds": P, "ms": round(ms1, 4), "folds_per_s": round(P / ms1 * 1e3, 1),
                "roofline": {"bound": "hbm", "achieved": round(bts / ms1 / 1e6, 1), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round(bts / ms1 / 1e6 / PEAK_HBM_GBS, 4),
                             "bytes_per_fold": "s*n*(K+M+1) + 8n"}}
        # the step after the path: Improved Kernel PLS (20 components) on the
        # training matrices of this workload's folds, where the fold stage left them
        if supp is not None:
            from cvmatrix_amd.pls import pls_fit_batched, pls_plan
            (bx, by), _ = model.training_XTX_XTY_batched(batch)
            A_pls = 20
            pls_fit_batched(bx, by, A_pls)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tl = []
            for _ in range(10):
                e0.record(); pls_fit_batched(bx, by, A_pls, check=False); e1.record()
                torch.cuda.synchronize(); tl.append(e0.elapsed_time(e1))
            ms1 = float(np.median(tl))
            pp = pls_plan(P, K, M, A_pls, np.float64 if es == 8 else np.float32)
            supp[f"pls_fit, {A_pls} components ({args.workload})"] = {
                "folds": P, "ms": round(ms1, 4), "folds_per_s": round(P / ms1 * 1e3, 1),
                "us_per_component": round(ms1 / A_pls * 1e3, 2), "plan": pp}
            del bx, by
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not ho:
            from oracle.cvmatrix_oracle import run_cv

            try:
                from threadpoolctl import threadpool_info

                thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
            except Exception:  # noqa: BLE001
                thr = os.cpu_count()
            times = []
            a0 = time.perf_counter()
            while len(times) < 3 or (time.perf_counter() - a0 < 10.0 and len(times) < 12):
                a = time.perf_counter()
                run_cv(X, Y, w if weighted else None, folds, *flags, ddof=1, dtype=dtype)
                times.append(time.perf_counter() - a)
            cpu_s = float(np.median(times))
            # one BLAS thread, one pass: lines up with the reference's published single-thread
            # numbers (benchmarks/README.md:5)
            one_thread = None
            try:
                from threadpoolctl import threadpool_limits

                with threadpool_limits(limits=1):
                    a = time.perf_counter()
                    run_cv(X, Y, w if weighted else None, folds, *flags, ddof=1, dtype=dtype)
                    one_thread = round(P / (time.perf_counter() - a), 3)
            except Exception:  # noqa: BLE001
                pass
            cpu = {"value": round(P / cpu_s, 3), "unit": "folds/s", "cores": int(thr),
                   "single_thread_value": one_thread,
                   "kind": "port",
                   "sample": f"the full {args.workload} workload (ctor+Partitioner+fit+{P} folds) "
                             f"{len(times)} times, median {cpu_s:.2f} s per pass "
                             f"({sum(times):.1f} s of CPU work), NumPy oracle "
                             f"(oracle/cvmatrix_oracle.py) on the host, BLAS threads={thr}, "
                             f"host cores={os.cpu_count()}"}
        result = {
            # BASELINE.json's metric string at its own workload (C3); other workloads say their shape
            "metric": ("folds/sec (training_XTX_XTY, center+scale) at N=1e5,K=512" if args.workload == "C3"
                       and not args.rows else f"folds/sec (training_XTX_XTY) at N={N},K={K} [{args.workload}]"),
            "value": round(value, 2), "unit": "folds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if es == 8 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={N} rows/GPU, K={K}, M={M}, {P} folds/GPU "
                                   f"(arange(N)%P), {'weighted' if weighted else 'unweighted'}, "
                                   f"center/scale X,Y={flags[0]}, fit + batched training_XTX_XTY per step "
                                   "(lazy fit: one sweep serves both calls)",
                       "parallelism": f"folds+rows sharded over {world} GPU(s); one all-reduce of [G|H|stats]"},
            "fit_ms": round(fit_ms, 4), "fold_stage_ms": round(fold_ms, 4),
            "update_only_folds_per_s": round(P / (fold_ms * 1e-3), 1),
            "two_stage_ms_per_step": round(two_ms, 4),
            "two_stage_folds_per_s": round(P * world / (two_ms * 1e-3), 1),
            "per_fold_call_ms_per_step": round(loop_ms, 4),
            "per_fold_call_folds_per_s": round(P * world / (loop_ms * 1e-3), 1),
            "parity": parity, "roofline": roofline, "cpu_baseline": cpu,
            "supplementary_hbm_regime": supp,
        }
        result = {k: (None if isinstance(v, float) and v != v else v) for k, v in result.items()}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
