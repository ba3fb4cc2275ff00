#!/usr/bin/env python3
"""bench.py -- folds/sec of the cvmatrix hot path on MI355X.

One "step" = one full cross-validation pass over device-resident inputs, the quantity the
reference's benchmark times (benchmarks/benchmark.py:101-158): CVMatrix.fit() (full-data
Gram + column statistics) followed by training_XTX_XTY for every fold (one batched call).
Workload = BASELINE.json configs[2] ("C3"): N=100000, K=512, M=16, 10 folds
(folds = arange(N) % P), weighted, center+scale X and Y, float64, inputs from
default_rng(42).random exactly as benchmarks/benchmark.py:223-233.

`value` = folds of the whole job / max-over-ranks time.  The timed path (`--path sweep`,
default) is fit() deferred + a batched call whose folds partition the rows: ONE sweep of the
Gram kernel yields the full-data matrices and every fold's matrices; the
eager two-stage path (fit kernel, then fold update) is timed next to it (two_stage_*,
fit_ms, fold_stage_ms), and so are the reference's one-call-per-fold loop and the
reference's whole benchmark protocol from host arrays.

Multi-GPU (launched by torch.distributed.run, one rank per GPU):
  --scaling strong (default)  BASELINE.json's metric: the SAME N x K x M problem and the
      same P folds on 1/2/4/8 GPUs.  Fold f -> rank (LPT on fold sizes = round-robin for
      equal folds); a rank holds the rows of its own folds only.
        --mode row_sharded (default): every rank sweeps its own folds' rows; ONE RCCL
            all-reduce of [G | H | column stats] (2.2 MB at C3) gives the full-data
            matrices; the per-fold finalize needs no communication.
        --mode replicated (the north_star variant): every rank holds all rows, rank 0 runs
            the fit stage and broadcasts [G | H | stats]; each rank updates its own folds.
      Ceiling P / ceil(P / G) folds-rounds: C3 (10 folds) 1x, 2x, 3.3x, 5x at 1/2/4/8 GPUs.
  --scaling weak  every rank owns its own N rows and P folds made of them (a world*N-row,
      world*P-fold cross-validation); one all-reduce.  The metric string says so.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings)."""

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from cvmatrix_amd.fp32_gate import fp32_bound  # noqa: E402  (the float32 accuracy contract's one definition)

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


def synth_device_rows(torch, dev, rows, N, K, M, tdt, seed, block=8192):
    """Rows ``rows`` (ascending global row numbers, a torch int64 tensor on ``dev``) of a
    synthetic N x K / N x M / N problem that is the same on every rank whatever the world
    size: row block b is drawn from a generator seeded with (seed, b).  Used for the shapes
    whose inputs are not generated on the host (C4: 8.2 GB)."""
    n_loc = int(rows.numel())
    X = torch.empty((n_loc, K), dtype=tdt, device=dev)
    Y = torch.empty((n_loc, M), dtype=tdt, device=dev)
    w = torch.empty((n_loc,), dtype=tdt, device=dev)
    gen = torch.Generator(device=dev)
    blk = torch.div(rows, block, rounding_mode="floor")
    o = 0
    for b in range((N + block - 1) // block):
        sel = rows[blk == b] - b * block
        if sel.numel() == 0:
            continue
        nb = min(block, N - b * block)
        gen.manual_seed(seed * 1000003 + b)
        xb = torch.rand((nb, K), dtype=tdt, device=dev, generator=gen)
        yb = torch.rand((nb, M), dtype=tdt, device=dev, generator=gen)
        wb = torch.rand((nb,), dtype=tdt, device=dev, generator=gen)
        n = int(sel.numel())
        X[o:o + n], Y[o:o + n], w[o:o + n] = xb[sel], yb[sel], wb[sel]
        o += n
    return X, Y, w


def direct_fold_check(torch, dist, world, Xd, Yd, wd, val_local, owner, rank, ddof, flags, got, dev,
                      yardstick=False):
    """Size-independent parity property at full size: the training-set matrices of ONE fold
    recomputed from scratch the naive way (tests/naive_cvmatrix.py's definition: centre and
    scale the training rows, then multiply) in float64 by library GEMMs -- nothing shared
    with the product path.  ``val_local``: the fold's validation rows in this rank's local
    numbering (only meaningful on ``owner``; the other ranks' rows are all training rows).
    Returns the norm-wise relative errors (XTX, XTY, mean_X, std_X) on every rank."""
    f64 = torch.float64
    n_loc = Xd.shape[0]
    keep = torch.ones(n_loc, dtype=torch.bool, device=dev)
    if rank == owner and val_local is not None:
        keep[val_local] = False
    Xt, Yt = Xd[keep].to(f64), Yd[keep].to(f64)
    wt = wd[keep].to(f64) if wd is not None else torch.ones(Xt.shape[0], dtype=f64, device=dev)
    K, M = Xt.shape[1], Yt.shape[1]
    cX, cY, sX, sY = flags

    def allsum(t):
        if world > 1:
            dist.all_reduce(t)
        return t

    head = allsum(torch.stack([wt.sum(), (wt != 0).sum().to(f64)]))
    sw, nz = head[0], head[1]
    s1 = allsum(torch.cat([(Xt * wt[:, None]).sum(0), (Yt * wt[:, None]).sum(0)]))
    muX, muY = s1[:K] / sw, s1[K:] / sw
    if not (cX or cY):
        muXc, muYc = torch.zeros_like(muX), torch.zeros_like(muY)
    else:
        # cvmatrix.py:1001-1010: XTX is centred with mu_X only if center_X; XTY's rank-1 term
        # uses both means whenever either flag is set
        muXc, muYc = muX, muY
    div = (nz - ddof) * sw / nz
    v2 = allsum(torch.cat([((Xt - muX) ** 2 * wt[:, None]).sum(0), ((Yt - muY) ** 2 * wt[:, None]).sum(0)]))
    sdX, sdY = (v2[:K] / div).sqrt(), (v2[K:] / div).sqrt()
    Xc = Xt - muXc if cX else Xt
    Xs = Xc / sdX if sX else Xc
    refX = allsum((Xs * wt[:, None]).T @ Xs)
    # XTY: (X - muX)^T W (Y - muY) when either centring flag is set
    Xc2 = (Xt - muX) if (cX or cY) else Xt
    Yc2 = (Yt - muY) if (cX or cY) else Yt
    Xs2 = Xc2 / sdX if sX else Xc2
    Ys2 = Yc2 / sdY if sY else Yc2
    refY = allsum((Xs2 * wt[:, None]).T @ Ys2)
    errs = torch.zeros(4, dtype=f64, device=dev)
    if rank == owner:
        gx, gy, gmu, gsd = got
        dx, dy = gx.to(f64) - refX, gy.to(f64) - refY
        errs[0] = torch.maximum(dx.abs().max() / refX.abs().max(), dx.norm() / refX.norm())
        errs[1] = torch.maximum(dy.abs().max() / refY.abs().max(), dy.norm() / refY.norm())
        if gmu is not None:
            errs[2] = ((gmu.to(f64).reshape(-1) - muX).abs() / muX.abs()).max()
        if gsd is not None:
            errs[3] = ((gsd.to(f64).reshape(-1) - sdX).abs() / sdX.abs()).max()
    if world > 1:
        dist.all_reduce(errs, op=dist.ReduceOp.MAX)
    del Xt, Yt, Xs, Xs2, Ys2, Xc, Xc2, Yc2
    out = [float(e) for e in errs.cpu()]
    if yardstick and world == 1:
        # the float32 restatement of the reference's algorithm against the same float64 result
        yx, yy = fp32_algorithm_error(torch, Xd, Yd, wd, val_local, ddof, flags, dev)
        dx, dy = yx - refX, yy - refY
        out.append(float(torch.maximum(dx.abs().max() / refX.abs().max(), dx.norm() / refX.norm())))
        out.append(float(torch.maximum(dy.abs().max() / refY.abs().max(), dy.norm() / refY.norm())))
    return out


def fp32_algorithm_error(torch, Xd, Yd, wd, val, ddof, flags, dev):
    """The yardstick of the float32 parity bound ("error must not exceed 2x
    NumPy-fp32's own error"): the reference's algorithm (full-data Gram minus the validation rows'
    Gram, then the mean/std correction: cvmatrix.py:1001-1010, 1119-1128) restated in plain float32
    torch ops (library GEMMs, float32 accumulation) on the same inputs, against the from-scratch
    float64 result.  Returns the norm-wise relative errors (XTX, XTY) of that float32 run."""
    f32, f64 = torch.float32, torch.float64
    X, Y = Xd.to(f32), Yd.to(f32)
    w = wd.to(f32) if wd is not None else torch.ones(X.shape[0], dtype=f32, device=dev)
    cX, cY, sX, sY = flags
    WX = X * w[:, None]
    G, H = WX.T @ X, WX.T @ Y
    sw, nz = w.sum(), (w != 0).sum().to(f32)
    s_x, s_y = WX.sum(0), (Y * w[:, None]).sum(0)
    q_x, q_y = (WX * X).sum(0), (Y * Y * w[:, None]).sum(0)
    Xv, Yv, wv = X[val], Y[val], w[val]
    WXv = Xv * wv[:, None]
    Gt, Ht = G - WXv.T @ Xv, H - WXv.T @ Yv
    swt, nzt = sw - wv.sum(), nz - (wv != 0).sum().to(f32)
    sxt, syt = s_x - WXv.sum(0), s_y - (Yv * wv[:, None]).sum(0)
    qxt, qyt = q_x - (WXv * Xv).sum(0), q_y - (Yv * Yv * wv[:, None]).sum(0)
    mux, muy = sxt / swt, syt / swt
    div = (nzt - ddof) * swt / nzt
    sdx = torch.sqrt(torch.clamp((-2 * mux * sxt + swt * mux * mux + qxt) / div, min=0))
    sdy = torch.sqrt(torch.clamp((-2 * muy * syt + swt * muy * muy + qyt) / div, min=0))
    if cX:
        Gt = Gt - swt * torch.outer(mux, mux)
    if cX or cY:
        Ht = Ht - swt * torch.outer(mux, muy)
    if sX:
        Gt = Gt / torch.outer(sdx, sdx)
    if sX and sY:
        Ht = Ht / torch.outer(sdx, sdy)
    elif sX:
        Ht = Ht / sdx[:, None]
    elif sY:
        Ht = Ht / sdy[None, :]
    del WX, WXv, G, H
    return Gt.to(f64), Ht.to(f64)


def measure_hbm_traffic(workload, path):
    """HBM bytes per launch of the dominant Gram kernel, measured NOW: two child runs of this
    script's headline loop under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
    passes, as MI355X_MICROARCH.md's HBM section prescribes; counter KB -> bytes x1024, FETCH_SIZE
    doubled on gfx950).  Returns (bytes, fetch, write, launches) or None if the profiler is not
    usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return None
    out = {}
    base = tempfile.mkdtemp(prefix="cvm_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--headline-only", "--steps", "5", "--warmup", "2",
                   "--workload", workload, "--path", path, "--no-live-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                               timeout=240)
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = row["Kernel_Name"]
                        # the gathered Gram launch: wgram4_kernel<T, WEIGHTED, GATHER = true, FUSED = false>
                        if "wgram4_kernel" in k and row["Counter_Name"] == ctr and ", true, false>" in k:
                            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            out[ctr] = (sum(per.values()) / len(per), len(per))
        fetch = out["FETCH_SIZE"][0] * 1024.0 * 2.0
        write = out["WRITE_SIZE"][0] * 1024.0
        return fetch + write, fetch, write, out["FETCH_SIZE"][1]
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(base, ignore_errors=True)


def clock_probe_summary(stamps):
    """cvm_clock_probe's buffer -> shader clock and workgroup lifetimes of the last probed launch.  Per workgroup
    [s_memtime, s_memrealtime] at its start and when it ran out of work; s_memrealtime ticks at 100 MHz."""
    s = np.asarray(stamps).astype(np.uint64).reshape(-1, 4)
    s = s[(s[:, 1] > 0) & (s[:, 3] > s[:, 1])]
    if not len(s):
        return None
    dc = (s[:, 2] - s[:, 0]).astype(np.float64)
    dq = (s[:, 3] - s[:, 1]).astype(np.float64)
    mhz = dc / dq * 100.0
    span = float(s[:, 3].max() - s[:, 1].min())
    return {"effective_clock_mhz": round(float(np.median(mhz)), 1),
            "clock_mhz_min_max_over_workgroups": [round(float(mhz.min()), 1), round(float(mhz.max()), 1)],
            "workgroups": int(len(s)), "workgroup_life_us_mean": round(float(dq.mean()) / 100.0, 1),
            "workgroup_life_us_max": round(float(dq.max()) / 100.0, 1), "launch_span_us": round(span / 100.0, 1),
            "cu_time_used": round(float(dq.sum() / (len(s) * span)), 4)}


def self_launch(n_ranks):
    """Start `torch.distributed.run` with one rank per GPU as a child process, on a free port of the
    loopback interface; the ranks re-enter this script with WORLD_SIZE set.  Returns the child's exit
    code.  (The parent must not have initialised the GPU: it has not imported torch.)"""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50,
                    help="untimed steps; the GPU needs ~30 ms of work to reach its steady clocks")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--mode", default="row_sharded", choices=["row_sharded", "replicated"])
    ap.add_argument("--path", default="sweep", choices=["sweep", "two_stage"],
                    help="timed path: lazy fit + one sweep (default) or eager fit + fold update")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="run only warmup + the timed steps (no split timers, two-stage, per-fold-call, "
                         "supplementary or CPU legs): the command profiled under "
                         "rocprofv3 for profiles/, so that every launch in the trace is a "
                         "launch of the timed region")
    ap.add_argument("--fresh-outputs", action="store_true",
                    help="time the default API behaviour (every call allocates fresh result tensors) instead of "
                         "reuse_outputs=True (results written into buffers the model keeps while shapes repeat: "
                         "what a loop that consumes each step's results before the next wants; the per-rank step "
                         "of an 8-GPU job is shorter than those allocations)")
    ap.add_argument("--brief", action="store_true",
                    help="--headline-only plus the full-size from-scratch parity check: what the default N=1 run "
                         "starts as a child process for every other BASELINE workload (other_workloads)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the child runs of the other BASELINE workloads (C2, C4, C5) at N=1")
    ap.add_argument("--rows", type=int, default=0, help="override the global N (debug / tests)")
    ap.add_argument("--device-data", action="store_true",
                    help="generate the inputs on the device (default for C4)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="run ONE rank's share of a G-GPU strong-scaling job on this GPU, without a process "
                         "group: shard_folds(labels, G, rank), the exact per-rank step (sweep of the rank's own "
                         "folds -> fit finalize -> exchange STUB -> per-fold finalize), and print the per-rank "
                         "breakdown.  The stub adds the other ranks' (precomputed) share of [G|H|stats] in place "
                         "-- the arithmetic of the all-reduce's last step -- and then holds the stream for --comm-us")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which rank of --emulate-world (0 owns the most folds)")
    ap.add_argument("--comm-us", type=float, default=0.0,
                    help="--emulate-world: latency of the emulated collective, held on the stream after the add")
    ap.add_argument("--streams", type=int, default=2,
                    help="supplementary pipelined figure: consecutive steps alternate over this many HIP streams "
                         "(one model per stream, the same inputs), so that a step's finalize/exchange overlaps "
                         "the next step's sweep; 1 = skip")
    ap.add_argument("--with-breakdown", action="store_true",
                    help="with --headline-only: still run the per-rank breakdown and the pipelined figure")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-measure roofline.traffic with rocprofv3 child runs (two short passes, "
                         "~1 min); the figure of profiles/hbm_traffic.json is reported instead")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` (N > 1) without a launcher: start the ranks ourselves, the way the reference's
    # harness is one command (benchmarks/benchmark.py:293-308).  This process never touches the GPU -- torch is
    # not even imported yet -- it starts torch.distributed.run as a CHILD, lets rank 0's JSON line through and
    # leaves with the child's exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_world:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = args.emulate_world if args.emulate_world > 1 else 0
    if emu and world > 1:
        sys.exit("--emulate-world runs in ONE process (no torch.distributed.run)")
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} was started as one of {world} ranks: the launcher's "
                 "--nproc-per-node and --gpus must agree")
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

    from cvmatrix_amd import CVMatrix, Partitioner, _lib
    from cvmatrix_amd.distributed import ShardedCVMatrix, shard_folds

    lib = _lib.load()
    N, K, M, P, weighted, flags, dtype = WORKLOADS[args.workload]
    if args.rows:
        N = args.rows
    tdt = torch.float64 if dtype is np.float64 else torch.float32
    es = np.dtype(dtype).itemsize
    strong = args.scaling == "strong"
    if emu and not strong:
        sys.exit("--emulate-world emulates the strong-scaling layout")
    # the LAYOUT (which folds and rows this process holds) is that of `lay_world` ranks; `world` is
    # the number of processes that really exchange (1 under --emulate-world)
    lay_world, lay_rank = (emu, args.emulate_rank) if emu else (world, rank)
    mode = args.mode if (strong and lay_world > 1) else "row_sharded"
    device_data = args.device_data or args.workload == "C4"

    # ---- the problem, and this rank's share of it ------------------------------------------
    X = Y = w = None
    if strong:
        labels = np.arange(N) % P                      # the folds of the WHOLE problem
        keys, rows, local_labels = shard_folds(labels, lay_world, lay_rank)
        if mode == "replicated":
            rows_held = np.arange(N)
            part = Partitioner(labels)
            fold_lists = [part.get_validation_indices(k) for k in keys]
        else:
            rows_held = rows
            part = Partitioner(local_labels)
            fold_lists = [part.get_validation_indices(k) for k in keys]
        Xf = Yf = wf = None                            # (--emulate-world: the WHOLE problem, for the stub)
        if device_data:
            if emu:
                Xf, Yf, wf = synth_device_rows(torch, dev, torch.arange(N, device=dev), N, K, M, tdt, 42)
                sel_d = torch.from_numpy(rows_held).to(dev)
                Xd, Yd, wd = Xf[sel_d].contiguous(), Yf[sel_d].contiguous(), wf[sel_d].contiguous()
            else:
                Xd, Yd, wd = synth_device_rows(torch, dev, torch.from_numpy(rows_held).to(dev), N, K, M, tdt, 42)
        else:
            X, Y, w = synth(N, K, M, dtype, 42)
            sel = slice(None) if rows_held.size == N else rows_held
            Xd = torch.from_numpy(X[sel]).to(dev)
            Yd = torch.from_numpy(Y[sel]).to(dev)
            wd = torch.from_numpy(w[sel]).to(dev)
            if emu:
                Xf, Yf, wf = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev), torch.from_numpy(w).to(dev)
        total_folds_per_step = P
    else:
        keys = list(range(P))
        labels = np.arange(N) % P
        if device_data:
            Xd, Yd, wd = synth_device_rows(torch, dev, torch.arange(N, device=dev), N, K, M, tdt, 42 + rank)
        else:
            X, Y, w = synth(N, K, M, dtype, 42 + rank)
            Xd, Yd, wd = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev), torch.from_numpy(w).to(dev)
        part = Partitioner(labels)
        fold_lists = [part.get_validation_indices(k) for k in keys]
        total_folds_per_step = P * world
    if not weighted:
        wd = wf = None
    n_mine = len(fold_lists)

    # ---- --emulate-world: one rank of a `emu`-rank job, the exchange replaced by a stub ------
    # (tools/emulate.py: every code path of the real multi-GPU step runs; the collective is
    #  an in-place add of the other ranks' precomputed share, then --comm-us of held stream)
    if emu:
        import functools

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from emulate import EmulatedRank, others_share, sleep_cycles_for

        Model = functools.partial(EmulatedRank, emu_world=lay_world, emu_rank=lay_rank,
                                  others=others_share(flags, dtype, dev, mode, (Xf, Yf, wf), (Xd, Yd, wd)),
                                  sleep_cycles=sleep_cycles_for(args.comm_us))
    else:
        Model = ShardedCVMatrix

    # `model`: lazy fit; fit() + a batched call whose folds partition the (local) rows is served
    # by ONE sweep of the Gram kernel (full-data matrices = sum of the folds' validation
    # matrices, all-reduced over the ranks).  `eager`: fit kernel, then fold update.
    # (copy=False: the inputs are already private device tensors of this process.)
    reuse = not args.fresh_outputs
    model = Model(*flags, ddof=1, dtype=dtype, copy=False, device=dev, mode=mode, lazy_fit=True, reuse_outputs=reuse,
                  trust_tensor_versions=True)
    eager = Model(*flags, ddof=1, dtype=dtype, copy=False, device=dev, mode=mode, lazy_fit=False, reuse_outputs=reuse,
                  trust_tensor_versions=True)
    timed_model = model if args.path == "sweep" else eager
    model.fit(Xd, Yd, wd)
    eager.fit(Xd, Yd, wd)
    batch = model.prepare_folds(fold_lists) if n_mine else None

    def step_of(m, b=None):
        b = batch if b is None else b

        def step():
            m.fit(Xd, Yd, wd)
            if b is None:              # more ranks than folds: take part in the exchange only
                m.ensure_fit()
                return None
            return m.training_XTX_XTY_batched(b)
        return step

    step = step_of(timed_model)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the clock the chip holds under the Gram kernel, read by the kernel itself (cvm_clock_probe: two scalar clock
    # pairs per workgroup lifetime, nothing inside the item loop; every launch overwrites the buffer, what is read
    # after the timed region are the stamps of its LAST Gram launch).  The buffer is allocated HERE, before the
    # warm-up: an allocation + fill between the warm-up and the timed steps idles the GPU long enough for its clock
    # to fall, and a timed region of 20 steps (10 ms) then runs inside the ramp back (2.2 instead of 2.39 GHz,
    # 0.53 instead of 0.48 ms per step with `--steps 20 --warmup 5`, whatever the length of the warm-up)
    clock_buf = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    # the GPU needs ~30 ms of work to reach its steady clocks: bring it there whatever --warmup says
    # (a fixed number of steps, the same on every rank: the step contains a collective when N > 1)
    for _ in range(int(os.environ.get("CVM_BENCH_PREWARM", "75")) if args.workload in ("C2", "C3") else 3):
        out = step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step()
    lib.cvm_clock_probe(C.c_void_p(clock_buf.data_ptr()), C.c_size_t(clock_buf.numel() * 8))      # (host only)
    fence()
    lib.cvm_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    t1 = time.perf_counter()
    lib.cvm_clock_probe(None, C.c_size_t(0))
    clock = clock_probe_summary(clock_buf.cpu().numpy())
    ms_fit, ms_fold = C.c_double(), C.c_double()
    n_fit, n_fold = C.c_int64(), C.c_int64()
    lib.cvm_timing_read(C.byref(ms_fit), C.byref(n_fit), C.byref(ms_fold), C.byref(n_fold))
    lib.cvm_timing_enable(0)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-rank breakdown of a step (outside the timed region) -----------------------------
    # Isolated steps (device idle before each): events on the step's stream at its start, around the
    # exchange (ShardedCVMatrix._probe) and at its end; the Gram launch by the library's own events.
    # host_ms = wall time of issuing the step minus the time the host spends waiting for the
    # device inside it (the all-reduced sample counts the validity checks need, multi-GPU only).
    def breakdown(m, b, reps=20):
        marks, waited = {}, [0.0]

        def probe(label):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks[label] = e

        orig_resolve = m._resolve_totals

        def resolve():
            a = time.perf_counter()
            orig_resolve()
            waited[0] += time.perf_counter() - a

        m._probe, m._resolve_totals = probe, resolve
        st = step_of(m, b)
        acc = {"host_ms": [], "host_wait_ms": [], "latency_ms": [], "sweep_and_fit_finalize_ms": [],
               "exchange_ms": [], "fold_finalize_ms": [], "device_ms": []}
        lib.cvm_timing_enable(1)
        try:
            for _ in range(reps):
                fence()
                marks.clear(); waited[0] = 0.0
                e0, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                a = time.perf_counter()
                st()
                bt = time.perf_counter()
                e3.record()
                torch.cuda.synchronize()
                ct = time.perf_counter()
                acc["host_ms"].append((bt - a - waited[0]) * 1e3)
                acc["host_wait_ms"].append(waited[0] * 1e3)
                acc["latency_ms"].append((ct - a) * 1e3)
                acc["device_ms"].append(e0.elapsed_time(e3))
                if "exchange_begin" in marks:
                    acc["sweep_and_fit_finalize_ms"].append(e0.elapsed_time(marks["exchange_begin"]))
                    acc["exchange_ms"].append(marks["exchange_begin"].elapsed_time(marks["exchange_end"]))
                    acc["fold_finalize_ms"].append(marks["exchange_end"].elapsed_time(e3))
        finally:
            m._probe = None
            m._resolve_totals = orig_resolve
        g_ms, g_n, f_ms, f_n = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
        lib.cvm_timing_read(C.byref(g_ms), C.byref(g_n), C.byref(f_ms), C.byref(f_n))
        lib.cvm_timing_enable(0)
        out_ = {k: round(float(np.median(v)), 4) for k, v in acc.items() if v}
        gl = (g_ms.value + f_ms.value) / max(g_n.value + f_n.value, 1)
        out_["gram_ms"] = round(gl, 4)
        if "sweep_and_fit_finalize_ms" in out_:
            out_["fit_finalize_ms"] = round(out_["sweep_and_fit_finalize_ms"] - gl, 4)
            out_["finalize_ms"] = round(out_["fit_finalize_ms"] + out_["fold_finalize_ms"], 4)
        else:
            out_["finalize_ms"] = round(out_["device_ms"] - gl, 4)
        out_["what"] = (f"median of {reps} isolated steps of this rank (device idle before each); gram_ms: the library's "
                        "events around the Gram launch; exchange_ms: events around the collective on the step's stream; "
                        "host_ms: issuing the step, waits for the device excluded (host_wait_ms)")
        if world > 1:
            for k in list(out_):
                if k.endswith("_ms"):
                    tt = torch.tensor([out_[k]], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    out_[k] = round(float(tt.item()), 4)
            out_["what"] += "; max over ranks"
        return out_

    # ---- pipelined steps: consecutive steps alternate over S streams (one model each) ----------
    def pipelined(S, steps_, warm_):
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        ms_ = [Model(*flags, ddof=1, dtype=dtype, copy=False, device=dev, mode=mode, lazy_fit=(args.path == "sweep"),
                     reuse_outputs=reuse, trust_tensor_versions=True)
               for _ in range(S)]
        sts = []
        for m_, s_ in zip(ms_, streams):
            with torch.cuda.stream(s_):
                m_.fit(Xd, Yd, wd)
                b_ = m_.prepare_folds(fold_lists) if n_mine else None
                sts.append(step_of(m_, b_))
                if b_ is None:
                    m_.ensure_fit()
        keep = [None] * S
        fence()
        # (its own pre-warm, whatever --warmup / --steps say: new models on new streams, and the GPU needs
        #  ~30 ms of work for its steady state -- timed cold, two streams came out SLOWER than one: the
        #  round-3 driver run, 20 steps after 5)
        warm_ = max(warm_, 75 if args.workload in ("C2", "C3") else 3)
        steps_ = max(steps_, 100 if args.workload in ("C2", "C3") else 3)
        for i in range(warm_):
            with torch.cuda.stream(streams[i % S]):
                keep[i % S] = sts[i % S]()
        fence()
        a = time.perf_counter()
        for i in range(steps_):
            with torch.cuda.stream(streams[i % S]):
                keep[i % S] = sts[i % S]()
        fence()
        el = time.perf_counter() - a
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        same = None
        if out is not None and keep[0] is not None and args.path == "sweep":
            (ax, ay), _ = out
            (bx_, by_), _ = keep[0]
            same = bool(torch.equal(ax, bx_) and torch.equal(ay, by_))
        del keep, ms_
        return {"streams": S, "steps": steps_, "ms_per_step": round(el / steps_ * 1e3, 4),
                "folds_per_s": round(total_folds_per_step * steps_ / el, 2),
                "identical_to_single_stream": same,
                "what": f"the same step, consecutive steps alternating over {S} HIP streams (one model per stream, "
                        "the same resident inputs): a step's finalize / exchange overlaps the next step's sweep"}

    # split timers (outside the timed region): fit alone, fold stage alone
    def timed(fn, reps=10):
        fence()
        a = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - a) / reps * 1e3
        if world > 1:
            tt = torch.tensor([ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
        return ms

    ho = args.headline_only or args.brief
    if emu:
        # ---- one emulated rank: breakdown, pipelined figure, parity of THIS rank's folds, one line ----
        bd = breakdown(timed_model, batch)
        pipe = pipelined(args.streams, args.steps, args.warmup) if args.streams > 1 else None
        notes, ok = [], True
        if out is not None:
            (bx, by), bst = out
            if not args.rows and not device_data and args.workload in ("C2", "C3"):
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import parity_cases as pc
                    from conftest import load_npz

                    z = load_npz("g6_digest.npz")
                    for i, f in enumerate(keys):
                        st = tuple(None if s_ is None else s_[i] for s_ in bst)
                        pc.check_digest(z, args.workload.lower(), int(f), bx[i], by[i], st, 1e-10)
                    notes.append(f"folds {[int(f) for f in keys]} of this rank within 1e-10 norm-wise of the reference digests")
                except AssertionError as e:  # pragma: no cover
                    ok = False
                    notes.append(f"digest FAILED: {e}")
            # this rank's last fold against a from-scratch float64 computation over the WHOLE problem
            vglob = torch.from_numpy(np.asarray(rows_held)[np.asarray(fold_lists[-1])]).to(dev)
            got = (bx[-1], by[-1], None if bst[0] is None else bst[0][-1], None if bst[1] is None else bst[1][-1])
            errs = direct_fold_check(torch, dist, 1, Xf, Yf, wf, vglob, 0, 0, 1, flags, got, dev, yardstick=(es == 4))
            bound = 1e-10 if es == 8 else fp32_bound(errs[4])
            bound_y = 1e-10 if es == 8 else fp32_bound(errs[5])
            good = errs[0] <= bound and errs[1] <= bound_y and all(e <= max(bound, 1e-10) for e in errs[2:4])
            ok = ok and good
            notes.append(f"fold {keys[-1]} vs a from-scratch float64 computation over all {N} rows: XTX {errs[0]:.1e}, "
                         f"XTY {errs[1]:.1e}, mean {errs[2]:.1e}, std {errs[3]:.1e}{'' if good else ' FAILED'}")
        n_val = np.diff(batch.host_offsets).astype(np.float64) if batch is not None else np.zeros(0)
        f_tri = float((n_val * (K * (K + 1) + 2.0 * K * M)).sum())
        gram_ms = ms_fold.value / max(n_fold.value, 1)
        step_ms = elapsed / args.steps * 1e3
        ceiling = P / math.ceil(P / emu)
        line = {
            "emulated": True, "emulate_world": emu, "emulate_rank": lay_rank, "mode": mode, "path": args.path,
            "workload": args.workload, "folds_of_this_rank": [int(f) for f in keys], "rows_of_this_rank": int(Xd.shape[0]),
            "comm_us_injected": args.comm_us,
            "per_rank_step_ms": round(step_ms, 4),
            "predicted_job_folds_per_s": round(P / (step_ms * 1e-3), 1),
            "scaling_ceiling_vs_1gpu": round(ceiling, 3),
            "gram_ms_in_timed_steps": round(gram_ms, 4),
            "gram_frac_of_mfma_peak": round(f_tri / (gram_ms * 1e-3) / 1e12 / PEAK_TFLOPS[dtype], 4) if gram_ms > 0 else None,
            "breakdown": bd, "pipelined": pipe,
            "steps": args.steps, "warmup": args.warmup,
            "parity": ("ok: " if ok else "FAILED: ") + "; ".join(notes),
            "what": (f"rank {lay_rank} of a {emu}-GPU strong-scaling job on ONE GPU, no process group: the real per-rank "
                     "step with the collective replaced by an in-place add of the other ranks' share (+ --comm-us of "
                     "held stream); predicted_job_folds_per_s = P / per_rank_step (the slowest rank is rank 0: it owns "
                     "ceil(P/G) folds)"),
            "lib": lib.cvm_version().decode(),
        }
        print(json.dumps(line), flush=True)
        return line
    other = eager if timed_model is model else model
    other_step = step_of(other)
    fit_ms = fold_ms = two_ms = float("nan")
    ms_fit2, ms_fold2 = C.c_double(), C.c_double()
    n_fit2, n_fold2 = C.c_int64(), C.c_int64()
    other_out = None
    if not ho:
        for _ in range(5):
            other_out = other_step()
        lib.cvm_timing_enable(1)
        two_ms = timed(other_step, reps=20)
        lib.cvm_timing_read(C.byref(ms_fit2), C.byref(n_fit2), C.byref(ms_fold2), C.byref(n_fold2))
        lib.cvm_timing_enable(0)
        fit_ms = timed(lambda: eager.fit(Xd, Yd, wd))
        if batch is not None:
            fold_ms = timed(lambda: eager.training_XTX_XTY_batched(batch))

    # the reference's call pattern, fold by fold (benchmarks/benchmark.py:153-158, README.md:120-141):
    # fit, then one training_XTX_XTY(p.get_validation_indices(fold)) call per fold.  With the lazy
    # fit the first call recognises the Partitioner's index array, sweeps all of its folds once and
    # every call is served from that sweep's partials; the eager object pays a Gram launch per call.
    def loop_step_of(m):
        def loop_step():
            m.fit(Xd, Yd, wd)
            return [m.training_XTX_XTY(v) for v in fold_lists]
        return loop_step

    loop_ms = loop2_ms = float("nan")
    loop_out = None
    if not ho and n_mine:
        looper = ShardedCVMatrix(*flags, ddof=1, dtype=dtype, copy=False, device=dev, mode=mode, lazy_fit=True)
        ls = loop_step_of(looper)
        loop_out = ls()
        loop_ms = timed(ls, reps=10)
        ls2 = loop_step_of(eager)
        ls2()
        loop2_ms = timed(ls2, reps=5)

    # the same step with the other output policy (default: the API's fresh tensors per call), timed the same way
    alt = None
    if not ho:
        am = Model(*flags, ddof=1, dtype=dtype, copy=False, device=dev, mode=mode, lazy_fit=(args.path == "sweep"),
                   reuse_outputs=not reuse, trust_tensor_versions=True)
        am.fit(Xd, Yd, wd)
        ab = am.prepare_folds(fold_lists) if n_mine else None
        ast_ = step_of(am, ab)
        for _ in range(int(os.environ.get("CVM_BENCH_PREWARM", "75")) if args.workload in ("C2", "C3") else 3):
            ast_()
        alt_ms = timed(ast_, reps=100 if args.workload in ("C2", "C3") else 5)
        alt = {"outputs": "reused buffers (reuse_outputs=True)" if not reuse else "fresh tensors every call (the API's default)",
               "ms_per_step": round(alt_ms, 4), "folds_per_s": round(total_folds_per_step / (alt_ms * 1e-3), 1),
               "host_ms": breakdown(am, ab, reps=10)["host_ms"]}
        del am, ab, ast_
    # the same step through the API AS THE REFERENCE SPELLS IT (cvmatrix/cvmatrix.py:157-167: every optional
    # argument at its default; only the workload's own flags / dtype are passed), on the same resident tensors:
    # copy=True (private device copies of X, Y, weights at every fit), lazy fit, fresh result tensors every call,
    # the weights validated on every fit (on the device: no read-back, no host wait) -- and the same with
    # copy=False (inputs aliased; fit() is then eager like the reference's, two Gram launches per step) and with
    # copy=False + lazy_fit=True (one sweep).  None of them uses trust_tensor_versions / reuse_outputs.
    api = None
    if not ho and world == 1 and not emu and n_mine:
        api = {"what": "fit + one batched training_XTX_XTY per step on the resident tensors, timed like other_output_policy; "
                       "no trust_tensor_versions, no reuse_outputs: every fit re-validates its inputs"}
        for label, kw in (("CVMatrix()", {}), ("CVMatrix(copy=False)", {"copy": False}),
                          ("CVMatrix(copy=False, lazy_fit=True)", {"copy": False, "lazy_fit": True})):
            dm = CVMatrix(*flags, ddof=1, dtype=dtype, **kw)
            dm.fit(Xd, Yd, wd)
            db_ = dm.prepare_folds(fold_lists)
            dst = step_of(dm, db_)
            for _ in range(40 if args.workload in ("C2", "C3") else 2):
                dst()
            d_ms = timed(dst, reps=100 if args.workload in ("C2", "C3") else 5)
            api[label] = {"ms_per_step": round(d_ms, 4), "folds_per_s": round(total_folds_per_step / (d_ms * 1e-3), 1),
                          "host_ms": breakdown(dm, db_, reps=10)["host_ms"]}
            del dm, db_, dst
        torch.cuda.empty_cache()
    # per-rank breakdown of the timed step and the pipelined figure (every rank: both contain the exchange)
    bd = pipe = None
    if not ho or args.with_breakdown:
        bd = breakdown(timed_model, batch)
        if args.streams > 1:
            pipe = pipelined(args.streams, args.steps, min(args.warmup, 20))

    # ---- parity gate in the same run ---------------------------------------------------------
    # C2/C3 (host-generated inputs = the reference benchmark's): this rank's folds against the
    # reference digests (tests/golden/g6_digest.npz); every workload: one fold recomputed from
    # scratch the naive way at full size (direct_fold_check).
    parity = "not checked"
    if True:
        notes, ok = [], True
        results = [r for r in ((out, args.path), (other_out, "two_stage" if args.path == "sweep" else "sweep"))
                   if r[0] is not None]
        if not args.rows and not device_data and args.workload in ("C2", "C3") and strong:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import parity_cases as pc
                from conftest import load_npz

                z = load_npz("g6_digest.npz")
                checked = []
                for res, _name in results:
                    (bx, by), bst = res
                    for i, f in enumerate(keys):
                        if f in (0, 4, 9) or lay_world > 1:
                            st = tuple(None if s is None else s[i] for s in bst)
                            pc.check_digest(z, args.workload.lower(), int(f), bx[i], by[i], st, 1e-10)
                            checked.append(int(f))
                notes.append(f"folds {sorted(set(checked))} of this rank within 1e-10 norm-wise of the "
                             f"reference digests ({' and '.join(n for _, n in results)})")
            except AssertionError as e:  # pragma: no cover
                ok = False
                notes.append(f"digest FAILED on rank {rank}: {e}")
        # full-size property check on the last fold of rank 0 (a collective when world > 1)
        if (not ho or args.brief) and (mode == "row_sharded" or rank == 0):
            # (replicated: rank 0 holds every row and checks alone, nothing collective)
            cw = world if mode == "row_sharded" else 1
            try:
                got = None
                if rank == 0 and out is not None:
                    (bx, by), bst = out
                    got = (bx[-1], by[-1], None if bst[0] is None else bst[0][-1],
                           None if bst[1] is None else bst[1][-1])
                vloc = torch.from_numpy(np.asarray(fold_lists[-1])).to(dev) if (rank == 0 and n_mine) else None
                errs = direct_fold_check(torch, dist, cw, Xd, Yd, wd, vloc, 0, rank, 1, flags, got, dev,
                                         yardstick=(es == 4))
                if es == 8:
                    bound = bound_y = 1e-10
                elif len(errs) > 4:
                    # float32: at most 2x the error the reference's algorithm
                    # makes in plain float32 on the same problem (measured here, fp32_algorithm_error)
                    bound, bound_y = fp32_bound(errs[4]), fp32_bound(errs[5])      # (+ two float32 roundings: cvmatrix_amd/fp32_gate.py)
                else:
                    bound = bound_y = 1e-3              # (several ranks: SURVEY 8d's fixed allowance)
                good = errs[0] <= bound and errs[1] <= bound_y and all(e <= max(bound, 1e-10) for e in errs[2:4])
                ok = ok and good
                notes.append(f"fold {keys[-1] if n_mine and rank == 0 else '?'} vs a from-scratch float64 "
                             f"computation at full size: XTX {errs[0]:.1e}, XTY {errs[1]:.1e}, mean {errs[2]:.1e}, "
                             f"std {errs[3]:.1e} (bounds {bound:.1e} / {bound_y:.1e}"
                             + (f" = 2x the float32 restatement's own error {errs[4]:.1e} / {errs[5]:.1e}" if len(errs) > 4 else "")
                             + f"){'' if good else ' FAILED'}")
            except Exception as e:  # noqa: BLE001  pragma: no cover
                ok = False
                notes.append(f"direct check raised: {e!r}")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity = ("ok: " if int(flag.item()) else "FAILED: ") + "; ".join(notes)

    # the per-fold loop must give the batched sweep's bits
    per_fold_same = None
    if loop_out is not None and out is not None and args.path == "sweep":
        (bx, by), _ = out
        per_fold_same = all(torch.equal(loop_out[i][0][0], bx[i]) and torch.equal(loop_out[i][0][1], by[i])
                            for i in range(n_mine))
    result = None
    if rank == 0:
        value = total_folds_per_step * args.steps / elapsed
        n_val = np.diff(batch.host_offsets).astype(np.float64) if batch is not None else np.zeros(0)
        # algorithmic flops of one fold-stage Gram launch ON THIS RANK.  SURVEY.md 8(d) gives two
        # conventions; `achieved` uses the smaller, symmetric one (what has to be computed:
        # upper triangle of XTX + XTY), the dense one (what the reference's dgemm executes,
        # F = 2 n K (K+M)) is reported next to it.
        f_tri = float((n_val * (K * (K + 1) + 2.0 * K * M)).sum())
        f_dense = float((2.0 * n_val * K * (K + M)).sum())
        b_alg = float((es * n_val * (K + M + 1) + 8 * n_val).sum() + 2.0 * es * K * (K + M) * n_mine)
        gram_ms = ms_fold.value / max(n_fold.value, 1)
        fms, fn = (ms_fit2.value, n_fit2.value) if n_fit2.value else (ms_fit.value, n_fit.value)
        fit_gram_ms = fms / fn if fn else float("nan")
        two_gram_ms = ms_fold2.value / max(n_fold2.value, 1) if n_fold2.value else float("nan")
        peak = PEAK_TFLOPS[dtype]
        achieved = f_tri / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else float("nan")
        info = (C.c_int64 * 8)()
        fl = 0x3F
        executed = float("nan")
        if n_mine:
            lib.cvm_plan_fold(n_mine, int(n_val.max()), K, M, _lib.CVM_F64 if es == 8 else _lib.CVM_F32,
                              fl, C.c_size_t(1 << 40), info)
            executed = float(info[5]) * 2048.0 * float(np.ceil(n_val / 4.0).sum())
        traffic = traffic_src = None
        if world == 1 and not args.rows and not ho and not args.no_live_traffic:
            live = measure_hbm_traffic(args.workload, args.path)
            if live is not None:
                traffic = round(live[0])
                traffic_src = (f"measured in this run: child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` over "
                               f"`bench.py --headline-only --steps 5 --warmup 2` ({live[3]} launches of the gathered "
                               f"wgram4_kernel; fetch {live[1] / 1e6:.0f} MB with the gfx950 x2 correction, write "
                               f"{live[2] / 1e6:.0f} MB)")
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if traffic is None and os.path.exists(tpath) and world == 1 and not args.rows:
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("fold_gram_bytes_per_launch")
            if traffic:
                traffic_src = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "`bench.py --headline-only` (not re-measured in this run)")
        n_rows_local = int(Xd.shape[0])
        roofline = {
            "kernel": "wgram4_kernel<T,WEIGHTED,GATHER,FUSED> (gather + weighted Gram of this rank's folds, "
                      "1 launch/step; in the sweep path the same launch also yields the full-data matrices)",
            "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_source": traffic_src,
            "flops_per_launch": f_tri, "flops_convention": "n*K*(K+1) + 2*n*K*M per fold (symmetric)",
            "achieved_dense_convention": round(f_dense / (gram_ms * 1e-3) / 1e12, 3) if gram_ms > 0 else None,
            "mfma_executed_tflops": round(executed / (gram_ms * 1e-3) / 1e12, 3) if gram_ms > 0 else None,
            "avg_launch_ms": round(gram_ms, 4), "launches_timed": int(n_fold.value),
            "two_stage_fit_gram_avg_launch_ms": round(fit_gram_ms, 4),
            "two_stage_fit_gram_achieved": round((n_rows_local * (K * (K + 1) + 2.0 * K * M)) / (fit_gram_ms * 1e-3) / 1e12, 3),
            "two_stage_fold_gram_avg_launch_ms": round(two_gram_ms, 4),
            "algorithmic_hbm_bytes_per_launch": b_alg,
            "hbm_frac_if_bytes_bound": round(b_alg / (gram_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if gram_ms > 0 else None,
        }
        if clock:
            # (the peak is quoted at the nominal 2400 MHz; what the silicon gave this launch is peak x clock / 2400)
            roofline.update(clock)
            roofline["frac_of_clock_adjusted_peak"] = round(achieved / (peak * clock["effective_clock_mhz"] / 2400.0), 4)
            roofline["clock_source"] = ("cvm_clock_probe: s_memtime / s_memrealtime x 100 MHz around each workgroup's life in the "
                                        "LAST timed launch of the product kernel, median over its workgroups (calibration: "
                                        "tools/mfma_peak.hip reads 64.00 / 32.00 cycles per float64 / float32 MFMA with the same "
                                        "stamps; profiles/r6/mfma_peak_clock_calibration.txt); cu_time_used = sum of workgroup "
                                        "lifetimes / (workgroups x launch span)")
        roofline = {k: (None if isinstance(v, float) and v != v else v) for k, v in roofline.items()}

        # the reference's whole benchmark protocol (benchmarks/benchmark.py:101-158, 293-308):
        # constructor + Partitioner + fit from HOST arrays + one call per fold, wall time of one
        # cold pass -- the host->device copy of X, Y, weights is inside
        proto = None
        if world == 1 and not ho and X is not None:
            def proto_run(batched):
                a = time.perf_counter()
                m_ = CVMatrix(*flags, ddof=1, dtype=dtype, copy=True)
                p_ = Partitioner(np.arange(N) % P)
                m_.fit(X, Y, w if weighted else None)
                if batched:
                    r_ = m_.training_XTX_XTY_batched(p_)
                else:
                    r_ = [m_.training_XTX_XTY(p_.get_validation_indices(f)) for f in p_.folds_dict]
                torch.cuda.synchronize()
                del r_
                return time.perf_counter() - a
            proto_run(False)
            tl = sorted(proto_run(False) for _ in range(3))
            tb = sorted(proto_run(True) for _ in range(3))
            proto = {"what": "ctor + Partitioner + fit(host arrays, copy=True) + training_XTX_XTY per fold, "
                             "median of 3 cold passes (host->device copy of X inside)",
                     "loop_folds_per_s": round(P / tl[1], 1), "loop_s": round(tl[1], 5),
                     "batched_folds_per_s": round(P / tb[1], 1), "batched_s": round(tb[1], 5)}

        # supplementary, HBM-bound regime (BASELINE.md section 2 "C5-hbm"): K=4096, M=1,
        # float32, folds of 16 rows -> the direct small-fold kernels; and leave-one-out at
        # the reference's published shape (N=1e5, K=500, M=10; benchmarks/README.md:11-20)
        supp = None
        if world == 1 and not args.rows and not ho:
            supp = {}

            def back_to_back(call, reps=8, samples=3, warm_s=0.04):
                """ms per call, measured like the headline: >= 30 ms of the same work first (the GPU's
                clocks and the memory system's state), then `reps` calls between ONE pair of events."""
                a_ = time.perf_counter()
                while time.perf_counter() - a_ < warm_s:
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tl_ = []
                for _ in range(samples):
                    e0.record()
                    for _k in range(reps):
                        call()
                    e1.record()
                    torch.cuda.synchronize()
                    tl_.append(e0.elapsed_time(e1) / reps)
                return float(np.median(tl_))

            stream_ptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for name, (n_, k_, m_, nv_, nf_, dt_) in {
                "C5-hbm (K=4096,M=1,f32,n_val=16)": (20000, 4096, 1, 16, 48, np.float32),
                "K=4096,M=1,f64,n_val=16": (20000, 4096, 1, 16, 48, np.float64),
                "LOOCV (K=500,M=10,f64,n_val=1)": (100000, 500, 10, 1, 2000, np.float64),
            }.items():
                tt = torch.float64 if dt_ is np.float64 else torch.float32
                gen = torch.Generator(device=dev); gen.manual_seed(1)
                Xs = torch.rand((n_, k_), dtype=tt, device=dev, generator=gen)
                Ys = torch.rand((n_, m_), dtype=tt, device=dev, generator=gen)
                ws_ = torch.rand((n_,), dtype=tt, device=dev, generator=gen)
                ms_ = CVMatrix(dtype=dt_, copy=False, device=dev, lazy_fit=False)
                ms_.fit(Xs, Ys, ws_)
                bs_ = ms_.prepare_folds([np.arange(i * nv_, (i + 1) * nv_) for i in range(nf_)])

                def call_():
                    o_ = ms_.training_XTX_XTY_batched(bs_)
                    del o_
                ms1 = back_to_back(call_)
                # the call's kernels by the library's own events (4 calls back to back)
                lib.cvm_timing_enable(1)
                for _k in range(4):
                    call_()
                torch.cuda.synchronize()
                kms, kn = (C.c_double * 4)(), (C.c_int64 * 4)()
                lib.cvm_timing_read_kinds(kms, kn)
                lib.cvm_timing_enable(0)
                # what this box's memory system takes when nothing but the outputs is written: one launch of
                # nontemporal 16-byte stores over a buffer of the outputs' size, timed the same way
                sz = np.dtype(dt_).itemsize
                obytes = (nf_ * k_ * (k_ + m_) * sz + 15) // 16 * 16
                probe = torch.empty(obytes, dtype=torch.uint8, device=dev)
                pp_ = C.c_void_p(probe.data_ptr())
                fill_ms = back_to_back(lambda: lib.cvm_fill_probe(pp_, C.c_size_t(obytes), stream_ptr), warm_s=0.02)
                fill_gbs = obytes / fill_ms / 1e6
                del probe
                bts = nf_ * (sz * nv_ * (k_ + m_ + 1) + 8 * nv_ + 2 * sz * k_ * (k_ + m_))
                # what the memory system must move at least when the full-data matrices stay in
                # cache (they are the same for every fold): the outputs + the rows, G and H once
                bts_mem = nf_ * (sz * nv_ * (k_ + m_ + 1) + 8 * nv_ + sz * k_ * (k_ + m_)) + sz * k_ * (k_ + m_)
                # `frac` counts the bytes the memory system has to move (the counters of
                # profiles/r*/small_folds_pmc_summary.json agree with it to a few per cent); the
                # per-fold formula of SURVEY 8(d) also bills a read of G and H per fold, which the
                # caches serve -- its GB/s is reported for reference, without a fraction (it is not
                # a roofline figure: it can exceed the peak)
                ach = bts_mem / ms1 / 1e6
                supp[name] = {"folds": nf_, "ms": round(ms1, 4), "folds_per_s": round(nf_ / ms1 * 1e3, 1),
                              "timing": "median of 3 samples of 8 calls back to back between one pair of events, after "
                                        ">= 40 ms of the same calls (like the headline's steps)",
                              "route": ("res8_apply_kernel (round 6: G resident in the register files of 512 persistent workgroups, both "
                                        "triangles computed directly; float32, K = 2048 or a multiple of 4096, >= 16 folds per workgroup "
                                        "set) behind res_pack_kernel"
                                        if (dt_ is np.float32 and k_ % 4096 == 0 and nv_ <= 16
                                            and nf_ >= 16 and os.environ.get("CVM_RESIDENT", "2") != "0")
                                        else ("small_rows_kernel" if nv_ <= 2 and k_ <= 512 else "small_apply_kernel")),
                              "kernel_ms": {"small_stats_kernel": round(kms[2] / max(kn[2], 1), 4),
                                            "update_kernels": round(kms[3] / max(kn[3], 1), 4),
                                            "what": "the library's events around the statistics kernel and around the update "
                                                    "kernels of a call (operand blocks + resident kernel, or tile / whole-rows kernel "
                                                    "+ XTY panels), mean of 4 calls"},
                              "fill_probe_GBps": round(fill_gbs, 1),
                              "roofline": {"bound": "hbm", "achieved": round(ach, 1),
                                           "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                           "frac": round(ach / PEAK_HBM_GBS, 4),
                                           "frac_of_fill": round(ach / fill_gbs, 4),
                                           "bytes_per_launch": "folds*(s*n*(K+M+1) + 8n + s*K*(K+M)) + s*K*(K+M): outputs and rows "
                                                               "once per fold, G and H once per launch",
                                           "per_fold_formula_GBps": round(bts / ms1 / 1e6, 1),
                                           "per_fold_formula": "s*n*(K+M+1) + 8n + 2*s*K*(K+M) per fold (bills a read of G,H per fold "
                                                               "that the caches serve: not a roofline fraction)"}}
                del Xs, Ys, ws_, ms_, bs_
        # statistics only (training_statistics): the column-statistics kernel
        # streams the validation rows once -> HBM-bound
        if supp is not None and batch is not None:
            # (on the eager object: after a sweep the lazy one derives the statistics from the
            #  partials it still holds, without touching the rows)
            ms1 = back_to_back(lambda: eager.training_statistics_batched(batch), warm_s=0.02)
            bts = float((es * n_val * (K + M + 1) + 8 * n_val).sum())
            supp[f"training_statistics ({args.workload})"] = {
                "folds": P, "ms": round(ms1, 4), "folds_per_s": round(P / ms1 * 1e3, 1),
                "timing": "8 calls back to back per sample, 3 samples, after 20 ms of the same calls",
                "roofline": {"bound": "hbm", "achieved": round(bts / ms1 / 1e6, 1), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round(bts / ms1 / 1e6 / PEAK_HBM_GBS, 4),
                             "bytes_per_fold": "s*n*(K+M+1) + 8n"}}
        # mid-size folds (between the HBM regime and the headline's ten big folds): the same rows cut
        # into 100, 300 and 1000 folds, the batched fold stage of the eager object (one unit per fold: folds of
        # 1000 and 333 rows finish in the Gram kernel's fused epilogue -- 333 rows is the trough just above the
        # hand-over --, folds of 100 rows take mid_tile_kernel; host.hpp: mid_default_maxn).
        # `frac` = roofline time / measured time, the roofline time being the larger of the algorithmic flops at
        # the MFMA peak and the bytes that MUST move at the HBM peak -- outputs and validation rows once per
        # fold, G and H once per LAUNCH (the accounting of the HBM-regime block above: the caches serve the
        # per-fold re-reads of G, the counters of profiles/r4/mid_tile say so).  SURVEY 8(d)'s per-fold formula
        # also bills a read of G, H per fold; the fraction by that formula is kept beside it under
        # `frac_survey_8d` (round 4 reported it as `frac`: 0.59 at P = 1000 where this accounting says 0.35).
        if supp is not None and batch is not None and args.workload in ("C2", "C3") and X is not None:
            for Pm in (100, 300, 1000):
                nvm = N // Pm
                foldsm = [np.arange(f, N, Pm)[:nvm] for f in range(Pm)]
                bm = eager.prepare_folds(foldsm)
                def callm_():
                    o_ = eager.training_XTX_XTY_batched(bm)
                    del o_
                ms1 = back_to_back(callm_, reps=6)
                fl = Pm * nvm * (K * (K + 1) + 2.0 * K * M)
                bt_8d = Pm * (es * nvm * (K + M + 1) + 8 * nvm + 2.0 * es * K * (K + M))
                bt = Pm * (es * nvm * (K + M + 1) + 8 * nvm + 1.0 * es * K * (K + M)) + 1.0 * es * K * (K + M)
                peak_fl = PEAK_TFLOPS[dtype] * 1e12
                t_fl, t_bt = fl / peak_fl * 1e3, bt / (PEAK_HBM_GBS * 1e9) * 1e3
                t_8d = bt_8d / (PEAK_HBM_GBS * 1e9) * 1e3
                mid_limit = (256 if es == 8 else 320) if K < 768 else ((200 if es == 8 else (320 if K <= 1024 else 256)) if K <= 2048 else 0)
                supp[f"mid-size folds ({args.workload} rows, P={Pm}, n_val={nvm})"] = {
                    "folds": Pm, "ms": round(ms1, 4), "folds_per_s": round(Pm / ms1 * 1e3, 1),
                    "timing": "6 calls back to back per sample, 3 samples, after 40 ms of the same calls",
                    "route": ("mid_tile_kernel (64x64 tiles, several workgroups per CU)"
                              if nvm <= mid_limit else
                              "wgram4_kernel<.., FUSED> (statistics formed inside the launch)"),
                    "roofline": {"bound": "mfma" if t_fl >= t_bt else "hbm", "flops_ms_at_peak": round(t_fl, 4),
                                 "bytes_ms_at_peak": round(t_bt, 4), "frac": round(max(t_fl, t_bt) / ms1, 4),
                                 "flops": "n*(K(K+1) + 2KM) per fold",
                                 "bytes": "folds*(s*n*(K+M+1) + 8n + s*K*(K+M)) + s*K*(K+M): outputs and rows once per "
                                          "fold, G and H once per launch",
                                 "frac_survey_8d": round(max(t_fl, t_8d) / ms1, 4),
                                 "bytes_survey_8d": "s*n*(K+M+1) + 8n + 2*s*K*(K+M) per fold (bills a read of G, H per fold "
                                                    "that the caches serve)"}}
                del bm, foldsm
        # the step after the path: Improved Kernel PLS (20 components) on the
        # training matrices of this workload's folds, where the fold stage left them
        if supp is not None and batch is not None:
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
        # the other BASELINE.json GPU workloads at N=1, each as a child run of this script (`--brief`: the
        # pre-warm, 2 warm-up and 3 timed sweep steps, the Gram launch by the library's events, the
        # in-run parity gate incl. the from-scratch check at full size) -- so that the driver's line
        # carries all four configurations, not C3 alone
        others = None
        if world == 1 and not args.rows and not ho and not args.no_other_workloads and not emu:
            import subprocess

            others = {}
            for wl_ in ("C2", "C4", "C5"):
                if wl_ == args.workload:
                    continue
                # (timed steps enough for the chip's steady state: the round-5 line's 2 + 3 steps read C2 at a clock still
                #  rising and C4 1-3 % under its steady rate)
                st_, wu_ = {"C2": ("100", "30"), "C3": ("100", "30"), "C4": ("8", "3"), "C5": ("6", "2")}[wl_]
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl_, "--steps", st_, "--warmup", wu_,
                       "--brief", "--no-live-traffic"] + (["--device-data"] if wl_ == "C5" else [])
                a_ = time.perf_counter()
                try:
                    r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                    line_ = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
                    j_ = json.loads(line_[-1]) if line_ else None
                except Exception as e:  # noqa: BLE001
                    j_, r_ = None, None
                    others[wl_] = {"error": repr(e)}
                if j_ is not None:
                    rf_ = j_["roofline"]
                    others[wl_] = {"workload": j_["config"]["workload"], "data": j_["data"], "dtype": j_["dtype"],
                                   "steps": j_["steps"], "warmup": j_["warmup"],
                                   "ms_per_step": j_["ms_per_step"], "folds_per_s": j_["value"],
                                   "gram_avg_launch_ms": rf_["avg_launch_ms"], "gram_launches_timed": rf_["launches_timed"],
                                   "gram_flops_per_launch": rf_["flops_per_launch"], "achieved_TFLOPs": rf_["achieved"],
                                   "peak_TFLOPs": rf_["peak"], "frac": rf_["frac"],
                                   "effective_clock_mhz": rf_.get("effective_clock_mhz"),
                                   "frac_of_clock_adjusted_peak": rf_.get("frac_of_clock_adjusted_peak"),
                                   "cu_time_used": rf_.get("cu_time_used"), "parity": j_["parity"],
                                   "wall_s": round(time.perf_counter() - a_, 1)}
                elif wl_ not in others:
                    others[wl_] = {"error": f"rc={r_.returncode}: {(r_.stderr or '')[-300:]}"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not ho:
            from oracle.cvmatrix_oracle import run_cv

            try:
                from threadpoolctl import threadpool_info

                thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
            except Exception:  # noqa: BLE001
                thr = os.cpu_count()
            # a BOUNDED sample of the same workload: the whole workload when one pass takes a few
            # seconds (C2/C3), else the first rows of it (same K, M, P), scaled by rows
            n_cpu = N if (X is not None and N * K * (K + M) <= 6e10) else max(P * 64, int(6e10 / (K * (K + M))))
            if X is not None:
                Xc, Yc, wc = X[:n_cpu], Y[:n_cpu], w[:n_cpu]
            else:
                Xc, Yc, wc = synth(n_cpu, K, M, dtype, 42)
            fc = np.arange(n_cpu) % P
            times = []
            a0 = time.perf_counter()
            while len(times) < 3 or (time.perf_counter() - a0 < 10.0 and len(times) < 12):
                a = time.perf_counter()
                run_cv(Xc, Yc, wc if weighted else None, fc, *flags, ddof=1, dtype=dtype)
                times.append(time.perf_counter() - a)
            cpu_s = float(np.median(times)) * (N / n_cpu)
            # one BLAS thread, one pass: lines up with the reference's published single-thread
            # numbers (benchmarks/README.md:5)
            one_thread = None
            try:
                from threadpoolctl import threadpool_limits

                with threadpool_limits(limits=1):
                    a = time.perf_counter()
                    run_cv(Xc, Yc, wc if weighted else None, fc, *flags, ddof=1, dtype=dtype)
                    one_thread = round(P / ((time.perf_counter() - a) * (N / n_cpu)), 3)
            except Exception:  # noqa: BLE001
                pass
            cpu = {"value": round(P / cpu_s, 3), "unit": "folds/s", "cores": int(thr),
                   "single_thread_value": one_thread,
                   "kind": "port",
                   "sample": (f"the {'full' if n_cpu == N else (f'first {n_cpu} rows of the' if X is not None else f'a host-generated {n_cpu}-row sample of the')} {args.workload} workload "
                              f"(ctor+Partitioner+fit+{P} folds) {len(times)} times, median "
                              f"{float(np.median(times)):.2f} s per pass ({sum(times):.1f} s of CPU work"
                              f"{'' if n_cpu == N else f', scaled by N/{n_cpu} rows'}), NumPy oracle "
                              f"(oracle/cvmatrix_oracle.py) on the host, BLAS threads={thr}, "
                              f"host cores={os.cpu_count()}")}
        Nfmt = f"{N:.0e}".replace("e+0", "e") if N in (100000, 1000000) else str(N)
        if strong:
            base_metric = ("folds/sec (training_XTX_XTY, center+scale) at N=1e5,K=512" if args.workload == "C3"
                           and not args.rows else f"folds/sec (training_XTX_XTY) at N={Nfmt},K={K} [{args.workload}]")
            par = (f"strong scaling: the {P} folds of ONE {N}-row problem dealt over {world} GPU(s) "
                   f"({n_mine} on rank 0), " +
                   ("each rank sweeps its own folds' rows, one all-reduce of [G|H|stats]" if mode == "row_sharded"
                    else "all rows on every rank, rank 0 fits, one broadcast of [G|H|stats]"))
            wl = (f"{args.workload}: N={N}, K={K}, M={M}, {P} folds (arange(N)%P), "
                  f"{'weighted' if weighted else 'unweighted'}, center/scale X,Y={flags[0]}, "
                  f"fit + batched training_XTX_XTY per step ({args.path} path)")
        else:
            base_metric = (f"folds/sec (training_XTX_XTY) at N={world}x{N} rows, K={K}, {world}x{P} folds "
                           f"[{args.workload}, weak scaling]")
            par = f"weak scaling: every GPU its own {N} rows and {P} folds; one all-reduce of [G|H|stats]"
            wl = (f"{args.workload} x {world}: N={N} rows/GPU, K={K}, M={M}, {P} folds/GPU (arange(N)%P), "
                  f"{'weighted' if weighted else 'unweighted'}, center/scale X,Y={flags[0]}, "
                  f"fit + batched training_XTX_XTY per step ({args.path} path)")
        ceiling = P / math.ceil(P / world) if strong else float(world)
        result = {
            "metric": base_metric,
            "value": round(value, 2), "unit": "folds/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64" if es == 8 else "f32",
            "data": "synthetic" + (" (device-generated, block-seeded)" if device_data else
                                   " (default_rng(42), benchmarks/benchmark.py:223-233)"),
            "config": {"workload": wl, "parallelism": par,
                       "inputs": "resident in HBM before the timed region (torch tensors, copy=False); every step's fit() is handed the "
                                 "same unmodified tensors and told so (trust_tensor_versions=True: it does not read the weights "
                                 "back to re-validate them -- not the API's default, which re-reads its inputs like the reference)",
                       "outputs": ("written into buffers the model keeps while shapes repeat (reuse_outputs=True): "
                                   "a step allocates nothing" if reuse else "fresh tensors every call (the API's default)")},
            "scaling_ceiling_vs_1gpu": round(ceiling, 3),
            "fit_ms": round(fit_ms, 4), "fold_stage_ms": round(fold_ms, 4),
            "update_only_folds_per_s": round(total_folds_per_step / (fold_ms * 1e-3), 1),
            ("two_stage_ms_per_step" if args.path == "sweep" else "sweep_ms_per_step"): round(two_ms, 4),
            ("two_stage_folds_per_s" if args.path == "sweep" else "sweep_folds_per_s"):
                round(total_folds_per_step / (two_ms * 1e-3), 1),
            "per_fold_call_ms_per_step": round(loop_ms, 4),
            "per_fold_call_folds_per_s": round(total_folds_per_step / (loop_ms * 1e-3), 1),
            "per_fold_call_identical_to_batched": per_fold_same,
            "per_fold_call_two_stage_folds_per_s": round(total_folds_per_step / (loop2_ms * 1e-3), 1),
            "reference_protocol": proto,
            "step_breakdown": bd, "other_output_policy": alt, "api_defaults": api, "pipelined": pipe,
            "parity": parity, "roofline": roofline, "cpu_baseline": cpu,
            "supplementary_hbm_regime": supp,
            "other_workloads": others,
            "lib": lib.cvm_version().decode(),
        }
        result = {k: (None if isinstance(v, float) and v != v else v) for k, v in result.items()}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
