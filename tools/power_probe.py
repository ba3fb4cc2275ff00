"""Socket power and clocks (rocm-smi samples every 0.25 s) while one kind of step runs back to
back for a few seconds, next to the Gram kernel's average launch time (hipEvent pairs) and its
executed MFMA rate.  Settles what limits the Gram kernel at the BASELINE shapes: scheduling, the
loop, or the power/clock management.

  python tools/power_probe.py [C3 C3fit C3fold C4 C5 ...]   ->  table on stdout
"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner, _lib  # noqa: E402

SHAPES = {"C3": (100000, 512, 16, 10, torch.float64), "C4": (1000000, 1024, 32, 64, torch.float64),
          "C5": (200000, 4096, 1, 20, torch.float32), "C2": (100000, 512, 16, 10, torch.float64)}
dev = torch.device("cuda:0")
lib = _lib.load()


def sample_smi(samples, stop):
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                                 text=True, timeout=10).stdout
            samples.append((time.time(), out))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), "ERR " + str(e)))
        time.sleep(0.25)


def parse(js):
    import json
    try:
        d = json.loads(js)
        c = d[sorted(d)[0]]
        pw = [float(v) for k, v in c.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))]
        sclk = [v for k, v in c.items() if "sclk" in k.lower()]
        return (pw[0] if pw else float("nan")), (sclk[0] if sclk else "?")
    except Exception:  # noqa: BLE001
        return float("nan"), "?"


def run(name, seconds=4.0):
    base = name[:2]
    N, K, M, P, tdt = SHAPES[base]
    kind = name[2:] or "sweep"
    g = torch.Generator(device=dev); g.manual_seed(0)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
    npdt = np.float64 if tdt == torch.float64 else np.float32
    lazy = CVMatrix(copy=False, device=dev, lazy_fit=True, dtype=npdt)
    eager = CVMatrix(copy=False, device=dev, lazy_fit=False, dtype=npdt)
    lazy.fit(X, Y, w); eager.fit(X, Y, w)
    b = lazy.prepare_folds(Partitioner(np.arange(N) % P))

    def step():
        if kind == "sweep":
            lazy.fit(X, Y, w); o = lazy.training_XTX_XTY_batched(b)
        elif kind == "fit":
            eager.fit(X, Y, w); o = None
        elif kind == "fold":
            o = eager.training_XTX_XTY_batched(b)
        else:
            eager.fit(X, Y, w); o = eager.training_XTX_XTY_batched(b)
        del o

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    a = time.time(); step(); torch.cuda.synchronize(); one = time.time() - a
    reps = min(3500, max(10, int(seconds / max(one, 1e-4))))
    samples, stop = [], [False]
    t = threading.Thread(target=sample_smi, args=(samples, stop)); t.start()
    time.sleep(0.6)
    lib.cvm_timing_enable(1)
    t0 = time.time()
    done = 0
    while done < reps:
        for _ in range(min(200, reps - done)):
            step()
        done += min(200, reps - done)
        if done % 4000 == 0:
            torch.cuda.synchronize()
            # the recorder holds 8192 launches: drain it now and then
    torch.cuda.synchronize()
    t1 = time.time()
    ms_fit, ms_fold, n_fit, n_fold = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib.cvm_timing_read(C.byref(ms_fit), C.byref(n_fit), C.byref(ms_fold), C.byref(n_fold))
    lib.cvm_timing_enable(0)
    time.sleep(0.4)
    stop[0] = True; t.join()
    busy = [parse(s) for ts, s in samples if t0 + 0.3 <= ts <= t1]
    idle = [parse(s) for ts, s in samples if ts < t0 - 0.1]
    pw = [p for p, _ in busy if p == p]
    info = (C.c_int64 * 8)()
    cdt = _lib.CVM_F64 if tdt == torch.float64 else _lib.CVM_F32
    nv = N // P
    lib.cvm_plan_fold(P, nv, K, M, cdt, 0x3F, C.c_size_t(1 << 40), info)
    exec_flops = float(info[5]) * 2048.0 * np.ceil(nv / 4.0) * P      # per pass over all rows
    out = {"name": name, "ms_per_step": (t1 - t0) / reps * 1e3, "reps": reps}
    for lab, ms, n in (("fit_gram", ms_fit, n_fit), ("fold_gram", ms_fold, n_fold)):
        if n.value:
            avg = ms.value / n.value
            out[lab + "_ms"] = avg
            out[lab + "_exec_tflops"] = exec_flops / (avg * 1e-3) / 1e12
    out["power_w_busy"] = (min(pw), float(np.mean(pw)), max(pw)) if pw else None
    out["power_w_idle"] = [p for p, _ in idle][:2]
    out["sclk_busy"] = sorted(set(str(s) for _, s in busy))
    out["gram_duty"] = ((ms_fit.value + ms_fold.value) / ((t1 - t0) * 1e3)) if (n_fit.value + n_fold.value) <= 8192 else None
    del X, Y, w, lazy, eager, b
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or ["C3", "C3fit", "C3fold", "C3two", "C4", "C5"]
    for nm in names:
        r = run(nm)
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
