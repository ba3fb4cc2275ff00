"""The shader clock the chip holds UNDER the shipped Gram kernel, read by the shipped kernel itself
(`cvm_clock_probe`: two scalar clock pairs per workgroup lifetime, nothing in the item loop), next to
`rocm-smi`'s sclk / socket power sampled DURING the same >= 3 s of back-to-back launches, the launch's
duration by the library's events, and what both say about the roofline fraction:

    frac               = algorithmic TFLOP/s / nominal peak (78.6 float64, 157.3 float32: 2.4 GHz)
    frac_clock_adjusted = algorithmic TFLOP/s / (nominal peak x effective clock / 2400 MHz)

  python tools/clock_product.py [C3 C3fit C3fold C2 C4 C5 ...] [--seconds 3]

(DESIGN 4.1 said 2.37-2.40 GHz, DESIGN 7 / tools/README 2.19-2.27, the round-5 diagnostic
build 1.94, rocm-smi 2.40 -- none of them read in the product build.)"""
import ctypes as C
import json
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

SHAPES = {"C3": (100000, 512, 16, 10, torch.float64, True), "C4": (1000000, 1024, 32, 64, torch.float64, True),
          "C5": (200000, 4096, 1, 20, torch.float32, True), "C2": (100000, 512, 16, 10, torch.float64, False)}
PEAK = {torch.float64: 78.6, torch.float32: 157.3}
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
        time.sleep(0.2)


def parse_smi(js):
    try:
        d = json.loads(js)
        c = d[sorted(d)[0]]
        pw = [float(v) for k, v in c.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))]
        sclk = [int(re.sub(r"[^0-9]", "", str(v))) for k, v in c.items() if "sclk" in k.lower() and re.search(r"[0-9]", str(v))]
        return (pw[0] if pw else float("nan")), (sclk[0] if sclk else -1)
    except Exception:  # noqa: BLE001
        return float("nan"), -1


def read_probe(buf):
    """median / min / max clock (MHz) and duration (us) over the workgroups of the last probed launch"""
    s = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4)
    s = s[(s[:, 3] > s[:, 1]) & (s[:, 1] > 0)]
    if not len(s):
        return None
    dc = (s[:, 2] - s[:, 0]).astype(np.float64)
    dq = (s[:, 3] - s[:, 1]).astype(np.float64)
    mhz = dc / dq * 100.0
    span = (s[:, 3].max() - s[:, 1].min()) / 100.0
    return {"workgroups": int(len(s)), "clock_mhz_median": round(float(np.median(mhz)), 1),
            "clock_mhz_min": round(float(mhz.min()), 1), "clock_mhz_max": round(float(mhz.max()), 1),
            "clock_mhz_time_weighted": round(float(dc.sum() / dq.sum() * 100.0), 1),
            "wg_life_us_mean": round(float(dq.mean() / 100.0), 1), "wg_life_us_max": round(float(dq.max() / 100.0), 1),
            "launch_span_us": round(float(span), 1),
            "cu_time_used": round(float(dq.sum() / (len(s) * (s[:, 3].max() - s[:, 1].min()))), 4)}


def run(name, seconds):
    base = name[:2]
    N, K, M, P, tdt, weighted = SHAPES[base]
    kind = name[2:] or "sweep"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g) if weighted else None
    npdt = np.float64 if tdt == torch.float64 else np.float32
    fl = (True,) * 4 if weighted else (False,) * 4
    kw = dict(dtype=npdt, copy=False, device=dev, reuse_outputs=True, trust_tensor_versions=True)
    lazy = CVMatrix(*fl, lazy_fit=True, **kw)
    eager = CVMatrix(*fl, lazy_fit=False, **kw)
    lazy.fit(X, Y, w)
    eager.fit(X, Y, w)
    b = lazy.prepare_folds(Partitioner(np.arange(N) % P))

    def step():
        if kind == "sweep":
            lazy.fit(X, Y, w)
            lazy.training_XTX_XTY_batched(b)
        elif kind == "fit":
            eager.fit(X, Y, w)
        else:
            eager.training_XTX_XTY_batched(b)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    a = time.time()
    step()
    torch.cuda.synchronize()
    one = time.time() - a
    reps = min(20000, max(10, int(seconds / max(one, 1e-4))))
    probe = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
    _lib.check(lib.cvm_clock_probe(probe.data_ptr(), probe.numel() * 8), "cvm_clock_probe")
    samples, stop = [], [False]
    th = threading.Thread(target=sample_smi, args=(samples, stop))
    th.start()
    try:
        # the first two thirds bring the chip to its steady state; the events cover the last third
        for _ in range(2 * reps // 3):
            step()
        torch.cuda.synchronize()
        lib.cvm_timing_enable(1)
        t0 = time.time()
        n3 = max(reps // 3, 5)
        for _ in range(n3):
            step()
        torch.cuda.synchronize()
        t1 = time.time()
    finally:
        stop[0] = True
        th.join()
        lib.cvm_clock_probe(None, 0)
    ms4, n4 = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.cvm_timing_read_kinds(ms4, n4)
    lib.cvm_timing_enable(0)
    k = 0 if kind == "fit" else 1
    gram_ms = ms4[k] / max(n4[k], 1)
    pr = read_probe(probe)
    if RAW_DIR:
        os.makedirs(RAW_DIR, exist_ok=True)
        np.save(os.path.join(RAW_DIR, f"stamps_{name}.npy"), probe.cpu().numpy().reshape(-1, 4))
    rows = float(N)
    flops = rows * (K * (K + 1) + 2.0 * K * M)
    tf = flops / (gram_ms * 1e-3) / 1e12
    busy = [parse_smi(js) for t, js in samples if t0 - 0.5 * (t1 - t0) <= t <= t1]
    sclk = sorted(v for _, v in busy if v > 0)
    pw = sorted(p for p, _ in busy if p == p)
    out = {"name": name, "reps": reps, "ms_per_step": round((t1 - t0) / n3 * 1e3, 4), "gram_launch_ms": round(gram_ms, 4),
           "gram_tflops_symmetric": round(tf, 2), "frac_of_nominal_peak": round(tf / PEAK[tdt], 4),
           "probe": pr,
           "rocm_smi_sclk_mhz": (sclk[0], sclk[len(sclk) // 2], sclk[-1]) if sclk else None,
           "rocm_smi_power_w": (pw[0], pw[len(pw) // 2], pw[-1]) if pw else None,
           "lib": lib.cvm_version().decode()}
    if pr:
        out["frac_of_clock_adjusted_peak"] = round(tf / (PEAK[tdt] * pr["clock_mhz_median"] / 2400.0), 4)
    print(json.dumps(out), flush=True)
    del X, Y, w, lazy, eager, b
    torch.cuda.empty_cache()
    return out


RAW_DIR = None

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--raw" in sys.argv:
        RAW_DIR = sys.argv[sys.argv.index("--raw") + 1]
        args = [a for a in args if a != RAW_DIR]
    secs = 3.0
    if "--seconds" in sys.argv:
        secs = float(sys.argv[sys.argv.index("--seconds") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--seconds") + 1]]
    for nm in (args or ["C3", "C3fit", "C3fold", "C2", "C4", "C5"]):
        run(nm, secs)
