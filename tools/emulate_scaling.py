#!/usr/bin/env python3
"""Predicted strong-scaling curve from ONE GPU: `bench.py --emulate-world G` for G = 1, 2, 4, 8
(rank 0 of each job: it owns ceil(P/G) folds, the slowest rank), per workload and injected
collective latency.  Writes the JSON lines and a markdown table.

    python tools/emulate_scaling.py [--workloads C3,C4] [--comm-us 0,30] [--out profiles/r3/emulated_scaling]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(workload, G, comm_us, steps, warmup, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps),
           "--warmup", str(warmup), "--no-cpu-baseline", "--no-live-traffic", *extra]
    if G > 1:
        cmd += ["--emulate-world", str(G), "--comm-us", str(comm_us)]
    else:
        cmd += ["--headline-only", "--with-breakdown"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stderr[-3000:]}")
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="C3,C4")
    ap.add_argument("--comm-us", default="0,30")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "emulated_scaling"))
    ap.add_argument("--mode", default="row_sharded")
    args = ap.parse_args()
    lines, table = [], []
    for wl in args.workloads.split(","):
        steps, warmup = (200, 50) if wl in ("C2", "C3") else (5, 2)
        base = None
        for G in [int(g) for g in args.worlds.split(",")]:
            for cu in ([0.0] if G == 1 else [float(c) for c in args.comm_us.split(",")]):
                ln = run(wl, G, cu, steps, warmup, ["--mode", args.mode] if G > 1 else [])
                ln["_workload"], ln["_G"], ln["_comm_us"] = wl, G, cu
                lines.append(ln)
                if G == 1:
                    step = ln["ms_per_step"]
                    base = step
                    b, pp = ln.get("step_breakdown") or {}, ln.get("pipelined") or {}
                    table.append((wl, G, cu, step, ln["roofline"]["avg_launch_ms"], None, None, b.get("finalize_ms"),
                                  b.get("host_ms"), 1.0, 1.0, pp.get("ms_per_step")))
                else:
                    b, pp = ln["breakdown"], ln.get("pipelined") or {}
                    step = ln["per_rank_step_ms"]
                    table.append((wl, G, cu, step, b.get("gram_ms"), b.get("fit_finalize_ms"), b.get("exchange_ms"),
                                  b.get("fold_finalize_ms"), b.get("host_ms"), base / step,
                                  ln["scaling_ceiling_vs_1gpu"], pp.get("ms_per_step")))
                print(json.dumps(ln), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out + ".json", "w") as f:
        json.dump(lines, f, indent=1)
    hdr = ("| workload | GPUs | comm stub µs | per-rank step ms | gram ms | fit finalize ms | exchange ms | fold finalize ms | "
           "host ms | predicted speed-up | ceiling | pipelined (2 streams) ms |")
    rows = [hdr, "|" + "---|" * 12]
    for t in table:
        rows.append("| " + " | ".join("—" if v is None else (f"{v:.4g}" if isinstance(v, float) else str(v)) for v in t) + " |")
    with open(args.out + ".md", "w") as f:
        f.write("\n".join(rows) + "\n")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
