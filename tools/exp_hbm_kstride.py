"""HBM regime: does the power-of-two row pitch of K = 4096 cost store rate?  The same 16-row folds at neighbouring K
(multiples of 64 so that the same kernels and tile counts per row apply), TB/s of bytes that must move.
    python tools/exp_hbm_kstride.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_hbm import run, fill

if __name__ == "__main__":
    fill()
    for dt in (np.float32, np.float64):
        for K in (3840, 3968, 4032, 4096, 4160, 4224, 4352):
            nf = int(48 * (4096 / K) ** 2 + 0.5)
            run(f"K={K} M=1 {np.dtype(dt).name} n=16", 20000, K, 1, 16, nf, dt)
