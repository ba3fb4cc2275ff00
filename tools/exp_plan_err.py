import os, sys, subprocess, json
import numpy as np
if len(sys.argv) > 1:
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    import torch, cvmatrix_amd as amd
    from oracle.cvmatrix_oracle import OracleCVMatrix
    K, M = 132, 1
    rng = np.random.default_rng(500 + K + M)
    N = 9000
    X = (rng.random((N, K)) + 0.1).astype(np.float32)
    Y = rng.random((N, M)).astype(np.float32)
    w = rng.random(N).astype(np.float32)
    w[rng.choice(N, 300, replace=False)] = 0
    perm = rng.permutation(N)
    folds = [perm[:2500], perm[2500:4000], perm[4000:]]
    flags = (True,)*4
    m = amd.CVMatrix(*flags, dtype=np.float32)
    o = OracleCVMatrix(*flags, dtype=np.float64)
    m.fit(X, Y, w); o.fit(X.astype(np.float64), Y.astype(np.float64), w.astype(np.float64))
    (bx, by), bst = m.training_XTX_XTY_batched(folds)
    out = []
    for f in range(3):
        (rx, ry), _ = o.training_XTX_XTY(folds[f])
        ey = np.abs(by[f].double().cpu().numpy() - ry).max() / np.abs(ry).max()
        ex = np.abs(bx[f].double().cpu().numpy() - rx).max() / np.abs(rx).max()
        out.append((float(ex), float(ey)))
    print(json.dumps(out))
else:
    for lazy in ("1", "0"):
        for sp in ["", "1,1", "2,2", "7,2", "7,7", "1,4", "1,16", "16,1", "3,2", "2,3"]:
            env = dict(os.environ, CVM_LAZY_FIT=lazy)
            if sp: env["CVM_FORCE_SPLITS"] = sp
            r = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
            print(lazy, sp or "auto", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
