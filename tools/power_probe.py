import subprocess, sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvmatrix_amd import CVMatrix, Partitioner
dev = torch.device("cuda:0")
N, K, M, P = 100000, 512, 16, 10
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
m = CVMatrix(copy=False, device=dev, lazy_fit=False); m.fit(X, Y, w)
b = m.prepare_folds(Partitioner(np.arange(N) % P))
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.time(), "ERR " + str(e)))
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
time.sleep(1.0)
t0 = time.time()
for _ in range(3000):
    m.fit(X, Y, w); o = m.training_XTX_XTY_batched(b); del o
torch.cuda.synchronize()
t1 = time.time()
time.sleep(1.0)
stop = True; t.join()
print("busy window", t0, t1, "ms/step", (t1 - t0) / 3000 * 1e3)
for ts, s in samples:
    print("%.2f %s %s" % (ts - t0, "BUSY" if t0 <= ts <= t1 else "idle", s[:400]))
