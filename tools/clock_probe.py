"""Effective shader clock and per-workgroup busy time of the (persistent) Gram kernel, from the
-DCVM_STAMPS build: every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz)
at its start and end.   python tools/clock_probe.py tools/libcvmhip_stamps.so [fit|fold|sweep]"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVM_SKIP_HASH_CHECK"] = "1"
import cvmatrix_amd._lib as L
L.LIB_PATH = sys.argv[1]
kind = sys.argv[2] if len(sys.argv) > 2 else "sweep"
from cvmatrix_amd import CVMatrix, Partitioner
dev = torch.device("cuda:0")
N, K, M, P = 100000, 512, 16, 10
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
lazy = CVMatrix(copy=False, device=dev, lazy_fit=True); eager = CVMatrix(copy=False, device=dev, lazy_fit=False)
lazy.fit(X, Y, w); eager.fit(X, Y, w)
b = lazy.prepare_folds(Partitioner(np.arange(N) % P))
def step():
    if kind == "sweep": lazy.fit(X, Y, w); return lazy.training_XTX_XTY_batched(b)
    if kind == "fit": return eager.fit(X, Y, w)
    return eager.training_XTX_XTY_batched(b)
lib = L.load()
for _ in range(300): step()
torch.cuda.synchronize()
lib.cvm_timing_enable(1)
for _ in range(50): step()
torch.cuda.synchronize()
a_, b_, na, nb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
lib.cvm_timing_read(C.byref(a_), C.byref(na), C.byref(b_), C.byref(nb)); lib.cvm_timing_enable(0)
ms = (a_.value / max(na.value, 1)) if kind == "fit" else (b_.value / max(nb.value, 1))
buf2 = (C.c_ulonglong * (1024 * 8 * 4))()
lib.cvm_debug_stamps2(buf2)
b2 = np.frombuffer(buf2, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)[:256]
b2 = b2[b2[:, 0, 2] > 0]
cyc, ticks = b2[:, 0, 0], b2[:, 0, 1]
t0 = b2[:, :, 2].min(); t1 = b2[:, :, 3].max()
print(f"{kind} splits={os.environ.get('CVM_FORCE_SPLITS')}: Gram launch {ms:.4f} ms; workgroups {len(b2)}; "
      f"clock {cyc.sum() / ticks.sum() / 10:.3f} GHz; span {(t1 - t0) / 100:.1f} us; busy/WG mean {ticks.mean() / 100:.1f} "
      f"min {ticks.min() / 100:.1f} max {ticks.max() / 100:.1f} us; CU-time used {ticks.sum() / (256 * (t1 - t0)):.3f}")
