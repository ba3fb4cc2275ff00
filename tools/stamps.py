"""Diagnostic: phase cycle shares of the Gram kernel (build with -DCVM_STAMPS)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = sys.argv[1]
from cvmatrix_amd import CVMatrix, Partitioner
rng = np.random.default_rng(42)
# (STAMP_N / _K / _M / _P / _DTYPE: other shapes -- e.g. the scaled C5: STAMP_N=50000 STAMP_K=4096 STAMP_M=1 STAMP_P=5 STAMP_DTYPE=f32)
N, K, M, P = (int(os.environ.get("STAMP_N", "100000")), int(os.environ.get("STAMP_K", "512")), int(os.environ.get("STAMP_M", "16")),
              int(os.environ.get("STAMP_P", "10")))
dt = np.float32 if os.environ.get("STAMP_DTYPE", "f64") == "f32" else np.float64
X, Y, w = rng.random((N, K), dtype=dt), rng.random((N, M), dtype=dt), rng.random(N, dtype=dt)
if os.environ.get("UNWEIGHTED"): w = None
m = CVMatrix(dtype=dt, lazy_fit=False); m.fit(X, Y, w)
b = m.prepare_folds(Partitioner(np.arange(N) % P))
for _ in range(3): m.training_XTX_XTY_batched(b)
L.load().cvm_timing_enable(1)
for _ in range(5): m.training_XTX_XTY_batched(b)
torch.cuda.synchronize()
_mf, _mo, _nf, _no = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
L.load().cvm_timing_read(C.byref(_mf), C.byref(_nf), C.byref(_mo), C.byref(_no))
print("fold-stage Gram launch ms", _mo.value / max(_no.value, 1), "CVM_DEBUG", os.environ.get("CVM_DEBUG"))
lib = L.load(); print(lib.cvm_version().decode())
buf = (C.c_ulonglong * (1024 * 8 * 4))()
lib.cvm_debug_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)
a = a[:500]
st = np.maximum(a[:, :, 3], 1)
print("workgroups", a.shape[0], "stages/WG", a[:, 0, 3].mean())
for name, i in (("phase A", 0), ("phase B", 1), ("phase C", 2)):
    per = a[:, :, i] / st
    print(f"{name:10s} cycles/stage by wave {np.round(per.mean(0))}")
tot = (a[:, :, 0] + a[:, :, 1] + a[:, :, 2]) / st
print("total cycles/stage by wave", np.round(tot.mean(0)))
print("compute waves 0-3: A=-, B=compute, C=barrier wait; DMA loaders 4-6: A=issue, B=vmcnt wait, C=barrier wait")

buf2 = (C.c_ulonglong * (1024 * 8 * 4))()
lib.cvm_debug_stamps2(buf2)
b2 = np.frombuffer(buf2, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)
b2 = b2[b2[:, 0, 2] > 0]
cyc, ticks = b2[:, 0, 0], b2[:, 0, 1]
print("workgroups seen", len(b2))
print("per-WG (wave 0): shader cycles mean %.0f, 100MHz ticks mean %.1f -> clock %.3f GHz" % (cyc.mean(), ticks.mean(), cyc.mean() / ticks.mean() / 10))
t0 = b2[:, :, 2].min(); t1 = b2[:, :, 3].max()
print("kernel span (first WG start -> last WG end) %.1f us; WG duration mean %.1f us, max %.1f us" % ((t1 - t0) / 100, ticks.mean() / 100, ticks.max() / 100))
st_ = np.sort((b2[:, 0, 2] - t0) / 100)
if len(st_) > 256:
    print("WG start times us: round 1 last %.1f; round 2 first %.1f mean %.1f last %.1f" % (st_[255], st_[256], st_[256:].mean(), st_[-1]))
en_ = (b2[:, 0, 3] - t0) / 100
print("WG end times us: quantiles", np.round(np.quantile(en_, [0, .25, .5, .75, 1]), 1))
buf3 = (C.c_ulonglong * (1024 * 8 * 2))()
lib.cvm_debug_stamps3(buf3)
b3 = np.frombuffer(buf3, dtype=np.uint64).reshape(1024, 8, 2).astype(np.float64)
b3 = b3[b3[:, 0, 0] > 0]
print("compute waves: prologue cycles by wave", np.round(b3[:, :4, 0].mean(0)), "epilogue", np.round(b3[:, :4, 1].mean(0)))
# diagonal vs off-diagonal workgroups (C3 shape: 10 tiles per unit, diagonal ones are 0, 4, 7, 9)
buf2b = np.frombuffer(buf2, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)
n_items = 500
ipx = (n_items + 7) // 8
blk = np.arange(1024)
item = (blk & 7) * ipx + (blk >> 3)
valid = ((blk >> 3) < ipx) & (item < n_items) & (buf2b[:, 0, 2] > 0)
it = item % 10
isd = np.isin(it, [0, 4, 7, 9])
for nm, m_ in (("diagonal", valid & isd), ("off-diagonal", valid & ~isd)):
    c = buf2b[m_, 0, 0]
    print(f"{nm:13s} workgroups {m_.sum():4d}: mean {c.mean():9.0f} shader cycles = {c.mean() / 125:6.0f} per 16-row stage")
