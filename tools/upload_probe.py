"""Host -> device upload of the C3 inputs (N=1e5 x K=512 float64 = 410 MB): what the reference protocol's 9-10 ms
are made of.  (a) pageable ndarray -> device, as fit() does it; (b) the same from pinned memory; (c) staging
through two pinned buffers in chunks, the host copy by 1 / 4 / 8 threads; (d) page-locking the user's array."""
import os, sys, time, threading
import numpy as np, torch

N, K = 100000, 512
X = np.random.default_rng(0).random((N, K))
dev = torch.device("cuda")
torch.cuda.init()
x0 = torch.from_numpy(X).to(dev); torch.cuda.synchronize(); del x0


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    return min(ts)


mb = X.nbytes / 1e6
t = best(lambda: torch.from_numpy(X).to(dev))
print(f"(a) pageable ndarray -> device      {t*1e3:7.2f} ms  {mb/t/1e3:6.1f} GB/s")
Xp = torch.from_numpy(X).pin_memory()
t = best(lambda: Xp.to(dev, non_blocking=True))
print(f"(b) pinned tensor -> device         {t*1e3:7.2f} ms  {mb/t/1e3:6.1f} GB/s")
for nthr in (1, 4, 8):
    for chunk_rows in (4096, 16384):
        bufs = [torch.empty((chunk_rows, K), dtype=torch.float64).pin_memory() for _ in range(2)]
        evs = [torch.cuda.Event() for _ in range(2)]
        out = torch.empty((N, K), dtype=torch.float64, device=dev)

        def staged():
            for i, r0 in enumerate(range(0, N, chunk_rows)):
                r1 = min(N, r0 + chunk_rows)
                b = bufs[i & 1]
                if i >= 2:
                    evs[i & 1].synchronize()
                dst = b.numpy()[: r1 - r0]
                if nthr == 1:
                    np.copyto(dst, X[r0:r1])
                else:
                    step = (r1 - r0 + nthr - 1) // nthr
                    th = [threading.Thread(target=np.copyto, args=(dst[j:j + step], X[r0 + j:r0 + min(j + step, r1 - r0)]))
                          for j in range(0, r1 - r0, step)]
                    [x.start() for x in th]; [x.join() for x in th]
                out[r0:r1].copy_(b[: r1 - r0], non_blocking=True)
                evs[i & 1].record()
        t = best(staged, reps=3)
        print(f"(c) staged, {nthr} copy thread(s), {chunk_rows:5d}-row chunks  {t*1e3:7.2f} ms  {mb/t/1e3:6.1f} GB/s")
a = time.perf_counter()
rc = torch.cuda.cudart().cudaHostRegister(X.ctypes.data, X.nbytes, 0)
t = time.perf_counter() - a
print(f"(d) hipHostRegister of the array     {t*1e3:7.2f} ms (rc {rc})")
if int(rc) == 0:
    Xr = torch.from_numpy(X)
    t = best(lambda: Xr.to(dev, non_blocking=True))
    print(f"    registered ndarray -> device     {t*1e3:7.2f} ms  {mb/t/1e3:6.1f} GB/s")
    torch.cuda.cudart().cudaHostUnregister(X.ctypes.data)
