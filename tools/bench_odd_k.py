"""Shapes whose rows are not whole 16-byte pieces (odd K, odd M in float64): the general kernels
(copy=False: the caller's arrays as they are) against padded private copies (argument `copy`)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from cvmatrix_amd import CVMatrix, Partitioner
dev="cuda"
COPY = len(sys.argv) > 1 and sys.argv[1] == "copy"     # copy=True: private device copies, padded to 16-byte rows
for (N,K,M,P,dt) in ((100000,512,16,10,torch.float64),(100000,511,16,10,torch.float64),(100000,510,15,10,torch.float64),(100000,701,3,10,torch.float64),(100000,702,3,10,torch.float64),(100000,511,16,10,torch.float32),(100000,512,16,10,torch.float32)):
    g=torch.Generator(device=dev); g.manual_seed(0)
    X=torch.rand((N,K),dtype=dt,device=dev,generator=g); Y=torch.rand((N,M),dtype=dt,device=dev,generator=g); w=torch.rand((N,),dtype=dt,device=dev,generator=g)
    m=CVMatrix(dtype=np.float64 if dt==torch.float64 else np.float32,copy=COPY,lazy_fit=True)
    m.fit(X,Y,w)
    b=m.prepare_folds(Partitioner(np.arange(N)%P))
    for _ in range(3):
        m.fit(X,Y,w); o=m.training_XTX_XTY_batched(b); del o
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(20):
        m.fit(X,Y,w)                      # (lazy: stores / copies the inputs; not timed)
        torch.cuda.synchronize()
        e0.record(); o=m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); del o
    ms=float(np.median(ts))
    fl=N*(K*(K+1)+2*K*M)
    print(f"N={N} K={K} M={M} P={P} {dt}: {ms:.3f} ms (sweep + fold stage)  {P/ms*1e3:.0f} folds/s  {fl/ms/1e9:.1f} TFLOP/s(sym)", flush=True)
