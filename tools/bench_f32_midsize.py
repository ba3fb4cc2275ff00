"""Mid-size folds in float32 (one unit per fold: the Gram kernel's fused epilogue; CVM_NO_FUSED=1: the
two-stage route): batched training_XTX_XTY, N rows in P strided folds."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix
dev="cuda"
for (N,K,M,P) in ((100000,512,16,1000),(100000,512,16,3000),(100000,512,16,300),(200000,1024,4,1000)):
    g=torch.Generator(device=dev); g.manual_seed(0)
    X=torch.rand((N,K),dtype=torch.float32,device=dev,generator=g); Y=torch.rand((N,M),dtype=torch.float32,device=dev,generator=g); w=torch.rand((N,),dtype=torch.float32,device=dev,generator=g)
    m=CVMatrix(dtype=np.float32,copy=False,lazy_fit=False); m.fit(X,Y,w)
    b=m.prepare_folds([np.arange(i,N,P) for i in range(P)])
    o=m.training_XTX_XTY_batched(b); del o; torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(5):
        e0.record(); o=m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)); del o
    print(f"fp32 N={N} K={K} M={M} P={P}: {np.median(ts):.3f} ms  {P/np.median(ts)*1e3:.0f} folds/s", flush=True)
