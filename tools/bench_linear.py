import torch, torch.nn.functional as F, os
shapes=[(15744,256,1024),(15744,1024,256),(15744,256,768),(15744,256,256),(15744,512,512),(15744,256,132),(15744,256,80),(268800,256,256),(268800,256,80),(12800,256,1024)]
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for backend in ("default","cublas"):
    if backend=="cublas":
        try: torch.backends.cuda.preferred_blas_library("cublas")
        except Exception as ex: print("cannot switch", ex); break
    print("backend", backend, torch.backends.cuda.preferred_blas_library())
    for M,K,N in shapes:
        x=torch.randn(M,K,device="cuda",dtype=torch.bfloat16,requires_grad=True)
        w=(torch.randn(N,K,device="cuda",dtype=torch.bfloat16)*0.05).requires_grad_(True)
        b=torch.zeros(N,device="cuda",dtype=torch.bfloat16,requires_grad=True)
        gy=torch.randn(M,N,device="cuda",dtype=torch.bfloat16)
        def f():
            y=F.linear(x,w,b)
        def fb():
            y=F.linear(x,w,b); y.backward(gy); x.grad=None; w.grad=None; b.grad=None
        tf,tfb=timeit(f),timeit(fb)
        fl=2*M*K*N
        print(f"  M{M} K{K} N{N}: fwd {tf:7.1f} us ({fl/tf/1e6:5.0f} TF)  fwd+bwd {tfb:7.1f} us ({3*fl/tfb/1e6:5.0f} TF)")
