import torch, time
dev='cuda'
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for lib in ('hipblaslt','rocblas'):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print('cannot set', lib, ex); continue
    for dt in (torch.float32, torch.bfloat16):
        for (M,K,N) in [(200,256,256),(200,256,2048),(200,2048,256),(200,256,768),(43008,256,288),(43008,256,256),(43008,256,1024),(43008,1024,256),(32768,256,512)]:
            x=torch.randn(M,K,device=dev,dtype=dt); w=torch.randn(N,K,device=dev,dtype=dt); b=torch.randn(N,device=dev,dtype=dt)
            t=bench(lambda: torch.nn.functional.linear(x,w,b))
            print(lib, dt, (M,K,N), '%.1f us'%t, '%.1f TF'%(2*M*K*N/t/1e6), flush=True)
