import torch, time, os, sys
dev='cuda:0'
def bench(M,K,N,reps=50):
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); b=torch.randn(N,device=dev)
    for _ in range(5): torch.nn.functional.linear(x,w,b)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): torch.nn.functional.linear(x,w,b)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    return dt*1e6, 2*M*K*N/dt/1e12
print('tunable', os.environ.get('PYTORCH_TUNABLEOP_ENABLED'))
for M in (512,1024,1600,2048,4096,10000,16384):
    for (K,N) in ((615,1024),(1024,1024)):
        us,tf=bench(M,K,N)
        print(f'M={M} K={K} N={N}: {us:.1f} us  {tf:.1f} TFLOP/s', flush=True)
