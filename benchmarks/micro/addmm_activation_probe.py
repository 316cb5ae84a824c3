import torch, time
dev='cuda:0'
torch.manual_seed(0)
for M in (512, 4096):
    x=torch.randn(M,615,device=dev); w=torch.randn(1024,615,device=dev)*0.05; b=torch.randn(1024,device=dev)
    ref=torch.relu(torch.nn.functional.linear(x,w,b))
    try:
        out=torch._addmm_activation(b, x, w.t(), use_gelu=False)
    except Exception as e:
        print('unsupported', e); break
    print(M, 'max diff', float((ref-out).abs().max()), 'equal', bool(torch.equal(ref,out)))
    def t(f, n=200):
        for _ in range(20): f()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
    print(M, 'linear+relu us', round(t(lambda: torch.relu(torch.nn.functional.linear(x,w,b))),1), 'addmm_activation us', round(t(lambda: torch._addmm_activation(b,x,w.t(),use_gelu=False)),1))
