"""fp32 GEMM rates of the shapes one fused SACAuto.update issues (config 3:
W = 327, hidden 1024-1024, batch 4096), through the torch entry points the
fused learner uses (mm / addmm / _addmm_activation with out=).  One line per
shape: microseconds, TFLOP/s.  GPU only."""
import sys
import time

import torch

dev = 'cuda:0'
torch.manual_seed(0)


def t(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def row(name, flop, f):
    dt = t(f)
    print(f'{name:58s} {dt * 1e6:8.1f} us {flop / dt / 1e12:7.1f} TF', flush=True)


def fwd(M, K, N, ldx=None):
    """relu(x @ w.T + b): bias+ReLU epilogue into a preallocated output."""
    ldx = ldx or K
    xs = torch.randn(M, ldx, device=dev)
    x = xs[:, :K]
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    row(f'fwd  relu(x[{M}x{K}|ld{ldx}] w[{N}x{K}]^T + b)', 2 * M * K * N,
        lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False, out=out))
    row(f'fwd  addmm same', 2 * M * K * N, lambda: torch.addmm(b, x, w.t(), out=out))


def dgrad(M, N, K, ldz=None):
    """dz[M x N] @ w[N x K] -> [M x K]"""
    ldz = ldz or N
    dzs = torch.randn(M, ldz, device=dev)
    dz = dzs[:, :N]
    w = torch.randn(N, K, device=dev)
    out = torch.empty(M, K, device=dev)
    row(f'dgrad dz[{M}x{N}|ld{ldz}] w[{N}x{K}]', 2 * M * K * N, lambda: torch.mm(dz, w, out=out))


def wgrad(M, N, K, ldz=None, lda=None):
    """dz[M x N]^T @ a[M x K] -> [N x K]"""
    ldz, lda = ldz or N, lda or K
    dz = torch.randn(M, ldz, device=dev)[:, :N]
    a = torch.randn(M, lda, device=dev)[:, :K]
    out = torch.empty(N, K, device=dev)
    row(f'wgrad dz[{M}x{N}|ld{ldz}]^T a[{M}x{K}|ld{lda}]', 2 * M * K * N,
        lambda: torch.mm(dz.t(), a, out=out))


def bmm2(M, K, N):
    x = torch.randn(2, M, K, device=dev)
    w = torch.randn(2, N, K, device=dev)
    out = torch.empty(2, M, N, device=dev)
    row(f'bmm2 x[2x{M}x{K}] w[2x{N}x{K}]^T', 4 * M * K * N,
        lambda: torch.bmm(x, w.transpose(1, 2), out=out))
    # one strided batch over the two halves of a [M x 2K] activation
    xs = torch.randn(M, 2 * K, device=dev)
    xv = xs.view(M, 2, K).transpose(0, 1)
    row(f'bmm2 strided x[{M}x(2x{K})] w[2x{N}x{K}]^T', 4 * M * K * N,
        lambda: torch.bmm(xv, w.transpose(1, 2), out=out))


which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'fwd'):
    for M in (4096, 8192):
        for K in (327, 328, 330, 332, 336, 352):
            fwd(M, K, 1024)
    fwd(4096, 327, 1024, ldx=328)
    fwd(8192, 330, 2048)
    fwd(8192, 330, 2048, ldx=332)
    fwd(8192, 336, 2048)
    fwd(4096, 330, 2048)
    fwd(4096, 1024, 1024)
    fwd(8192, 1024, 1024)
    fwd(8192, 1024, 1024, ldx=2048)
    fwd(12288, 1024, 1024)
    fwd(4096, 1024, 6)
    fwd(4096, 1024, 1)
if which in ('all', 'bwd'):
    dgrad(4096, 1024, 1024)
    dgrad(8192, 1024, 1024)
    dgrad(8192, 1024, 1024, ldz=2048)
    dgrad(4096, 6, 1024)
    dgrad(4096, 1024, 3)
    wgrad(4096, 1024, 1024)
    wgrad(4096, 1024, 1024, ldz=2048, lda=2048)
    wgrad(4096, 1024, 327)
    wgrad(4096, 1024, 328)
    wgrad(4096, 1024, 336)
    wgrad(4096, 2048, 330)
    wgrad(4096, 2048, 332)
    wgrad(4096, 2048, 336)
    wgrad(4096, 6, 1024)
if which in ('all', 'bmm'):
    bmm2(4096, 1024, 1024)
    bmm2(8192, 1024, 1024)
