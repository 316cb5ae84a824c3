"""What a row count hipBLASLt has not seen costs: the policy's two hidden GEMMs
(W = 327 -> 1024 -> 1024, fp32, bias + ReLU epilogue) at fresh row counts
against repeated ones.  Host + device time per call, synchronised."""
import sys, time, torch
dev = 'cuda:0'
torch.manual_seed(0)
w1, b1 = torch.randn(1024, 327, device=dev) * 0.05, torch.randn(1024, device=dev)
w2, b2 = torch.randn(1024, 1024, device=dev) * 0.03, torch.randn(1024, device=dev)
big = torch.randn(70000, 327, device=dev)


def fwd(n):
    h = torch._addmm_activation(b1, big[:n], w1.t(), use_gelu=False)
    return torch._addmm_activation(b2, h, w2.t(), use_gelu=False)


def t(n, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fwd(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


fwd(1000); fwd(2000)
first = [t(n) for n in range(20011, 20011 + 40 * 97, 97)]          # 40 fresh row counts
again = [t(n) for n in range(20011, 20011 + 40 * 97, 97)]          # the same ones again
print(f'fresh row counts : mean {sum(first) / len(first):.3f} ms, max {max(first):.3f}')
print(f'seen row counts  : mean {sum(again) / len(again):.3f} ms, max {max(again):.3f}')
bucket = [t(-(-n // 512) * 512) for n in range(30011, 30011 + 40 * 97, 97)]
print(f'rounded up to 512: mean {sum(bucket) / len(bucket):.3f} ms (8 distinct shapes over 40 calls)')
