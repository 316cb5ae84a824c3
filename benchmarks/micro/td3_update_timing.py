"""TD3 / DDPG update at config 3's shapes: fused (shared/fused.py) against autograd."""
import sys, time, torch
sys.path.insert(0, '.')
from tracktolearn_amd.algorithms.td3 import TD3
from tracktolearn_amd.algorithms.ddpg import DDPG
dev = torch.device('cuda:0')
W, B = 327, 4096
g = torch.Generator().manual_seed(0)
batch = [torch.randn(B, W, generator=g).to(dev), torch.tanh(torch.randn(B, 3, generator=g)).to(dev),
         torch.randn(B, W, generator=g).to(dev), torch.rand(B, generator=g).to(dev),
         (torch.rand(B, generator=g) > 0.2).float().to(dev)]
for cls in (TD3, DDPG):
    for fused in (True, False):
        torch.manual_seed(0)
        alg = cls(W, 3, '1024-1024', n_actors=8, batch_size=B, replay_size=100, rng=None, device=dev)
        alg.use_fused_learner = fused
        for _ in range(6):
            alg.update(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            alg.update(batch)
        torch.cuda.synchronize()
        print(f'{cls.__name__} fused={fused}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per update', flush=True)
