"""SACAuto.update at config 3's shapes with hipBLASLt's default pick per GEMM
against PyTorch TunableOp's pick (every hipBLASLt / rocBLAS solution timed per
shape): how much GEMM time the library's heuristic leaves on the table."""
import json, os, sys, time, torch
sys.path.insert(0, '.')
from tracktolearn_amd.algorithms.sac_auto import SACAuto
dev = torch.device('cuda:0')
W, B = 327, 4096
g = torch.Generator().manual_seed(0)
batch = [torch.randn(B, W, generator=g).to(dev), torch.tanh(torch.randn(B, 3, generator=g)).to(dev),
         torch.randn(B, W, generator=g).to(dev), torch.rand(B, generator=g).to(dev),
         (torch.rand(B, generator=g) > 0.2).float().to(dev)]
torch.manual_seed(0)
alg = SACAuto(W, 3, '1024-1024', n_actors=8, batch_size=B, replay_size=100, rng=None, device=dev)


def timed(n=60):
    for _ in range(6):
        alg.update(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        alg.update(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'default_ms': timed()}
import torch.cuda.tunable as tn
tn.set_filename('gpurun_out/r04/tunableop_results.csv')
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(30)
tn.set_max_tuning_iterations(30)
t0 = time.perf_counter()
for _ in range(2):
    alg.update(batch)
torch.cuda.synchronize()
out['tuning_s'] = time.perf_counter() - t0
tn.tuning_enable(False)
out['tuned_ms'] = timed()
out['results'] = [list(map(str, r)) for r in tn.get_results()]
tn.enable(False)
out['default_again_ms'] = timed()
print(json.dumps(out))
