import sys, time, torch
sys.path.insert(0, '/root/repo')
from tracktolearn_amd.oracles.fused_net import FusedOracleNet
from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
torch.manual_seed(0)
for ff in (2048, 1024, 64):
    for layers in (4, 1):
        m = TransformerOracle(381, 1, 4, layers, 1e-4)
        for layer in m.bert.layers:
            layer.linear1 = torch.nn.Linear(32, ff); layer.linear2 = torch.nn.Linear(ff, 32)
        m = m.cuda().eval()
        net = FusedOracleNet(m)
        for rows in (1024, 4096):
            d = torch.randn(rows, 127, 3, device='cuda') * 0.3
            print(f'ff {ff} layers {layers} rows {rows}: {t(lambda: net(d)):.4f} ms', flush=True)
