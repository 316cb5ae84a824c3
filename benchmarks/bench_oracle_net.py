#!/usr/bin/env python3
"""TractOracle-Net forward: the fused kernel (`ttl_oracle_net_forward`) against the
PyTorch-ROCm module under fp16 autocast (what the reference runs), at the batch
sizes the oracle sees in BASELINE config 5.  One JSON line per batch size.

    python benchmarks/bench_oracle_net.py [rows ...]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FP16_MFMA_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: dense f16 / bf16 MFMA


def flop_per_streamline(n_layers, ff, d=32, tokens=128, issued_last_layer_tokens=32):
    """(module FLOP, FLOP the fused kernel issues: its last layer runs the
    attention output / feed-forward of one 32-token tile only)."""
    per_layer = 2 * tokens * d * 3 * d + 2 * 2 * tokens * tokens * d + 2 * tokens * d * d \
        + 2 * 2 * tokens * d * ff
    last = 2 * tokens * d * 3 * d + 2 * 2 * tokens * issued_last_layer_tokens * d \
        + 2 * issued_last_layer_tokens * d * d + 2 * 2 * issued_last_layer_tokens * d * ff
    return n_layers * per_layer, (n_layers - 1) * per_layer + last


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def kernel_name(rows):
    """Which of the two kernels ttl_oracle_net_forward picks (TTL_ORACLE_NET_WG=1 / 0
    force one; default: one workgroup per streamline up to 512 rows)."""
    force = os.environ.get('TTL_ORACLE_NET_WG')
    wg = rows <= 512 if force is None else force != '0'
    return 'workgroup per streamline' if wg else 'wavefront per streamline'


def measure(rows_list, with_module=True, device='cuda'):
    """One dict per batch size (bench.py's `config5.oracle_net`, and the lines this
    script prints)."""
    from tracktolearn_amd.oracles.fused_net import FusedOracleNet
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    torch.manual_seed(0)
    model = TransformerOracle(381, 1, 4, 4, 1e-4).to(device).eval()
    net = FusedOracleNet(model)
    full, issued = flop_per_streamline(4, 2048)
    out = []
    for rows in rows_list:
        dirs = torch.randn(rows, 127, 3, device=device) * 0.3

        def module():
            with torch.no_grad(), torch.autocast('cuda'):
                for lo in range(0, rows, 4096):             # OracleSingleton's batches
                    model(dirs[lo:lo + 4096])
        t_fused = timeit(lambda: net(dirs), 20 if rows <= 4096 else 8)
        line = {
            'rows': rows, 'kernel': kernel_name(rows), 'fused_ms': round(t_fused * 1e3, 4),
            'fused_TFLOPs_issued': round(rows * issued / t_fused / 1e12, 1),
            'fused_frac_of_fp16_mfma_peak': round(rows * issued / t_fused / 1e12 / FP16_MFMA_PEAK_TF, 4),
            'mflop_per_streamline_module': round(full / 1e6, 1),
            'mflop_per_streamline_issued': round(issued / 1e6, 1)}
        if with_module:
            t_mod = timeit(module, 5 if rows <= 4096 else 2)
            line.update(module_autocast_ms=round(t_mod * 1e3, 3),
                        speedup=round(t_mod / t_fused, 1),
                        module_TFLOPs=round(rows * full / t_mod / 1e12, 1))
        out.append(line)
    return out


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 4096, 16384]
    for rows in rows_list:
        print(json.dumps(measure([rows])[0]), flush=True)


if __name__ == '__main__':
    main()
