#!/usr/bin/env python3
"""Experiment: how much faster is the state gather when neighbouring rows are
neighbouring streamlines?  Same config-2 workload with the seeds (a) in random
order, (b) sorted by 8^3-voxel brick.  Prints k_state time per step."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(sort, xcd=False):
    env, subject = bench.make_env('cuda:0', 0)
    seeds = env.seeds
    if sort:
        v = np.floor(seeds + 0.5).astype(np.int64) // 8
        key = (v[:, 0] * 64 + v[:, 1]) * 64 + v[:, 2]
        order = np.argsort(key, kind='stable')
        if xcd:
            # rows of block j run on XCD j % 8: give each XCD a contiguous
            # spatial range (16 rows per block)
            n = len(order)
            blocks = order.reshape(-1, 16)
            nb = blocks.shape[0]
            dest = np.empty(nb, dtype=np.int64)
            per = nb // 8
            j = np.arange(nb)
            dest = (j % 8) * per + j // 8
            out = np.empty_like(blocks)
            out[j] = blocks[dest]
            order = out.reshape(-1)
        env.seeds = seeds[order]
    state = env.reset(0, bench.N_ACTOR)
    times = []
    for step in range(16):
        env.profile_begin(4, classes=('state',))
        a = env.scripted_actions(state, step, 1, bench.WOBBLE)
        env.step_device(a)
        state, _ = env.harvest()
        torch.cuda.synchronize()
        ms, n = env.profile_end()['state']
        times.append(round(ms * 1e3))
    print('sorted' if sort else 'random', 'xcd' if xcd else '', times, 'us')


run(False)
run(True)
run(True, True)
