#!/usr/bin/env python3
"""Is the gather's slow mode a matter of WHICH XCD reads which part of the
volume?  For every env instance: the gather's time with XCD x taking range
(x + rot) & 7 of the processing order, rot = 0..7 (TTL_XCD_ROTATE).

    python benchmarks/placement_probe7.py [instances]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TTL_VOLUME_CANDIDATES'] = '1'
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import rehandle, timed  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    subject = bench.make_subject()
    envs = []
    for i in range(n):
        env = bench.make_env(subject, 'cuda:0', 0)
        env.reset(0, bench.N_ACTOR)
        window(env)
        envs.append(env)
    for i, env in enumerate(envs):
        row = []
        for rot in range(8):
            os.environ['TTL_XCD_ROTATE'] = str(rot)
            rehandle(env)
            row.append(round(timed(env), 4))
        os.environ['TTL_XCD_ROTATE'] = '0'
        rehandle(env)
        print(json.dumps(dict(i=i, ms_by_rotation=row, again_rot0=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
