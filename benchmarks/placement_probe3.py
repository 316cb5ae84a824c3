#!/usr/bin/env python3
"""Follow-up of placement_probe.py: the first env instance of a process gathers
faster (0.178 ms) than later ones (0.20 ms).  Which buffer carries that?
(a) a later instance borrows the first one's packed SH volume tensor;
(b) the first instance is destroyed and a new one created, which gets the
    freed blocks back from the caching allocator.

    python benchmarks/placement_probe3.py
"""
import gc
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import rehandle, timed  # noqa: E402


def ptrs(env):
    return dict(sh=hex(env._sh_packed.data_ptr()), hist=hex(env._buf_streamlines.data_ptr()),
                ws=hex(env._buf_ws.data_ptr()), idx=hex(env._buf_idx.data_ptr()))


def main():
    subject = bench.make_subject()
    envs = []
    for i in range(4):
        env = bench.make_env(subject, 'cuda:0', 0)
        env.reset(0, bench.N_ACTOR)
        window(env)
        envs.append(env)
    for i, env in enumerate(envs):
        print(json.dumps(dict(i=i, ms=round(timed(env), 4), **ptrs(env))), flush=True)
    # (a) instance 2 gathers from instance 0's volume
    own = envs[2]._sh_packed
    envs[2]._sh_packed = envs[0]._sh_packed
    rehandle(envs[2])
    print(json.dumps(dict(case='instance 2 on the volume of instance 0', ms=round(timed(envs[2]), 4),
                          **ptrs(envs[2]))), flush=True)
    # and instance 0 from instance 2's volume
    envs[0]._sh_packed = own
    rehandle(envs[0])
    print(json.dumps(dict(case='instance 0 on the volume of instance 2', ms=round(timed(envs[0]), 4),
                          **ptrs(envs[0]))), flush=True)
    for i, env in enumerate(envs):
        print(json.dumps(dict(i=i, ms=round(timed(env), 4), **ptrs(env))), flush=True)
    # (b) which state-row blocks does a window use?
    st = envs[1]._new_state(bench.N_ACTOR)
    st2 = envs[1]._new_state(bench.N_ACTOR)
    print(json.dumps(dict(state_blocks=[hex(st.data_ptr()), hex(st2.data_ptr())])), flush=True)
    del st, st2
    print(torch.cuda.memory_summary(abbreviated=True)[:1500])


if __name__ == '__main__':
    main()
