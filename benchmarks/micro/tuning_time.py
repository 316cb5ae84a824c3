import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, bench
for r in (8, 16):
    os.environ['TTL_STATE_RING_CANDIDATES'] = str(r)
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.reset(0, bench.N_ACTOR)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flat = [x for row in env._sh_tuned for x in row]
    print(r, 'rings: first large reset', round(dt * 1e3), 'ms;', len(flat), 'pairs, best', min(flat), 'worst', max(flat), 'peak GiB', round(torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30, 1), flush=True)
    del env
