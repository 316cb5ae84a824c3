#!/usr/bin/env python3
"""Idle time between consecutive kernels of the bench loop, from a
`rocprofv3 --kernel-trace --output-format csv` directory:
    python benchmarks/trace_gaps.py <dir>
Prints, for the steady-state steps (k_state_dd launches of > 200 k rows), the
median duration of each kernel of a step and the median gap before it."""
import csv
import glob
import os
import sys
import statistics as st


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0].split('<')[0]


def main():
    path = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
                     int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)))
    rows.sort()
    gaps, durs = {}, {}
    for prev, cur in zip(rows, rows[1:]):
        if cur[2].startswith('k_') and prev[2].startswith('k_'):
            gaps.setdefault((prev[2], cur[2]), []).append(cur[0] - prev[1])
    for s, e, n, g in rows:
        if n.startswith('k_'):
            durs.setdefault(n, []).append(e - s)
    # GPU busy fraction over runs of kernels separated by less than 1 ms
    busy, span, start, last_end = 0, 0, None, None
    for s0, e0, n, g in rows:
        if not n.startswith('k_'):
            continue
        if last_end is None or s0 - last_end > 1_000_000:
            if start is not None:
                span += last_end - start
            start = s0
        busy += e0 - s0
        last_end = max(last_end or 0, e0)
    if start is not None:
        span += last_end - start
    print(f'kernels busy {busy / 1e6:.2f} ms of {span / 1e6:.2f} ms spanned ({100.0 * busy / max(span, 1):.1f} %)')
    # the same per size class of the step (rows of its k_advance launch)
    classes = {}
    cur = None
    for s0, e0, n, g in rows:
        if n == 'k_scripted_actions':
            cur = [g, s0, 0, e0]
            classes.setdefault('tmp', []).append(cur)
        if cur is not None and n.startswith('k_'):
            cur[2] += e0 - s0
            cur[3] = max(cur[3], e0)
    steps = classes.get('tmp', [])
    for lo, hi in ((131072, 1 << 30), (65536, 131072), (16384, 65536), (4096, 16384), (0, 4096)):
        sel = [(st_[2], nx[1] - st_[1]) for st_, nx in zip(steps, steps[1:])
               if lo <= st_[0] < hi and nx[1] - st_[1] < 1_000_000]
        if sel:
            b = sum(x for x, _ in sel)
            w = sum(y for _, y in sel)
            print(f'  steps of [{lo}, {hi}) rows: {len(sel)} steps, kernels busy {100.0 * b / w:.1f} % '
                  f'of the step-to-step time, {w / len(sel) / 1e3:.1f} us per step')
    print('kernel durations (ns): median over all launches')
    for n, v in sorted(durs.items()):
        print(f'  {n:24s} n={len(v):4d} median {st.median(v):9.0f}')
    print('gap between the end of one kernel and the start of the next (ns)')
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
        print(f'  {a:20s} -> {b:20s} n={len(v):4d} median {st.median(v):8.0f}  p90 {sorted(v)[int(0.9 * len(v))]:8.0f}')


if __name__ == '__main__':
    main()
