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
    print('kernel durations (ns): median over all launches')
    for n, v in sorted(durs.items()):
        print(f'  {n:24s} n={len(v):4d} median {st.median(v):9.0f}')
    print('gap between the end of one kernel and the start of the next (ns)')
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
        print(f'  {a:20s} -> {b:20s} n={len(v):4d} median {st.median(v):8.0f}  p90 {sorted(v)[int(0.9 * len(v))]:8.0f}')


if __name__ == '__main__':
    main()
