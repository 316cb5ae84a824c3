#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a `rocprofv3 --pmc ...` run wrote
(rocpd .db or csv output).

    python profiles/pmc_any.py <rocprof_output_dir> [kernel_substring]
"""
import collections
import csv
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row['Dispatch_Id'], row['Counter_Name'])
                per_dispatch[key] += float(row['Counter_Value'])
                names[row['Dispatch_Id']] = row['Kernel_Name']
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    for path in glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True):
        c = sqlite3.connect(path)
        per_dispatch = collections.defaultdict(float)
        names = {}
        for name, disp, ctr, val, du in c.execute(
                'select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
            per_dispatch[(disp, ctr)] += float(val)
            names[disp] = name
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
        for name, du in c.execute('select name, duration from kernels'):
            dur[name].append(du)
    for k in sorted(acc):
        if want not in k:
            continue
        print(k[:110], f'avg {sum(dur[k]) / max(len(dur[k]), 1) / 1e3:.1f} us' if dur[k] else '')
        for ctr in sorted(acc[k]):
            v = acc[k][ctr]
            print(f'    {ctr:40s} n={len(v):3d} avg={sum(v) / len(v):.6g}')


if __name__ == '__main__':
    main()
