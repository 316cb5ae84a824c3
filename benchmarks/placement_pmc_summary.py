#!/usr/bin/env python3
"""Splits the per-dispatch counters of a placement_pmc.py run into the fast and
the slow windows.   python benchmarks/placement_pmc_summary.py <rocprof dir> <marker json>"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, marker = sys.argv[1], json.load(open(sys.argv[2]))
    print(json.dumps(marker))
    if 'launches_before' not in marker:
        return
    rows = collections.defaultdict(dict)
    order = {}
    for path in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if 'k_state_dd' not in r['Kernel_Name']:
                    continue
                disp = int(r['Dispatch_Id'])
                rows[disp][r['Counter_Name']] = rows[disp].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    disps = sorted(rows)
    tail = disps[marker['launches_before']:]
    if len(tail) != 52:
        print('unexpected number of gather launches after the marker:', len(tail), 'of', len(disps))
        tail = disps[-52:]
    acc = {'fast': collections.defaultdict(list), 'slow': collections.defaultdict(list)}
    for w, name in enumerate(marker['sequence']):
        for disp in tail[13 * w + 1:13 * w + 13]:        # skip the reset launch
            for k, v in rows[disp].items():
                acc[name][k].append(v)
    for k in sorted(acc['fast']):
        f = sum(acc['fast'][k]) / len(acc['fast'][k])
        s = sum(acc['slow'][k]) / len(acc['slow'][k])
        print(f'{k:40s} fast {f:16.1f}  slow {s:16.1f}  slow/fast {s / f if f else float("nan"):.3f}')


if __name__ == '__main__':
    main()
