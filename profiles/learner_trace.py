#!/usr/bin/env python3
"""One SACAuto.update as the GPU saw it: the launches between a
k_build_learner_inputs and the second k_adam_polyak after it, from a rocprofv3
kernel trace of `benchmarks/bench_training.py --config c3` (update-only loop).

    python profiles/learner_trace.py <dir with *_kernel_trace.csv> > profiles/<tag>_learner_update_trace.txt
"""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if 'k_build_learner_inputs' in r['Kernel_Name']]
    i0 = starts[min(20, len(starts) - 1)]          # inside the 30-update timing loop
    seq = []
    for r in rows[i0:]:
        seq.append(r)
        if sum('k_adam_polyak' in x['Kernel_Name'] for x in seq) == 2:
            break
    t0 = prev = int(seq[0]['Start_Timestamp'])
    tot = gemm = gaps = 0
    print('  start_us   gap_us   dur_us  kernel')
    for r in seq:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('void (anonymous namespace)::', '') \
            .replace('(anonymous namespace)::', '')
        if name.startswith('Cijk'):
            gemm += e - s
            name = name[:14] + ' ... ' + name.split('UserArgs_')[1][:16] + ' (hipBLASLt fp32 GEMM)'
        print(f'{(s - t0) / 1e3:10.1f} {(s - prev) / 1e3:8.1f} {(e - s) / 1e3:8.1f}  {name[:100]}')
        tot += e - s
        gaps += s - prev
        prev = e
    print(f'launches {len(seq)}, kernel time {tot / 1e3:.1f} us (GEMMs {gemm / 1e3:.1f}, others '
          f'{(tot - gemm) / 1e3:.1f}), gaps {gaps / 1e3:.1f} us, span {(prev - t0) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
