#!/usr/bin/env python3
"""Summarise rocprofv3 outputs for the judged profiles.

    python profiles/pmc_summary.py <tag> <stats_dir> <fetch_dir> <write_dir> [json_name]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace
--stats), profiles/<tag>_pmc_traffic.txt and profiles/pmc_traffic.json (read
by bench.py to fill roofline.traffic).

Counter handling follows /opt/skills/guides (cdna_hip_programming.md section 7,
MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE come from SEPARATE
--pmc passes (TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2); both are
in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. reads exactly
half of a wide (16 B/lane) read stream, so the read side is doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
The guide calibrates that factor for 16-B-per-lane streams only; k_state's
gather is 16 B per lane, its stores are 4 B per lane (WRITE_SIZE is documented
exact for 16-B stores and float atomics, other widths "uncalibrated").
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'\s*([A-Za-z_0-9:]+(?:<[^()]*>)?)', name)
    return m.group(1).replace(' ', '') if m else name


def search_launches(dirname):
    """State-gather launches of the placement search in the run that wrote
    `dirname` (env.py:_tune_placement: a reset + four steps per pair, before
    anything else): 5 x the entries of `placement_candidates_ms` in the bench
    line the run printed (<dirname>.json).  They are excluded from the per-launch
    averages: many of them rewrite one buffer, which keeps its rows in the caches."""
    try:
        line = json.loads(open(dirname.rstrip('/') + '.json').read().strip().splitlines()[-1])
        tried = line.get('placement_candidates_ms') or line.get('placement_candidates_ms_config4')
        return 5 * sum(len(row) for row in tried or [])
    except (OSError, ValueError, IndexError):
        return 0


def gather_rows_log(dirname):
    """Rows every launch of the state gather really wrote, in launch order
    (`<dirname>.rows`, written by the library under TTL_GATHER_ROWS_LOG: with the
    fused step tail the gather's grid covers the slots of an uncompacted
    processing order, more than its rows).  [] if the pass wrote none."""
    path = dirname.rstrip('/') + '.rows'
    if not os.path.exists(path):
        return []
    return [int(x) for x in open(path).read().split()]


def counters(dirname, counter):
    """{kernel: [(counter value, grid size or -rows), ...]} in dispatch order;
    for the state gather the second entry is MINUS the true row count when the
    pass logged one per launch (see units_of)."""
    path = glob.glob(os.path.join(dirname, '**', '*_counter_collection.csv'),
                     recursive=True)[0]
    per_dispatch = collections.defaultdict(lambda: [0.0, 0, ''])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d = per_dispatch[int(r['Dispatch_Id'])]
            d[0] += float(r['Counter_Value'])
            d[1] = int(r['Grid_Size'])
            d[2] = short(r['Kernel_Name'])
    true_rows = gather_rows_log(dirname)
    n_gathers = sum(1 for d in per_dispatch.values() if d[2].startswith('k_state'))
    if len(true_rows) != n_gathers:
        true_rows = []
    rows = collections.defaultdict(list)
    seen = 0
    for disp in sorted(per_dispatch):
        v, g, k = per_dispatch[disp]
        if k.startswith('k_state'):
            if true_rows:
                g = -true_rows[seen]
            seen += 1
        rows[k].append((v, g))
    skip = search_launches(dirname)
    for k in rows:
        if k.startswith('k_state') and skip and len(rows[k]) > skip:
            rows[k] = rows[k][skip:]
    return rows


def durations_by_size(stats_dir):
    """Average duration of the state gather per launch size, from the kernel
    trace of the --stats pass: the aggregated stats mix the step launches of the
    timed windows (all rows active) with the placement search's launches on
    131 072 rows (env.py:_tune_placement), which pulls their average down."""
    paths = glob.glob(os.path.join(stats_dir, '**', '*_kernel_trace.csv'), recursive=True)
    if not paths:
        return []
    groups = collections.defaultdict(list)
    rows = sorted(csv.DictReader(open(paths[0])), key=lambda r: int(r['Start_Timestamp']))
    skip, seen = search_launches(stats_dir), 0
    true_rows = gather_rows_log(stats_dir)
    if len(true_rows) != sum(1 for r in rows if short(r['Kernel_Name']).startswith('k_state')):
        true_rows = []
    for r in rows:
        k = short(r['Kernel_Name'])
        if not k.startswith('k_state'):
            continue
        seen += 1
        grid = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
        m = re.match(r'k_state(?:_dd)?<(\d+)', k)
        units = grid // 256 * (4 * (64 // int(m.group(1))) if m else 20)
        if true_rows:
            units = true_rows[seen - 1]
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3      # us
        what = 'placement search (the first %d launches)' % skip if seen <= skip else \
            ('steps and resets of the windows (more than 200 000 rows)' if units > 200000
             else 'at most 200 000 rows')
        groups[(k, what)].append((dur, units))
    out = ['state gather by phase, rocprofv3 kernel trace of the --stats pass:']
    for (k, what), rows in sorted(groups.items()):
        out.append(f'{k} | {what} | {len(rows)} launches | '
                   f'{sum(u for _, u in rows) / len(rows):.0f} rows avg | '
                   f'{sum(d for d, _ in rows) / len(rows):.1f} us avg')
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    json_name = sys.argv[5] if len(sys.argv) > 5 else 'pmc_traffic.json'
    stats = glob.glob(os.path.join(stats_dir, '**', '*_kernel_stats.csv'),
                      recursive=True)[0]
    shutil.copy(stats, os.path.join(HERE, f'{tag}_kernel_stats.csv'))
    fetch = counters(fetch_dir, 'FETCH_SIZE')
    write = counters(write_dir, 'WRITE_SIZE')
    lines = ['kernel | launches | FETCH_SIZE KiB avg | WRITE_SIZE KiB avg | '
             'HBM bytes/launch = (2*FETCH + WRITE)*1024']
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith('k_'):
            continue
        f = [v for v, _ in fetch.get(k, [])]
        w = [v for v, _ in write.get(k, [])]
        fa = sum(f) / max(len(f), 1)
        wa = sum(w) / max(len(w), 1)
        hbm = (2 * fa + wa) * 1024
        out[k] = dict(launches=len(f), fetch_kib=fa, write_kib=wa,
                      hbm_bytes_per_launch=hbm)
        lines.append(f'{k} | {len(f)} | {fa:.1f} | {wa:.1f} | {hbm:.4g}')
    state = [k for k in out if k.startswith('k_state')]
    js = {'source': f'profiles/{tag}_pmc_traffic.txt', 'kernels': out}
    if state:
        k = max(state, key=lambda name: out[name]['launches'])
        js['k_state_kernel'] = k
        js['k_state_hbm_bytes_per_launch'] = out[k]['hbm_bytes_per_launch']
        # units (streamlines) per launch from the launch geometry the counter
        # rows carry: a 256-thread workgroup of k_state_dd<LPS,...> serves
        # 4 waves x (64 // LPS) streamlines
        m = re.match(r'k_state(?:_dd)?<(\d+)', k)
        rows_per_block = 4 * (64 // int(m.group(1))) if m else 20
        logged = any(g < 0 for _, g in fetch.get(k, []))
        units = sum(-g if g < 0 else g // 256 * rows_per_block for _, g in fetch.get(k, []))
        total = out[k]['hbm_bytes_per_launch'] * out[k]['launches']
        js['k_state_units_per_launch'] = units / max(out[k]['launches'], 1)
        js['k_state_hbm_bytes_per_unit'] = total / max(units, 1)
        js['k_state_units_source'] = 'rows logged by the library per launch' if logged \
            else 'grid sizes'
        lines.append(f'{k}: {js["k_state_units_per_launch"]:.0f} units/launch (from '
                     f'{"the rows the library logged per launch" if logged else "the grid sizes"}'
                     f') -> {js["k_state_hbm_bytes_per_unit"]:.1f} HBM bytes per unit')
    lines += durations_by_size(stats_dir)
    open(os.path.join(HERE, f'{tag}_pmc_traffic.txt'), 'w').write(
        '\n'.join(lines) + '\n')
    json.dump(js, open(os.path.join(HERE, json_name), 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
