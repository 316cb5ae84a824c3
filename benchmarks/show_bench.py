#!/usr/bin/env python3
"""Digest of a bench.py JSON line: python benchmarks/show_bench.py <file>"""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if j.get('windows') and j.get('roofline'):
    w, r = j['windows'], j['roofline']
    ep = j.get('whole_episode') or {}
    print(f"value {j['value'] / 1e6:.1f} M (min {w['value_min'] / 1e6:.1f}, max {w['value_max'] / 1e6:.1f}; "
          f"with events {((w.get('value_median_with_events') or 0) / 1e6):.1f}, "
          f"without {((w.get('value_median_without_events') or 0) / 1e6):.1f}) "
          f"k_state {r['avg_launch_ms']:.4f} ms x{r['launches']} frac {r['frac']} "
          f"other {r.get('other_kernels_ms_per_step')} "
          f"whole episode {ep.get('streamline_steps_per_s_rank0', 0) / 1e6:.1f} M")
if j.get('strong'):
    s = j['strong']
    print(f"strong: {s['value'] / 1e6:.1f} M at {s['n_actor_per_gpu']} per GPU (same run: {s['same_run_as_value']})")
if j.get('pipelined_halves'):
    p = j['pipelined_halves']
    print(f"pipelined halves: {p['value'] / 1e6:.1f} M ({p['ms_per_step']:.4f} ms/step, x{p.get('vs_value') or 0:.3f} of value)")
if j.get('config4'):
    c = j['config4']
    e = c['end_to_end']
    print(f"config4: step-only {c['step_only']['value'] / 1e6:.1f} M; end to end {e['value_end_to_end'] / 1e6:.1f} M "
          f"(track {e['track_ms']:.1f} ms + collate {e['collate_ms']:.1f} ms), track only {e['value_step_only'] / 1e6:.1f} M")
if j.get('roofline_hbm_regime'):
    r = j['roofline_hbm_regime']
    print(f"hbm regime: k_state {r['avg_launch_ms']:.4f} ms, {r['units_per_launch']:.0f} units, frac {r['frac']}, "
          f"shard {r.get('value_rank0_shard', 0) / 1e6:.1f} M")
if isinstance(j.get('other_shapes'), dict):
    for name, o in j['other_shapes'].items():
        if 'error' in o:
            print(f"{name}: {o['error']}")
            continue
        print(f"{name}: {o['value'] / 1e6:.1f} M ({o['ms_per_step'] * 1e3:.1f} us/step, k_state {o['k_state_ms']:.4f} ms; {o['loop']})")
def training(name, t):
    if isinstance(t, dict) and 'error' in t:
        print(f'{name}:', t['error'])
    elif isinstance(t, dict):
        g = t.get('graphed_update') or {}
        roof = t.get('roofline') or {}
        ph = t.get('phases_ms_per_step') or {}
        print(f"{name}: {t['train_step_ms']:.2f} ms per step at n_actor {t['n_actor']} "
              f"({t['train_streamline_steps_per_s'] / 1e6:.1f} M streamline-steps/s), update alone "
              f"{t['update_ms']:.2f} ms = {roof.get('achieved', 0):.1f} TF/s = "
              f"{roof.get('frac', 0):.3f} of the fp32 MFMA peak"
              + (f"; graphed update: step {g['train_step_ms']:.2f} ms, update {g['update_ms']:.2f} ms"
                 if g else ''))
        print('   phases (ms): ' + ', '.join(f'{k} {v:.3f}' for k, v in ph.items()))
        pf = t.get('policy_forward')
        if pf:
            print(f"   policy forward: {pf['achieved']:.1f} TF/s = {pf['frac']:.3f} of the fp32 MFMA peak "
                  f"at {pf['rows_per_step']:.0f} rows per step")
        if 'oracle_rows_scored_per_step' in t:
            print(f"   oracle: {t['oracle_rows_scored_per_step']:.0f} rows scored per step in "
                  f"{t['oracle_batches_per_step']:.2f} batches ({t.get('oracle_net')})")
        on = t.get('oracle_net_alone')
        if on:
            print('   oracle network alone: ' + ', '.join(
                f"{b['rows']} rows {b['fused_ms']:.3f} ms" for b in on['per_batch'])
                + f" = {on['roofline']['frac']:.3f} of the fp16 MFMA peak; module "
                f"{on['per_batch'][-1].get('module_autocast_ms')} ms")
        w = t.get('whole_batch_on_one_gpu')
        if w:
            print(f"   whole batch on one GPU ({w['n_actor']}): {w['train_step_ms']:.2f} ms per step, "
                  f"{w['train_streamline_steps_per_s'] / 1e6:.1f} M streamline-steps/s")


training('config 3 training step', j.get('config3_training'))
training('config 5 training step', j.get('config5'))
c = j.get('cpu_baseline')
if isinstance(c, dict):
    a = c.get('all_cores') or {}
    print(f"cpu baseline: {c['value'] / 1e3:.0f} k/s on 1 core; {a.get('value', 0) / 1e6:.2f} M/s on "
          f"{a.get('cores')} cores ({c.get('cpu_model')})")
