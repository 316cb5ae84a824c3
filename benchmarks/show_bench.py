#!/usr/bin/env python3
"""One-line digest of a bench.py JSON line: python benchmarks/show_bench.py <file>"""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w, r = j['windows'], j['roofline']
ep = j.get('whole_episode') or {}
print(f"value {j['value'] / 1e6:.1f} M (min {w['value_min'] / 1e6:.1f}, max {w['value_max'] / 1e6:.1f}; "
      f"with events {((w.get('value_median_with_events') or 0) / 1e6):.1f}, "
      f"without {((w.get('value_median_without_events') or 0) / 1e6):.1f}) "
      f"k_state {r['avg_launch_ms']:.4f} ms x{r['launches']} frac {r['frac']} "
      f"whole episode {ep.get('streamline_steps_per_s_rank0', 0) / 1e6:.1f} M")
