#!/usr/bin/env python3
"""stdin: the output of bench_traversal.py; prints one short line (mean / min / max ms per algorithm, relaxations and steps, levels)"""
import json
import sys

d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
parts = []
for k in ("bfs", "sssp"):
    if k in d:
        x = d[k]
        p = f"{k} mean {x['mean_ms']} min {x['min_ms']} max {x['max_ms']}"
        p += f" relax/edge {x.get('mean_relaxations_per_edge')} steps {x.get('mean_steps')}" if k == "sssp" else f" levels {x.get('mean_levels')}"
        parts.append(p)
print(" | ".join(parts))
