#!/usr/bin/env python3
"""Pretty-print a tools/bench_configs.py JSONL file."""
import json
import sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['name'][:62]:62s} N={d['N']:8d} {d['dtype']} {d['us_per_launch']:9.2f} us  {d['units_per_s']/1e9:8.3f} G/s  {d['GBps']:7.0f} GB/s  {d['frac_of_8TBps']:.3f}")
