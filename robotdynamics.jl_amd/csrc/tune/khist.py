#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps .s file:  khist.py file.s 'regex on demangled name'"""
import re, subprocess, collections, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.M | re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if not re.search(pat, dn): continue
    ops = collections.Counter(re.findall(r'^\s+([a-z_0-9]+)\s', body, re.M))
    tot = sum(ops.values())
    fp = sum(v for k, v in ops.items() if re.match(r'v_(fma|fmac|mul|add|pk_fma|pk_mul|pk_add|div|rcp|sqrt|rsq|ldexp|trig|sin|cos)\w*_f(64|32)', k))
    print(dn[:140]); print('  total instr', tot, ' fp-valu', fp)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 18
    print('   ', ', '.join(f'{k}:{v}' for k, v in ops.most_common(n)))
