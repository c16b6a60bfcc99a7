#!/usr/bin/env python3
"""Instruction mix of the large loops of one kernel in a -save-temps gfx950 .s file: tools/loop_mix.py <file.s> <kernel-name-substring> [min_instructions]."""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 300
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(key), s, re.M)
body = s[m.end():]
body = body[:re.search(r'^\.Lfunc_end\d+:', body, re.M).start()]
lines = [l.strip() for l in body.split('\n')]
labels = {re.match(r'^(\.LBB\d+_\d+):', l).group(1): n for n, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
for n, l in enumerate(lines):
    b = re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if not (b and b.group(1) in labels and labels[b.group(1)] < n):
        continue
    seg = [x for x in lines[labels[b.group(1)]:n] if x and not x.startswith(('.', ';'))]
    if len(seg) < minlen:
        continue
    c = collections.Counter()
    for x in seg:
        op = x.split()[0]
        if 'mfma' in op: c['MFMA ' + op] += 1
        elif op.startswith('v_'): c['VALU (all)'] += 1; c['  ' + re.sub(r'_e32|_e64|_dpp|_sdwa', '', op)] += 1
        elif op.startswith('ds_'): c['LDS ' + op] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): c['VMEM ' + op] += 1
        elif op.startswith('s_waitcnt'): c['s_waitcnt'] += 1
        elif op.startswith('s_'): c['SALU'] += 1
        else: c['other ' + op] += 1
    print("loop %s: %d instructions" % (b.group(1), len(seg)))
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:50]:
        print("   %-40s %d" % (k, v))
