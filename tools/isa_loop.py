#!/usr/bin/env python3
"""Print the instruction-kind sequence of the loop blocks of one kernel in a .s file.  usage: isa_loop.py file.s <mangled-substring>"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(pat) + r'\S*):.*?\.Lfunc_end', s, re.S | re.M)
body = m.group(0).split('\n')
blocks, cur, name = [], [], 'entry'
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.strip(); cur = []
    else:
        cur.append(l)
blocks.append((name, cur))
for name, b in blocks:
    if 'Loop' in name:
        seq = []
        for x in b:
            t = x.strip().split(' ')[0]
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            seq.append('M' if 'mfma' in t else 'r' if t.startswith('ds_read') else 'w' if t.startswith('ds_write') else
                       'G' if t.startswith(('global_load', 'buffer_load')) else 'S' if t.startswith(('global_store', 'buffer_store')) else
                       '|' if t == 's_waitcnt' else 'B' if t == 's_barrier' else 'v' if t.startswith('v_') else 's' if t.startswith('s_') else '?')
        print(name.split(';')[0].strip(), len(seq))
        print(''.join(seq))
