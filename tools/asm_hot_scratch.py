#!/usr/bin/env python3
"""Lists, per kernel of a .s file, the scratch (spill) instructions and which of them sit in a basic block of a loop that also
issues MFMAs in a pinned gap pattern ("hot").  Usage: tools/asm_hot_scratch.py file.s [kernel-name-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
kern, blocks, cur = None, [], None
out = {}
for ln in src:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        kern = m.group(1)
        out[kern] = []
        cur = {'label': 'entry', 'loop': False, 'mfma': 0, 'sched': 0, 'scratch': []}
        out[kern].append(cur)
        continue
    if kern is None:
        continue
    m = re.match(r'^(\.LBB\w+):(.*)', ln)
    if m:
        cur = {'label': m.group(1), 'loop': 'Loop' in m.group(2), 'mfma': 0, 'sched': 0, 'scratch': []}
        out[kern].append(cur)
        continue
    if re.match(r'^; %bb\.', ln):
        cur = {'label': ln.split()[1], 'loop': 'Loop' in ln, 'mfma': 0, 'sched': 0, 'scratch': []}
        out[kern].append(cur)
        continue
    t = ln.strip()
    if t.startswith('v_mfma'):
        cur['mfma'] += 1
    if 'sched_barrier' in t:
        cur['sched'] += 1
    if t.startswith('scratch_'):
        cur['scratch'].append(t.split()[0])
    if t.startswith('s_endpgm'):
        kern = None
for k, bl in out.items():
    if want not in k:
        continue
    tot = sum(len(b['scratch']) for b in bl)
    hot = [(b['label'], len(b['scratch'])) for b in bl if b['scratch'] and b['loop'] and b['sched'] >= 4]
    print(f'{k}: {tot} scratch instructions; in pinned loop blocks: {hot if hot else "none"}')
