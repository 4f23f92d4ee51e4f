#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: scratch/res_usage.py remarks.txt [name filter]"""
import re, sys
t = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None
rows = {}
for ln in t:
    m = re.search(r'remark:\s+Function Name: (\S+)', ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass', ln)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if flt in k:
        print(k[:70], 'VGPR', v.get('VGPRs'), 'AGPR', v.get('AGPRs'), 'spill', v.get('VGPRs Spill'), 'scratch', v.get('ScratchSize [bytes/lane]'), 'occ', v.get('Occupancy [waves/SIMD]'), 'LDS', v.get('LDS Size [bytes/block]'))
