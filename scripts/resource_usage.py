#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).
usage: resource_usage.py remarks.txt [substring ...]   -> VGPR AGPR scratch spill occupancy LDS name"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
rows = []
for b in blocks:
    name = b.split(' [')[0].split('\n')[0]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    rows.append((name, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'),
                 g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
print("VGPR AGPR scratch spill occ LDS  name")
for r, n in zip(rows, names):
    n = n.replace('vof::', '')
    if not filt and r[3] <= 0:
        continue
    if filt and not any(f in n for f in filt):
        continue
    print("%4d %4d %7d %5d %3d %6d  %s" % (r[1:] + (n[:170],)))
