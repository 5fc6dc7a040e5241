"""Kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file): name, VGPR, AGPR, scratch, occupancy."""
import re, subprocess, sys
rows = []; cur = {}
for l in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        if cur: rows.append(cur)
        cur = {'name': m.group(1)}
    for k, s in (('VGPRs', 'v'), ('AGPRs', 'a'), (r'ScratchSize \[bytes/lane\]', 'scr'), (r'Occupancy \[waves/SIMD\]', 'occ'), (r'LDS Size \[bytes/block\]', 'lds')):
        m = re.search(r' ' + k + r': (\d+)', l)
        if m: cur[s] = int(m.group(1))
rows.append(cur)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    if re.search(pat, n):
        print(f"{n[:70]:70s} v={r.get('v'):4d} a={r.get('a'):4d} scratch={r.get('scr'):4d} occ={r.get('occ')}")
