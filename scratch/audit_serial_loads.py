"""ISA audit: chains of dependent single loads.  For every kernel in a gfx950 .s file, count `global_load` instructions that are
followed by `s_waitcnt vmcnt(0)` before any other global load is issued (a load whose latency nothing else overlaps), and report
the longest run of such load->wait pairs in program order.  A run of k means k memory round trips back to back in one wave."""
import re, subprocess, sys
src = open(sys.argv[1]).read().split('\n')
kern = None; stats = {}
pending = 0; run = 0; best = 0; singles = 0
def flush():
    global kern, run, best, singles, pending
    if kern: stats[kern] = (singles, best)
    run = best = singles = pending = 0
for l in src:
    m = re.match(r'^(_Z\S+):', l)
    if m:
        flush(); kern = m.group(1); continue
    t = l.strip()
    if t.startswith(';') or not t: continue
    if t.startswith('.Lfunc_end'):
        flush(); kern = None; continue
    if kern is None: continue
    if re.match(r'(global|flat|buffer)_load', t):
        pending += 1
    elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        if pending == 1:
            singles += 1; run += 1; best = max(best, run)
        elif pending > 1:
            run = 0
        pending = 0
    elif t.startswith('s_barrier') or t.startswith('s_endpgm'):
        run = 0
flush()
names = subprocess.run(['c++filt'], input='\n'.join(stats), capture_output=True, text=True).stdout.split('\n')
for (k, (s, b)), n in sorted(zip(stats.items(), names), key=lambda x: -x[0][1][1]):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n).replace('void ', '')
    if b >= 2: print(f"{n[:80]:80s} single-load waits {s:4d}  longest chain {b}")
