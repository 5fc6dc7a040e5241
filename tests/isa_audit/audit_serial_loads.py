"""ISA audit: chains of dependent single loads.  For every kernel in a gfx950 .s file, count `global_load` instructions that are
followed by `s_waitcnt vmcnt(0)` before any other global load is issued (a load whose latency nothing else overlaps), and report
the longest run of such load->wait pairs in program order.  A run of k means k memory round trips back to back in one wave.

    hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o x.s csrc/x.hip ;  python tests/isa_audit/audit_serial_loads.py x.s
"""
import re
import subprocess
import sys


def chains(path):
    """{demangled kernel name: (single-load waits, longest chain)} for every kernel of the listing."""
    kern = None
    stats = {}
    pending = run = best = singles = 0
    for l in open(path).read().split('\n'):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            if kern:
                stats[kern] = (singles, best)
            kern = m.group(1)
            pending = run = best = singles = 0
            continue
        t = l.strip()
        if t.startswith(';') or not t or kern is None:
            continue
        if t.startswith('.Lfunc_end'):
            stats[kern] = (singles, best)
            kern = None
            continue
        if re.match(r'(global|flat|buffer)_load', t):
            pending += 1
        elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
            if pending == 1:
                singles += 1
                run += 1
                best = max(best, run)
            elif pending > 1:
                run = 0
            pending = 0
        elif t.startswith('s_barrier') or t.startswith('s_endpgm'):
            run = 0
    names = subprocess.run(['c++filt'], input='\n'.join(stats), capture_output=True, text=True).stdout.split('\n')
    out = {}
    for (k, v), n in zip(stats.items(), names):
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        out[n] = v
    return out


if __name__ == '__main__':
    for n, (s, b) in sorted(chains(sys.argv[1]).items(), key=lambda x: -x[1][1]):
        if b >= 2:
            print(f"{n[:80]:80s} single-load waits {s:4d}  longest chain {b}")
