"""ISA audit for the scorer: in every score_kernel instantiation of a --save-temps build, does the
loop around each per-tile s_barrier close on a SCALAR condition?

    hipcc -O3 ... --save-temps -c csrc/score.hip       ->  score-hip-amdgcn-amd-amdhsa-gfx950.s
    python tests/isa_audit/audit_barriers.py score-hip-amdgcn-amd-amdhsa-gfx950.s [name-regex]

A barrier is safe when every wave of the workgroup reaches it the same number of times.  hipcc guarantees that only for
control flow it KNOWS to be wave-uniform, which shows in the ISA as scalar loop control.  The listing's own loop
annotations are used ("; =>This Loop Header", "; in Loop: Header=BBn_m"): for every s_barrier that sits in a loop, every
branch of the loop's blocks that closes the loop (target = its header) or leaves it is classified
  scalar   s_branch / s_cbranch_scc*, or s_cbranch_vcc* where vcc = exec & (an SGPR pair set by s_cselect_b64 / s_and / s_or
           of such pairs, i.e. a scalar compare the compiler widened to a lane mask)
  VECTOR   s_cbranch_exec*, or s_cbranch_vcc* whose mask a v_cmp wrote: the trip count lives in VGPRs - the r02 symptom
and a barrier that an s_cbranch_exec* of the same loop can jump over is counted as 'under a lane mask'.
"""
import re
import sys


def kernels(lines):
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\S*score_kernel\S*):', lines[i])
        if m:
            j = i
            while not lines[j].startswith('.Lfunc_end'):
                j += 1
            yield m.group(1), lines[i:j]
            i = j
        i += 1


def sgpr_pair_is_scalar(body, n, reg, depth=0):
    """walk back from line n: was SGPR pair `reg` last written by scalar-compare machinery?"""
    if depth > 6:
        return None
    for k in range(n - 1, -1, -1):
        t = body[k].split(';')[0].strip()
        m = re.match(r'(\S+)\s+(s\[\d+:\d+\]|vcc)\s*,\s*(.*)', t)
        if not m or m.group(2) != reg:
            continue
        op, rest = m.group(1), m.group(3)
        if op.startswith('v_cmp'):
            return False
        if op == 's_cselect_b64' or op == 's_mov_b64':
            return True
        if op in ('s_and_b64', 's_andn2_b64', 's_or_b64', 's_orn2_b64', 's_xor_b64'):
            ok = True
            for src in re.findall(r's\[\d+:\d+\]|vcc', rest):
                r = sgpr_pair_is_scalar(body, k, src, depth + 1)
                if r is False:
                    return False
                ok = ok and (r is not False)
            return ok
        return None
    return None


def audit(body):
    # block label -> (line, loop header it belongs to or itself if header)
    blocks = []
    for n, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
        if m:
            blocks.append((n, m.group(1)))
    def loop_of(n):
        # nearest label at or before n; its annotation may sit on the label line or the following comment lines
        lab = None
        for ln, name in blocks:
            if ln <= n:
                lab = (ln, name)
        if lab is None:
            return None
        ln, name = lab
        text = body[ln] + ' ' + ' '.join(body[ln + 1:ln + 3])
        if 'Loop Header' in body[ln]:
            return name
        m = re.search(r'in Loop: Header=(BB\d+_\d+)', text)
        return '.L' + m.group(1) if m else None
    res = {'barriers': 0, 'in_loop': 0, 'scalar': 0, 'vector': 0, 'unknown': 0, 'masked': 0}
    label_line = {name: ln for ln, name in blocks}
    for n, l in enumerate(body):
        if not re.match(r'\s+s_barrier', l):
            continue
        res['barriers'] += 1
        hdr = loop_of(n)
        # a barrier that an EXEC-conditional forward branch can skip sits under a lane mask
        for k in range(n - 1, -1, -1):
            m = re.match(r'\s+s_cbranch_exec(z)\s+(\.LBB\d+_\d+)', body[k])     # (execnz = an out-of-line block that comes back)
            if m and label_line.get(m.group(2), -1) > n and loop_of(k) == hdr:
                res['masked'] += 1
                break
            if hdr is not None and loop_of(k) != hdr:
                break
        if hdr is None:
            continue
        res['in_loop'] += 1
        # every branch of the loop's own blocks that closes the loop (target = header) or leaves it (target outside)
        for k, b in enumerate(body):
            m = re.match(r'\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', b)
            if not m or loop_of(k) != hdr:
                continue
            tgt = m.group(2)
            tl = label_line.get(tgt)
            inside = tl is not None and loop_of(tl) == hdr and tgt != hdr
            if inside:
                continue                  # forward branch inside the loop body
            op = m.group(1)
            if op == 's_branch' or op.startswith('s_cbranch_scc'):
                res['scalar'] += 1
            elif op.startswith('s_cbranch_exec'):
                res['vector'] += 1
            else:
                r = sgpr_pair_is_scalar(body, k, 'vcc')
                res['scalar' if r else ('vector' if r is False else 'unknown')] += 1
    return res


def main():
    lines = open(sys.argv[1]).read().split('\n')
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    bad = n = 0
    for name, body in kernels(lines):
        if pat and not pat.search(name):
            continue
        r = audit(body)
        n += 1
        flag = r['vector'] or r['unknown'] or r['masked'] or (r['in_loop'] and not r['scalar'])
        bad += 1 if flag else 0
        short = re.search(r'score_kernelI(.*?)EEv', name).group(1).replace('EL', ',').replace('Li', '').replace('b', '').rstrip('E')
        print(f"{'!!' if flag else 'ok'} score_kernel<{short}>  barriers {r['barriers']} (in a loop: {r['in_loop']})  back edges: "
              f"scalar {r['scalar']} vector {r['vector']} unknown {r['unknown']}  barriers under a lane mask: {r['masked']}")
    print(f"{n} kernels, {bad} flagged")
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
