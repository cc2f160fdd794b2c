"""Static check on a gfx950 assembly listing: no register that an inline-asm load is still filling is read by a compiler spill or copy.

A buffer load issued from inline asm defines its destination registers as far as the compiler is concerned -- it does not know that the
data arrives later and is waited for with a hand-counted `s_waitcnt vmcnt(N)`.  Under register pressure it may spill (scratch_store) or
copy (v_mov) such a register right behind the load: the spilled value is whatever the register held before (gdn512_rows.hip, round 5:
two of twelve fragment registers, results garbage).  This script walks each kernel's listing in program order, keeps the queue of
outstanding vector-memory operations (vmcnt retires in issue order; `s_waitcnt vmcnt(N)` leaves the N youngest outstanding) and reports
every scratch_store / v_mov / v_accvgpr_write whose source overlaps the destination of an asm load (marked `; wfrag`, `; gop`, or any
load between ;;#ASMSTART / ;;#ASMEND) that is still outstanding.  Straight-line approximation: branches are ignored (the kernels'
pipelines are straight-line inside their tile loops).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S kernel.hip -o kernel.s && python tools/audit_inflight.py kernel.s
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
VMEM = re.compile(r'^\s*(buffer_|global_|scratch_|flat_)(load|store|atomic)')
WAIT = re.compile(r's_waitcnt.*vmcnt\((\d+)\)')


def regs(tok):
    out = set()
    for a, b, c in REG.findall(tok):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def audit(path):
    findings = []
    kernel, in_asm, queue = None, False, []     # queue: (is_asm_load, dest registers)
    for n, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r'^([A-Za-z_][\w$.]*):', s)
        if m and not s.startswith('.L'):
            kernel, queue = m.group(1), []
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        code = s.split(';')[0]
        w = WAIT.search(code)
        if w:
            keep = int(w.group(1))
            queue = queue[len(queue) - keep:] if keep else []
            continue
        if VMEM.match(code):
            ops = code.split(None, 1)[1] if ' ' in code else ''
            first = ops.split(',')[0]
            if 'store' in code.split()[0]:
                # a spill of a register that is being filled
                if code.startswith('scratch_store'):
                    src = regs(ops)
                    for is_asm, dst in queue:
                        if is_asm and src & dst:
                            findings.append((kernel, n, s))
                            break
                queue.append((False, set()))
            else:
                queue.append((in_asm and 'lds' not in code, regs(first)))
            continue
        if code.startswith(('v_mov_b', 'v_accvgpr_write', 'v_pk_mov')):
            ops = code.split(None, 1)[1]
            src = regs(','.join(ops.split(',')[1:]))
            for is_asm, dst in queue:
                if is_asm and src & dst:
                    findings.append((kernel, n, s))
                    break
    return findings


if __name__ == '__main__':
    bad = []
    for path in sys.argv[1:]:
        bad += [(path,) + f for f in audit(path)]
    for path, kernel, n, s in bad:
        print('{}:{}: [{}] reads a register an asm load is still filling: {}'.format(path, n, kernel, s))
    print('{} finding(s)'.format(len(bad)))
    sys.exit(1 if bad else 0)
