"""ISA audit for kernels whose LDS fragment reads are inline asm and are waited for LATER than the compiler can see
(conv_wgrad.hip: a slab's fragments are read one slab ahead; the wait is the `s_waitcnt lgkmcnt(0)` of the next step).

hipcc does not know that the destination registers of such a read are valid only behind that wait: a register copy, a use as an
operand or a spill placed between the read and the wait would move stale data.  This tool compiles the file to ISA and follows every
`ds_read_b64_tr_b16` destination through the control-flow graph (fall-through and branch targets) until the first
`s_waitcnt lgkmcnt(0)` on each path; any instruction on the way that reads or writes one of the destination registers is reported.

    python tools/audit_lds_inflight.py sc2-benchmark_amd/csrc/conv_wgrad.hip        (exit status 1 on findings)
"""
import os
import re
import subprocess
import sys
import tempfile

READ = re.compile(r'^\s*ds_read_b64_tr_b16\s+v\[(\d+):(\d+)\]')
WAIT = re.compile(r'^\s*s_waitcnt\b.*lgkmcnt\(0\)')
LABEL = re.compile(r'^(\.LBB[\w]+):')
BRANCH = re.compile(r'^\s*s_(c?branch\w*)\s+(\.LBB\w+)')
REGS = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs_of(line):
    body = line.split(';')[0]
    out = set()
    for m in REGS.finditer(body):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(listing):
    text = open(listing).read()
    findings, kernels = [], 0
    for km in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', text, flags=re.S | re.M):
        name, lines = km.group(1), km.group(2).splitlines()
        if not any(READ.match(l) for l in lines):
            continue
        kernels += 1
        labels = {LABEL.match(l).group(1): i for i, l in enumerate(lines) if LABEL.match(l)}
        for i, l in enumerate(lines):
            m = READ.match(l)
            if not m:
                continue
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            seen, work = set(), [i + 1]
            while work:
                j = work.pop()
                while j < len(lines) and j not in seen:
                    seen.add(j)
                    t = lines[j]
                    if WAIT.match(t):
                        break
                    if 's_endpgm' in t:
                        break
                    ins = t.strip()
                    if ins and not ins.startswith(('.', ';')) and not LABEL.match(t) and not READ.match(t) and regs_of(t) & dst:
                        findings.append('{}: line {}: `{}` touches v{} of the LDS read at line {} before its wait'.format(
                            name, j + 1, ins[:90], sorted(regs_of(t) & dst), i + 1))
                    b = BRANCH.match(t)
                    if b:
                        if b.group(2) in labels:
                            work.append(labels[b.group(2)])
                        if b.group(1) == 'branch':       # unconditional: no fall-through
                            break
                    j += 1
    return kernels, findings


def main():
    src = sys.argv[1]
    if src.endswith('.s'):
        kernels, findings = audit(src)
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory() as td:
            lst = os.path.join(td, 'k.s')
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip', '--cuda-device-only', '-S',
                                   '-I' + os.path.join(root, 'include'), '-I' + os.path.join(root, 'sc2-benchmark_amd', 'csrc'), src, '-o', lst],
                                  stderr=subprocess.DEVNULL)
            kernels, findings = audit(lst)
    for f in findings[:40]:
        print('INFLIGHT?', f)
    print('{}: {} kernel(s) with asm fragment reads, {} finding(s)'.format(os.path.basename(src), kernels, len(findings)))
    return 1 if findings or not kernels else 0


if __name__ == '__main__':
    sys.exit(main())
