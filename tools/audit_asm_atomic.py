"""Audit of the raw `global_atomic_add` claims in the persistent kernels (conv0_gdn96.hip, conv1x1_stream.hip, conv1x1_kres.hip, conv1x1_pair.hip).

Their destination VGPR is written when the atomic RETURNS, not at the asm statement; hipcc does not know that and may
copy or spill the register early (under register pressure it did so in conv2_gdn48.hip: stale claims, an endless unit
loop - that kernel now issues and waits in one statement).  This script compiles each file to ISA and prints, for every
atomic, the first instructions that read its destination: they must be the `v_add_u32 / v_cmp_eq_u32` of the tail, behind
the counted `s_waitcnt vmcnt(N)`.  Run after any change to those kernels:  python tools/audit_asm_atomic.py
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'sc2-benchmark_amd', 'csrc')
bad = 0
for f in ('conv0_gdn96.hip', 'conv1x1_stream.hip', 'conv1x1_kres.hip', 'conv1x1_pair.hip', 'conv2_gdn48.hip'):
    out = os.path.join(tempfile.gettempdir(), f + '.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip', '-S',
                           '--cuda-device-only', os.path.join(CSRC, f), '-o', out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    kern = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_ZN\S+):', l)
        if m:
            kern = m.group(1)
        m = re.search(r'global_atomic_add (v\d+),', l)
        if not m:
            continue
        reg = m.group(1)
        wait_seen = False
        for j in range(i + 1, min(i + 3000, len(lines))):
            if 's_waitcnt vmcnt' in lines[j]:
                wait_seen = True
            if re.search(r'\b%s\b' % reg, lines[j]):
                ok = wait_seen and re.search(r'v_add_u32|v_cmp_eq_u32', lines[j])
                print(('ok   ' if ok else 'BAD  ') + kern[:80], reg, '->', lines[j].strip()[:60])
                bad += 0 if ok else 1
                break
sys.exit(1 if bad else 0)
