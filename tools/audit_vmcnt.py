"""Development aid: lists the s_waitcnt vmcnt(N) of every loop of every kernel in a hipcc -S listing, with the number of
MFMAs and vector-memory instructions since the previous wait.  A small N in a K loop whose operands are fetched several
k-steps ahead means the compiler's scoreboard merge (conditional loads, loop back edges) has turned the prefetch into a
full drain: the case found in conv2x2_win (round 4: `vmcnt(2)` at every other slab = a wait for the window pieces issued
a few instructions earlier).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S csrc/<file>.hip -o /tmp/f.s
    python tools/audit_vmcnt.py /tmp/f.s [kernel-name substring]
    python tools/audit_vmcnt.py --copies csrc/<file>.hip [...]     (compiles; exit status 1 on a finding)
    python tools/audit_vmcnt.py --stores csrc/<file>.hip [...]     (compiles; exit status 1 on a finding)
    python tools/audit_vmcnt.py --counts csrc/<file>.hip [...]     (compiles; exit status 1 on a finding)

--counts: is every counted wait SMALL enough?  (audit_counts: every register an inline-asm load writes is followed through the
control-flow graph with the number of vector-memory operations issued behind it; a `vmcnt(N)` with N larger than that number does
not wait for it -- the class of commit 9bc486e, which --copies cannot see because it trusts the wait in front of the first MFMA.)

--copies: the second hazard of hand-counted waits.  A register written by an inline-asm `buffer_load_dwordx4` / `global_load_dwordx4` (between
;;#ASMSTART / ;;#ASMEND) holds its value only once the load has landed, which the compiler does not know: any instruction
other than an MFMA that READS such a register (a v_mov the register allocator placed to satisfy a tied operand or a phi)
may copy it while the load is in flight.  The kernels mark their weight-fragment loads with the asm comment `; wfrag`; the
check walks each kernel's listing in layout order, keeps a register 'hot' from a marked load until some other instruction
writes it, and reports every instruction but a v_mfma that reads a hot register.

A counted wait that lands only SOME of the marked loads names them: the asm comment `; landed v[a:b]` (conv_f32.hip: an empty asm
statement per operand register behind the s_waitcnt) takes those registers off the hot set.

--stores: `buffer_store_dwordx3/x4 vData, vOff, rsrc, sN` (SGPR soffset) directly followed by a VALU write of one of its data
registers.  hipcc's hazard recogniser exempts this form from the ">64-bit store data" hazard; gfx950 has the hazard all the same
(tools/micro/store_hazard.hip, profiles/r04_store_hazard.txt).  The kernels that store this way put wait states behind every
store (`buf_store16`); this mode checks the listing for any such store the compiler left unprotected.
"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith('.Lfunc_end'):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip('\n'))
    if name:
        yield name, body


def audit(body):
    out = []
    mfma = vmem = dma = 0
    depth = 0
    for i, ln in enumerate(body):
        s = ln.strip()
        m = re.search(r'Depth=(\d+)', ln)
        if s.startswith('.LBB') and m:
            depth = int(m.group(1))
        elif s.startswith('.LBB'):
            depth = 0
        if s.startswith('v_mfma'):
            mfma += 1
        elif re.match(r'(buffer|global)_(load|store|atomic)', s):
            if ' lds' in s:
                dma += 1
            else:
                vmem += 1
        w = re.search(r'vmcnt\((\d+)\)', s)
        if w or s.startswith('s_barrier'):
            out.append((i, depth, 'vmcnt(%s)' % w.group(1) if w else 'barrier', mfma, vmem, dma))
            mfma = vmem = dma = 0
    return out


def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def audit_copies(body):
    """-> list of (line, text) of instructions that may read a weight-fragment register while its load is in flight.

    Dataflow over the kernel's control-flow graph (labels, s_branch / s_cbranch_*): a register becomes HOT at a marked asm load
    (`; wfrag`) that writes it and stays hot until (a) another instruction writes it, or (b) a v_mfma reads it -- by the kernels'
    discipline the first MFMA that reads a fragment sits behind the counted wait for it, so from there on the value has landed
    and later copies are harmless.  Any other instruction that reads a hot register is reported (a v_mov or a scratch store the
    register allocator placed between a load and its wait), and so is any instruction that WRITES one: the load overwrites the
    new value when it lands (seen: fetches "past the end" that nothing consumes, their registers reused by the epilogue).
    `s_waitcnt vmcnt(0)` clears every register."""
    # ---- instructions and blocks
    insts = []   # (line index, text, is_marked_load, in_asm)
    inasm = False
    for i, ln in enumerate(body):
        s = ln.strip()
        if s.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if s.startswith(';;#ASMEND'):
            inasm = False
            continue
        if inasm and s.startswith('; landed'):     # asm comment of a counted wait: `; landed v[a:b]` = these registers' loads have landed
            insts.append((i, 'landed ' + s[len('; landed'):], inasm))
            continue
        if not s or s[0] == ';' or (s[0] == '.' and not s.startswith('.LBB')):
            continue
        insts.append((i, s, inasm))
    if not any(a and t.startswith(('buffer_load_dwordx4', 'global_load_dwordx4', 'buffer_load_dword')) and 'wfrag' in t for _, t, a in insts):
        return []
    blocks, cur, label_of = [], None, {}
    for idx, (i, t, a) in enumerate(insts):
        if t.startswith('.LBB'):
            name = t.split(':')[0]
            cur = {'name': name, 'insts': [], 'succ': []}
            label_of[name] = len(blocks)
            blocks.append(cur)
            continue
        if cur is None:
            cur = {'name': '<entry>', 'insts': [], 'succ': []}
            blocks.append(cur)
        cur['insts'].append((i, t, a))
        if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm'):
            nxt = {'name': '<ft%d>' % len(blocks), 'insts': [], 'succ': []}
            cur['term'] = t
            blocks.append(nxt)
            cur = nxt
    for b, blk in enumerate(blocks):
        term = blk.get('term', '')
        if term.startswith('s_endpgm'):
            continue
        if term.startswith('s_branch') or term.startswith('s_cbranch'):
            tgt = term.split()[1]
            if tgt in label_of:
                blk['succ'].append(label_of[tgt])
        if not term.startswith('s_branch') and b + 1 < len(blocks):
            blk['succ'].append(b + 1)

    def step(t, a, hot, report, i):
        toks = t.replace(',', ' ').split()
        op, args = toks[0], toks[1:]
        if op == 'landed':
            out = set(hot)
            for x in args:
                out -= regs_of(x)
            return out
        if a and op in ('buffer_load_dwordx4', 'global_load_dwordx4', 'buffer_load_dword') and 'wfrag' in t:
            return hot | regs_of(args[0])
        if op == 's_waitcnt' and ('vmcnt(0)' in t or 'wfrag-landed' in t):
            return set()   # everything has landed (vmcnt(0)), or the kernel says so: a counted wait whose budget covers every marked
            #                load still in flight carries the asm comment `; wfrag-landed` (conv2_gdn48: fetches some waves never use)
        if op.startswith('s_'):
            return hot
        is_store = op.startswith(('buffer_store', 'global_store', 'ds_write', 'flat_store', 'scratch_store'))
        dst = set() if is_store or op.startswith('s_') else (regs_of(args[0]) if args else set())
        srcs = set()
        for x in (args if is_store else args[1:]):
            srcs |= regs_of(x)
        if op.startswith('v_mfma'):
            if (dst & (hot - srcs)) and report is not None:
                report.append((i, t + '   <- overwrites a register whose load is in flight'))
            return hot - srcs - dst
        if ((srcs | dst) & hot) and report is not None:
            report.append((i, t + ('   <- overwrites a register whose load is in flight' if dst & hot else '')))
        return hot - dst

    # ---- fixpoint of the hot set at block entry (union over predecessors)
    hot_in = [set() for _ in blocks]
    work = list(range(len(blocks)))
    while work:
        b = work.pop()
        hot = set(hot_in[b])
        for i, t, a in blocks[b]['insts']:
            hot = step(t, a, hot, None, i)
        for sidx in blocks[b]['succ']:
            if not hot <= hot_in[sidx]:
                hot_in[sidx] |= hot
                work.append(sidx)
    out = []
    for b, blk in enumerate(blocks):
        hot = set(hot_in[b])
        for i, t, a in blk['insts']:
            hot = step(t, a, hot, out, i)
    return sorted(set(out))


def _cfg(body):
    """-> list of blocks {'insts': [(line, text, in_asm)], 'succ': [block index]} of one kernel listing."""
    insts, inasm = [], False
    for i, ln in enumerate(body):
        s = ln.strip()
        if s.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if s.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not s or s[0] == ';' or (s[0] == '.' and not s.startswith('.LBB')):
            continue
        insts.append((i, s, inasm))
    blocks, cur, label_of = [], None, {}
    for i, t, a in insts:
        if t.startswith('.LBB'):
            cur = {'insts': [], 'succ': []}
            label_of[t.split(':')[0]] = len(blocks)
            blocks.append(cur)
            continue
        if cur is None:
            cur = {'insts': [], 'succ': []}
            blocks.append(cur)
        cur['insts'].append((i, t, a))
        if t.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            cur['term'] = t
            cur = {'insts': [], 'succ': []}
            blocks.append(cur)
    for b, blk in enumerate(blocks):
        term = blk.get('term', '')
        if term.startswith('s_endpgm'):
            continue
        if term.startswith(('s_branch', 's_cbranch')):
            tgt = term.split()[1]
            if tgt in label_of:
                blk['succ'].append(label_of[tgt])
        if not term.startswith('s_branch') and b + 1 < len(blocks):
            blk['succ'].append(b + 1)
    return blocks


_VMEM = re.compile(r'(buffer|global|flat|scratch)_(load|store|atomic)')
CAP = 64


def audit_counts(body):
    """-> list of (line, text): instructions that read or overwrite a register whose INLINE-ASM load (or returning atomic) the
    counted `s_waitcnt vmcnt(N)` in front of them does not cover -- the "too large N" class (commit 9bc486e: the first tile of
    conv0_gdn_f32_persist waited vmcnt(12) with no store behind its loads yet, i.e. for nothing).

    Model = the one the kernels are written to: vector-memory operations retire in issue order, `vmcnt(N)` returns when at most N
    are outstanding.  For every register written by an asm `global_load / buffer_load / returning global_atomic` the analysis
    carries d = the number of vector-memory operations (loads, stores, atomics, LDS-DMA; asm or compiler-issued) issued after
    it, the MINIMUM over all paths of the control-flow graph (fixpoint); `vmcnt(N)` retires the registers with d >= N.  A
    register that is still outstanding when an instruction reads it, or overwrites it, is a finding.  Compiler-visible loads are
    not tracked: hipcc's own scoreboard waits for those."""
    blocks = _cfg(body)

    def is_asm_ret(t, a):
        if not a:
            return False
        op = t.split()[0]
        if op.startswith(('global_load_dword', 'buffer_load_dword', 'global_load_ushort', 'global_load_ubyte')) and ' lds' not in t:
            return True
        return op.startswith('global_atomic') and (' sc0' in t or ' glc' in t)

    def step(t, a, st, report, i):
        toks = t.replace(',', ' ').split()
        op, args = toks[0], toks[1:]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                st = {r: d for r, d in st.items() if d < n}
            return st
        if op.startswith('s_'):
            return st
        vm = bool(_VMEM.match(op))
        is_store = op.startswith(('buffer_store', 'global_store', 'ds_write', 'flat_store', 'scratch_store'))
        dst = set() if is_store else (regs_of(args[0]) if args else set())
        srcs = set()
        for x in (args if is_store else args[1:]):
            srcs |= regs_of(x)
        tracked = is_asm_ret(t, a)
        hit = (srcs | (set() if tracked else dst)) & set(st)
        if hit and report is not None:
            report.append((i, t + '   <- v%s: asm load possibly in flight (vmcnt budget %d)' % (sorted(hit)[0], min(st[r] for r in hit))))
        if vm:
            st = {r: min(d + 1, CAP) for r, d in st.items()}
        st = {r: d for r, d in st.items() if r not in dst}
        if tracked:
            for r in dst:
                st[r] = 0
        return st

    state_in = [None] * len(blocks)
    state_in[0] = {}
    work = [0]
    while work:
        b = work.pop()
        st = dict(state_in[b])
        for i, t, a in blocks[b]['insts']:
            st = step(t, a, st, None, i)
        for sidx in blocks[b]['succ']:
            old = state_in[sidx]
            if old is None:
                state_in[sidx] = dict(st)
                work.append(sidx)
            else:
                merged = dict(old)
                changed = False
                for r, d in st.items():
                    if r not in merged or d < merged[r]:
                        merged[r] = d
                        changed = True
                if changed:
                    state_in[sidx] = merged
                    work.append(sidx)
    out = []
    for b, blk in enumerate(blocks):
        if state_in[b] is None:
            continue
        st = dict(state_in[b])
        for i, t, a in blk['insts']:
            st = step(t, a, st, out, i)
    return sorted(set(out))


def audit_stores(body):
    """-> list of (store, next instruction) where a VALU write of the store's data registers follows it directly."""
    ins = [ln.strip() for ln in body]
    ins = [t for t in ins if t and t[0] not in ';.' and not t.endswith(':')]
    out = []
    for i, t in enumerate(ins[:-1]):
        if not t.startswith(('buffer_store_dwordx4', 'buffer_store_dwordx3')):
            continue
        toks = t.replace(',', ' ').split()
        if len(toks) < 5 or not re.match(r's\d+$', toks[4]):
            continue
        nx = ins[i + 1].replace(',', ' ').split()
        if nx[0].startswith('v_') and not nx[0].startswith('v_cmp') and len(nx) > 1 and regs_of(nx[1]) & regs_of(toks[1]):
            out.append((t, ins[i + 1]))
    return out


def compile_to_isa(src):
    import os
    import subprocess
    import tempfile
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + '.audit.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-D__HIP_PLATFORM_AMD__=1', '-x', 'hip',
                           '-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)
    return out


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--stores':
        import os
        bad = 0
        for src in sys.argv[2:]:
            n_file = 0
            for name, body in kernels(compile_to_isa(src)):
                for st, nx in audit_stores(body):
                    print('STORE? %s: %s -> %s' % (name[:80], st, nx))
                    n_file += 1
            bad += n_file
            print('%s: %s' % (os.path.basename(src), 'ok' if not n_file else '%d findings' % n_file))
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == '--counts':
        import os
        bad = 0
        for src in sys.argv[2:]:
            n_file = 0
            for name, body in kernels(compile_to_isa(src)):
                found = audit_counts(body)
                for i, text in found[:6]:
                    print('COUNT? %s line %d: %s' % (name[:90], i, text))
                if len(found) > 6:
                    print('       ... %d more in this kernel' % (len(found) - 6))
                n_file += len(found)
            bad += n_file
            print('%s: %s' % (os.path.basename(src), 'ok' if not n_file else '%d findings' % n_file))
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 1 and sys.argv[1] == '--copies':
        import os
        import subprocess
        import tempfile
        bad = 0
        for src in sys.argv[2:]:
            out = compile_to_isa(src)
            n_file = 0
            for name, body in kernels(out):
                found = audit_copies(body)
                for i, text in found[:6]:
                    print('COPY? %s line %d: %s' % (name[:90], i, text))
                if len(found) > 6:
                    print('       ... %d more in this kernel' % (len(found) - 6))
                n_file += len(found)
            bad += n_file
            print('%s: %s' % (os.path.basename(src), 'ok' if not n_file else '%d findings' % n_file))
        sys.exit(1 if bad else 0)
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, body in kernels(sys.argv[1]):
        if pat not in name:
            continue
        print('==', name)
        for i, depth, what, mfma, vmem, dma in audit(body):
            if depth > 0:
                print('  line %5d depth %d %-10s after %3d mfma, %2d vmem, %2d lds-dma' % (i, depth, what, mfma, vmem, dma))
