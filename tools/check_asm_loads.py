#!/usr/bin/env python3
"""Audit of the assembly of the kernels with hand-counted waits (pasero_amd/csrc/gemm8p.hip, gemmln.hip, gemmbs.hip), run
by __graft_entry__.build() and the CPU tests:

  1. between the PK8P_LOOP_BEGIN / PK8P_LOOP_END markers no `s_waitcnt` may drain the vector-memory counter below the
     hand-placed counted waits (hipcc adds `vmcnt(0)` in front of LDS accesses it cannot disambiguate from an LDS-DMA
     in flight: that would serialise the prefetch, silently);
  2. the transposed LDS reads are inline asm, so the compiler neither waits for them nor knows when their destination
     registers become valid: from each `ds_read_b64_tr_b16` to the next `s_waitcnt lgkmcnt(0)` nothing else may read or
     write those registers (a compiler copy there would move stale data);
  2b. the same for inline-asm `buffer_load_dword*` to registers (gemmpw.hip's side operands: the hardware does not interlock
     a register with a load in flight): from the load to the next `s_waitcnt vmcnt(..)` nothing may touch its destination
     (checked over the whole kernel body, in program order);
  3. no scratch (spill) traffic inside the K loop's blocks (spills in the prologue / epilogue, or between the markers
     but outside every loop block — executed once — are reported, not refused).
The smallest counted wait a kernel places in its loop is 6 unless the kernel says otherwise with a `; PK8P_MIN_VMCNT n`
marker; rule 2 also covers a `PKBS_BFRAG_BEGIN / _END` region (gemmbs.hip's one-time transposed reads of the B panel).

Usage: check_asm_loads.py <file.s>     (exit code 1 and a report if a rule is broken)"""
import re
import sys

notes = []


def regs(tok: str):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def operands(line: str):
    body = line.split(';')[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else '', []
    return parts[0], [t.strip() for t in re.split(r',\s*', parts[1])]


def audit(path: str):
    lines = open(path).read().splitlines()
    problems, kernels = [], 0
    global notes
    notes = []
    i = 0
    while i < len(lines):
        m = re.match(r'^(_ZN[^:]*(?:gemm8p_(?:group_|ln_|hm2_|pw_|pt_)?|gemmbs_)kernel[^:]*):', lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        kernels += 1
        j = i
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):  # (not the first s_endpgm: a kernel may return early)
            j += 1
        body = lines[i:j]
        i = j
        n_scratch = sum('scratch_' in ln for ln in body)
        try:
            b = next(k for k, ln in enumerate(body) if 'PK8P_LOOP_BEGIN' in ln)
            e = next(k for k, ln in enumerate(body) if 'PK8P_LOOP_END' in ln)
        except StopIteration:
            problems.append(f'{name}: loop markers not found')
            continue
        if n_scratch:
            notes.append(f'{name}: {n_scratch} scratch instructions outside the K loop')
        min_vmcnt = 6
        for ln in body:
            mm = re.search(r'PK8P_MIN_VMCNT (\d+)', ln)
            if mm:
                min_vmcnt = int(mm.group(1))
        spans = [(b, e, True)]
        try:
            spans.append((next(k for k, ln in enumerate(body) if 'PKBS_BFRAG_BEGIN' in ln),
                          next(k for k, ln in enumerate(body) if 'PKBS_BFRAG_END' in ln), False))
        except StopIteration:
            pass
        # rule 2b: asm buffer loads (between ;;#ASMSTART / ;;#ASMEND) over the whole body
        in_asm, vpend = False, {}
        for k, ln in enumerate(body):
            if 'ASMSTART' in ln:
                in_asm = True
                continue
            if 'ASMEND' in ln:
                in_asm = False
                continue
            op, ops = operands(ln)
            if not op or op.endswith(':') or op.startswith('.') or op.startswith(';'):
                continue
            if op == 's_waitcnt' and 'vmcnt' in ln:
                vpend.clear()
                continue
            if op in ('s_branch', 's_endpgm', 's_setpc_b64'):
                vpend.clear()
                continue
            if in_asm and op.startswith('buffer_load_dword') and ops and 'lds' not in ln:
                for r in regs(ops[0]):
                    vpend[r] = k
                continue
            if vpend and ops:
                touched = set().union(*[regs(t) for t in ops])
                bad = touched & set(vpend)
                if bad:
                    problems.append(f'{name}: line {k}: `{ln.strip()}` touches v{sorted(bad)} before a vmcnt wait covers the asm '
                                    f'load at line {vpend[sorted(bad)[0]]}')
                    for r in bad:
                        del vpend[r]
        pending = {}  # register -> line of the asm tr read that wrote it
        looping = False  # inside a block LLVM marks as part of a loop
        # rule 3 tells a spill inside the loop from one that runs once by LLVM's label annotations ("Loop Header" / "in Loop"):
        # an assembly written without them (no verbose-asm comments, another format) would turn every in-loop spill into a
        # note — so a K loop between the markers with no annotated label at all fails the audit instead of passing it
        if not any(re.match(r'^\.LBB\d+_\d+:.*Loop', body[k]) for k in range(b, e)):
            problems.append(f'{name}: no label with a loop annotation between the loop markers: cannot tell in-loop spills '
                            f'from run-once ones (assembly without verbose-asm comments?)')
        for k, in_loop in [(k, lp) for (s0, s1, lp) in spans for k in range(s0, s1)]:
            ln = body[k]
            if not in_loop:  # the B-fragment region: only rule 2
                op, ops = operands(ln)
                if op == 's_waitcnt' and 'lgkmcnt(0)' in ln:
                    pending.clear()
                elif op == 'ds_read_b64_tr_b16':
                    for r in regs(ops[0]):
                        pending[r] = k
                elif ops:
                    touched = set().union(*[regs(t) for t in ops])
                    bad = touched & set(pending)
                    if bad:
                        problems.append(f'{name}: line {k}: `{ln.strip()}` touches v{sorted(bad)} before the lgkmcnt(0) '
                                        f'that covers the asm read at line {pending[sorted(bad)[0]]}')
                continue
            mlab = re.match(r'^\.LBB\d+_\d+:(.*)$', ln)
            if mlab:  # LLVM annotates the labels of blocks that belong to a loop ("Loop Header" / "in Loop:")
                looping = 'Loop' in mlab.group(1)
            if 'scratch_' in ln:
                if looping:
                    problems.append(f'{name}: line {k}: spill traffic inside the K loop: `{ln.strip()}`')
                else:  # between the markers but executed once: ahead of the first iteration or behind the last
                    notes.append(f'{name}: line {k}: spill outside the loop blocks (runs once): `{ln.strip()}`')
                n_scratch -= 1
            op, ops = operands(ln)
            if op in ('s_branch', 's_endpgm', 's_setpc_b64'):
                pending.clear()  # what follows is only reached by a jump: not in program order behind these reads
                continue
            if op == 's_waitcnt':
                mm = re.search(r'vmcnt\((\d+)\)', ln)
                if mm and int(mm.group(1)) < min_vmcnt:
                    problems.append(f'{name}: line {k}: `{ln.strip()}` inside the K loop drains the LDS-DMA prefetch')
                if 'lgkmcnt(0)' in ln:
                    pending.clear()
                continue
            if op == 'ds_read_b64_tr_b16':
                for r in regs(ops[0]):
                    pending[r] = k
                touched = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
            else:
                touched = set().union(*[regs(t) for t in ops]) if ops else set()
                if op.startswith('ds_read') and ops:  # another LDS read may reuse... its destination only
                    touched = set().union(*[regs(t) for t in ops[1:]]) | (regs(ops[0]) & set(pending))
            bad = touched & set(pending)
            if bad and op != 'ds_read_b64_tr_b16':
                problems.append(f'{name}: line {k}: `{ln.strip()}` touches v{sorted(bad)} before the lgkmcnt(0) that '
                                f'covers the asm read at line {pending[sorted(bad)[0]]}')
    if kernels == 0:
        problems.append('no audited kernels found in ' + path)
    return kernels, problems


if __name__ == '__main__':
    n, probs = audit(sys.argv[1])
    for p in probs:
        print('PROBLEM:', p)
    for t in notes:
        print('note:', t)
    print(f'{n} kernels audited, {len(probs)} problems')
    sys.exit(1 if probs else 0)
