#!/usr/bin/env python3
"""Scalar-memory audit of the shipped kernels: which s_load / s_buffer_load instructions read anything
but the kernel-argument segment, and is the scalar data cache invalidated before them?

Why.  The scalar data cache (K$) is not coherent with vector stores, and on this platform a kernel was
observed reading a previous solve's data through it from a reused workspace address (DESIGN.md 4.1c;
profiles/r02_pipe2d_bringup.txt).  Rule enforced here (tests/test_smem_audit.py runs this on the built
objects):

    a kernel may read memory other than its argument segment through the scalar unit only if it
    executes `s_dcache_inv` first (xinv_fresh_scalar_cache_wg: one wavefront of the workgroup invalidates,
    a workgroup barrier, then the loads) -- checked as: an s_dcache_inv precedes, in program order, the
    first scalar load whose base is not the argument segment.

How.  Every code object is taken out of build/obj/*.o (clang-offload-bundler), disassembled
(llvm-objdump) and scanned kernel by kernel.  The argument-segment pointer is the user SGPR pair the
kernel descriptor enables (kernel_code_properties, bits 0-3); pairs that only ever receive a copy of it
(s_mov_b64) count as the argument segment too.  Everything else is listed.

  python tools/smem_audit.py [--list] [--out FILE] [substring ...]
exit status 1 when a kernel breaks the rule.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def code_objects(objdir):
    """-> [(object file name, path of the extracted gfx950 code object)] (temporary files, kept by the caller)."""
    out = []
    td = tempfile.mkdtemp(prefix='smem_audit_')
    for f in sorted(os.listdir(objdir)):
        if not f.endswith('.o'):
            continue
        o = os.path.join(objdir, f)
        fat = os.path.join(td, f + '.fat')
        co = os.path.join(td, f + '.co')
        subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, o], capture_output=True)
        if not os.path.exists(fat):
            continue
        r = subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat,
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append((f, co))
    return out


def kernarg_sgpr(co):
    """kernel name -> first SGPR of the argument-segment pointer, from the kernel descriptors (<name>.kd)."""
    syms = subprocess.run([LLVM + '/llvm-readelf', '-s', '-W', co], capture_output=True, text=True).stdout
    secs = subprocess.run([LLVM + '/llvm-readelf', '-S', '-W', co], capture_output=True, text=True).stdout
    ro = None
    for ln in secs.splitlines():
        m = re.search(r'\]\s+\.rodata\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)', ln)
        if m:
            ro = (int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16))
    data = open(co, 'rb').read()
    res = {}
    for ln in syms.splitlines():
        p = ln.split()
        if len(p) >= 8 and p[-1].endswith('.kd') and ro:
            addr = int(p[1], 16)
            off = ro[1] + (addr - ro[0])
            props, = struct.unpack_from('<H', data, off + 56)
            if not (props >> 3) & 1:
                res[p[-1][:-3]] = None
                continue
            res[p[-1][:-3]] = 4 * (props & 1) + 2 * ((props >> 1) & 1) + 2 * ((props >> 2) & 1)
    return res


SLOAD = re.compile(r'^\s*(s_(?:buffer_)?load_dword\S*)\s+(\S+),\s*(s\[(\d+):(\d+)\]),\s*(.*?)\s*(?://.*)?$')
SMOV64 = re.compile(r'^\s*s_mov_b64\s+s\[(\d+):(\d+)\],\s*s\[(\d+):(\d+)\]')
WRITES = re.compile(r'^\s*s_\w+\s+(s\[(\d+):(\d+)\]|s(\d+))\b')


ADDR = re.compile(r'//\s*([0-9A-Fa-f]+):')
BRANCH = re.compile(r'^\s*(s_cbranch_\w+|s_branch)\s+\S+.*<[^>+]+(?:\+0x([0-9a-fA-F]+))?>')
NOWRITE = re.compile(r'^\s*s_(cmp|cbranch|branch|waitcnt|barrier|nop|endpgm|sleep|setprio|dcache|store|setreg|bitcmp|sendmsg|'
                     r'icache|ttrace|sethalt|setkill|setvskip|trap|rfe|inc_perf|dec_perf)')


def _sregs(tok):
    """'s[4:7]' -> [4,5,6,7]; 's3' -> [3]; anything else -> []"""
    m = re.fullmatch(r's\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', tok)
    return [int(m.group(1))] if m else []


def _vregs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return [int(m.group(1))] if m else []


def _is_imm(tok):
    return re.fullmatch(r'-?(0x[0-9a-fA-F]+|\d+)', tok) is not None


def transfer(ln, st):
    """Effect of one instruction on the symbolic state  {('s', n): 'KLO'|'KHI', ('v', n, lane): 'KLO'|'KHI'}:
    which scalar registers (and SGPR-spill lanes of vector registers) hold the low / high half of a pointer
    INTO the kernel-argument segment.  Copies (s_mov_b32/b64, v_writelane/v_readlane spills) and constant
    offsets (s_add_u32 imm + s_addc_u32 0: the implicit-argument pointer) keep it; any other write clears."""
    txt = ln.split('//')[0].strip()
    if not txt:
        return st
    p = txt.replace(',', ' ').split()
    op, args = p[0], p[1:]
    st = dict(st)

    def kill_s(regs):
        for r in regs:
            st.pop(('s', r), None)

    def kill_v(regs):
        for k in [k for k in st if k[0] == 'v' and k[1] in regs]:
            st.pop(k)

    if op == 's_mov_b64' and len(args) == 2 and _sregs(args[0]) and _sregs(args[1]):
        d, sr = _sregs(args[0]), _sregs(args[1])
        vals = [st.get(('s', r)) for r in sr]
        kill_s(d)
        for r, v in zip(d, vals):
            if v:
                st[('s', r)] = v
        return st
    if op == 's_mov_b32' and len(args) == 2 and _sregs(args[0]):
        v = st.get(('s', _sregs(args[1])[0])) if _sregs(args[1]) else None
        kill_s(_sregs(args[0]))
        if v:
            st[('s', _sregs(args[0])[0])] = v
        return st
    if op == 's_add_u32' and len(args) == 3 and _sregs(args[0]):
        a, b = args[1], args[2]
        v = None
        # pointer arithmetic on the argument-segment pointer stays inside the segment (an index into an
        # array member of the argument struct, the implicit-argument pointer)
        if _sregs(a) and st.get(('s', _sregs(a)[0])) == 'KLO' and not (_sregs(b) and st.get(('s', _sregs(b)[0]))):
            v = 'KLO'
        if _sregs(b) and st.get(('s', _sregs(b)[0])) == 'KLO' and not (_sregs(a) and st.get(('s', _sregs(a)[0]))):
            v = 'KLO'
        kill_s(_sregs(args[0]))
        if v:
            st[('s', _sregs(args[0])[0])] = v
        return st
    if op == 's_addc_u32' and len(args) == 3 and _sregs(args[0]):
        a, b = args[1], args[2]
        v = None
        if _sregs(a) and st.get(('s', _sregs(a)[0])) == 'KHI' and not (_sregs(b) and st.get(('s', _sregs(b)[0]))):
            v = 'KHI'
        if _sregs(b) and st.get(('s', _sregs(b)[0])) == 'KHI' and not (_sregs(a) and st.get(('s', _sregs(a)[0]))):
            v = 'KHI'
        kill_s(_sregs(args[0]))
        if v:
            st[('s', _sregs(args[0])[0])] = v
        return st
    if op == 'v_writelane_b32' and len(args) == 3 and _vregs(args[0]) and _is_imm(args[2]):
        vn, lane = _vregs(args[0])[0], int(args[2], 0)
        st.pop(('v', vn, lane), None)
        v = st.get(('s', _sregs(args[1])[0])) if _sregs(args[1]) else None
        if v:
            st[('v', vn, lane)] = v
        return st
    if op == 'v_readlane_b32' and len(args) == 3 and _sregs(args[0]) and _vregs(args[1]) and _is_imm(args[2]):
        v = st.get(('v', _vregs(args[1])[0], int(args[2], 0)))
        kill_s(_sregs(args[0]))
        if v:
            st[('s', _sregs(args[0])[0])] = v
        return st
    if NOWRITE.match(txt) or op.startswith('s_cmp') or op.startswith('s_bitcmp'):
        return st
    # generic: the first operand (and vcc/exec writers do not matter here) is the destination
    if args:
        if re.match(r'^(global_store|flat_store|buffer_store|scratch_store|ds_write|ds_add|ds_max|ds_min|global_atomic\w*$)', op):
            return st
        kill_s(_sregs(args[0]))
        kill_v(_vregs(args[0]))
        # instructions with a second (carry / compare) SGPR destination: v_add_co_u32 v, s[..], ...; v_div_scale
        if op.startswith('v_') and len(args) > 1 and _sregs(args[1]) and re.match(r'^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64|mad_i64)', op):
            kill_s(_sregs(args[1]))
    return st


def _join(a, b):
    return {k: v for k, v in a.items() if b.get(k) == v}


def scan(co):
    """-> {kernel: dict(kernarg=.., nonkarg=[(index, text)], first_inv=index or None, ninstr=..)}

    Which SGPR pairs point into the argument segment is a forward must-analysis over the kernel's control-flow
    graph (basic blocks from the branch targets llvm-objdump prints): the user SGPR pair the descriptor enables
    at entry, then `transfer` per instruction; at a join only what holds on every incoming edge."""
    ka = kernarg_sgpr(co)
    dis = subprocess.run([LLVM + '/llvm-objdump', '-d', '--no-show-raw-insn', co], capture_output=True, text=True).stdout
    kernels, cur, name = {}, None, None
    for ln in dis.splitlines():
        m = re.match(r'^([0-9a-f]+) <(\S+)>:', ln)
        if m:
            name = m.group(2)
            cur = kernels.setdefault(name, []) if name in ka else None
            continue
        if cur is not None and ln.strip() and not ln.startswith('Disassembly'):
            cur.append(ln)
    out = {}
    for k, lines in kernels.items():
        base = ka.get(k)
        n = len(lines)
        addr = []
        for ln in lines:
            m = ADDR.search(ln)
            addr.append(int(m.group(1), 16) if m else None)
        start = addr[0]
        index_of = {a: i for i, a in enumerate(addr) if a is not None}
        leaders = {0}
        succ = [[] for _ in range(n)]
        for i, ln in enumerate(lines):
            m = BRANCH.match(ln)
            if m:
                t = index_of.get(start + int(m.group(2) or '0', 16))
                if t is not None:
                    succ[i].append(t); leaders.add(t)
                if m.group(1) != 's_branch' and i + 1 < n:
                    succ[i].append(i + 1)
                if i + 1 < n:
                    leaders.add(i + 1)
            elif re.match(r'^\s*s_endpgm', ln):
                if i + 1 < n:
                    leaders.add(i + 1)
            elif i + 1 < n:
                succ[i].append(i + 1)
        lead = sorted(leaders)
        blk_of, blocks = {}, []
        for bi, l in enumerate(lead):
            e = lead[bi + 1] if bi + 1 < len(lead) else n
            blocks.append((l, e))
            for i in range(l, e):
                blk_of[i] = bi
        preds = [[] for _ in blocks]
        for bi, (l, e) in enumerate(blocks):
            for t in succ[e - 1]:
                preds[blk_of[t]].append(bi)
        TOP = None                                         # not reached yet
        IN = [TOP] * len(blocks)
        OUT = [TOP] * len(blocks)
        IN[0] = {} if base is None else {('s', base): 'KLO', ('s', base + 1): 'KHI'}
        work = [0]
        while work:
            bi = work.pop(0)
            if bi != 0:
                acc = TOP
                for p in preds[bi]:
                    if OUT[p] is TOP:
                        continue
                    acc = dict(OUT[p]) if acc is TOP else _join(acc, OUT[p])
                IN[bi] = acc
            if IN[bi] is TOP:
                continue
            cur_st = IN[bi]
            l, e = blocks[bi]
            for i in range(l, e):
                cur_st = transfer(lines[i], cur_st)
            if OUT[bi] is TOP or cur_st != OUT[bi]:
                OUT[bi] = cur_st
                for t in succ[e - 1]:
                    if blk_of[t] not in work:
                        work.append(blk_of[t])
        nonk, first_inv = [], None
        for bi, (l, e) in enumerate(blocks):
            cur_st = IN[bi] if IN[bi] is not TOP else {}
            for i in range(l, e):
                ln = lines[i]
                m = SLOAD.match(ln)
                if m:
                    b = int(m.group(4))
                    is_k = cur_st.get(('s', b)) == 'KLO' and cur_st.get(('s', b + 1)) == 'KHI' and \
                        not m.group(1).startswith('s_buffer')
                    if not is_k:
                        nonk.append((i, ' '.join(ln.split('//')[0].split())))
                cur_st = transfer(ln, cur_st)
        nonk.sort()
        for i, ln in enumerate(lines):
            if re.match(r'^\s*s_dcache_inv\b', ln):
                first_inv = i
                break
        out[k] = dict(kernarg=base, nonkarg=nonk, first_inv=first_inv, ninstr=n,
                      nsload=sum(1 for ln in lines if SLOAD.match(ln)))
    return out


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, r))


def audit(objdir=None, pats=()):
    objdir = objdir or os.path.join(ROOT, 'build', 'obj')
    rows, bad = [], []
    for fname, co in code_objects(objdir):
        res = scan(co)
        dm = demangle(list(res))
        for k in sorted(res, key=lambda x: dm[x]):
            d = res[k]
            name = dm[k]
            if pats and not all(p in name for p in pats):
                continue
            first_nonk = d['nonkarg'][0][0] if d['nonkarg'] else None
            ok = (first_nonk is None) or (d['first_inv'] is not None and d['first_inv'] < first_nonk)
            rows.append((fname, name, d, ok))
            if not ok:
                bad.append((fname, name))
    return rows, bad


def main():
    args = [a for a in sys.argv[1:]]
    out = None
    if '--out' in args:
        i = args.index('--out'); out = args[i + 1]; del args[i:i + 2]
    listing = '--list' in args
    pats = [a for a in args if not a.startswith('--')]
    rows, bad = audit(pats=pats)
    lines = []
    nk = sum(1 for r in rows if r[2]['nonkarg'])
    lines.append('# scalar-memory audit: %d kernels, %d with scalar loads outside the argument segment, %d breaking the rule'
                 % (len(rows), nk, len(bad)))
    lines.append('# columns: object | kernel | instructions | scalar loads | of which outside the argument segment | '
                 'index of the first s_dcache_inv | index of the first such load | verdict')
    for fname, name, d, ok in rows:
        fn = d['nonkarg'][0][0] if d['nonkarg'] else '-'
        verdict = 'argument segment only' if not d['nonkarg'] else ('invalidates first' if ok else 'NOT INVALIDATED')
        lines.append('%s | %s | %d | %d | %d | %s | %s | %s' % (fname, name[:170], d['ninstr'], d['nsload'], len(d['nonkarg']),
                                                                 d['first_inv'] if d['first_inv'] is not None else '-', fn, verdict))
        if listing or not ok:
            for i, t in d['nonkarg'][:12]:
                lines.append('        [%d] %s' % (i, t))
            if len(d['nonkarg']) > 12:
                lines.append('        ... %d more' % (len(d['nonkarg']) - 12))
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    sys.stdout.write(txt if (listing or len(rows) < 60) else '\n'.join(l for l in lines if l.startswith('#') or 'NOT INVAL' in l) + '\n')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
