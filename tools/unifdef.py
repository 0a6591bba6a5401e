#!/usr/bin/env python3
"""A small `unifdef`: resolve the preprocessor conditionals of a source file whose condition is made ONLY of the given
macros (value, or U = undefined), delete the branches not taken and the `#ifndef X / #define X v / #endif` blocks that gave
the resolved macros their defaults; everything else is left alone.  Used in round 6 to take the measured-and-not-kept
experiment switches of rounds 2-5 out of the kernel headers (their record: profiles/HISTORY.md); tools/device_code_hash.py
shows that the compiled device code did not change.

  python tools/unifdef.py FILE [-o OUT] NAME=VALUE ... NAME=U ...
"""
import re
import sys


def cond_value(expr, macros):
    """-> True / False when `expr` contains only known macros, else None."""
    e = re.sub(r'/\*.*?\*/', ' ', expr)
    e = re.sub(r'//.*$', ' ', e).strip()
    names = set(re.findall(r'[A-Za-z_]\w*', e)) - {'defined'}
    if not names or not names <= set(macros):
        return None
    def repl_defined(m):
        return '1' if macros[m.group(1)] != 'U' else '0'
    e = re.sub(r'defined\s*\(\s*(\w+)\s*\)', repl_defined, e)
    e = re.sub(r'defined\s+(\w+)', repl_defined, e)
    for n in sorted(names, key=len, reverse=True):
        v = macros[n]
        e = re.sub(r'\b%s\b' % n, '0' if v == 'U' else '(%s)' % v, e)
    e = e.replace('&&', ' and ').replace('||', ' or ')
    e = re.sub(r'!(?!=)', ' not ', e)
    try:
        return bool(eval(e, {'__builtins__': {}}, {}))
    except Exception:
        return None


def process(lines, macros):
    out = []
    # stack entries: dict(kind='resolved'|'kept', taken=bool (a branch was already taken), active=bool (emit this branch))
    stack = []
    i = 0
    n = len(lines)

    def emitting():
        return all(f['active'] for f in stack)

    while i < n:
        ln = lines[i]
        # a directive may continue over backslash-newlines, and a trailing /* comment may run over several lines
        m = re.match(r'\s*#\s*(if|ifdef|ifndef|elif|else|endif|define|undef)\b(.*)', ln)
        if not m:
            if emitting():
                out.append(ln)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        j = i
        full = ln
        def unterminated(s):
            s2 = re.sub(r'/\*.*?\*/', '', s, flags=re.S)
            return '/*' in s2
        while (full.rstrip('\n').endswith('\\') or unterminated(full)) and j + 1 < n:
            j += 1
            full += lines[j]
        block = lines[i:j + 1]
        text = re.sub(r'/\*.*?\*/', ' ', full[full.index(kw) + len(kw):], flags=re.S).replace('\\\n', ' ')
        if kw in ('if', 'ifdef', 'ifndef'):
            # the default-giving block `#ifndef X` `#define X ...` `#endif` of a resolved macro
            if kw == 'ifndef' and text.strip() in macros and emitting():
                k = j + 1
                # find the matching #endif; the block must hold only the #define of the same macro (and comments)
                depth, k2, ok = 1, k, True
                body = []
                while k2 < n and depth:
                    mm = re.match(r'\s*#\s*(if|ifdef|ifndef|endif)\b', lines[k2])
                    if mm:
                        depth += -1 if mm.group(1) == 'endif' else 1
                    if depth:
                        body.append(lines[k2])
                    k2 += 1
                btxt = re.sub(r'/\*.*?\*/', ' ', ''.join(body), flags=re.S)
                btxt = re.sub(r'//.*', ' ', btxt)
                if re.fullmatch(r'\s*#\s*define\s+%s\b[^\n]*\s*' % re.escape(text.strip()), btxt.replace('\\\n', ' ')):
                    i = k2
                    continue
            if kw == 'if':
                val = cond_value(text, macros)
            else:
                name = text.strip()
                val = None if name not in macros else ((macros[name] != 'U') == (kw == 'ifdef'))
            if val is None or not emitting():
                stack.append({'kind': 'kept', 'taken': False, 'active': True if emitting() else False, 'dead_parent': not emitting()})
                if emitting():
                    out.extend(block)
            else:
                stack.append({'kind': 'resolved', 'taken': val, 'active': val, 'dead_parent': False})
        elif kw == 'elif':
            f = stack[-1]
            if f['kind'] == 'kept':
                if emitting():
                    out.extend(block)
            else:
                if f['taken']:
                    f['active'] = False
                else:
                    val = cond_value(text, macros)
                    if val is None:
                        raise SystemExit('cannot resolve #elif %s at line %d after a resolved #if' % (text.strip(), i + 1))
                    f['active'] = val
                    f['taken'] = val
        elif kw == 'else':
            f = stack[-1]
            if f['kind'] == 'kept':
                if emitting():
                    out.extend(block)
            else:
                f['active'] = not f['taken']
                f['taken'] = True
        elif kw == 'endif':
            f = stack.pop()
            if f['kind'] == 'kept' and not f['dead_parent'] and emitting():
                out.extend(block)
        else:                                            # define / undef
            if emitting():
                out.extend(block)
        i = j + 1
    if stack:
        raise SystemExit('unbalanced conditionals')
    return out


def main():
    args = sys.argv[1:]
    path = args.pop(0)
    outp = path
    if args and args[0] == '-o':
        args.pop(0); outp = args.pop(0)
    macros = dict(a.split('=', 1) for a in args)
    lines = open(path).read().splitlines(keepends=True)
    res = process(lines, macros)
    open(outp, 'w').write(''.join(res))
    print('%s: %d -> %d lines' % (path, len(lines), len(res)))


if __name__ == '__main__':
    main()
