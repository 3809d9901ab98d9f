#!/usr/bin/env python3
"""Wraps over-long C++/HIP source lines at token boundaries (never inside a string, a character literal or a // comment's code part), so that the token stream
-- and therefore the compiled code -- is unchanged.  A trailing // comment of a long line moves to its own line(s) ABOVE the code; a long comment line is
re-flowed; long string literals are cut into adjacent literals.  Preprocessor lines and lines of a multi-line macro are left alone (listed).
    python tools/wrap_lines.py [--limit 180] [--check] file ...
--check: only report lines above the limit (exit code 1 if any)."""
import sys

LIMIT = 180


def split_code_comment(line):
    """index of the trailing // comment outside strings, or -1"""
    i, n, q = 0, len(line), None
    while i < n:
        c = line[i]
        if q:
            if c == '\\': i += 2; continue
            if c == q: q = None
        else:
            if c in '"\'': q = c
            elif c == '/' and i + 1 < n and line[i + 1] == '/': return i
            elif c == '/' and i + 1 < n and line[i + 1] == '*':
                j = line.find('*/', i + 2)
                if j < 0: return -1
                i = j + 2; continue
        i += 1
    return -1


def wrap_comment(text, indent, limit):
    """text without the leading //; returns lines"""
    words = text.strip().split(' ')
    out, cur = [], indent + '//'
    for w in words:
        if len(cur) + 1 + len(w) > limit and cur.strip() != '//':
            out.append(cur); cur = indent + '//'
        cur += ' ' + w
    out.append(cur)
    return out


def break_points(code):
    """candidate break positions (index AFTER which a newline may go) with a priority: 0 = after ';' at paren depth 0, 1 = after '{' or before '}' at paren depth 0,
    2 = after ', ' / before ' && ' ' || ' ' ? ' ' : ' and around ' = ' at any depth, 3 = any space outside a literal"""
    pts, i, n, q, depth = [], 0, len(code), None, 0
    while i < n:
        c = code[i]
        if q:
            if c == '\\': i += 2; continue
            if c == q: q = None
            i += 1; continue
        if c in '"\'': q = c
        elif c == '/' and i + 1 < n and code[i + 1] == '*':
            j = code.find('*/', i + 2); i = (j + 2) if j >= 0 else n; continue
        elif c in '([': depth += 1
        elif c in ')]': depth -= 1
        elif c == ';' and depth == 0 and i + 1 < n and code[i + 1] == ' ': pts.append((i + 1, 0))
        elif c == '{' and depth == 0 and i + 1 < n and code[i + 1] == ' ': pts.append((i + 1, 1))
        elif c == ',' and i + 1 < n and code[i + 1] == ' ': pts.append((i + 1, 2))
        elif c == ' ':
            rest = code[i + 1:i + 4]
            if rest.startswith('&& ') or rest.startswith('|| ') or rest.startswith('? ') or rest.startswith(': '): pts.append((i, 2))
            else: pts.append((i, 3))
        i += 1
    return pts


def cut_strings(code, limit, indent):
    """a single literal longer than the room there is: cut into adjacent literals at a space inside it"""
    return code     # (not needed so far: no literal in the tree is longer than a line)


def wrap_code(code, indent, limit):
    out, first = [], True
    cont = indent + '    '
    cur = code
    while len(cur) > limit:
        pts = break_points(cur)
        lo = len(indent) + 24
        best = None
        for pr in (0, 1, 2, 3):
            cand = [p for p, q in pts if q == pr and lo <= p <= limit]
            if cand: best = max(cand); break
        if best is None:
            break                                   # nothing to cut at: left as it is (reported by --check)
        head, tail = cur[:best].rstrip(), cur[best:].lstrip()
        out.append(head)
        cur = (indent if False else cont) + tail
        first = False
    out.append(cur)
    return out


def process(path, limit, check):
    src = open(path).read().split('\n')
    out, bad, in_macro, changed = [], [], False, False
    for ln, line in enumerate(src, 1):
        stripped = line.lstrip()
        cont_macro = in_macro
        in_macro = line.rstrip().endswith('\\')
        if len(line) <= limit:
            out.append(line); continue
        if check:
            bad.append((ln, len(line))); out.append(line); continue
        if stripped.startswith('#') or cont_macro or in_macro:
            bad.append((ln, len(line))); out.append(line); continue
        indent = line[:len(line) - len(stripped)]
        if stripped.startswith('//'):
            out.extend(wrap_comment(stripped[2:], indent, limit)); changed = True; continue
        if stripped.startswith('/*') or stripped.startswith('*'):
            bad.append((ln, len(line))); out.append(line); continue
        ci = split_code_comment(line)
        code = line if ci < 0 else line[:ci].rstrip()
        if ci >= 0:
            out.extend(wrap_comment(line[ci + 2:], indent, limit))
        pieces = wrap_code(code, indent, limit)
        for p in pieces:
            if len(p) > limit: bad.append((ln, len(p)))
        out.extend(pieces); changed = True
    if changed and not check:
        open(path, 'w').write('\n'.join(out))
    return bad


def main():
    args = sys.argv[1:]
    limit, check = LIMIT, False
    files = []
    i = 0
    while i < len(args):
        if args[i] == '--limit': limit = int(args[i + 1]); i += 2
        elif args[i] == '--check': check = True; i += 1
        else: files.append(args[i]); i += 1
    rc = 0
    for f in files:
        bad = process(f, limit, check)
        for ln, n in bad:
            print("%s:%d: %d characters" % (f, ln, n)); rc = 1
    sys.exit(rc)


if __name__ == '__main__':
    main()
