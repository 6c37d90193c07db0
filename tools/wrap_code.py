#!/usr/bin/env python3
"""Break over-long C++ / HIP code lines at 160 columns without changing a token: trailing // comments move to a line of their own above, the statement is broken after
", " / an operator outside string literals (continuation lines indented by 4 more), and a string literal that alone overflows is cut into adjacent literals at a space.
Preprocessor lines are left alone.  usage: wrap_code.py <file> ...   (compile afterwards: this tool knows tokens, not the language)"""
import sys

LIMIT = 160
BREAKS = (", ", " && ", " || ", " + ", " - ", " ? ", " : ", " << ", " = ", "; ")


def scan(line):
    """state per character: True inside a string / char literal"""
    inside, out, q, i = False, [], "", 0
    while i < len(line):
        c = line[i]
        if inside:
            out.append(True)
            if c == "\\":
                out.append(True); i += 2; continue
            if c == q:
                inside = False
        else:
            if c in "\"'":
                inside, q = True, c; out.append(True)
            else:
                out.append(False)
        i += 1
    return out[:len(line)] + [False] * (len(line) - len(out))


def comment_start(line, st):
    for i in range(len(line) - 1):
        if not st[i] and line[i:i + 2] == "//":
            return i
    return -1


def wrap(line):
    if len(line) <= LIMIT or line.lstrip().startswith(("#", "//", "/*", "* ")) or line.rstrip().endswith("\\"):      # (comment lines: tools/wrap_comments.py)
        return [line]
    indent = line[:len(line) - len(line.lstrip())]
    st = scan(line)
    out = []
    cs = comment_start(line, st)
    if cs > 0 and line[:cs].strip():
        out.append(indent + line[cs:].rstrip())
        line = line[:cs].rstrip(); st = scan(line)
        if len(out[0]) > LIMIT:
            out = []            # (a comment that is itself too long: wrap_comments.py's business)
            return [line + "  " + ""] if False else [indent + l for l in []] or wrap_long_comment(indent, out, line, cs)
    cont = indent + "    "
    first = True
    while len(line) > LIMIT:
        cut = -1
        depth, d = [], 0                      # parenthesis depth in front of every character ("; " inside a for (...) header is no place to break)
        for i, ch in enumerate(line):
            depth.append(d)
            if not st[i]:
                d += ch == "("; d -= ch == ")"
        # statement boundaries first ("; " outside every parenthesis, in the last 60 columns), then the other places
        for want in (("; ",), BREAKS):
            lo = LIMIT - 62 if want == ("; ",) else len(indent) + 20
            for i in range(min(LIMIT - 2, len(line) - 1), lo, -1):
                if st[i]:
                    continue
                for b in want:
                    if line[i - len(b) + 1:i + 1] == b and not any(st[i - len(b) + 1:i + 1]) and not (b in ("; ", " = ") and depth[i] > 0):
                        cut = i + 1; break
                if cut > 0:
                    break
            if cut > 0:
                break
        if cut < 0:
            # a string literal that overflows on its own: cut it at a space into two adjacent literals
            for i in range(LIMIT - 3, len(indent) + 20, -1):
                if st[i] and line[i] == " " and st[i + 1] and line[i - 1] != "\\":
                    out.append(line[:i + 1] + "\"")
                    line = cont + "\"" + line[i + 1:]
                    st = scan(line); cut = 0
                    break
            if cut < 0:
                break
            continue
        out.append(line[:cut].rstrip())
        line = (cont if first or True else indent) + line[cut:].lstrip()
        st = scan(line); first = False
    out.append(line)
    return out


def wrap_long_comment(indent, out, line, cs):
    return [line]


for path in sys.argv[1:]:
    src = open(path).read().split("\n")
    res, changed = [], 0
    for l in src:
        w = wrap(l)
        if len(w) > 1:
            changed += 1
        res.extend(w)
    if changed:
        open(path, "w").write("\n".join(res))
    print(path, "wrapped", changed, "lines;", sum(1 for l in res if len(l) > LIMIT), "still longer than", LIMIT)
