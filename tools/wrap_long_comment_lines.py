#!/usr/bin/env python3
"""Split pure comment lines longer than 160 characters at a word boundary (same prefix on the continuation): the last resort after tools/wrap_comments.py
(paragraph re-flow) for block-comment bodies (' * ...') and lines it leaves alone.  usage: wrap_long_comment_lines.py FILE..."""
import re
import sys

LIMIT = 160
PREFIX = re.compile(r"^(\s*(?://+|\*|#)\s?)(\s*)")


def split(line):
    m = PREFIX.match(line)
    if not m or line.lstrip().startswith("#define") or line.lstrip().startswith("#pragma") or line.lstrip().startswith("#include"):
        return [line]
    out = []
    pre = m.group(1) + m.group(2)
    while len(line) > LIMIT:
        cut = line.rfind(" ", len(pre) + 20, LIMIT + 1)
        if cut < 0:
            break
        out.append(line[:cut].rstrip())
        line = pre + line[cut + 1:]
    out.append(line)
    return out


for path in sys.argv[1:]:
    src = open(path).read().split("\n")
    res = []
    n = 0
    for l in src:
        if len(l) > LIMIT:
            parts = split(l)
            n += len(parts) > 1
            res.extend(parts)
        else:
            res.append(l)
    open(path, "w").write("\n".join(res))
    left = sum(len(l) > LIMIT for l in res)
    print(f"{path}: split {n} lines; {left} still longer than {LIMIT}")
