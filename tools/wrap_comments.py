#!/usr/bin/env python3
"""Re-flow comment paragraphs (runs of comment-only lines, // ... or # ..., same indentation) that contain a line longer than 160 columns -- or an orphan of one or two
words left behind by an earlier line-by-line wrap -- to at most 160 columns.  Code lines, tables, indented continuation lines (two or more spaces after the marker)
and list items start new paragraphs and keep their own first-line prefix.  usage: wrap_comments.py <file> ..."""
import re
import sys
import textwrap

LIMIT = 160


def split(line, py):
    body = line.lstrip()
    mark = "//" if body.startswith("//") else "#" if py and body.startswith("#") and not body.startswith("#!") else None
    if mark is None or (mark == "//" and body.startswith("///")):
        return None
    indent = line[:len(line) - len(body)]
    text = body[len(mark):]
    return indent, mark, text


for path in sys.argv[1:]:
    py = path.endswith((".py", ".sh"))
    lines = open(path).read().split("\n")
    out, i, changed = [], 0, 0
    while i < len(lines):
        sp = split(lines[i], py)
        if sp is None:
            out.append(lines[i]); i += 1; continue
        indent, mark, text = sp
        if re.search(r"[-=]{8,}", text):                       # a section rule ("// ---- title ------"): never part of a paragraph; trimmed to the limit if too long
            l = lines[i]
            if len(l) > LIMIT:
                l = re.sub(r"([-=]{8,})\s*$", lambda m: m.group(1)[:max(8, len(m.group(1)) - (len(l) - LIMIT))], l.rstrip())
            out.append(l); i += 1; continue
        # a paragraph: this line + following comment-only lines of the same indent whose text starts with exactly one space and no list / table marker
        para = [text]; j = i + 1
        while j < len(lines):
            s2 = split(lines[j], py)
            if s2 is None or s2[0] != indent or s2[1] != mark:
                break
            t2 = s2[2]
            if not t2.startswith(" ") or t2.startswith("  ") or re.match(r" (\*|-|\d+[.)]|\||[A-Za-z0-9_]+ {2,})", t2) or not t2.strip() or re.search(r"[-=]{8,}", t2):
                break
            para.append(t2); j += 1
        block = lines[i:j]
        too_long = any(len(l) > LIMIT for l in block)
        orphan = any(len(block[k].split()) <= 3 and len(block[k - 1]) >= LIMIT - 25 for k in range(1, len(block)))
        # ... or a short line in the middle of flowing text (what a repaired section rule leaves behind)
        orphan = orphan or any(len(block[k]) < LIMIT - 50 and len(block[k + 1]) >= LIMIT - 15 for k in range(len(block) - 1))
        if not (too_long or orphan) or not text.strip():
            out.extend(block); i = j; continue
        lead = text[:len(text) - len(text.lstrip())] or " "
        joined = " ".join(p.strip() for p in para)
        joined = re.sub(r"  +", "  ", joined)
        first_w = LIMIT - len(indent) - len(mark) - len(lead)
        cont_lead = lead if len(lead) <= 2 else " " * len(lead)
        wrapped = textwrap.wrap(joined, first_w, break_long_words=False, break_on_hyphens=False)
        out.extend(indent + mark + (lead if n == 0 else cont_lead) + w for n, w in enumerate(wrapped))
        changed += 1; i = j
    if changed:
        open(path, "w").write("\n".join(out))
    print(path, "re-flowed", changed, "comment paragraphs;", sum(1 for l in out if len(l) > LIMIT and split(l, py) is not None), "comment lines still longer than", LIMIT)
