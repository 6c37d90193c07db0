#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file to at most 160 columns: paragraphs and list items that contain a longer line are re-wrapped (list items with a hanging
indent); tables, headings, fenced code and indented code are left alone.  usage: wrap_md.py <file> ..."""
import re
import sys
import textwrap

LIMIT = 160
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    out, i, fenced, changed = [], 0, False, 0
    item = re.compile(r"^(\s*)([*\-+]|\d+[.)])\s+")
    def special(l):
        return not l.strip() or l.lstrip().startswith(("|", "#", "```", ">")) or l.startswith("    ")
    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fenced = not fenced; out.append(l); i += 1; continue
        if fenced or special(l):
            out.append(l); i += 1; continue
        m = item.match(l)
        first_prefix = m.group(0) if m else ""
        cont_prefix = " " * len(first_prefix) if m else ""
        block = [l[len(first_prefix):] if m else l]; j = i + 1
        while j < len(lines) and not special(lines[j]) and not item.match(lines[j]) and not fenced:
            block.append(lines[j].strip()); j += 1
        if any(len(x) > LIMIT for x in lines[i:j]):
            text = " ".join(b.strip() for b in block)
            w = textwrap.wrap(text, LIMIT - len(first_prefix), break_long_words=False, break_on_hyphens=False)
            out.extend((first_prefix if n == 0 else cont_prefix) + x for n, x in enumerate(w)); changed += 1
        else:
            out.extend(lines[i:j])
        i = j
    if changed:
        open(path, "w").write("\n".join(out))
    print(path, "re-flowed", changed, "paragraphs;", sum(1 for x in out if len(x) > LIMIT and not x.lstrip().startswith("|")), "non-table lines still longer than", LIMIT)
