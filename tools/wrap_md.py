#!/usr/bin/env python3
"""Re-wraps the long lines of a Markdown file at WIDTH columns without changing what it renders to: fenced code, tables, headings and
reference-style lines are left alone; a wrapped list item continues under its text.  python tools/wrap_md.py DESIGN.md [width]"""
import re, sys, textwrap
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 150
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence; out.append(line); continue
    if fence or len(line) <= W or line.lstrip().startswith(("|", "#", ">")) or line.startswith("    "):
        out.append(line); continue
    m = re.match(r"^(\s*)((?:[-*+]|\d+[.)])\s+|\*\([^)]*\)\*\s+)?", line)
    indent, bullet = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(bullet):]
    sub = indent + " " * len(bullet)
    wrapped = textwrap.wrap(body, width=W - len(sub), break_long_words=False, break_on_hyphens=False)
    # a continuation line must not start with something Markdown reads as a new block
    fixed = []
    for w in wrapped:
        if fixed and re.match(r"^([-*+]\s|\d+[.)]\s|#|>|\|)", w):
            fixed[-1] += " " + w
        else:
            fixed.append(w)
    out.append(indent + bullet + fixed[0])
    out += [sub + w for w in fixed[1:]]
open(path, "w").write("\n".join(out))
