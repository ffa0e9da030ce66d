#!/usr/bin/env python3
"""Re-flows the prose of a Markdown file at WIDTH columns without changing what it renders to: paragraphs and list items are joined and wrapped
again (so that repeated runs leave no orphan words behind); fenced code, tables, headings, quotes, indented (4+) lines and blank lines are left
alone; a wrapped list item continues under its text.  python tools/wrap_md.py DESIGN.md [width]"""
import re, sys, textwrap
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 150
BULLET = re.compile(r"^(\s{0,3})((?:[-*+]|\d+[.)])\s+)")
lines = open(path).read().split("\n")
out, fence, i = [], False, 0
def verbatim(l):
    return (not l.strip()) or l.lstrip().startswith(("|", "#", ">", "```")) or l.startswith("    ") or l.startswith("\t")
def wrap(first_prefix, sub, text):
    w = textwrap.wrap(" ".join(text.split()), width=W - len(sub), break_long_words=False, break_on_hyphens=False) or [""]
    fixed = []
    for x in w:                     # a continuation line must not start with something Markdown reads as a new block
        if fixed and re.match(r"^([-*+]\s|\d+[.)]\s|#|>|\||```)", x):
            fixed[-1] += " " + x
        else:
            fixed.append(x)
    return [first_prefix + fixed[0]] + [sub + x for x in fixed[1:]]
while i < len(lines):
    l = lines[i]
    if l.lstrip().startswith("```"):
        fence = not fence; out.append(l); i += 1; continue
    if fence or verbatim(l):
        out.append(l); i += 1; continue
    m = BULLET.match(l)
    if m:
        indent, bullet = m.group(1), m.group(2)
        sub = indent + " " * len(bullet)
        text = [l[len(indent) + len(bullet):]]
        i += 1
        # continuation lines of this item: indented under its text (or lazy, unindented prose), up to a blank line, a new item or a verbatim line
        while i < len(lines) and lines[i].strip() and not BULLET.match(lines[i]) and not lines[i].lstrip().startswith(("|", "#", ">", "```")) and not (lines[i].startswith("    ") and not lines[i].startswith(sub)):
            text.append(lines[i].strip()); i += 1
        out += wrap(indent + bullet, sub, " ".join(text))
        continue
    # a paragraph: up to a blank line, a list item or a verbatim line; its indentation (0-3 spaces, e.g. under a list item) is kept
    indent = re.match(r"^(\s{0,3})", l).group(1)
    text = [l.strip()]
    i += 1
    while i < len(lines) and not verbatim(lines[i]) and not BULLET.match(lines[i]) and not lines[i].lstrip().startswith("```"):
        text.append(lines[i].strip()); i += 1
    out += wrap(indent, indent, " ".join(text))
open(path, "w").write("\n".join(out))
