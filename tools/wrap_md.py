#!/usr/bin/env python3
"""Reflows the prose of a Markdown file to 120 columns: paragraphs and list items are joined and re-wrapped (hanging indent kept);
tables, code fences, headings and blank lines are left alone.  usage: python tools/wrap_md.py FILE [FILE...]"""
import re
import sys
import textwrap

WIDTH = 120
BULLET = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")


def flush(group, out):
    if not group:
        return
    m = BULLET.match(group[0])
    if m:
        first, rest = m.group(1) + m.group(2), m.group(1) + " " * len(m.group(2))
        body = group[0][len(first):]
    else:
        lead = re.match(r"^(\s*)", group[0]).group(1)
        first = rest = lead
        body = group[0][len(lead):]
    text = " ".join([body.strip()] + [g.strip() for g in group[1:]])
    text = re.sub(r"(?<=[.:;)])  +", "  ", text)  # keep the double space after a sentence end as written
    out.extend(textwrap.wrap(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()])
    group.clear()


def wrap_file(path):
    out, group, fence = [], [], False
    for line in open(path, encoding="utf-8").read().split("\n"):
        stripped = line.lstrip()
        if stripped.startswith("```"):
            flush(group, out)
            fence = not fence
            out.append(line)
            continue
        if fence or not stripped or stripped.startswith(("|", "#", "<", ">")):
            flush(group, out)
            out.append(line)
            continue
        if BULLET.match(line):  # a new list item starts a new group
            flush(group, out)
        group.append(line)
    flush(group, out)
    open(path, "w", encoding="utf-8").write("\n".join(out))


for p in sys.argv[1:]:
    wrap_file(p)
