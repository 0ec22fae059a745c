#!/usr/bin/env python3
"""Wrap the over-long prose lines of a Markdown file at 160 columns: paragraphs and list items (hanging indent), block quotes; tables, headings and fenced code stay
as they are (a table row is one line by definition).    python scripts/wrap_markdown.py DESIGN.md [...]"""
import re
import sys
import textwrap

LIMIT = 160


def wrap(path):
    out, fence, n = [], False, 0
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= LIMIT or line.lstrip().startswith(("|", "#")):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[*+-]|\d+[.)])\s+|>\s?)?(.*)$", line)
        indent, marker, text = m.group(1), m.group(2) or "", m.group(3)
        hang = indent + ("> " if marker.startswith(">") else " " * len(marker))
        parts = textwrap.wrap(text, width=LIMIT - len(hang), break_long_words=False, break_on_hyphens=False)
        out.append(indent + marker + parts[0])
        out.extend(hang + p for p in parts[1:])
        n += 1
    open(path, "w").write("\n".join(out))
    print(f"{path}: {n} lines wrapped at {LIMIT} columns")


if __name__ == "__main__":
    for f in sys.argv[1:]:
        wrap(f)
