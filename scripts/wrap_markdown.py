#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file at 160 columns: paragraphs and list items (hanging indent), block quotes; tables, headings, fenced code and blank lines stay as
they are (a table row is one line by definition).  A line that starts a list item, a `**bold lead-in**` or a block quote starts a new run; the lines after it up to the
next such line are its continuation -- which is how Markdown reads them anyway: the rendering does not change.    python scripts/wrap_markdown.py DESIGN.md [...]"""
import re
import sys
import textwrap

LIMIT = 160
START = re.compile(r"^(\s*)((?:[*+-]|\d+[.)])\s+|>\s?)(.*)$")


def flush(run, out):
    if not run:
        return
    indent, marker, texts = run
    hang = indent + ("> " if marker.startswith(">") else " " * len(marker))
    parts = textwrap.wrap(" ".join(t.strip() for t in texts if t.strip()), width=LIMIT - len(hang), break_long_words=False, break_on_hyphens=False) or [""]
    out.append(indent + marker + parts[0])
    out.extend(hang + p for p in parts[1:])


def wrap(path):
    out, fence, run = [], False, None
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            flush(run, out); run = None
            fence = not fence
            out.append(line)
            continue
        if fence or not line.strip() or line.lstrip().startswith(("|", "#")) or re.match(r"^\s*(-{3,}|={3,})\s*$", line):
            flush(run, out); run = None
            out.append(line)
            continue
        m = START.match(line)
        if m:
            flush(run, out)
            run = (m.group(1), m.group(2), [m.group(3)])
        elif line.lstrip().startswith("**") or run is None:
            flush(run, out)
            lead = re.match(r"^(\s*)", line).group(1)
            run = (lead if len(lead) <= 1 else "", "", [line.strip()]) if run is None or line.lstrip().startswith("**") else run
            if len(lead) > 1 and run[2] == [line.strip()]:
                run = (lead, "", [line.strip()])
        else:
            run[2].append(line)
    flush(run, out)
    open(path, "w").write("\n".join(out))
    print(f"{path}: re-flowed at {LIMIT} columns, {len(out)} lines")


if __name__ == "__main__":
    for f in sys.argv[1:]:
        wrap(f)
