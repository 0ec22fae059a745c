#!/usr/bin/env python3
"""Wrap over-long PURE comment lines (`// ...`, nothing but the comment on the line) of C++/HIP sources at 160 columns, keeping the indentation and a bullet's hanging indent.
A code line that is over-long because of its trailing comment gets the comment moved onto lines of its own above it.    python scripts/wrap_comments.py [--check] file ..."""
import re
import sys
import textwrap

LIMIT = 160


def wrap_file(path, check):
    out, changed = [], 0
    for line in open(path).read().split("\n"):
        m = re.match(r"^(\s*)//( *)(.*)$", line)
        if not m and len(line) > LIMIT and not line.rstrip().endswith("\\"):
            # a code line with a trailing comment: the comment moves onto lines of its own above the code (same indentation)
            t = re.match(r"^(\s*)(\S.*?\S)\s{2,}//\s?(.*)$", line)
            if t and t.group(2).count('"') % 2 == 0 and "//" not in t.group(2):
                indent, code, text = t.groups()
                parts = textwrap.wrap(text, width=LIMIT - len(indent) - 3, break_long_words=False, break_on_hyphens=False)
                out.extend(indent + "// " + q for q in parts)
                out.append(indent + code)
                changed += 1
                continue
        if not m or len(line) <= LIMIT or m.group(3).startswith(("!", "/")):
            out.append(line)
            continue
        indent, gap, text = m.groups()
        bullet = re.match(r"^((?:[*-]|\d+[.)]|\(\w\))\s+)", text)
        hang = " " * len(bullet.group(1)) if bullet else ""
        first = indent + "//" + gap
        width = LIMIT - len(first)
        parts = textwrap.wrap(text, width=width, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False)
        out.extend(first + p for p in parts)
        changed += 1
    if changed and not check:
        open(path, "w").write("\n".join(out))
    return changed


if __name__ == "__main__":
    check = "--check" in sys.argv
    total = 0
    for f in [a for a in sys.argv[1:] if a != "--check"]:
        n = wrap_file(f, check)
        if n:
            print(f"{f}: {n} comment lines {'over' if check else 'wrapped at'} {LIMIT} columns")
        total += n
    sys.exit(1 if check and total else 0)
