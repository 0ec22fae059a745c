#!/usr/bin/env python3
"""Re-flow blocks of `#` comment lines of Python sources at 160 columns: consecutive pure-comment lines of one indentation are one paragraph (a line that starts
with a bullet, `(`-less code-like text is not special-cased: blocks holding a line over the limit are re-flowed whole, the others are left alone); a code line that
is over-long because of its trailing comment gets the comment moved onto lines of its own above it.    python scripts/reflow_py_comments.py file ..."""
import re
import sys
import textwrap

LIMIT = 160


def reflow(path):
    lines = open(path).read().split("\n")
    out, i, changed = [], 0, 0
    while i < len(lines):
        m = re.match(r"^(\s*)# ?(.*)$", lines[i])
        if m and not lines[i].lstrip().startswith("#!"):
            indent = m.group(1)
            j, block = i, []
            while j < len(lines):
                q = re.match(r"^(\s*)# ?(.*)$", lines[j])
                if not q or q.group(1) != indent or not q.group(2).strip():
                    break
                block.append(q.group(2))
                j += 1
            if block and any(len(lines[k]) > LIMIT for k in range(i, j)):
                text = " ".join(b.strip() for b in block)
                out.extend(indent + "# " + p for p in textwrap.wrap(text, width=LIMIT - len(indent) - 2, break_long_words=False, break_on_hyphens=False))
                changed += 1
                i = j
                continue
            if block:
                out.extend(lines[i:j]); i = j
                continue
        line = lines[i]
        if len(line) > LIMIT:
            t = re.match(r"^(\s*)(\S.*?\S)\s{2,}#\s?(.*)$", line)
            if t and t.group(2).count('"') % 2 == 0 and t.group(2).count("'") % 2 == 0 and "#" not in t.group(2):
                indent, code, text = t.groups()
                out.extend(indent + "# " + p for p in textwrap.wrap(text, width=LIMIT - len(indent) - 2, break_long_words=False, break_on_hyphens=False))
                out.append(indent + code)
                changed += 1
                i += 1
                continue
        out.append(line)
        i += 1
    if changed:
        open(path, "w").write("\n".join(out))
    return changed


if __name__ == "__main__":
    for p in sys.argv[1:]:
        n = reflow(p)
        if n:
            print(f"{p}: {n} comment blocks re-flowed at {LIMIT} columns")
