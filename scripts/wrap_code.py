#!/usr/bin/env python3
"""Break over-long CODE lines (> 160 columns) of C++/HIP sources at an argument boundary: the last ", " (or " && " / " || " / " ? " / " : ") outside quotes before column
158, the continuation aligned behind the line's first opening parenthesis (or indented by 8 where that would leave too little room).  Preprocessor lines, comment lines and
lines with a trailing comment are left alone; a line whose only break point lies inside a string literal gets the literal split at a space (adjacent literals concatenate) (scripts/wrap_comments.py handles those).    python scripts/wrap_code.py file ..."""
import re
import sys

LIMIT = 160


def split_once(line):
    if len(line) <= LIMIT:
        return None
    stripped = line.lstrip()
    if stripped.startswith(("#", "//", "*", "/*")) or line.rstrip().endswith("\\") or "//" in line:
        return None
    indent = len(line) - len(stripped)
    first = line.find("(")
    cont = first + 1 if 0 <= first < 90 else indent + 8
    cands, strcands, quote, depth = [], [], None, 0   # (depth, position) of the break points outside quotes; spaces inside string literals
    i = 0
    while i < min(len(line), LIMIT - 3):
        c = line[i]
        if quote:
            if c == "\\":
                i += 1
            elif c == quote:
                quote = None
            elif c == " " and quote == '"' and i > cont + 20:
                strcands.append(i)
        elif c in "\"'":
            quote = c
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        elif i > cont + 20:
            if line.startswith(", ", i) or (line.startswith("; ", i) and depth == 0):
                cands.append((depth, i + 1))
            elif any(line.startswith(op, i) for op in (" && ", " || ", " ? ", " : ")) or any(line.startswith(op, i) for op in (" & ", " | ", " + ")):
                cands.append((depth, i))
        i += 1
    # the shallowest break point, and among those the last one that leaves a head of some length
    good = [c for c in cands if c[1] >= 60] or cands
    if good:
        d = min(c[0] for c in good)
        target = max(0.62 * len(line), len(line) - (LIMIT - cont - 12))   # a balanced split, not a stub of a tail
        best = min((c[1] for c in good if c[0] == d), key=lambda pos: abs(pos - target))
        if d <= 2:
            return line[:best].rstrip(), " " * cont + line[best:].lstrip()
    if strcands:   # a long string literal: adjacent literals concatenate -- break it at a space
        last = strcands[-1]
        return line[:last + 1] + '"', " " * cont + '"' + line[last + 1:]
    return None


def wrap(path):
    out, n = [], 0
    for line in open(path).read().split("\n"):
        while True:
            r = split_once(line)
            if not r:
                break
            out.append(r[0])
            line = r[1]
            n += 1
        out.append(line)
    if n:
        open(path, "w").write("\n".join(out))
        print(f"{path}: {n} breaks")


if __name__ == "__main__":
    for f in sys.argv[1:]:
        wrap(f)
