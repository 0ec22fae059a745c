#!/usr/bin/env python3
"""Print one kernel's gfx950 assembly (or a summary of it) from a `hipcc -save-temps=obj` .s file.
usage: kernel_asm.py file.s <substring of the mangled name> [spills|count|dump]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "count"
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if pat in m.group(1)]
for name in names:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    print("==", name, len(body), "lines")
    if mode == "dump":
        print("\n".join(body))
    elif mode == "spills":
        for n, l in enumerate(body):
            if "scratch_" in l:
                print(n, l.strip())
    else:
        cnt = {}
        for l in body:
            l = l.strip()
            if not l or l.startswith((";", ".", "_Z")) or l.endswith(":"):
                continue
            op = l.split()[0]
            key = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
            cnt[key] = cnt.get(key, 0) + 1
        print(cnt)
