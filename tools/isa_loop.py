#!/usr/bin/env python3
"""Summarise the innermost loop of a kernel in a hipcc `-S` dump: loads (L), scalar loads (S), waits (W), fp ops (f).
usage: isa_loop.py file.s <mangled-name-prefix>"""
import itertools
import re
import sys

s = open(sys.argv[1]).read()
lines = s.split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print("loops (line ranges):", loops)
a, b = max(loops, key=lambda t: t[1] - t[0])
seq = []
for l in body[a:b + 1]:
    l = l.strip()
    if l.startswith("global_load") or l.startswith("buffer_load"):
        seq.append("L")
    elif l.startswith("global_store"):
        seq.append("ST")
    elif l.startswith("ds_read") or l.startswith("ds_load"):
        seq.append("D")
    elif l.startswith("s_waitcnt"):
        seq.append("W(" + l.split(None, 1)[1] + ")")
    elif re.match(r"v_(fma|mul|add|pk_fma|pk_mul|pk_add)_f(64|32)", l):
        seq.append("f")
    elif l.startswith("s_load"):
        seq.append("S")
    elif l.startswith("scratch_"):
        seq.append("SCR")
out = []
for k, g in itertools.groupby(seq):
    n = len(list(g))
    out.append(k if n == 1 else "%s*%d" % (k, n))
print(" ".join(out))
