#!/usr/bin/env python3
"""Re-wrap the paragraphs and bullets of a markdown file to 118 columns (headers, tables and code fences untouched).
usage: wrap_md.py FILE"""
import re
import sys
import textwrap

p = sys.argv[1]
lines = open(p, encoding="utf-8").read().split("\n")
res, buf, fence = [], [], False


def flush():
    global buf
    if not buf:
        return
    text = " ".join(x.strip() for x in buf)
    first = buf[0]
    indent = len(first) - len(first.lstrip())
    if re.match(r"^(\* |\d+\. )", text) and indent == 0:
        w = textwrap.fill(text, width=118, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False)
    else:
        w = textwrap.fill(text, width=118, initial_indent=" " * indent, subsequent_indent=" " * indent, break_long_words=False, break_on_hyphens=False)
    res.append(w)
    buf = []


for l in lines:
    if l.startswith("```"):
        flush()
        fence = not fence
        res.append(l)
        continue
    if fence or l.strip() == "" or l.startswith("#") or l.startswith("|"):
        flush()
        res.append(l)
        continue
    if re.match(r"^(\* |\d+\. )", l):
        flush()
        buf = [l]
        continue
    buf.append(l)
flush()
open(p, "w", encoding="utf-8").write("\n".join(res))
print(max(len(x) for x in "\n".join(res).split("\n")))
