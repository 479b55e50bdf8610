#!/usr/bin/env python3
"""Rewrite every file-backed number of DESIGN.md (the **number** (`file[key].path` / scale) cells that
tests/test_design_numbers.py checks) from the file it cites, keeping the number of decimals shown.  Run after the profiles
were re-collected; the test then passes by construction, and `git diff DESIGN.md` shows what moved."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_design_numbers as t  # noqa: E402

path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
cache, out, last, changed = {}, [], 0, 0
for m in t.CITE.finditer(text):
    f = t._resolve(m["file"], t.DOCS["DESIGN.md"])
    if f not in cache:
        cache[f] = json.load(open(f))
    val = t._lookup(cache[f], m["key"], m["path"])
    if m["scale"]:
        val = val / float(m["scale"]) if m["op"] == "/" else val * float(m["scale"])
    shown = m["num"]
    digits = len(shown.split(".")[1]) if "." in shown else 0
    new = f"{val:.{digits}f}"
    if new != shown:
        changed += 1
        print(f"{m['file']}{m['path']}: {shown} -> {new}")
    out.append(text[last:m.start("num")])
    out.append(new)
    last = m.end("num")
out.append(text[last:])
open(path, "w").write("".join(out))
print(changed, "numbers rewritten")
