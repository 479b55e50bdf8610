"""Every number DESIGN.md (the design as built, round 6) quotes from a tracked JSON file must be IN that file.

A cell that quotes a measurement writes it as   **number** (`file.json[key].field.sub` / scale)   — the bold number, then in
backticks the tracked file (under profiles/r06/, then profiles/), an optional top-level [key] (keys of valu.json / traffic.json
contain ':' and ','), a dotted path, and an optional '/ scale' or 'x scale'.  This test parses all of them and fails on a
mismatch beyond the rounding of the printed digits (plus 0.2 %), so that the document cannot drift from the files the way
round 2's did (391 vs 374.4 VALU per wave-step).  It also pins profiles/valu.json and profiles/traffic.json — the copies
bench.py reads — to the round's collection under profiles/r06/.  (profiles/r03/NOTES.md, round 3's text, was held to the same
rule until the per-round copies of the counter files it cites were pruned in round 5; it is frozen history.)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"\*\*(?P<num>[-+]?[0-9][0-9.,]*(?:e[-+]?[0-9]+)?)\*\*[^`|\n]{0,60}?\(`(?P<file>[A-Za-z0-9_./-]+\.json)"
                  r"(?:\[(?P<key>[^\]]+)\])?(?P<path>(?:\.[A-Za-z0-9_]+)+)`(?:\s*(?P<op>[/x])\s*(?P<scale>[0-9.e+-]+))?(?=[);,])")
# a bold number followed by a backticked .json path that the pattern above does NOT parse is a malformed citation
LOOSE = re.compile(r"\*\*[-+]?[0-9][0-9.,e+-]*\*\*[^`|\n]{0,60}?\(`[A-Za-z0-9_./-]+\.json")


DOCS = {"DESIGN.md": (os.path.join("profiles", "r06"), "profiles")}


def _resolve(name, bases):
    for base in bases:
        path = os.path.join(ROOT, base, name)
        if os.path.exists(path):
            return path
    raise AssertionError(f"{name} is cited but is not a tracked file under {bases}")


def _lookup(doc, key, path):
    if key is not None:
        doc = doc[key]
    for part in path.strip(".").split("."):
        doc = doc[int(part)] if isinstance(doc, list) else doc[part]
    return float(doc)


import pytest  # noqa: E402


@pytest.mark.parametrize("doc", sorted(DOCS))
def test_document_numbers_are_in_the_files_they_cite(doc):
    with open(os.path.join(ROOT, doc)) as fh:
        text = fh.read()
    cites = list(CITE.finditer(text))
    assert len(cites) >= 12, f"only {len(cites)} file-backed numbers found in {doc}: the citation format changed?"
    assert len(cites) == len(LOOSE.findall(text)), "a bold number cites a .json file in a form this test cannot parse"
    # ... and a citation that sits on the NEXT line, or further than 60 characters behind its number, is not parsed either: a
    # bold number whose following text reaches a backticked .json path before the next bold number must be a parsed one
    parsed = {m.start() for m in cites}
    for m in re.finditer(r"\*\*[-+]?[0-9][0-9.,]*(?:e[-+]?[0-9]+)?\*\*", text):
        if m.start() in parsed:
            continue
        tail = text[m.end():m.end() + 200].split("**")[0]
        assert not re.search(r"\(`[A-Za-z0-9_./-]+\.json", tail), (
            f"{doc} line {text.count(chr(10), 0, m.start()) + 1}: {m.group(0)} is followed by a .json citation this test does "
            f"not parse (put `(file.json.path)` right behind the number, on the same line)")
    cache, bad = {}, []
    for m in cites:
        path = _resolve(m["file"], DOCS[doc])
        if path not in cache:
            with open(path) as fh:
                cache[path] = json.load(fh)
        try:
            val = _lookup(cache[path], m["key"], m["path"])
        except (KeyError, IndexError, TypeError) as exc:
            bad.append(f"{m.group(0)}: no such entry ({exc!r})")
            continue
        if m["scale"]:
            val = val / float(m["scale"]) if m["op"] == "/" else val * float(m["scale"])
        shown = m["num"].replace(",", "")
        digits = len(shown.split(".")[1].split("e")[0]) if "." in shown else 0
        exp = float("1" + shown[shown.index("e"):]) if "e" in shown else 1.0
        tol = 0.5 * 10.0 ** (-digits) * exp + 2e-3 * abs(val)
        if abs(float(shown) - val) > tol:
            bad.append(f"{doc} says {shown}, {m['file']}{'[' + m['key'] + ']' if m['key'] else ''}{m['path']} holds {val:.6g}")
    assert not bad, "\n".join(bad)


def test_bench_reads_this_rounds_counter_files():
    for name in ("valu.json", "traffic.json"):
        with open(os.path.join(ROOT, "profiles", name)) as a, open(os.path.join(ROOT, "profiles", "r06", name)) as b:
            assert json.load(a) == json.load(b), f"profiles/{name} is not profiles/r06/{name}"


def test_design_md_stays_a_design_document():
    """The design as built in at most 28 KB (review, round 3: 59 KB, half of it A/B history — that lives in profiles/r0N/ now;
    round 5 added the small-ensemble kernel and what the N > 1 line carries: 20 -> 24 KB; round 6 the compensated fp32 form, the
    octet kernel, the placement finding and the host binding: 24 -> 28 KB)."""
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 28 * 1024
