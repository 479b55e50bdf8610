"""The measurement tools are part of the evidence chain (profiles/README.md names the command behind every file):
keep them importable-as-source.  Running them needs the GPU box; this only checks that they still parse and that the
collection script references files that exist."""
import ast
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tools_and_entry_points_parse():
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + [os.path.join(ROOT, n) for n in
                                                              ("bench.py", "__graft_entry__.py", "example/concentrations.py")]
    assert len(files) >= 15
    for path in files:
        with open(path) as fh:
            ast.parse(fh.read(), filename=path)


def test_collect_script_references_existing_tools():
    with open(os.path.join(ROOT, "tools", "collect_profiles.sh")) as fh:
        text = fh.read()
    for rel in sorted(set(re.findall(r"\$R/((?:tools/)?[A-Za-z0-9_]+\.py)", text))):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    # rocprofv3 is always handed the program itself (python3 <script>), never a shell / env wrapper, and --pmc passes
    # carry --kernel-trace only
    for line in text.splitlines():
        if line.strip().startswith("rocprofv3"):
            assert re.search(r" -- python3 \$R/", line), line
            if "--pmc" in line:
                assert "--sys-trace" not in line and "--hip-trace" not in line and "--hsa-trace" not in line and "--stats" not in line
