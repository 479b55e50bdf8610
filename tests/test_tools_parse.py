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


def test_bench_numpy_worker_runs_without_torch_or_gpu():
    """bench.py --numpy-worker: one process of the cpu_baseline's `numpy_nproc` leg.  It must start without importing torch
    (16 of them run beside the GPU process) and print its compute seconds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, runpy; sys.argv = ['bench.py', '--numpy-worker', 'multigas', '3', '4000', '30', '1000', '3000']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules, 'the worker imported torch'\n" % os.path.join(root, "bench.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("numpy-worker")]
    assert len(line) == 1 and float(line[0].split()[1]) > 0.0
