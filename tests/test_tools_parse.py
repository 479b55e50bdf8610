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


def test_pmc_traffic_reducer_on_a_synthetic_counter_table(tmp_path):
    """tools/pmc_traffic.reduce — the reducer behind profiles/traffic.json AND behind the traffic figure bench.py measures itself
    (benchlib.traffic.live_traffic) — on a hand-made rocprofv3 counter table: KiB units, the calibration on the copy kernel of known byte
    count in the same pass (which is where the guide's gfx950 FETCH_SIZE correction comes from), means over dispatches."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
            '"Start_Timestamp","End_Timestamp"\n')
    copy_elems = 1 << 20
    known = copy_elems * 8.0

    def table(path, counter, copy_kib, step_kib):
        os.makedirs(path)
        rows = []
        for i, (name, val) in enumerate([("fiveeq::stream_copy_kernel(long, double const*, double*)", v) for v in copy_kib]
                                        + [("void fiveeq::step_kernel<double, 4, 1, 1, false, false>(...)", v) for v in step_kib]
                                        + [("void at::native::vectorized_elementwise_kernel<4>(...)", 123.0)]):
            rows.append(f'{i},{i},"Agent 2",1,77,77,1000,5,"{name}",64,0,0,8,0,48,"{counter}",{val},{1000 * i},{1000 * i + 500}\n')
        with open(os.path.join(path, "1_counter_collection.csv"), "w") as fh:
            fh.write(head + "".join(rows))

    # the copy's reads are counted at HALF their bytes (the gfx950 FETCH_SIZE behaviour), its writes exactly
    table(str(tmp_path / "f" / "run"), "FETCH_SIZE", [known / 2 / 1024] * 5, [3000.0, 3100.0, 2900.0])
    table(str(tmp_path / "w" / "run"), "WRITE_SIZE", [known / 1024] * 5, [2000.0] * 3)
    rec = pmc_traffic.reduce(str(tmp_path / "f"), str(tmp_path / "w"), copy_elems)
    assert rec["copy_calibration"]["fetch_factor"] == 2.0 and rec["copy_calibration"]["write_factor"] == 1.0
    assert rec["fetch_bytes"] == 3000.0 * 1024 * 2.0 and rec["write_bytes"] == 2000.0 * 1024
    assert rec["hbm_bytes_per_launch"] == rec["fetch_bytes"] + rec["write_bytes"] and rec["step_dispatches"] == [3, 3]
    assert rec["copy_calibration"]["copy_dispatches"] == [5, 5]
    import pytest
    os.remove(str(tmp_path / "w" / "run" / "1_counter_collection.csv"))
    with pytest.raises(KeyError):
        pmc_traffic.reduce(str(tmp_path / "f"), str(tmp_path / "w"), copy_elems)          # a pass without dispatches: loud, bench falls back
