"""roofline.traffic measured by the bench run itself: child `rocprofv3 --pmc` passes started BEFORE the bench process touches the
GPU (this module imports neither torch nor the engine)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def live_traffic(kind, dtype, members, timeout_s=240):
    """HBM bytes per launch of the per-step kernel from PMC counters, collected BY THIS RUN: `rocprofv3 --pmc FETCH_SIZE
    --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes: the two do not fit one; no other trace domain), the
    program directly after `--` (python3 tools/pmc_workload.py: five calibration copies of known byte count, then 60 timesteps of
    this workload), reduced exactly like the committed figure (tools/pmc_traffic.py: the guide's gfx950 correction comes out of
    the calibration on the copy).  Child processes of a process that has not touched the GPU.  {"hbm_bytes_per_launch", ...} or
    {"error": ...}: a failure costs the line nothing but the live figure (the committed one is reported instead, labelled)."""
    import shutil
    import subprocess
    import tempfile
    t0 = time.perf_counter()
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix="fiveeq_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", FIVEEQ_SIDE_STREAM_PROBE="0")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, counter), "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), str(members), kind, dtype]
            out = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if out.returncode != 0:
                return {"error": f"rocprofv3 --pmc {counter} exited {out.returncode}: {out.stderr[-300:]}"}
        rec = pmc_traffic.reduce(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), 1 << 27)
        rec["seconds"] = time.perf_counter() - t0
        rec["source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
                         f"tools/pmc_workload.py {members} {kind} {dtype} as child processes before the GPU was touched, calibrated on the "
                         "copy kernel of known byte count in the same pass (tools/pmc_traffic.py)")
        return rec
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
