#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSV output to HBM bytes per launch of the step kernel, with the gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
  * FETCH_SIZE / WRITE_SIZE are in KiB;
  * FETCH_SIZE under-reports wide coalesced streaming reads (x2 for 16 B/lane); other widths are
    uncalibrated -> we calibrate on fiveeq::stream_copy_kernel (same 8 B/lane access shape, known
    byte count: n*8 read, n*8 written, buffers far beyond the 256 MiB Infinity Cache) IN THE SAME PASS
    and scale the step kernel's counters by (known bytes / counted bytes) of the copy.

    python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <copy_elems> <key> [out.json] [step|fused]
"""
import csv
import glob
import json
import os
import sys


KERNEL = "fiveeq::step_kernel"


def per_kernel(dirname, counter):
    """mean counter value per dispatch, by kernel"""
    acc = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                key = "step" if KERNEL in name else ("copy" if "fiveeq::stream_copy_kernel" in name else None)
                if key:
                    acc.setdefault(key, []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def reduce(fetch_dir, write_dir, copy_elems, kernel=None):
    """HBM bytes per launch of `kernel` (default: the step kernel) from a FETCH_SIZE pass and a WRITE_SIZE pass, each calibrated
    on the copy kernel of known byte count in the same pass.  Raises KeyError when a pass holds no dispatch of either kernel."""
    global KERNEL
    if kernel is not None:
        KERNEL = kernel
    f = per_kernel(fetch_dir, "FETCH_SIZE")
    w = per_kernel(write_dir, "WRITE_SIZE")
    known = copy_elems * 8.0
    f_raw, w_raw = f["step"][0] * 1024.0, w["step"][0] * 1024.0
    f_cal = known / (f["copy"][0] * 1024.0)
    w_cal = known / (w["copy"][0] * 1024.0)
    return {
        "fetch_bytes_raw": f_raw, "write_bytes_raw": w_raw,
        "copy_calibration": {"known_bytes_each_way": known, "fetch_counted": f["copy"][0] * 1024.0,
                             "write_counted": w["copy"][0] * 1024.0, "fetch_factor": f_cal, "write_factor": w_cal,
                             "copy_dispatches": [f["copy"][1], w["copy"][1]]},
        "fetch_bytes": f_raw * f_cal, "write_bytes": w_raw * w_cal,
        "hbm_bytes_per_launch": f_raw * f_cal + w_raw * w_cal,
        "step_dispatches": [f["step"][1], w["step"][1]],
    }


def main():
    fetch_dir, write_dir, copy_elems, key = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    out_path = sys.argv[5] if len(sys.argv) > 5 else None
    rec = reduce(fetch_dir, write_dir, copy_elems, "fiveeq::fused_kernel" if len(sys.argv) > 6 and sys.argv[6] == "fused" else None)
    print(json.dumps({key: rec}, indent=1))
    if out_path:
        doc = {}
        if os.path.exists(out_path):
            with open(out_path) as fh:
                doc = json.load(fh)
        doc[key] = rec
        with open(out_path, "w") as fh:
            json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main()
