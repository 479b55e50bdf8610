#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc SQ_* passes to per-kernel means and to VALU wave-instructions per wave-step.

    python tools/pmc_valu.py <summary.csv> <valu.json> <steps_per_fused_launch> <pass_dir> [...]

summary.csv  : kernel, counter, dispatches, mean value per dispatch   (committed under profiles/<round>/)
valu.json    : {"step:f64:4,1,1": {"valu_per_wave_step": ..., ...}, "fused:f64:4,1,1": {...}, ...}  (bench.py reads it)

Units (MI355X_MICROARCH.md): SQ_INSTS_* count wave-instructions; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count
quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over shader engines.  One wave64 VALU instruction occupies its
SIMD for >= 4 cycles, so  issue fraction = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel time x 2.4 GHz).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    out_csv, out_json, fused_steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for d in sys.argv[4:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"]
                    if "fiveeq::" not in name:
                        continue
                    # packed fp32 lanes show up as "float __vector(2)": call them float2 before cutting the argument list off
                    k = name.replace("float __vector(2)", "float2").split("(")[0].replace("void ", "").strip()
                    acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    with open(out_csv, "w") as fh:
        fh.write("kernel,counter,dispatches,mean_per_dispatch\n")
        for (k, c), v in sorted(acc.items()):
            fh.write(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.6g}\n")
        for k, v in sorted(dur.items()):
            fh.write(f"\"{k}\",DURATION_US_UNDER_PMC,{len(v)},{sum(v) / len(v):.6g}\n")
    doc = {}
    if os.path.exists(out_json):
        with open(out_json) as fh:
            doc = json.load(fh)
    kernels = {k for k, _ in acc}
    for k in sorted(kernels):
        # fused_kernel<V, P0, P1, P2, INV, BINS>: the plain forward form only
        ms = re.match(r"fiveeq::small_kernel<(double|float), (\d), (\d), false>", k)      # <T, P0, lanes per member, STATS>
        m = re.match(r"fiveeq::(step|fused)_kernel<(double|float2|float), (\d), (\d), (\d)((?:, (?:true|false))*)>", k)
        mm_ = re.match(r"fiveeq::small_multi_kernel<(double|float), (\d), (\d), (\d), false(?:, false)?>", k)   # one lane per member (<.., STATS, COMP>)
        mo = re.match(r"fiveeq::small_octet_kernel<(double|float)>", k)                             # 4 + 1 + 1: one member per octet of lanes
        comp = m is not None and m.group(1) == "fused" and m.group(6).replace(" ", "") == ",false,false,true"      # the compensated fp32 form
        if mo:
            m = re.match(r"(small) (\w+) (\d) (\d) (\d)()", f"small {mo.group(1)} 4 1 1")
        if comp:
            m = re.match(r"(fused) (\w+) (\d) (\d) (\d)()", f"fused {m.group(2)} {m.group(3)} {m.group(4)} {m.group(5)}")
        if ms:
            m = re.match(r"(small) (\w+) (\d) (0) (0)()", f"small {ms.group(1)} {ms.group(2)} 0 0")
        if mm_:
            m = re.match(r"(small) (\w+) (\d) (\d) (\d)()", f"small {mm_.group(1)} {mm_.group(2)} {mm_.group(3)} {mm_.group(4)}")
        if not m or "true" in m.group(6):
            continue
        mean = lambda c: (sum(acc[(k, c)]) / len(acc[(k, c)])) if (k, c) in acc else None   # noqa: E731
        valu, waves = mean("SQ_INSTS_VALU"), mean("SQ_WAVES")
        if not valu or not waves:
            continue
        steps = 1 if m.group(1) == "step" else fused_steps
        per_wave = 128 if m.group(2) == "float2" else 64            # members of one wave: packed lanes carry two each
        tag = {"double": "f64", "float": "f32", "float2": "f32x2"}[m.group(2)]
        key = f"{m.group(1)}:{tag}:{m.group(3)},{m.group(4)},{m.group(5)}"
        if ms:
            per_wave = 64 // int(ms.group(3))                       # a quad of lanes per member: 16 members per wave
            key += f":{ms.group(3)}"
        if mm_:
            key += ":1"
        if mo:
            per_wave = 8
            key += ":8"
        if comp:
            key += ":comp"
        rec = {"kernel": k, "valu_per_wave_step": valu / waves / steps, "members_per_wave": per_wave,
               "valu_per_member_step": valu / waves / steps / per_wave, "waves": waves, "steps_per_launch": steps,
               "dispatches": len(acc[(k, "SQ_INSTS_VALU")])}
        # packed fp32 share of the stream (for the fp32 kernels' nominal issue time: a v_pk_* instruction holds its SIMD for 4
        # cycles, a scalar fp32 / integer one for 2).  SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F32 count INSTRUCTIONS (a packed one
        # once); SQ_INSTS_VALU_FLOPS_FP32 counts per-lane flops (FMA 2, packed x2): the excess over an all-scalar stream
        # is what the packed forms add — 2 per packed FMA, 1 per packed multiply or add.
        fl, nf, nm, na, ntr = (mean("SQ_INSTS_VALU_FLOPS_FP32"), mean("SQ_INSTS_VALU_FMA_F32"), mean("SQ_INSTS_VALU_MUL_F32"),
                               mean("SQ_INSTS_VALU_ADD_F32"), mean("SQ_INSTS_VALU_TRANS_F32"))
        if None not in (fl, nf, nm, na, ntr) and (2 * nf + nm + na) > 0:
            share = max(0.0, min(1.0, (fl - (2 * nf + nm + na + ntr)) / (2 * nf + nm + na)))
            rec["packed_per_wave_step"] = share * (nf + nm + na) / waves / steps
            rec["fma_mul_add_trans_per_wave_step"] = [x / waves / steps for x in (nf, nm, na, ntr)]
        gui, durs = mean("GRBM_GUI_ACTIVE"), dur.get(k)
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; only meaningful over a long kernel (the time-fused family): a
        # 20-us per-step launch is dominated by the counter's start / stop window
        if gui and durs and sum(durs) / len(durs) > 500.0:
            rec["clock_GHz_under_load"] = gui / 8.0 / (sum(durs) / len(durs) * 1e-6) / 1e9
        for c, nm in (("SQ_INSTS_SALU", "salu_per_wave_step"), ("SQ_INSTS_LDS", "lds_per_wave_step"),
                      ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave_step"), ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave_step")):
            if mean(c) is not None:
                rec[nm] = mean(c) / waves / steps
        wc = mean("SQ_WAVE_CYCLES")
        if wc:
            for c, nm in (("SQ_ACTIVE_INST_VALU", "valu_active_frac_of_wave_cycles"), ("SQ_WAIT_ANY", "wait_any_frac"),
                          ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac")):
                if mean(c) is not None:
                    rec[nm] = mean(c) / wc
        doc[key] = rec
    with open(out_json, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
