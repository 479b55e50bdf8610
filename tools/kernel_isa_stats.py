#!/usr/bin/env python3
"""Static per-kernel facts from the gfx950 assembly hipcc emits: VGPR/SGPR/LDS/scratch and the
instruction mix (VALU by type, SALU, LDS, global).  Runs in the CPU container (cross-compile).

    python tools/kernel_isa_stats.py [substring ...]      # default: the 4+1+1 instantiations
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    want = sys.argv[1:] or ["Li4ELi1ELi1E"]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "--save-temps", "-o",
               os.path.join(tmp, "lib.so"), os.path.join(ROOT, "fiveeqscm_amd", "csrc", "fiveeq_capi.hip")]
        cmd += [f"-D{d}" for d in os.environ.get("FIVEEQ_DEFS", "").split() if d]
        subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "fiveeq_capi-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not any(w in name for w in want):
            continue
        demangled = subprocess.run(["c++filt", name], capture_output=True,
                                   text=True).stdout.strip().split("(")[0]
        code = body.split(".section")[0]
        get = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, body) or [None, "?"])[1]   # noqa: E731
        tail = asm[m.end():m.end() + 4000]
        scratch = (re.search(r"; ScratchSize: (\d+)", tail) or [None, "?"])[1]
        occ = (re.search(r"; Occupancy: (\d+)", tail) or [None, "?"])[1]
        lds = (re.search(r"; LDSByteSize: (\d+)", tail) or [None, "?"])[1]
        cnt = lambda pat: len(re.findall(r"^\s+" + pat, code, re.M))   # noqa: E731
        print(f"{demangled}\n   vgpr {get('next_free_vgpr')} sgpr {get('next_free_sgpr')} lds {lds} scratch {scratch} "
              f"occupancy {occ}\n   VALU {cnt('v_')} (fma_f64 {cnt('v_fma_f64')} mul_f64 {cnt('v_mul_f64')} "
              f"add_f64 {cnt('v_add_f64')} fma_f32 {cnt('v_fma_f32|v_fmac_f32')} pk {cnt('v_pk_')} readlane "
              f"{cnt('v_readlane|v_writelane')}) SALU {cnt('s_')} LDS {cnt('ds_')} global {cnt('global_')} "
              f"waitcnt {cnt('s_waitcnt')}")


if __name__ == "__main__":
    main()
