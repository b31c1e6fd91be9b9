#!/usr/bin/env python3
"""Static instruction census of one kernel of the built library (what the node / leaf steps are made of):
    python profiles/isa_stats.py [lib.so] 'k_trace<true,false,false>' [--dump]
Extracts the gfx950 code object from the HIP fat binary (llvm-objdump --offloading, into /tmp), disassembles it and
counts instruction classes with their MEASURED issue cost (profiles/issue_peak.json: full-rate VALU 2 cycles per
wave64 instruction at >= 2 waves per SIMD, half-rate VALU 4, transcendental 8)."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin"
HALF = ("v_max", "v_min", "v_med3", "v_cndmask", "v_cmp", "v_lshl", "v_lshr", "v_ashr", "v_pk_", "v_div_", "v_bfe", "v_readlane", "v_readfirstlane",
        "v_mbcnt", "v_cvt", "v_mul_lo", "v_mul_hi", "v_mad_u64", "v_ldexp", "v_frexp", "v_rndne", "v_fract", "v_floor", "v_trunc")
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if len(args) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuturenderer_amd", "libtutu_hip.so")
    want = args[-1].replace(" ", "")
    tmp = tempfile.mkdtemp(prefix="isa_")
    loc = os.path.join(tmp, os.path.basename(lib))
    subprocess.run(["cp", lib, loc], check=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", loc], capture_output=True, cwd=tmp)
    co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, co)], capture_output=True, text=True).stdout
    cur, body = None, {}
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            cur = demangle(m.group(1)).replace(" ", "")
            body[cur] = []
        elif cur and ln.startswith("\t"):
            body[cur].append(ln.strip().split("//")[0].strip())
    hits = [k for k in body if want in k]
    if not hits:
        print("no kernel matches; have:", *sorted(k for k in body if "k_" in k)[:40], sep="\n  ")
        return
    for k in hits:
        ins = [x.split()[0] for x in body[k] if x]
        c = Counter()
        for i in ins:
            if i.startswith("v_"):
                c["valu_trans" if i.startswith(TRANS) else ("valu_half" if i.startswith(HALF) else "valu_full")] += 1
                if i.startswith("v_mov"):
                    c["  of which v_mov"] += 1
                if i.startswith("v_cndmask"):
                    c["  of which v_cndmask"] += 1
            elif i.startswith("s_cbranch") or i.startswith("s_branch"):
                c["branch"] += 1
            elif i.startswith("s_waitcnt") or i.startswith("s_nop"):
                c["waitcnt/nop"] += 1
            elif i.startswith("s_") and ("exec" in i or i.endswith("_b64")):
                c["salu_mask"] += 1
            elif i.startswith("s_"):
                c["salu_other"] += 1
            elif i.startswith("ds_"):
                c["lds"] += 1
            elif i.startswith(("global_", "flat_", "buffer_", "scratch_")):
                c["vmem"] += 1
            else:
                c["other"] += 1
        valu = c["valu_full"] + c["valu_half"] + c["valu_trans"]
        print(f"{k}: {len(ins)} instructions; VALU {valu} (issue cycles at >= 2 waves/SIMD: {2 * c['valu_full'] + 4 * c['valu_half'] + 8 * c['valu_trans']})")
        for n, v in sorted(c.items(), key=lambda kv: kv[0].strip()):
            print(f"   {n:22s} {v}")
        if "--dump" in sys.argv:
            print("\n".join(body[k]))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
