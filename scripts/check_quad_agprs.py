#!/usr/bin/env python3
"""gemm_quad_kernel keeps its accumulators in a[0:255] ACROSS inline-asm statements, so the compiler itself must never touch an AGPR in that
kernel (no AGPR spill slots, no copies). Compiles csrc/mdr_encoder.hip to assembly and checks every gemm_quad_kernel instantiation:
outside ;;#ASMSTART / ;;#ASMEND no instruction may name an AGPR, and nothing may spill to scratch. Exit code 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    bad = []
    kernels = 0
    for m in re.finditer(r"^(_Z\w*gemm_quad_kernel\w*):[^\n]*\n(.*?)\n\s*s_endpgm", asm_text, re.S | re.M):
        kernels += 1
        inside = False
        for line in m.group(2).split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                inside = True
            elif t.startswith(";;#ASMEND"):
                inside = False
            elif not inside and t and not t.startswith((";", ".")):
                if re.search(r"\ba\[?\d+", t) or "accvgpr" in t or t.startswith("scratch_"):
                    bad.append((m.group(1), t))
    return kernels, bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "enc.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "multihop_dense_retrieval_amd", "csrc", "mdr_encoder.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        kernels, bad = check(open(out).read())
    print(f"{kernels} gemm_quad_kernel instantiations, {len(bad)} compiler-generated AGPR / scratch instructions")
    for k, t in bad[:20]:
        print("  ", k[-40:], t)
    return 0 if kernels >= 3 and not bad else 1


if __name__ == "__main__":
    sys.exit(main())
