"""Build libmdrhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m multihop_dense_retrieval_amd.build [--force]

The .so lands next to this file so it travels with the source tree (it is git-ignored, not
gpurun-ignored). hipcc cross-compiles for gfx950 without a GPU present.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmdrhip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inl")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_lib(force=False, verbose=True, defines=(), out=None):
    """defines / out: build a measurement VARIANT of the library next to the product one (e.g. defines=["MDR_MIPS_DMA_AUX=2"],
    out="libmdrhip_nt.so"); a process selects it with MDR_LIB_PATH (scripts/measure/gpu_ab.sh). The product build takes neither."""
    target = LIB if out is None else os.path.join(HERE, out)
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-Wno-unused-variable", "-I", INCLUDE] + ["-D" + d for d in defines] + sources() + ["-o", target + ".tmp"]
    if verbose:
        print("[mdr build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build_lib(force="--force" in sys.argv, defines=defs, out=outs[0] if outs else None))
