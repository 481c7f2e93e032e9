#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) and SQ counters of the MIPS screen kernels at 5M rows -> gpurun_out/<tag>/
#   sequential loop: mips_screen8_kernel  / mips_screen_kernel   (16 queries per wave, nq = 100 per call; int8 tier / fp16 screen)
#   pipelined loop : mips_screen8w_kernel / mips_screen32_kernel (32 queries per wave, nq = 200 per call)
# (fp16 screen alone: MDR_MIPS_I8=0 in the environment)
# Counters are collected in their own passes with --kernel-trace only (no other trace domain).
set -u
TAG=${1:-pmcs}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for MODE in sequential pipelined; do
  FLAG=""; [ $MODE = sequential ] && FLAG="--sequential"
  SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE")
  [ "${PMC_FETCH_ONLY:-0}" = 1 ] && SETS=("FETCH_SIZE")   # the pass profiles/pmc_traffic.json is made from
  for PMC in "${SETS[@]}"; do
    N=$(echo $PMC | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p -o p -- python $REPO/bench.py --rows 5000000 --steps 3 --warmup 1 --no-encoder --no-cpu-baseline --no-sequential --no-verify $FLAG > $OUT/log_${MODE}_$N.txt 2>&1
    P=$(find $OUT/p -name "*counter_collection.csv" | head -1)
    if [ -n "$P" ]; then head -1 "$P" > $OUT/${MODE}_$N.csv; grep -E "mips_(screen|screen32|screen8|screen8w|refine|refine8|star8)" "$P" >> $OUT/${MODE}_$N.csv; fi
    rm -rf $OUT/p
  done
done
python - $OUT $REPO <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections, os, json, hashlib
fetch = collections.defaultdict(lambda: collections.defaultdict(list))  # loop mode -> kernel -> FETCH_SIZE KB per launch
for f in sorted(glob.glob(sys.argv[1] + "/*_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        kern = ("screen8w" if "screen8w" in n else "screen8" if "screen8_kernel" in n else "screen32" if "screen32" in n
                else "screen" if "mips_screen_kernel" in n else "star8" if "star8" in n else "refine8" if "refine8" in n else "refine")
        mode = "main" if ("<24, 1" in n or "<12, 1" in n) else "sample" if ("<24, 0" in n or "<12, 0" in n) else ""
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        agg[(kern + " " + mode, r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
        if r["Counter_Name"] == "FETCH_SIZE":
            fetch[os.path.basename(f).split("_")[0]][kern + " " + mode].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        mean = sum(x for x, _ in v) / len(v)
        extra = ""
        if c == "FETCH_SIZE":  # KB; gfx950: x2 (MI355X_MICROARCH.md, HBM)
            extra = f"  -> {mean * 1024 * 2 / 1e9:.3f} GB per launch (x2 corrected), {mean * 1024 * 2 / (sum(d for _, d in v) / len(v)):.1f} GB/s at the profiled duration"
        print(f"{os.path.basename(f):44s} {k:16s} {c:28s} mean {mean:.6g} n={len(v)} avg_ns={sum(d for _, d in v) / len(v):.0f}{extra}")
# ---- profiles/pmc_traffic.json: physical bytes of ONE search call of each kernel family / algorithmic bytes, + the kernel source hash
def mean(v): return sum(v) / len(v) if v else 0.0
repo = sys.argv[2]
alg = 5_000_000 * 768 * 4.0
sys.path.insert(0, repo)
import bench
sha = bench.src_sha16()  # over csrc/mdr_mips* (the same function bench.py checks freshness with)
path = os.path.join(repo, "profiles", "pmc_traffic.json")
try: tbl = json.load(open(path))
except Exception: tbl = {"kernels": {}}
def per_call(mode, main, parts):  # KB -> bytes, x2 (gfx950); helper kernels may run twice per call (star8)
    f = fetch[mode]
    if not f.get(main): return None
    return (mean(f[main]) + sum(mult * mean(f.get(k, [])) for k, mult in parts)) * 1024 * 2
for name, mode, main, parts in (("mips_screen8_kernel", "sequential", "screen8 main", (("screen8 sample", 1), ("refine8 ", 1), ("star8 ", 2))),
                                ("mips_screen8w_kernel", "pipelined", "screen8w main", (("screen8w sample", 1), ("refine8 ", 1), ("star8 ", 2))),
                                ("mips_screen_kernel", "sequential", "screen main", (("screen sample", 1), ("refine ", 1))),
                                ("mips_screen32_kernel", "pipelined", "screen32 main", (("screen32 sample", 1), ("refine ", 1)))):
    b = per_call(mode, main, parts)
    if b is None or b < 0.05 * alg: continue  # that family did not run in this pass (it is the tier behind the int8 one)
    tbl["kernels"][name] = {"ratio": round(b / alg, 4), "source": f"profiles/{os.environ.get('PMC_PROFILE_TAG', os.path.basename(sys.argv[1]))}_{mode}_FETCH_SIZE.csv",
                            "csrc_sha16": sha, "note": f"{main} {mean(fetch[mode][main]):.6g} KB + helpers, x2 -> {b / 1e9:.3f} GB per search call at 5M x 768"}
    print("pmc_traffic.json:", name, tbl["kernels"][name])
json.dump(tbl, open(os.path.join(sys.argv[1], "pmc_traffic.json"), "w"), indent=1)
PY
