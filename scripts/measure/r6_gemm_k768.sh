#!/bin/bash
# round 6: where a tile-round of the K = 768 GEMMs goes today, at the hop-2 forward's row count: s_memtime timeline of the persistent 256 x 256 kernel
# (-DMDR_GEMM_ABL=5 build), the vendor library on the same shapes (reference point only), and the in-forward rocprofv3 averages of the default bench command
set -u
TAG=${1:-r06gemm}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
M=${M:-21200}
echo "== product library, isolated (20 back-to-back launches) + hipBLASLt via torch on the same shapes"
GEMM_SHAPES=qkv:2304:768:0,ffn1:3072:768:1,out:768:768:3 GEMM_TORCH_REF=1 timeout 300 python scripts/measure/gpu_gemm_bench.py $M 6 7 2>&1 | grep -v amdgpu.ids | tee $OUT/isolated.txt
echo "== timeline build"
MDR_LIB_PATH=$REPO/multihop_dense_retrieval_amd/libmdrhip_gemm_timeline.so GEMM_SHAPES=qkv:2304:768:0,ffn1:3072:768:1 timeout 300 python scripts/measure/gpu_gemm_bench.py $M 6 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline.txt
echo "== in-forward kernel averages (default bench command under rocprofv3)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python $REPO/bench.py --no-cpu-baseline --no-anisotropic --structured > $OUT/bench.json 2> $OUT/bench.err
S=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/bench_default_kernel_stats.csv
python - $OUT/bench_default_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("gemm_", "attention", "layernorm", "mips_screen8")):
        short = n.replace("void mdr::(anonymous namespace)::", "").replace("mdr::(anonymous namespace)::", "").split("(")[0]
        print(f"{short:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:9.1f} pct {r['Percentage']}")
PY
rm -rf $OUT/prof
