#!/bin/bash
# round 6, fourth visit: query split with the addends computed in the idle steps of the wide kernel's chain -- tests, A/B, full-length encoder-geometry corpus
set -u
TAG=${1:-r06v4}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest (MIPS + CLI)"
timeout 1800 python -m pytest tests/test_mips_i8_gpu.py tests/test_mips_gpu.py tests/test_mips_fullsize_gpu.py tests/test_cli_reference_gpu.py tests/test_retrieval_agreement_gpu.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_mips.txt; tail -5 $OUT/pytest_mips.txt
grep -o "query split, nq.*\|case [0-9] beam.*\|clustered 5 M.*" $OUT/pytest_mips.txt | cut -c1-260
echo "== structured A/B"
for CB in 0 1; do
  MDR_MIPS_I8_CB=$CB timeout 900 python bench.py --mode structured > $OUT/structured_cb$CB.json 2> $OUT/structured_cb$CB.err
done
timeout 900 python bench.py --mode structured > $OUT/structured_auto.json 2> $OUT/structured_auto.err
python - $OUT <<'PY'
import json, sys
for tag in ("cb0", "cb1", "auto"):
    try:
        r = json.loads(open(f"{sys.argv[1]}/structured_{tag}.json").read().strip().splitlines()[-1])["structured"]
    except Exception as e:
        print(tag, "failed", e); continue
    for name, v in r.items():
        for nq in ("nq100", "nq200"):
            x = v[nq]
            print(f"{tag:5s} {name:17s} {nq}: {x['ms_per_search']:.4f} ms  int8 decided {x['int8_tier_decided']}  emitted {x['candidates_emitted']:7d} rescored {x['candidates_rescored']:6d}  agree {x['top1_agreement_up_to_exact_ties']}")
PY
echo "== full-length encoder-geometry corpus (20..300 tokens, 5 M passages)"
for CB in 0 1; do
  MDR_MIPS_I8_CB=$CB timeout 1500 python bench.py --mode structured --structured encoder --geom-len 20 300 > $OUT/structured_full_length_cb$CB.json 2> $OUT/structured_full_length_cb$CB.err
  python - $OUT/structured_full_length_cb$CB.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["structured"]["encoder_geometry"]
print(r["encode_seconds"], r["stats"])
for nq in ("nq100", "nq200"):
    x = r[nq]
    print(f"  {nq}: {x['ms_per_search']:.4f} ms  int8 decided {x['int8_tier_decided']}  emitted {x['candidates_emitted']:7d} rescored {x['candidates_rescored']:6d}  agree {x['top1_agreement_up_to_exact_ties']}")
PY
done
du -sh $OUT
