#!/usr/bin/env python3
"""bench.py with a rejected loop variant swapped in for the main pipeline (scripts/measure/loop_variants.py). Own flags: --loop, --hop1-group, --lane-cus;
everything else goes to bench.py unchanged."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

ap = argparse.ArgumentParser(add_help=False)
ap.add_argument("--loop", choices=["deep", "pipelined", "shift"], default="pipelined")
ap.add_argument("--hop1-group", type=int, default=1)
ap.add_argument("--lane-cus", type=int, default=0)
own, rest = ap.parse_known_args()
sys.argv = [os.path.join(ROOT, "bench.py")] + rest

import bench  # noqa: E402
import loop_variants  # noqa: E402
from multihop_dense_retrieval_amd import mhop  # noqa: E402

loop_variants.VariantTwoHop.MODE = own.loop
loop_variants.VariantTwoHop.HOP1_GROUP = own.hop1_group
loop_variants.VariantTwoHop.LANE_CUS = own.lane_cus
mhop.SyntheticTwoHop = loop_variants.VariantTwoHop
bench.main()
