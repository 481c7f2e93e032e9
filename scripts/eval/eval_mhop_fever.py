#!/usr/bin/env python3
"""Same path, arguments and outputs as the reference's scripts/eval/eval_mhop_fever.py; MI355X-native inside."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd.eval_mhop_fever import main  # noqa: E402

if __name__ == "__main__":
    main()
