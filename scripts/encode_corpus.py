#!/usr/bin/env python3
"""Same path, flags and outputs as the reference's scripts/encode_corpus.py; MI355X-native inside."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multihop_dense_retrieval_amd.encode_corpus import main  # noqa: E402

if __name__ == "__main__":
    main()
