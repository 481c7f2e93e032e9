"""MI355X-native implementation of the iterative-retrieval hot path of
facebookresearch/multihop_dense_retrieval (scripts/eval/eval_mhop_retrieval.py +
scripts/encode_corpus.py): RoBERTa-base encoder forward and two-hop beam-search brute-force MIPS,
hand-written HIP for gfx950 behind a C ABI (include/mdr_hip.h). See DESIGN.md."""

__version__ = "0.1.0"
