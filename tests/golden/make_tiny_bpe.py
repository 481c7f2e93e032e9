#!/usr/bin/env python3
"""Generates tests/golden/tiny_bpe/{vocab.json,merges.txt}: a tiny byte-level BPE (the RoBERTa tokenizer family: GPT-2
byte-to-unicode alphabet, `Ġ` word-start marker, special tokens <s>=0 <pad>=1 </s>=2 <unk>=3 <mask>) trained on the few
sentences below with the `tokenizers` library. No network, no pretrained files: the real roberta-base vocab.json / merges.txt
do not exist offline, but every code path of the tokenizer CLASS (pair template, longest-first truncation,
padding, prefix-space rule, byte fallback for unseen characters) is the same with 400 merges as with 50 000.

    python tests/golden/make_tiny_bpe.py
"""
import os

from tokenizers import ByteLevelBPETokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
TEXT = [
    "The quick brown fox jumps over the lazy dog near the river bank.",
    "Multi-hop dense retrieval answers open-domain questions by reading two passages in a row.",
    "Who directed the film that starred the actor born in 1950 in Lyon?",
    "Paris is the capital of France. It lies on the Seine, north of Orléans.",
    "The 2012 Summer Olympics were held in London; the stadium seats 80,000 people.",
    "Which band released the album recorded at the studio founded by the producer of Thriller?",
    "A passage has a title and a text. An empty text falls back to the title.",
    "What is the population of the city where the author of Les Misérables was born?",
    "Retrieval, encoder, index, beam, top-k, inner product, embedding, question, answer.",
    "Zürich, São Paulo and Kraków host universities; 3.14 is not 22/7.",
]


def main():
    tok = ByteLevelBPETokenizer()
    tok.train_from_iterator(TEXT * 25, vocab_size=600, min_frequency=2, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"])
    out = os.path.join(HERE, "tiny_bpe")
    os.makedirs(out, exist_ok=True)
    tok.save_model(out)
    print(sorted(os.listdir(out)), tok.get_vocab_size())


if __name__ == "__main__":
    main()
