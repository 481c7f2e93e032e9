"""Flag sets of the corpus-encoder CLI: `encode_args()` = `common_args()` + two flags, flag for flag as
/root/reference/mdr/retrieval/config.py:14-69,107-112 (training-only flags are accepted and ignored so
existing command lines keep working)."""
import argparse


def common_args():
    p = argparse.ArgumentParser()
    # task
    p.add_argument("--train_file", type=str, default="../data/nq-with-neg-train.txt")
    p.add_argument("--predict_file", type=str, default="../data/nq-with-neg-dev.txt")
    p.add_argument("--num_workers", default=30, type=int)
    p.add_argument("--do_train", default=False, action="store_true")
    p.add_argument("--do_predict", default=False, action="store_true")
    # model
    p.add_argument("--model_name", default="bert-base-uncased", type=str)
    p.add_argument("--init_checkpoint", type=str, default="")
    p.add_argument("--max_c_len", default=512, type=int)
    p.add_argument("--max_q_len", default=50, type=int)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--fp16_opt_level", type=str, default="O1")
    p.add_argument("--no_cuda", default=False, action="store_true")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--max_q_sp_len", default=50, type=int)
    p.add_argument("--sent-level", action="store_true")
    p.add_argument("--rnn-retriever", action="store_true")
    p.add_argument("--predict_batch_size", default=512, type=int)
    p.add_argument("--shared-encoder", action="store_true")
    # multi vector scheme / momentum / NQ trial: training-side switches, parsed for compatibility
    p.add_argument("--multi-vector", type=int, default=1)
    p.add_argument("--scheme", type=str, default="none")
    p.add_argument("--momentum", action="store_true")
    p.add_argument("--init-retriever", type=str, default="")
    p.add_argument("--k", type=int, default=38400)
    p.add_argument("--m", type=float, default=0.999)
    p.add_argument("--nq-multi", action="store_true")
    return p


def encode_args(argv=None):
    p = common_args()
    p.add_argument("--embed_save_path", type=str, default="")
    p.add_argument("--is_query_embed", action="store_true")
    # addition of this build (off by default): also write <embed_save_path>.bf16.npy, the matrix as bf16 bit patterns
    p.add_argument("--save_bf16", action="store_true")
    # addition (default behaviour is result-identical to the reference): passages are tokenised in windows of
    # predict_batch_size x this many items, sorted by token count inside a window and batched by length; every embedding
    # still lands in its own row. 1 = the reference's order-of-appearance batches.
    p.add_argument("--length_bucket_window", type=int, default=16)
    return p.parse_args(argv)
