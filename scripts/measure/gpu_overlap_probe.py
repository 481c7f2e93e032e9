"""Probe (GPU box): can the fused corpus pass of the pipelined loop hide beside the two encoder forwards?
Times, on fixed inputs with no data dependencies between the three pieces of work:
  enc    = hop-2 forward (main stream) + hop-1 forward (side stream), as in SyntheticTwoHop._step_pipelined
  search = one fused 200-query search at 5M rows
  both   = enc and search launched together (search on a third stream)
Prints ms per iteration of each; `both` close to `enc` means the search fits in the encoders' gaps, close to enc + search means the
two only share CU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from multihop_dense_retrieval_amd import mhop

sys.argv = [sys.argv[0], "--rows", os.environ.get("ROWS", "5000000")]
args = bench.parse()
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
sidx, local, lo, hi, GB, planted, rows_sum = bench.build_pipeline(args, 1, 0, device, None, False)
pipe = mhop.SyntheticTwoHop(sidx, batch=args.batch, beam=1, topk=1, dim=args.dim, device=device, max_q_len=args.max_q_len,
                            max_q_sp_len=args.max_q_sp_len, use_encoder=True, planted_rows=rows_sum, pipelined=True)
for _ in range(3):
    out = pipe.step()
torch.cuda.synchronize()
ids2, mask2 = out["ids2"], out["mask2"]
e = torch.cat([out["q2"], out["q"]], 0).contiguous()
side, srch = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
main = torch.cuda.current_stream()


def enc():
    start = torch.cuda.Event(); start.record()
    side.wait_event(start)
    with torch.cuda.stream(side):
        pipe._encode(pipe.q_ids, pipe.q_mask, lane=1)
        done = torch.cuda.Event(); done.record()
    pipe._encode(ids2, mask2)
    main.wait_event(done)


def search(stream=None):
    if stream is None:
        local.search_device(e, 1)
        return
    start = torch.cuda.Event(); start.record()
    stream.wait_event(start)
    with torch.cuda.stream(stream):
        local.search_device(e, 1)
        done = torch.cuda.Event(); done.record()
    return done


def both():
    d = search(srch)
    enc()
    main.wait_event(d)


def both_late():  # search enqueued after the encoders (other launch order)
    start = torch.cuda.Event(); start.record()
    side.wait_event(start); srch.wait_event(start)
    with torch.cuda.stream(side):
        pipe._encode(pipe.q_ids, pipe.q_mask, lane=1)
        d1 = torch.cuda.Event(); d1.record()
    pipe._encode(ids2, mask2)
    with torch.cuda.stream(srch):
        local.search_device(e, 1)
        d2 = torch.cuda.Event(); d2.record()
    main.wait_event(d1); main.wait_event(d2)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {}
for name, f in (("enc", enc), ("search", search), ("both", both), ("both_late", both_late), ("enc_again", enc)):
    res[name] = round(timeit(f), 3)
print("overlap probe (ms per iteration):", res)
print("sum enc + search = %.3f; gain if pipelined one stage deeper = %.3f ms per step" % (res["enc"] + res["search"], res["enc"] + res["search"] - min(res["both"], res["both_late"])))

# ---- hop-1 questions of G future batches encoded as ONE forward every G-th step (offline evaluation knows all questions up front)
for G in (1, 2, 4, 8):
    ids_g, mask_g = pipe.q_ids.repeat(G, 1), pipe.q_mask.repeat(G, 1)
    pipe.encoder.encode_q(ids_g, mask_g, None, lane=1)
    pipe.encoder.encode_q(ids_g, mask_g, None, lane=1)
    torch.cuda.synchronize()
    state = {"i": 0}

    def enc_grouped():
        do = state["i"] % G == 0
        state["i"] += 1
        start = torch.cuda.Event(); start.record()
        if do:
            side.wait_event(start)
            with torch.cuda.stream(side):
                pipe.encoder.encode_q(ids_g, mask_g, None, lane=1)
                done = torch.cuda.Event(); done.record()
        pipe._encode(ids2, mask2)
        if do:
            main.wait_event(done)

    def alone():
        pipe.encoder.encode_q(ids_g, mask_g, None, lane=1)

    t_alone = timeit(alone, 10)
    t = timeit(enc_grouped, 8 * max(G, 3))
    print(f"hop-1 group of {G} batches ({100 * G} questions): forward alone {t_alone:.3f} ms = {t_alone / G:.3f} per batch; encoder stage per step {t:.3f} ms")
