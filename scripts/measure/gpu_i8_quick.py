import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multihop_dense_retrieval_amd import index as mi
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = mi.IndexFlatIP(768, device=dev)
idx.reserve(rows)
g = torch.Generator(device=dev).manual_seed(0)
chunks = []
for lo in range(0, rows, 250_000):
    x = torch.randn((min(250_000, rows - lo), 768), generator=g, device=dev)
    idx.add(x)
    if lo < 1_000_000: chunks.append(x)
X = torch.cat(chunks)
NQ = int(os.environ.get("I8_NQ", "100"))
for name, q in (("random", torch.randn((NQ, 768), generator=g, device=dev)), ("planted", X[:NQ] + 0.05 * torch.randn((NQ, 768), generator=g, device=dev))):
    res = {}
    for v in (3, 4, 2):
        idx.set_variant(v)
        D, I = idx.search_device(q, 1)
        torch.cuda.synchronize()
        t = idx.telemetry(NQ, 1)
        for _ in range(2): idx.search_device(q, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): idx.search_device(q, 1)
        e1.record(); torch.cuda.synchronize()
        res[v] = (D.clone(), I.clone())
        print(f"{name} rows {rows} variant {v} {idx.last_kernel():30s} {e0.elapsed_time(e1)/10:7.3f} ms  telemetry {t}", flush=True)
    for v in (3, 4):
        same_i = bool((res[v][1] == res[2][1]).all()); dmax = float((res[v][0] - res[2][0]).abs().max())
        print(f"   variant {v} vs exact: ids equal {same_i}  max |dD| {dmax:.3e}")

if os.environ.get("I8_STAMPS"):  # needs a -DMDR_I8_ABL=9 build selected with MDR_LIB_PATH
    import ctypes
    from multihop_dense_retrieval_amd import _lib
    buf = (ctypes.c_uint64 * 8)()
    _lib.check(_lib.lib().mdr_test_i8_stamps(buf, 1))
    n = max(1, buf[7])
    print("int8 wide kernel, cycles per stage (wave 0): " + "  ".join(f"{nm} {buf[i] / n:7.1f}" for i, nm in enumerate(("wait+barrier", "exch+dma", "chain", "epilogue", "share"))) + f"  total {sum(buf[:5]) / n:7.1f}")
