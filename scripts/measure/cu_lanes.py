"""CU-partitioned encoder lanes (round 5; a measured negative: profiles/r05_cu_partitioned_lanes_negative.txt), outside the product. Needs a measurement build of
the library that exports mdr_stream_create_cu_range (include/mdr_hip_measure.h):

    python -m multihop_dense_retrieval_amd.build -DMDR_CU_LANES=1 --out=libmdrhip_cu_lanes.so
    MDR_LIB_PATH=.../libmdrhip_cu_lanes.so python scripts/measure/bench_loops.py --lane-cus 64

partition_lanes(model, side_cus): lane 1 (the short forward of the pipelined loop) gets the last `side_cus` CUs of the device, lane 0 the rest; each lane is bound to
its own CU-masked stream through the product's generic lane -> stream binding (retriever.bind_lane_stream); that build sizes the persistent GEMM grids by the CUs of
the stream it is launched (or captured) on."""
import ctypes

import torch

from multihop_dense_retrieval_amd import _lib

_CU_RANGE_STREAMS = {}  # (device, cu_lo, cu_hi) -> (torch.cuda.ExternalStream, raw pointer); never destroyed: torch's caching allocator keeps per-stream state


def partition_lanes(model, side_cus):
    L = _lib.lib()
    if not hasattr(L, "mdr_stream_create_cu_range"):
        raise RuntimeError("this libmdrhip has no CU-lane hooks: build with -DMDR_CU_LANES=1 and point MDR_LIB_PATH at it")
    fn = L.mdr_stream_create_cu_range
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    for lane in (0, 1):
        model.bind_lane_stream(lane, None)
    side_cus = int(side_cus)
    if side_cus <= 0:
        return
    n = torch.cuda.get_device_properties(model.device).multi_processor_count
    if side_cus % 8 or not 8 <= side_cus <= n - 8:
        raise ValueError(f"side_cus must be a multiple of 8 in [8, {n - 8}]")
    for lane, (lo, hi) in ((0, (0, n - side_cus)), (1, (n - side_cus, n))):
        key = (model.device.index or 0, lo, hi)
        if key not in _CU_RANGE_STREAMS:
            ptr = ctypes.c_void_p()
            _lib.check(fn(key[0], lo, hi, ctypes.byref(ptr)))
            _CU_RANGE_STREAMS[key] = (torch.cuda.ExternalStream(ptr.value, device=model.device), ptr.value)
        model.bind_lane_stream(lane, _CU_RANGE_STREAMS[key][0])
