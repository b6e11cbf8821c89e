#!/usr/bin/env python3
"""Decode micro-benchmark. No argument: bench.decode_bench at batch 32 / 416x416 and batch 128 / 608x608.
Argument 0 | 1: ONLY yolo_decode3_ex(write_back = arg) at batch 32, 416x416, 80 classes (the PMC passes of tools/profile_round.sh:
one kernel variant per process, so that a counter average is the average of that variant)."""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import yolo_for_turbines_amd as yt
if len(sys.argv) > 1:
    import ctypes as C
    from yolo_for_turbines_amd import _lib as L
    wb, batch, size, nc = int(sys.argv[1]), 32, 416, 80
    g = [size // 32, size // 16, size // 8]
    gen = torch.Generator().manual_seed(11)
    dev = torch.device("cuda:0")
    preds = [torch.randn((batch, 3, gg, gg, 5 + nc), generator=gen).to(dev) for gg in g]
    anchors = [torch.rand((3, 2), generator=gen).to(dev) * gg for gg in g]
    n_total = sum(3 * gg * gg for gg in g)
    out = torch.empty((batch, n_total, 6), dtype=torch.float32, device=dev)
    pp = (C.c_void_p * 3)(*[p.data_ptr() for p in preds])
    st = (C.c_int64 * 15)(*[v for p in preds for v in p.stride()])
    ap = (C.c_void_p * 3)(*[a.data_ptr() for a in anchors])
    gg3 = (C.c_int * 3)(*g)
    for _ in range(10):
        L.check(L.lib().yolo_decode3_ex(pp, st, ap, gg3, batch, nc, wb, out.data_ptr(), n_total, L.current_stream()), "yolo_decode3_ex")
    torch.cuda.synchronize()
    print("decode3 write_back", wb, "x10 done")
else:
    print(json.dumps(bench.decode_bench(yt, torch.device("cuda:0"))))
    print(json.dumps(bench.decode_bench(yt, torch.device("cuda:0"), batch=128, size=608)))
