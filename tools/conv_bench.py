#!/usr/bin/env python3
"""Single-layer conv micro-benchmark (tuning aid; calls the C-ABI directly).

  python tools/conv_bench.py --layer c52_3x3 --tile 0 --reps 20
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -- python3 tools/conv_bench.py ...

Layers are the dominant YOLOv3 shapes at batch 32, 416x416.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yolo_for_turbines_amd import _lib as L  # noqa: E402

LAYERS = {   # name: (H, cin, cout, k, stride)
    "c416_first": (416, 3, 32, 3, 1),
    "c416_s2": (416, 32, 64, 3, 2),
    "c208_1x1": (208, 64, 32, 1, 1),
    "c208_3x3": (208, 32, 64, 3, 1),
    "c208_s2": (208, 64, 128, 3, 2),
    "c104_1x1": (104, 128, 64, 1, 1),
    "c104_3x3": (104, 64, 128, 3, 1),
    "c104_s2": (104, 128, 256, 3, 2),
    "c52_1x1": (52, 256, 128, 1, 1),
    "c52_3x3": (52, 128, 256, 3, 1),
    "c52_s2": (52, 256, 512, 3, 2),
    "c26_1x1": (26, 512, 256, 1, 1),
    "c26_3x3": (26, 256, 512, 3, 1),
    "c26_s2": (26, 512, 1024, 3, 2),
    "c13_1x1": (13, 1024, 512, 1, 1),
    "c13_3x3": (13, 512, 1024, 3, 1),
    "c13_head": (13, 1024, 255, 1, 1),
}


def run(name, batch, tile, reps, residual, dev, dtype="fp32"):
    H, cin, cout, k, s = LAYERS[name]
    lib = L.lib()
    cpad = (cin + 3) // 4 * 4
    Ho = (H + 2 * (k // 2) - k) // s + 1
    code, tdt = {"fp32": (L.F32, torch.float32), "fp16": (L.F16, torch.float16), "bf16": (L.BF16, torch.bfloat16)}[dtype]
    x = torch.randn(batch * H * H * cpad, device=dev).to(tdt)
    w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5
    wp = torch.empty(lib.yolo_packed_weight_bytes(cout, cin, k, code), dtype=torch.uint8, device=dev)
    stream = L.current_stream()
    L.check(lib.yolo_pack_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, code, stream))
    scale = torch.rand(cout, device=dev) + 0.5
    shift = torch.randn(cout, device=dev) * 0.1
    y = torch.empty(batch * Ho * Ho * cout, device=dev, dtype=tdt)
    r = torch.randn(batch * Ho * Ho * cout, device=dev).to(tdt) if residual else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    d = L.ConvDesc(n=batch, h=H, w=H, cin=cin, cout=cout, ksize=k, stride=s, x_ld=cpad, x_off=0, y_ld=cout, y_off=0,
                   r_ld=cout, r_off=0, act=L.ACT_LEAKY, out_mode=L.OUT_NHWC, dtype=code,
                   flags=(L.FLAG_RESIDUAL if residual else 0) | L.FLAG_NANCHECK, tile=tile)

    need = lib.yolo_conv_workspace_bytes(d)                       # > 0: Winograd (tile 13, or the heuristic's choice for tile 0)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)

    def launch():
        L.check(lib.yolo_conv_fwd_ws(d, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(r),
                                     y.data_ptr(), ws.data_ptr() if need else 0, need, flag.data_ptr(), stream), "conv")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gflop = 2.0 * batch * Ho * Ho * cout * cin * k * k / 1e9
    picked = (lib.yolo_conv_pick_tile(d) if dtype == "fp32" else 0) if tile == 0 else tile
    print(f"{name:12s} tile={picked} {ms * 1e3:8.1f} us  {gflop / ms:7.2f} TFLOP/s  ({gflop:.1f} GFLOP)", flush=True)
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="all")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tile", default="0", help="tile id, comma list, or 'all'")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--dtype", default="fp32")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = list(LAYERS) if a.layer == "all" else a.layer.split(",")
    nt = L.lib().yolo_conv_num_tiles()
    tiles = list(range(1, nt + 1)) if a.tile == "all" else [int(t) for t in a.tile.split(",")]
    for n in names:
        for t in tiles:
            run(n, a.batch, t, a.reps, a.residual, dev, a.dtype)


if __name__ == "__main__":
    main()
