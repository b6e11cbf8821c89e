#!/usr/bin/env python3
"""bench.py — images/s of the YOLOv3 forward hot path on MI355X (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: batch 32, 416x416, fp32 inference, 80-class COCO
head, one MI355X; synthetic input already resident in HBM, seeded random weights of the exact
architecture (no pretrained file exists offline).  One "step" = one forward of the batch through
the drop-in ``YOLOv3`` module: NCHW->NHWC boundary kernel, 75 fused convolution launches, the NaN
sticky-flag check (one host sync, as the reference's forward does 27).  N>1: one process per GPU
(torchrun), every rank runs the same per-GPU batch on its own images (weak scaling, images are
independent: no data-path collective), barrier + max-over-ranks timing.

Prints ONE JSON line. Extra objects:
  roofline      the conv kernel family (v_mfma_f32_32x32x2_f32 implicit GEMM): algorithmic conv
                FLOPs of one step / sum of the per-launch durations measured with HIP events on the
                launch stream, vs the 157.3 TFLOP/s dense f32 matrix peak.
  cpu_baseline  the oracle (CPU restatement of the reference, torch CPU ops) timed on this host on
                a bounded sample of the same workload (rank 0, N=1 only).
  nms           secondary metric of BASELINE.json: boxes/s of batched NMS at 10,000 post-threshold
                boxes per image (device path vs the list-based CPU port).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, capped at the
    GPU box's share (16 per GPU) — os.cpu_count() reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_H16_MFMA_TFLOPS = 2500.0         # dense bf16 / fp16 MFMA peak (not the 2:1-sparsity marketing figure)
TRAFFIC_F32_C52 = 146.8e6             # PMC: 58.2 MB read (x2-corrected) + 88.6 MB written, conv_patch_f32<3,64> 128->256 @52x52 B=32
TRAFFIC_H16_C52 = 71.8e6              # PMC: 27.5 MB read (x2-corrected) + 44.3 MB written, conv3_dma_h16 128->256 @52x52 B=32 (profiles/r02/pmc_hbm_traffic.txt)
GFLOP_PER_IMAGE_416_NC80 = 65.864     # BASELINE.md §2 (75 convs, 2*Ho*Wo*Cout*Cin*k^2)


def seeded_model(yt, nc, device, seed=0, gain=0.8):
    """Random weights of the exact architecture: W ~ N(0, gain/fan_in), non-trivial BN stats."""
    m = yt.YOLOv3(num_classes=nc)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                fan_in = mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1]
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (gain / fan_in) ** 0.5)
                if mod.bias is not None:
                    mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            elif isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
    return m.to(device).eval()


def conv_flops(plan):
    """Algorithmic FLOPs of every launch in the plan (one step), per launch."""
    out = []
    for op in plan.prog.ops:
        cv = op["block"].conv
        out.append(2.0 * plan.prog.B * op["Ho"] * op["Wo"] * cv.out_channels * cv.in_channels * op["k"] ** 2)
    return out


def per_launch_times(plan, reps, x):
    """Per-launch durations (ms) with HIP events on the launch stream, `reps` passes."""
    from yolo_for_turbines_amd import _lib as L
    lib = L.lib()
    stream = L.current_stream()
    n = len(plan.table)
    acc = np.zeros(n)
    alive = []                                   # fresh head outputs (the table's old ones belong to a past call)
    for k in range(plan.prog.n_pred):
        i, g, c3 = plan.pred_ops[k]
        alive.append(torch.empty((plan.prog.B, 3, g, g, c3), dtype=torch.float32, device=plan.device))
        plan.table[i].y = alive[-1].data_ptr()
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            if i < plan.first:                      # layers[0] runs as the stem kernel straight from NCHW
                plan.load_input(x, stream)
            else:
                e = plan.table[i]
                L.check(lib.yolo_conv_fwd_ws(e.d, e.x, e.w_packed, e.scale, e.shift, e.residual, e.y, e.workspace, e.workspace_bytes,
                                             plan.nan_flag.data_ptr(), stream), "yolo_conv_fwd_ws")
            evs[i + 1].record()
        torch.cuda.synchronize()
        acc += np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n)])
    return acc / reps


def conv_roofline(model, x, dtype, args, elapsed, gflop_img):
    """MFMA roofline of the 3x3 launches of one forward plan: algorithmic conv FLOPs / HIP-event launch durations."""
    plan = next(iter(model._engine._plans.values()))
    flops = conv_flops(plan)
    with torch.no_grad():
        times_ms = per_launch_times(plan, max(1, min(args.steps, 5)), x)
    log(f"per-launch event timing done ({dtype})")
    if args.per_layer:
        for i, (op, t, f) in enumerate(zip(plan.prog.ops, times_ms, flops)):
            cv = op["block"].conv
            log(f"  op{i:2d} {cv.in_channels:4d}->{cv.out_channels:4d} k{op['k']} s{op['s']} {op['x'].H:3d}->{op['Ho']:3d} "
                f"{t * 1e3:8.1f} us {f / t / 1e9:7.1f} TF")
    is3 = np.array([op["k"] == 3 and i >= plan.first for i, op in enumerate(plan.prog.ops)])   # MFMA 3x3 launches
    f3, t3 = float(np.sum(np.array(flops)[is3])), float(np.sum(times_ms[is3])) * 1e-3
    fall, tall = float(np.sum(flops)), float(np.sum(times_ms)) * 1e-3
    ach = f3 / t3 / 1e12
    fp32 = dtype == "fp32"
    # launches that run as Winograd F(2x2, 3x3) (conv_wino_f32: input-transform pass + 16 GEMMs + output transform): the
    # algorithm multiplies 16 / 36 of the direct convolution's products, so its ALGORITHMIC flops are the direct ones / 2.25
    from yolo_for_turbines_amd import _lib as L
    isw = np.array([i >= plan.first and L.lib().yolo_conv_workspace_bytes(plan.table[i].d) > 0 for i in range(len(flops))])
    if fp32 and isw.any():
        return wino_roofline(plan, flops, times_ms, is3, isw, args, elapsed, gflop_img)
    peak = PEAK_F32_MFMA_TFLOPS if fp32 else PEAK_H16_MFMA_TFLOPS
    at_cfg1 = args.batch == 32 and args.size == 416
    out = {
        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(ach / peak, 4),
        # HBM bytes of one launch of the dominant kernel (128->256 3x3 at 52x52, batch 32) from the PMC passes committed under
        # profiles/ (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md)
        "traffic": (PMC_TRAFFIC.get("conv_patch_f32" if fp32 else "conv3_dma_h16", (TRAFFIC_F32_C52 if fp32 else TRAFFIC_H16_C52, ""))[0])
        if at_cfg1 else None,
        "traffic_source": (PMC_TRAFFIC.get("conv_patch_f32" if fp32 else "conv3_dma_h16",
                                           (0, "profiles/r01/pmc_hbm_traffic.txt" if fp32 else "profiles/r02/pmc_hbm_traffic.txt"))[1] +
                           " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/conv_bench.py on this kernel; NOT "
                           "measured by the run that printed this line: a process cannot collect PMC counters on itself)"),
        "traffic_unit": ("HBM bytes per launch (PMC, conv_patch_f32<3,64> 128->256 @52x52); algorithmic 134.1e6" if fp32 else
                         "HBM bytes per launch (PMC, conv3_dma_h16 128->256 @52x52); algorithmic 67.1e6"),
        "kernel": ("conv_patch_f32 / conv_igemm_f32 (3x3 launches, v_mfma_f32_32x32x2_f32)" if fp32 else
                   f"conv3_dma_h16 (stride-1 3x3) + conv1_dma_h16 with gathered rows (stride-2 3x3) + conv3_ws_h16 (<= 64 channels), v_mfma_f32_32x32x16_{'f16' if dtype == 'fp16' else 'bf16'}"),
        "launches_per_step": int(is3.sum()), "avg_launch_us": round(t3 / int(is3.sum()) * 1e6, 2),
        "algorithmic_gflop_per_step": round(f3 / 1e9, 2),
        "all_conv_launches": {"achieved": round(fall / tall / 1e12, 2), "launches_per_step": len(flops),
                              "sum_kernel_ms": round(tall * 1e3, 3), "gflop_per_step": round(fall / 1e9, 2)},
        "whole_step_tflops": round(fall * args.steps / elapsed / 1e12, 2) if gflop_img else None,
    }
    return out


def wino_roofline(plan, flops, times_ms, is3, isw, args, elapsed, gflop_img):
    """fp32 forward with the Winograd launches. `achieved` follows SURVEY 8d to the letter: ALGORITHMIC flops of the layers
    (2*H*W*Cout*Cin*9, the direct convolution's) / the HIP-event time of their launch pairs (transform pass + GEMM kernel) -
    and can therefore exceed the f32 matrix peak, because F(2x2,3x3) multiplies 16 / 36 of those products. The matrix cores'
    own utilisation (EXECUTED flops = algorithmic / 2.25 over the same time) is `mfma_utilisation`; the remaining direct
    3x3 launches (stride 2, 32 input channels: conv_igemm_f32 / conv_patch_f32) are listed beside it."""
    fl, t = np.array(flops), times_ms * 1e-3
    fw_direct, tw = float(fl[isw].sum()), float(t[isw].sum())
    fw_exec = fw_direct / 2.25
    isd = is3 & ~isw
    fd, td = float(fl[isd].sum()), float(t[isd].sum())
    f3, t3 = float(fl[is3].sum()), float(t[is3].sum())
    fall, tall = float(fl.sum()), float(t.sum())
    executed_all = fall - fw_direct + fw_exec
    at_cfg1 = args.batch == 32 and args.size == 416
    tg, tx = PMC_TRAFFIC.get("conv_wino_f32", (None, "")), PMC_TRAFFIC.get("wino_xform_f32", (None, ""))
    tr = (tg[0] + tx[0], tg[1]) if tg[0] and tx[0] else (None, "")
    return {
        "bound": "mfma", "achieved": round(fw_direct / tw / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(fw_direct / tw / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        "frac_note": "above 1 by construction: SURVEY 8d prices a 3x3 layer at the direct convolution's 2*H*W*Cout*Cin*9 flops and these "
                     "launches run Winograd F(2x2,3x3), which multiplies 16/36 of those products - fewer multiplications, not faster "
                     "matrix cores; the cores' own utilisation is mfma_utilisation",
        "executed_tflops": round(fw_exec / tw / 1e12, 2),
        "mfma_utilisation": round(fw_exec / tw / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        "kernel": "conv_wino_f32 (+ wino_xform_f32): fp32 3x3 stride 1 with >= 64 input channels as Winograd F(2x2,3x3), 16 GEMMs on "
                  "v_mfma_f32_32x32x2_f32; time = both launches of a layer",
        "launches_per_step": int(isw.sum()), "avg_launch_us": round(tw / int(isw.sum()) * 1e6, 2),
        "algorithmic_gflop_per_step": round(fw_direct / 1e9, 2), "executed_gflop_per_step": round(fw_exec / 1e9, 2),
        "traffic": tr[0] if at_cfg1 else None,
        "traffic_source": tr[1] or "no PMC pass of conv_wino_f32 committed yet",
        "traffic_unit": "HBM bytes per layer (PMC: wino_xform_f32 + conv_wino_f32, 128->256 @52x52 with a residual); algorithmic 222.7e6 "
                        "(input, residual, output, filters) + 2 x 177.2e6 for the transformed tiles written and read once",
        "direct_3x3_launches": {"kernel": "conv_patch_f32 / conv_igemm_f32 (stride 2, 32 input channels)", "launches_per_step": int(isd.sum()),
                                "achieved": round(fd / td / 1e12, 2) if td else None,
                                "frac": round(fd / td / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if td else None,
                                "avg_launch_us": round(td / max(1, int(isd.sum())) * 1e6, 2)},
        "all_3x3_launches": {"achieved": round(f3 / t3 / 1e12, 2), "frac": round(f3 / t3 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                             "launches_per_step": int(is3.sum())},
        "all_conv_launches": {"achieved": round(fall / tall / 1e12, 2), "executed": round(executed_all / tall / 1e12, 2),
                              "launches_per_step": len(flops), "sum_kernel_ms": round(tall * 1e3, 3),
                              "gflop_per_step": round(fall / 1e9, 2), "executed_gflop_per_step": round(executed_all / 1e9, 2)},
        "whole_step_tflops": round(fall * args.steps / elapsed / 1e12, 2) if gflop_img else None,
        "whole_step_tflops_executed": round(executed_all * args.steps / elapsed / 1e12, 2) if gflop_img else None,
    }


# HBM bytes per launch of the dominant kernels from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS round's
# binary (bench.py cannot run a PMC pass on itself): kernel -> (bytes, source file). Filled in by tools/profile_round.sh pmc.
PMC_TRAFFIC = {}
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03", "pmc_traffic.json")) as _f:
        PMC_TRAFFIC = {k: (v["bytes"], v["source"]) for k, v in json.load(_f).items()}
except (OSError, ValueError, KeyError):
    pass


def nms_bench(yt, device, images=16, n=10000, nc=80, reps=5):
    from tests import golden_inputs as gi       # seeded box generators (data only)

    def leg(batch):
        t = torch.from_numpy(batch).to(device)
        yt.nms_indices(t, 0.45, 0.5, "center")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            keep, count = yt.nms_indices(t, 0.45, 0.5, "center")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        return dt, float(count.float().mean())

    batch = np.stack([gi.boxes_uniform(n, nc, 1000 + b) for b in range(images)])
    dt, kept = leg(batch)
    out = dict(boxes_per_s=images * n / dt, images=images, boxes_per_image=n, classes=nc, ms_per_batch=dt * 1e3, kept_mean=kept)
    # the fine-tune class count (2 classes: N^2 / 4 same-class pairs per image) and the clustered generator (SURVEY 8d Config 5 b)
    for name, b2 in (("uniform_2_classes", np.stack([gi.boxes_uniform(n, 2, 1000 + b) for b in range(images)])),
                     ("clustered_2_classes", np.stack([gi.boxes_clustered(n, 2, 1000 + b, jitter=0.15) for b in range(images)])),
                     # SURVEY 8d Config 5 (b) as written: 60 objects per image, ~167 jittered boxes each (sigma = 5 % of the size)
                     ("clustered_80_classes", np.stack([gi.boxes_clustered(n, nc, 1000 + b) for b in range(images)]))):
        dt2, kept2 = leg(b2)
        out[name] = dict(boxes_per_s=images * n / dt2, ms_per_batch=dt2 * 1e3, kept_mean=kept2)
    return out, batch


def decode_bench(yt, device, batch=32, size=416, nc=80, reps=20):
    """cells_to_boxes (utils.py:86-148) for the three scales of one batch: an HBM-bound pass. Algorithmic bytes
    (SURVEY 8d): B * sum(3 g^2) * ((5+nc)*4 read + 4*4 written back in place + 6*4 boxes written)."""
    g = [size // 32, size // 16, size // 8]
    gen = torch.Generator().manual_seed(11)
    preds = [torch.randn((batch, 3, gg, gg, 5 + nc), generator=gen).to(device) for gg in g]
    anchors = [torch.rand((3, 2), generator=gen).to(device) * gg for gg in g]
    n_total = sum(3 * gg * gg for gg in g)
    out = torch.empty((batch, n_total, 6), dtype=torch.float32, device=device)

    import ctypes as C
    from yolo_for_turbines_amd import _lib as L
    pp = (C.c_void_p * 3)(*[p.data_ptr() for p in preds])
    st = (C.c_int64 * 15)(*[v for p in preds for v in p.stride()])
    ap = (C.c_void_p * 3)(*[a.data_ptr() for a in anchors])
    gg3 = (C.c_int * 3)(*g)

    def timed(write_back):                                 # what yt.detect() launches: the three scales in one kernel
        def run():
            L.check(L.lib().yolo_decode3_ex(pp, st, ap, gg3, batch, nc, write_back, out.data_ptr(), n_total, L.current_stream()),
                    "yolo_decode3_ex")
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timed(0)                                          # detect() / detect_images(): predictions left alone
    ms_wb = timed(1)                                       # cells_to_boxes semantics: sigmoid / exp written back in place
    nbytes = batch * n_total * ((5 + nc) * 4 + 6 * 4)
    nbytes_wb = nbytes + batch * n_total * 4 * 4
    traffic = PMC_TRAFFIC.get("decode3") if (batch, size, nc) == (32, 416, 80) else None
    return {"workload": f"batch {batch}, {size}x{size}, {nc} classes: {n_total} boxes/image, 3 scales in 1 launch, no in-place "
                        "write-back (the detect path)", "ms": round(ms, 4),
            "boxes_per_s": round(batch * n_total / ms * 1e3, 1), "algorithmic_bytes": nbytes,
            "roofline": {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(nbytes / ms / 1e6 / 8000.0, 4),
                         "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None},
            "with_write_back": {"workload": "cells_to_boxes semantics (utils.py:106-110): 16 more bytes written into every cell",
                                "ms": round(ms_wb, 4), "algorithmic_bytes": nbytes_wb,
                                "achieved_GBps": round(nbytes_wb / ms_wb / 1e6, 1)}}


COCO_ANCHORS = [[(0.28, 0.22), (0.38, 0.48), (0.9, 0.78)], [(0.07, 0.15), (0.15, 0.11), (0.14, 0.29)],
                [(0.02, 0.03), (0.04, 0.07), (0.08, 0.06)]]


def gi_boxes(batch, mean_boxes=9, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    anc = np.asarray(COCO_ANCHORS).reshape(9, 2)
    out = []
    for _ in range(batch):
        n = max(1, rng.poisson(mean_boxes))
        a = rng.integers(0, 9, n)
        wh = np.clip(anc[a] * np.exp(0.3 * rng.standard_normal((n, 2))), 0.01, 0.95)
        xy = rng.uniform(0.01, 0.99, (n, 2))
        out.append(np.concatenate([xy, wh, rng.integers(0, 80, (n, 1))], 1).astype(np.float32).tolist())
    return out


def aux_legs(yt, device, result, world, args):
    """Secondary measurements next to the path (rank 0 only, no collectives): decode bandwidth, target builder, mAP."""
    result["decode"] = decode_bench(yt, device)
    # ground-truth tensor builder (dataset.py:119-161) for one batch-64 416x416 batch of seeded COCO-shaped boxes
    tb = gi_boxes(64)
    yt.build_targets(tb, COCO_ANCHORS, 416)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        yt.build_targets(tb, COCO_ANCHORS, 416)
    torch.cuda.synchronize()
    dt_t = (time.perf_counter() - t0) / 10
    result["targets"] = {"workload": "batch 64, 416x416, ~9 boxes/image incl. host list -> device copy", "ms": round(dt_t * 1e3, 3),
                         "images_per_s": round(64 / dt_t, 1)}
    # mAP of one evaluation pass (utils.py:193-274): 128 images, 20 classes, ~8 ground truths and 50 kept boxes per image
    rng_m = np.random.Generator(np.random.PCG64(9))
    pb, tbx = [], []
    for img in range(128):
        for _ in range(8):
            cls = int(rng_m.integers(0, 20))
            bb = [float(np.float32(v)) for v in (*rng_m.uniform(0.2, 0.8, 2), *rng_m.uniform(0.05, 0.3, 2))]
            tbx.append([img, *bb, 1.0, cls])
            for _ in range(5):
                pb.append([img, *[float(np.float32(v + 0.03 * rng_m.standard_normal())) for v in bb],
                           float(np.float32(rng_m.uniform(0.2, 1))), cls])
        for _ in range(10):
            pb.append([img, *[float(np.float32(v)) for v in rng_m.uniform(0.1, 0.9, 4)], float(np.float32(rng_m.uniform(0.1, 0.9))),
                       int(rng_m.integers(0, 20))])
    pt, tt = torch.tensor(pb).to(device), torch.tensor(tbx).to(device)
    m_val = float(yt.calc_mAP(pt, tt, 0.5, "center", 20))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        yt.calc_mAP(pt, tt, 0.5, "center", 20)
    torch.cuda.synchronize()
    dt_m = (time.perf_counter() - t0) / 5
    result["map"] = {"workload": f"{len(pb)} detections, {len(tbx)} ground truths, 128 images, 20 classes", "ms": round(dt_m * 1e3, 3),
                     "detections_per_s": round(len(pb) / dt_m, 1), "mAP": round(m_val, 6)}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import metrics as om
        sub_p, sub_t = [r for r in pb if r[0] < 16], [r for r in tbx if r[0] < 16]
        t0 = time.perf_counter()
        om.calc_map(sub_p, sub_t, 0.5, "center", 20)
        dt_c = time.perf_counter() - t0
        result["map"]["cpu_port_detections_per_s"] = round(len(sub_p) / dt_c, 1)
        result["map"]["cpu_port_sample"] = f"first 16 images ({len(sub_p)} detections), oracle/metrics.py restatement of calc_mAP"


def launch_argv(n_gpus, port, script_args):
    """Command line of the N-rank child: the launcher of the bench contract, this script, the caller's own flags."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(script_args)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def kfd_gpu_count():
    """GPUs of this node from the KFD topology in sysfs (nodes with SIMDs), without touching HIP: the launcher parent must
    not bring up the runtime it then leaves to its children. None when the topology cannot be read."""
    import glob
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(ln.split()[:2] for ln in open(path) if len(ln.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if seen else None


def self_launch(args):
    import subprocess
    script_args = [a for a in sys.argv[1:] if a != "--dry-launch"]
    argv = launch_argv(args.gpus, int(os.environ.get("MASTER_PORT") or free_port()), script_args)
    if args.dry_launch:
        print(json.dumps({"argv": argv}))
        return 0
    # A profiler's preloaded library initialises the GPU before this process runs its first line; starting the ranks from
    # here would then be a launcher hop out of a GPU-initialised process. Profile multi-rank runs by starting torchrun first.
    if any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise SystemExit("bench.py --gpus N was started under a profiler: launch `python -m torch.distributed.run ... bench.py` "
                         "yourself (see --dry-launch) and put the profiler in front of that")
    n_dev = kfd_gpu_count()                                # sysfs only: torch.cuda.device_count() may open the HIP runtime
    if n_dev is not None and n_dev < args.gpus:            # (unreadable topology: let the ranks report)
        raise SystemExit(f"--gpus {args.gpus} but this node shows {n_dev} GPU(s)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    log(f"starting {args.gpus} ranks: {' '.join(argv)}")
    child = subprocess.run(argv, env=env, stdout=subprocess.PIPE, text=True)     # stderr passes through
    lines = [ln for ln in child.stdout.splitlines() if ln.lstrip().startswith("{")]
    if child.returncode != 0 or not lines:
        sys.stderr.write(child.stdout)
        raise SystemExit(child.returncode or 1)
    print(lines[-1], flush=True)                           # rank 0's ONE JSON line
    return 0


def cpu_train_step_baseline(size, batch=4, nc=2):
    """One fine-tune step of the CPU port (train.py:41-69 sequence: train-mode forward, 3 x per-scale loss, backward, SGD with
    momentum / weight decay) at BASELINE.md's CPU shape: batch 4, 2 classes, 416x416. Timed on the second step (the first
    pays allocator / oneDNN first-touch)."""
    from oracle import loss as oloss
    from oracle import net as onet
    from tests import golden_inputs as gi
    anchors = gi.TRAIN_CASE["anchors"]
    sd = onet.synth_state_dict(3, 3, nc, gain=0.8)
    x = onet.synth_input(4, batch, size)
    tg = [torch.from_numpy(t) for t in gi.synth_targets(batch, size, nc, anchors, 5)]
    sa = torch.tensor(anchors) * torch.tensor([size // 32, size // 16, size // 8]).view(3, 1, 1)
    par = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = dict(sd)
    full.update(par)
    opt = torch.optim.SGD(list(par.values()), lr=1e-4, momentum=0.9, weight_decay=5e-4)
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        opt.zero_grad()
        pr = onet.forward(full, x, nc, "leaky_relu", training=True, new_stats={})
        sum(sum(oloss.yolo_loss(pr[i], tg[i].clone(), sa[i])) for i in range(3)).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    return {"value": round(batch / times[-1], 3), "unit": "images/s", "first_step_images_per_s": round(batch / times[0], 3),
            "sample": f"1 timed step (after 1 untimed) of batch {batch}, {nc} classes, {size}x{size} fp32: forward(train-mode BN) + 3 x loss + "
                      "backward + SGD (oracle/net.py + oracle/loss.py under torch autograd)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--torch-sgd", action="store_true", help="fine-tune legs: torch.optim.SGD instead of yt.SGD (same update, ~19 launches)")
    ap.add_argument("--tile", type=int, default=0, help="force a conv tile id (tuning)")
    ap.add_argument("--per-layer", action="store_true", help="print per-launch times to stderr")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16", "bf16"],
                    help="compute dtype of the forward leg (BASELINE configs[1] = fp32; fp16 / bf16 = configs 4-5 arithmetic)")
    ap.add_argument("--config5", action="store_true", help="(default on; kept for compatibility)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip BASELINE config 5 per-GPU shape: batch 16, 608x608 fp16 forward + decode + per-image NMS")
    ap.add_argument("--no-h16", action="store_true", help="skip the secondary bf16 forward leg of the default (fp32) run")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip BASELINE config 3 shape: batch 64 multi-scale fine-tune steps (S cycles through 320..608)")
    ap.add_argument("--train-steps", type=int, default=5, help="timed fine-tune steps (0 = skip the fwd+bwd leg)")
    ap.add_argument("--train-classes", type=int, default=2, help="fine-tune head (BASELINE configs[2-3]: 2-class turbine head)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 outside torchrun: print the child command line (JSON) instead of starting it")
    args = ap.parse_args()

    # ---- N > 1 from a plain `python bench.py --gpus N`: start the N ranks OURSELVES, as a child process, before anything in
    # this process touches the GPU (no HIP call, no library load: replacing or forking a GPU-initialised process is not
    # allowed on this pool). The child is the launcher the contract names (`python -m torch.distributed.run`, one rank per
    # GPU, rendezvous on 127.0.0.1); its rank 0 prints the single JSON line, which is relayed verbatim, and its exit code
    # becomes ours. Under torchrun (WORLD_SIZE set) this block is skipped and the process is one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    # stdout carries exactly ONE JSON line: libraries (RCCL prints a version banner on fd 1) are sent to
    # stderr for the whole run, the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import yolo_for_turbines_amd as yt
    from yolo_for_turbines_amd import dist as ydist
    rank, local_rank, world = ydist.env_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = ydist.init("nccl", device)                              # "nccl" is RCCL on ROCm; None for a single process

    model = seeded_model(yt, args.classes, device)
    if args.tile:
        model._engine.tile_override = args.tile
    model._engine.compute_dtype = args.dtype
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand((args.batch, 3, args.size, args.size), generator=g).to(device)   # resident in HBM before timing
    log(f"model + input ready on {device}")
    with torch.no_grad():
        elapsed = ydist.timed_steps(lambda: model(x), args.steps, args.warmup, dist, device)
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps")
    # ---- the same forward in bf16 (BASELINE configs[3-5] arithmetic; conv3_dma_h16 is its dominant kernel): a secondary
    # line of the default run, never `value`
    h16, m16, el16 = None, None, 0.0
    if args.dtype == "fp32" and not args.no_h16:
        m16 = seeded_model(yt, args.classes, device)
        m16._engine.compute_dtype = "bf16"
        with torch.no_grad():
            # two timed regions of `steps` each, both reported, the faster one is `value`: a 3.5 ms step is short enough for one
            # host hiccup (50 ms once in ~10 runs on the pool's boxes) to halve a single region; the fp32 headline above is
            # one region, as the contract says
            regions = [ydist.timed_steps(lambda: m16(x), args.steps, args.warmup, dist, device) for _ in range(2)]
        el16 = min(regions)
        h16 = {"metric": "images/sec at 416x416 (fwd)", "dtype": "bf16", "unit": "images/s",
               "value": round(args.batch * world * args.steps / el16, 2), "ms_per_step": round(el16 / args.steps * 1e3, 4),
               "timed_regions_ms_per_step": [round(r / args.steps * 1e3, 4) for r in regions],
               "note": "same weights, input and step count as the fp32 headline; 16-bit activations / weights, fp32 accumulation and heads"}
        log(f"bf16 forward leg: {h16['value']} img/s")
    # ---------------------------------------------------------------- fwd+bwd leg (fine-tune step)
    train = None
    if args.train_steps > 0:
        from tests import golden_inputs as gi           # seeded synthetic targets / turbine anchors (data only)
        tm = seeded_model(yt, args.train_classes, device, seed=1).train()
        if dist is not None:
            ydist.data_parallel(tm, dist)
        anchors = gi.TRAIN_CASE["anchors"]
        grids = [args.size // 32, args.size // 16, args.size // 8]
        sa = (torch.tensor(anchors) * torch.tensor(grids).view(3, 1, 1)).to(device)
        tg = [torch.from_numpy(t).to(device) for t in gi.synth_targets(args.batch, args.size, args.train_classes, anchors, 3 + rank)]
        # train.py:171-172: SGD(model.parameters(), lr, momentum, weight_decay). yt.SGD is the same optimizer (a torch.optim.SGD
        # subclass: same state, same bits per step, tests/test_gpu_parity.py::test_sgd_step_same_bits_as_torch) with the update
        # as one HIP launch; --torch-sgd times PyTorch's own multi-tensor implementation instead
        opt = (torch.optim.SGD if args.torch_sgd else yt.SGD)(tm.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
        n_par = sum(p.numel() for p in tm.parameters())
        step_desc = ("zero_grad + forward(train-mode BN) + 3 x per-scale loss + backward + SGD (" +
                     ("torch.optim.SGD" if args.torch_sgd else "yt.SGD: one launch, same bits as torch.optim.SGD") + ")")

        def make_step(lf, autocast_dtype):              # train.py:41-69: zero_grad, autocast forward, 3 x loss, backward, SGD
            # the reference's YOLOLoss overwrites its targets (loss.py:70) and train.py gets fresh ones from the loader every batch;
            # this loop reuses one set, so that loss gets a copy per step. FusedYOLOLoss leaves its inputs alone: no copy
            fresh = (lambda t: t.clone()) if isinstance(lf, yt.YOLOLoss) else (lambda t: t)

            def train_step():
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
                    preds = tm(x)                       # the loss sits inside the autocast block, as in train.py:53-65
                    loss = sum(sum(lf(preds[i], fresh(tg[i]), sa[i])) for i in range(3))
                loss.backward()
                opt.step()
            return train_step

        def entry(t_el, name, loss_name):
            return {"metric": "images/sec at 416x416 (fwd+bwd)", "value": round(args.batch * world * args.train_steps / t_el, 2),
                    "unit": "images/s", "ms_per_step": round(t_el / args.train_steps * 1e3, 3), "steps": args.train_steps,
                    "per_gpu_batch": args.batch, "num_classes": args.train_classes, "dtype": name, "step": step_desc, "loss": loss_name,
                    "parallelism": f"dp{world}" + (f": bucketed RCCL all-reduce of {n_par * 4 / 1e6:.1f} MB fp32 gradients" if world > 1 else ""),
                    "algorithmic_tflops": round(3 * 65.297 * (args.size / 416.0) ** 2 * args.batch * world * args.train_steps / t_el / 1e3, 2)}

        def graph_leg(autocast_dtype):                  # the same step captured ONCE into a HIP graph and replayed
            gstep = yt.GraphedTrainStep(tm, opt, sa, x, tg, autocast_dtype=autocast_dtype)
            t_el = ydist.timed_steps(lambda: gstep(x, tg), args.train_steps, 2, dist, device)
            gstep.release()                             # un-pin the train plan
            del gstep
            return t_el

        def train_leg(autocast_dtype):
            name = "f32" if autocast_dtype is None else "bf16"
            t_f = ydist.timed_steps(make_step(yt.FusedYOLOLoss(), autocast_dtype), args.train_steps, 2, dist, device)
            log(f"train leg ({name}, fused loss kernels): {t_f:.3f} s for {args.train_steps} steps")
            e = entry(t_f, name, "FusedYOLOLoss (3 HIP kernels per scale; same values/gradients as loss.py:29-81)")
            t_p = ydist.timed_steps(make_step(yt.YOLOLoss(), autocast_dtype), args.train_steps, 2, dist, device)
            e["with_pytorch_loss"] = entry(t_p, name, "YOLOLoss (the reference's boolean-mask PyTorch ops)")
            # the same eager step with the NaN guards of model.py:175,183-184 reported a step later instead of inside the forward
            # (nan_check = "deferred": no host sync between forward and backward; same exceptions). Default stays the reference's.
            tm._engine.nan_check = "deferred"
            try:
                t_d = ydist.timed_steps(make_step(yt.FusedYOLOLoss(), autocast_dtype), args.train_steps, 2, dist, device)
                tm._engine.flush_nan()
            finally:
                tm._engine.nan_check = True
            e["deferred_nan_guard"] = entry(t_d, name, "FusedYOLOLoss; eager, NaN guards read without a host sync inside the step")
            if dist is None:                            # no process group: capture never sees a collective (a forced 1-rank
                try:                                    # RCCL group inside a capture crashed once in ~10 runs)
                    t_g = graph_leg(autocast_dtype)
                    e["hip_graph"] = entry(t_g, name, "FusedYOLOLoss; whole step replayed as one HIP graph")
                except Exception as ex:                 # capture is an optimisation, never the measured default
                    log(f"graph capture skipped: {ex!r}")
                    torch.cuda.synchronize()
            return e

        train = train_leg(None)
        # BASELINE configs[3] arithmetic: the same step under torch.autocast(bf16) (train.py:53) -> 16-bit kernels for
        # activations and activation gradients, fp32 master weights / statistics / parameter gradients
        train["bf16_autocast"] = train_leg(torch.bfloat16)
        # ---- BASELINE configs[2] shape: batch 64, 2 classes, S switching between sizes (train.py:45-46; the list
        # there is 416..608, SURVEY 8d widens it to 320..608). 2 timed steps per size after 1 untimed at the new size
        # (plan build + first-touch of ~60 GB of buffers is a per-switch cost the reference pays every 10 batches).
        if not args.no_config3 and world == 1:
            sizes, b3 = list(range(320, 609, 32)), 64        # SURVEY 8d Config 3: all ten sizes 320, 352, ..., 608
            per_size, tot_img, tot_t = {}, 0, 0.0
            lf3 = yt.FusedYOLOLoss()
            for S3 in sizes:
                x3 = torch.rand((b3, 3, S3, S3), generator=g).to(device)
                g3 = [S3 // 32, S3 // 16, S3 // 8]
                sa3 = (torch.tensor(anchors) * torch.tensor(g3).view(3, 1, 1)).to(device)
                tg3 = [torch.from_numpy(t).to(device) for t in gi.synth_targets(b3, S3, args.train_classes, anchors, 7)]

                def step3():
                    opt.zero_grad(set_to_none=True)
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        preds = tm(x3)
                        loss = sum(sum(lf3(preds[i], tg3[i], sa3[i])) for i in range(3))
                    loss.backward()
                    opt.step()
                t3_el = ydist.timed_steps(step3, 2, 1, dist, device)
                per_size[str(S3)] = round(b3 * 2 / t3_el, 1)
                tot_img += b3 * 2
                tot_t += t3_el
                del x3, tg3
            train["config3_multiscale"] = {"workload": "batch 64, 2 classes, bf16 autocast, S in 320..608 step 32 (ten sizes), 2 timed "
                                                       "steps per size after 1 untimed step at the new size",
                                           "value": round(tot_img / tot_t, 2), "unit": "images/s", "images_per_s_by_size": per_size}
            log(f"config3 leg: {tot_img / tot_t:.1f} img/s")
        del tm, opt
        torch.cuda.empty_cache()

    # ---------------------------------------------------------------- config 5 leg (optional)
    cfg5 = None
    if not args.no_config5:
        m5 = seeded_model(yt, 80, device, seed=2)
        m5._engine.compute_dtype = "fp16"
        x5 = torch.rand((16, 3, 608, 608), generator=g).to(device)
        anchors5 = [[(0.28, 0.22), (0.38, 0.48), (0.9, 0.78)], [(0.07, 0.15), (0.15, 0.11), (0.14, 0.29)],
                    [(0.02, 0.03), (0.04, 0.07), (0.08, 0.06)]]
        sa5 = [torch.tensor(a).to(device) * gsz for a, gsz in zip(anchors5, (19, 38, 76))]

        def step5():                                    # forward + decode + NMS, one host sync per batch (the NaN guards of the forward)
            yt.detect_images(m5, x5, sa5, 0.45, 0.5, "center")
        t5 = ydist.timed_steps(step5, 10, 2, dist, device)
        b5, keep5, cnt5 = yt.detect_images(m5, x5, sa5, 0.45, 0.5, "center")
        cfg5 = {"workload": "BASELINE configs[4] per-GPU shape: batch 16, 608x608 fp16 forward + decode (22,743 boxes/image) + per-image NMS",
                "value": round(16 * world * 10 / t5, 2), "unit": "images/s", "ms_per_step": round(t5 / 10 * 1e3, 3),
                # what the NMS of THIS leg actually consumed (random-init weights: far fewer than the 10,000 post-threshold
                # boxes per image of SURVEY 8d Config 5, which the `nms` legs below use)
                "post_threshold_boxes_mean": round(float((b5[..., 4] > 0.5).sum(1).float().mean()), 1),
                "kept_boxes_mean": round(float(cnt5.float().mean()), 1)}
        del m5
        torch.cuda.empty_cache()

    images = args.batch * world * args.steps
    value = images / elapsed
    gflop_img = GFLOP_PER_IMAGE_416_NC80 if (args.size == 416 and args.classes == 80) else None

    result = {
        "metric": "images/sec at 416x416 (fwd)", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.dtype], "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch {args.batch}/GPU {args.size}x{args.size} {args.dtype} inference, "
                               f"{args.classes}-class head, YOLOv3 forward (75 fused conv launches)",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "image_size": args.size,
                   "num_classes": args.classes, "parallelism": f"image-sharded x{world} (no collective)"},
    }

    if train is not None:
        result["train"] = train
    if cfg5 is not None:
        result["config5"] = cfg5
    if rank == 0:
        # ------------------------------------------------------------ roofline (dominant kernel)
        result["roofline"] = conv_roofline(model, x, args.dtype, args, elapsed, gflop_img)
        if h16 is not None:
            h16["roofline"] = conv_roofline(m16, x, "bf16", args, el16, gflop_img)
            result["forward_bf16"] = h16
        # ------------------------------------------------------------------ NMS secondary metric
        nms_batch = None
        if not args.no_nms:
            result["nms"], nms_batch = nms_bench(yt, device)
            try:                                            # auxiliary legs must never cost the headline line
                aux_legs(yt, device, result, world, args)
            except Exception as ex:                         # pragma: no cover
                result["aux_error"] = repr(ex)
            log("nms + decode + targets + mAP bench done")
        # -------------------------------------------------------------------------- CPU baseline
        if world == 1 and not args.no_cpu_baseline:
            from oracle import net as onet
            from oracle import postprocess as opp
            torch.set_num_threads(host_cores())
            log(f"cpu baseline on {torch.get_num_threads()} threads")
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            xb = x[:8].cpu()
            with torch.no_grad():
                onet.forward(sd, xb[:1], args.classes)          # first-touch
                t0 = time.perf_counter()
                reps = 0
                while reps < 2 or (time.perf_counter() - t0 < 10 and reps < 8):
                    onet.forward(sd, xb, args.classes)
                    reps += 1
                dt = time.perf_counter() - t0
            log("cpu forward baseline done")
            cores = torch.get_num_threads()
            cb = result["cpu_baseline"] = {
                "value": round(8 * reps / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x batch 8 of the same {args.size}x{args.size} fp32 forward (oracle/net.py, torch CPU ops)"}
            # the other CPU legs BASELINE.md section 3 names, each on a bounded sample (a few seconds of CPU work in all)
            with torch.no_grad():
                t0 = time.perf_counter()
                for _ in range(3):
                    onet.forward(sd, xb[:1], args.classes)
                cb["forward_batch1"] = {"value": round(3 / (time.perf_counter() - t0), 2), "unit": "images/s",
                                        "sample": f"3 x batch 1 of the {args.size}x{args.size} fp32 forward"}
            cb["train_step"] = cpu_train_step_baseline(args.size)
            log("cpu fine-tune step baseline done")
            if nms_batch is not None:
                from tests import golden_inputs as gi_b
                legs = {"uniform_80_classes": nms_batch[0],
                        "uniform_2_classes": gi_b.boxes_uniform(nms_batch.shape[1], 2, 1000),
                        "clustered_80_classes": gi_b.boxes_clustered(nms_batch.shape[1], 80, 1000)}
                cb["nms"] = {}
                for name_l, boxes_l in legs.items():
                    t0 = time.perf_counter()
                    kept = opp.nms_list(np.asarray(boxes_l).tolist(), 0.45, 0.5, "center")
                    dt = time.perf_counter() - t0
                    cb["nms"][name_l] = {"boxes_per_s": round(nms_batch.shape[1] / dt, 1), "kept": len(kept)}
                cb["nms"]["sample"] = "1 image x 10,000 post-threshold boxes per leg, list-based port (oracle/postprocess.py:nms_list, utils.py:150-191), 1 thread"
                result["nms"]["cpu_port_boxes_per_s"] = cb["nms"]["uniform_80_classes"]["boxes_per_s"]
                result["nms"]["cpu_port_sample"] = "1 image x 10,000 boxes, list-based port (oracle/postprocess.py:nms_list)"
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
