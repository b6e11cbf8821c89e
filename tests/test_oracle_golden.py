"""Pins the ORACLE (``oracle/``) against golden vectors produced by the imported reference
(``tests/gen_golden.py``).  CPU only.  If these fail the oracle cannot be trusted as a
checker for the HIP path."""
import numpy as np
import pytest
import torch

from oracle import net as onet
from oracle import postprocess as opp
from tests import golden_inputs as gi

F32 = np.float32


# ----------------------------------------------------------------------------- network
@pytest.mark.parametrize("name", ["nc80_s96_b2_leaky", "nc2_s128_b1_leaky", "nc80_s96_b1_mish"])
def test_net_forward_full(golden, name):
    g = golden("net_fwd")
    c = gi.NET_CASES[name]
    sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
    x = onet.synth_input(c["xseed"], c["batch"], c["size"])
    taps = {}
    with torch.no_grad():
        preds = onet.forward(sd, x, c["nc"], c["act"], taps=taps)
    for i, p in enumerate(preds):
        ref = g[f"{name}/p{i}"]
        assert tuple(p.shape) == ref.shape
        np.testing.assert_allclose(p.contiguous().numpy(), ref, rtol=0, atol=2e-5)
    for k in gi.TAP_KEYS:
        got = taps[k].contiguous().reshape(-1)[::gi.TAP_STRIDE].numpy()
        np.testing.assert_allclose(got, g[f"{name}/tap/{k}"], rtol=0, atol=2e-5)


def test_net_forward_416_sampled(golden):
    g = golden("net_fwd")
    name = "nc80_s416_b1_leaky"
    c = gi.NET_CASES[name]
    sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
    x = onet.synth_input(c["xseed"], c["batch"], c["size"])
    with torch.no_grad():
        preds = onet.forward(sd, x, c["nc"], c["act"])
    for i, p in enumerate(preds):
        p = p.contiguous()
        np.testing.assert_allclose(p.reshape(-1)[::gi.SAMPLE_STRIDE].numpy(), g[f"{name}/p{i}_sample"], rtol=0, atol=5e-5)
        s = g[f"{name}/p{i}_sums"]
        assert abs(float(p.double().sum()) - s[0]) <= 1e-5 * s[1]
        assert abs(float(p.double().abs().sum()) - s[1]) <= 1e-5 * s[1]


def test_state_dict_spec_matches_reference(golden):
    g = golden("loader")
    keys = [k for k, _ in onet.state_dict_spec(3, 80)]
    assert keys == list(g["state_dict_keys"])
    assert len(keys) == 438
    n_params = sum(int(np.prod(s)) for k, s in onet.state_dict_spec(3, 80)
                   if "running" not in k and "num_batches" not in k)
    assert n_params == 61949149
    n2 = sum(int(np.prod(s)) for k, s in onet.state_dict_spec(3, 2) if "running" not in k and "num_batches" not in k)
    assert n2 == 61529119


def test_darknet_stream_layout(golden):
    """The stream written by the oracle is read back by the reference loader at exactly the
    offsets stored in the golden map (full file), 62,001,757 floats in total."""
    g = golden("loader")
    keys, offs, cnts = list(g["full/keys"]), g["full/offsets"], g["full/counts"]
    assert int(offs[-1] + cnts[-1]) == 62001757
    assert keys[0] == "layers.0.batch_norm.bias" and int(offs[1]) == 32      # beta then gamma
    assert keys[4] == "layers.0.conv.weight" and int(offs[4]) == 128 and int(cnts[4]) == 864
    assert bool(g["full/loaded"].all())
    loaded = g["conv74/loaded"]
    last = [k for k, l in zip(g["conv74/keys"], loaded) if l][-1]
    assert last == "layers.8.layers.4.1.conv.weight"       # `.conv.74` cutoff quirk


# ------------------------------------------------------------------------ CNN block level
@pytest.mark.parametrize("i", range(len(gi.BLOCK_CONFIGS)))
def test_block_eval_and_train(golden, i):
    g = golden("blocks")
    cin, cout, k, s, bn, h = gi.BLOCK_CONFIGS[i]
    p = gi.block_params(i, cin, cout, k, bn)
    x = torch.from_numpy(gi.block_input(i, cin, h))
    cv = dict(prefix="b", cin=cin, cout=cout, k=k, stride=s, bn=bn)
    sd = {"b.conv.weight": torch.from_numpy(p["w"])}
    if bn:
        sd.update({"b.batch_norm.weight": torch.from_numpy(p["gamma"]), "b.batch_norm.bias": torch.from_numpy(p["beta"]),
                   "b.batch_norm.running_mean": torch.from_numpy(p["mean"]),
                   "b.batch_norm.running_var": torch.from_numpy(p["var"])})
    else:
        sd["b.conv.bias"] = torch.from_numpy(p["bias"])
    for act in (("leaky_relu", "mish") if bn else ("leaky_relu",)):
        tag = f"cfg{i}/{act}"
        with torch.no_grad():
            y = onet.cnn_block(sd, cv, x, act)
        np.testing.assert_allclose(y.reshape(-1)[::gi.BLOCK_STRIDE].numpy(), g[f"{tag}/eval"], rtol=0, atol=1e-5)
        if bn:
            stats = {}
            with torch.no_grad():
                y = onet.cnn_block(sd, cv, x, act, training=True, new_stats=stats)
            np.testing.assert_allclose(y.reshape(-1)[::gi.BLOCK_STRIDE].numpy(), g[f"{tag}/train"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(stats["b.batch_norm.running_mean"].numpy(), g[f"{tag}/new_mean"], atol=1e-6)
            np.testing.assert_allclose(stats["b.batch_norm.running_var"].numpy(), g[f"{tag}/new_var"], atol=1e-6)


# ------------------------------------------------------------------------------- decode
@pytest.mark.parametrize("name", list(gi.DECODE_CASES))
def test_decode(golden, name):
    g = golden("decode")
    pred, anchors = gi.decode_input(name)
    p = torch.from_numpy(pred.copy())
    boxes = opp.cells_to_boxes(p, torch.from_numpy(anchors), gi.DECODE_CASES[name]["g"])
    np.testing.assert_array_equal(boxes.numpy(), g[f"{name}/boxes"])          # same torch ops -> bitwise
    np.testing.assert_array_equal(p.numpy(), g[f"{name}/mutated"])
    # C restatement (libm expf): tolerance only; class ids exact
    c = opp.decode_c(pred, anchors, gi.DECODE_CASES[name]["g"])
    np.testing.assert_allclose(c[..., :5], g[f"{name}/boxes"][..., :5], rtol=2e-6, atol=1e-7)
    np.testing.assert_array_equal(c[..., 5], g[f"{name}/boxes"][..., 5])


def test_decode_box_order_and_kat(golden):
    g = golden("kat")
    z = torch.zeros((5, 3, 3, 3, 8))
    b = opp.cells_to_boxes(z, torch.tensor([[0.28, 0.22], [0.38, 0.48], [0.9, 0.78]]), 3)
    assert tuple(b.shape) == (5, 27, 6)                                       # utils_test.py:34-40
    np.testing.assert_array_equal(b.numpy(), g["c2b_zeros"])


# ---------------------------------------------------------------------------------- IoU
def test_iou_kats(golden):
    g = golden("kat")
    b = torch.tensor([0.5, 0.5, 0.25, 0.25])
    v = opp.calc_iou(b, b)
    np.testing.assert_array_equal(v.numpy(), g["iou_self"])
    assert float(v) == pytest.approx(0.0625 / (0.0625 + 1e-6), rel=1e-6) and float(v) != 1.0
    rng = np.random.Generator(np.random.PCG64(99))
    a = rng.random((200, 4), dtype=F32)
    c = rng.random((200, 4), dtype=F32)
    np.testing.assert_array_equal(opp.calc_iou(torch.from_numpy(a), torch.from_numpy(c), "center").numpy(), g["iou_center"])
    np.testing.assert_array_equal(opp.calc_iou(torch.from_numpy(a), torch.from_numpy(c), "corners").numpy(), g["iou_corners"])
    for j in range(200):                                  # numpy form, bitwise
        assert opp.iou_np(a[j], c[j:j + 1], True)[0] == g["iou_center"][j]
        assert opp.iou_np(a[j], c[j:j + 1], False)[0] == g["iou_corners"][j]


# ---------------------------------------------------------------------------------- NMS
SMALL = [n for n in gi.NMS_CASES if "10000" not in n]


@pytest.mark.parametrize("name", SMALL)
def test_nms_numpy_indices(golden, name):
    g = golden("nms")
    boxes, iou_thr, obj_thr, fmt = gi.nms_boxes(name)
    keep = opp.nms_indices(boxes.tolist(), iou_thr, obj_thr, fmt)
    np.testing.assert_array_equal(keep, g[f"{name}/keep"])


@pytest.mark.parametrize("name", list(gi.NMS_CASES))
def test_nms_c_indices(golden, name):
    g = golden("nms")
    boxes, iou_thr, obj_thr, fmt = gi.nms_boxes(name)
    keep = opp.nms_indices_c(boxes, iou_thr, obj_thr, fmt)
    np.testing.assert_array_equal(keep, g[f"{name}/keep"])


@pytest.mark.parametrize("name", ["empty", "one", "u65_nc2", "adversarial", "mixed1500_nc3", "c2000_nc80"])
def test_nms_list_port(golden, name):
    g = golden("nms")
    boxes, iou_thr, obj_thr, fmt = gi.nms_boxes(name)
    kept = opp.nms_list(boxes.tolist(), iou_thr, obj_thr, fmt)
    want = boxes[g[f"{name}/keep"]]
    assert len(kept) == len(want)
    if len(kept):
        np.testing.assert_array_equal(np.asarray(kept, F32), want)


# --------------------------------------------------------------------------- train step
def test_train_step_losses_and_grads(golden):
    """The oracle's network under autograd + a restated YOLO loss reproduces the reference's
    loss parts and sampled gradients (G7)."""
    from oracle import loss as oloss
    g = golden("train_step")
    c = gi.TRAIN_CASE
    sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = dict(sd)
    full.update(params)
    x = onet.synth_input(c["xseed"], c["batch"], c["size"])
    tg = [torch.from_numpy(t) for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
    grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
    sa = torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)
    stats = {}
    preds = onet.forward(full, x, c["nc"], "leaky_relu", training=True, new_stats=stats)
    parts = torch.stack([torch.stack(oloss.yolo_loss(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
    np.testing.assert_allclose(parts.detach().numpy(), g["leaky/loss_parts"], rtol=2e-4, atol=1e-5)
    parts.sum().backward()
    for k in ("layers.0.conv.weight", "layers.29.pred_block.1.conv.bias", "layers.0.batch_norm.weight"):
        got = params[k].grad
        want = g[f"leaky/grad/{k}"]
        got = got.reshape(-1)[::gi.TRAIN_GRAD_STRIDE].numpy() if got.numel() > 4096 else got.numpy()
        scale = max(1e-6, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-3 * scale)
    np.testing.assert_allclose(stats["layers.0.batch_norm.running_mean"].numpy(), g["leaky/rm0"], atol=1e-6)


@pytest.mark.parametrize("tag,act", [("leaky", "leaky_relu"), ("mish", "mish")])
def test_train_trajectory_three_steps(golden, tag, act):
    """Three iterations of the reference's loop body (train.py:41-82) on the oracle: momentum buffers, weight decay, the
    per-batch LinearLR warm-up, running-statistic accumulation and num_batches_tracked, against the imported reference's
    trajectory (tests/golden/train_traj.npz)."""
    from oracle import loss as oloss
    g = golden("train_traj")
    c = gi.TRAIN_CASE
    sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = dict(sd)
    full.update(params)
    x = onet.synth_input(c["xseed"], c["batch"], c["size"])
    tg = [torch.from_numpy(t) for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
    grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
    sa = torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)
    order = [k for k, _ in onet.state_dict_spec(3, c["nc"]) if k in params]     # = model.parameters() order
    opt = torch.optim.SGD([params[k] for k in order], **gi.TRAJ_OPT)
    sched = torch.optim.lr_scheduler.LinearLR(opt, **gi.TRAJ_SCHED)
    nbt = 0
    for step in range(gi.TRAJ_STEPS):
        opt.zero_grad()
        stats = {}
        preds = onet.forward(full, x, c["nc"], act, training=True, new_stats=stats)
        parts = torch.stack([torch.stack(oloss.yolo_loss(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
        np.testing.assert_allclose(parts.detach().numpy(), g[f"{tag}/loss_parts"][step], rtol=5e-4, atol=1e-5)
        parts.sum().backward()
        assert abs(opt.param_groups[0]["lr"] - g[f"{tag}/lrs"][step]) < 1e-12
        opt.step()
        sched.step()
        full.update(stats)                                 # running statistics carry over to the next iteration
        nbt += 1
    norms = np.array([float(params[k].grad.double().norm()) for k in order])
    np.testing.assert_allclose(norms, g[f"{tag}/gradnorm_step3"], rtol=2e-3, atol=1e-6 * float(norms.max()))
    np.testing.assert_allclose(full["layers.0.batch_norm.running_mean"].numpy(), g[f"{tag}/rm0"], atol=1e-6)
    np.testing.assert_allclose(full["layers.0.batch_norm.running_var"].numpy(), g[f"{tag}/rv0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(full["layers.28.batch_norm.running_var"].numpy(), g[f"{tag}/rv_last"], rtol=1e-3, atol=1e-6)
    assert nbt == int(g[f"{tag}/nbt0"])
    for k in gi.TRAJ_WEIGHT_KEYS:
        np.testing.assert_allclose(params[k].detach().reshape(-1)[::7].numpy(), g[f"{tag}/w/{k}"], rtol=0, atol=2e-6)
    pn = np.array([float(params[k].detach().double().norm()) for k in order])
    np.testing.assert_allclose(pn, g[f"{tag}/param_norms"], rtol=1e-5)
    mn = np.array([float(opt.state[params[k]]["momentum_buffer"].double().norm()) for k in order])
    np.testing.assert_allclose(mn, g[f"{tag}/momentum_norms"], rtol=2e-3, atol=1e-6 * float(mn.max()))


def test_net_forward_in_channels_1(golden):
    """in_channels = 1 (model.py:151): the oracle against the reference built with a one-channel input."""
    g = golden("net_in1")
    c = gi.NET_IN1
    sd = onet.synth_state_dict(c["wseed"], c["in_channels"], c["nc"], gain=gi.NET_GAIN)
    x = onet.synth_input(c["xseed"], c["batch"], c["size"], c["in_channels"])
    with torch.no_grad():
        preds = onet.forward(sd, x, c["nc"], c["act"])
    for i, p in enumerate(preds):
        np.testing.assert_allclose(p.contiguous().numpy(), g[f"p{i}"], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ target builder (dataset.py:119-161)
@pytest.mark.parametrize("case", list(gi.TARGET_CASES))
def test_targets_oracle_equals_reference_getitem(golden, case):
    """oracle/targets.py against the tensors the reference's own YOLODataset.__getitem__ produced for the same
    seeded box lists (empty image, crowded cells, boxes on exact cell corners, duplicates)."""
    from oracle import targets as otg
    g = golden("targets")
    c = gi.TARGET_CASES[case]
    for b, boxes in enumerate(gi.target_boxes(case)):
        got = otg.build_targets_image(boxes, c["anchors"], c["size"])
        for s_i in range(3):
            want = g[f"{case}/img{b}/scale{s_i}"]
            assert got[s_i].shape == want.shape
            np.testing.assert_array_equal(got[s_i].numpy(), want)


# ------------------------------------------------------------------ mAP (utils.py:193-274)
@pytest.mark.parametrize("case", list(gi.MAP_CASES))
def test_map_oracle_vs_reference(golden, case):
    from oracle import metrics as om
    g = golden("kat")
    pb, tb = gi.map_boxes(case)
    nc = gi.MAP_CASES[case]["nc"]
    assert abs(float(om.calc_map(pb, tb, 0.5, "center", nc)) - float(g[f"map_{case}"])) <= 1e-6
    assert abs(float(om.calc_map(pb, tb, 0.75, "center", nc)) - float(g[f"map_{case}_iou75"])) <= 1e-6


def test_letterbox_restatement_is_sane():
    """oracle/preprocess.py (parity with cv2 unpinned): geometry rules and the fixed-point bilinear within one grey level
    of a float bilinear resize; identity when the image already has the target size."""
    from oracle import preprocess as opre
    rng = np.random.Generator(np.random.PCG64(3))
    img = rng.integers(0, 256, (375, 500, 3), dtype=np.uint8)
    out, (h, w, nh, nw, top, left) = opre.letterbox(img, 416)
    assert (nh, nw, top, left) == (312, 416, 52, 0) and out.shape == (3, 416, 416)
    assert float(out[:, :top].max()) == 0.0 and float(out[:, top + nh:].max()) == 0.0
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    ref = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(opre.resize_linear_u8(img, nh, nw).astype(np.float32) - ref).max() <= 1.0
    sq = rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)
    o2, _ = opre.letterbox(sq, 96)
    np.testing.assert_array_equal(o2, sq.transpose(2, 0, 1).astype(np.float32) * np.float32(1 / 255))
