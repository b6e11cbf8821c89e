"""Generate ``tests/golden/*.npz`` by IMPORTING THE REFERENCE (build container only).

Run from the repo root:  ``python tests/gen_golden.py``

Recipe: ``albumentations`` / ``cv2`` are imported at module top by the
reference's ``config.py`` / ``utils.py`` only for data augmentation; they are absent here,
so ``MagicMock`` stand-ins go into ``sys.modules`` before ``/root/reference/code`` is put on
``sys.path``.  Nothing from the reference is copied: the fixtures hold OUTPUT numbers only;
inputs are regenerated from seeds by ``tests/golden_inputs.py``.
"""
from __future__ import annotations

import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _m in ("albumentations", "albumentations.pytorch", "cv2"):
    sys.modules[_m] = MagicMock()
sys.path.insert(0, "/root/reference/code")

import model as ref_model      # noqa: E402  (the reference)
import utils as ref_utils      # noqa: E402
import loss as ref_loss        # noqa: E402

from oracle import net as onet              # noqa: E402
from tests import golden_inputs as gi       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


def sums(t: torch.Tensor):
    d = t.double()
    return np.array([float(d.sum()), float(d.abs().sum())])


def ref_net(nc, act, sd):
    m = ref_model.YOLOv3(num_classes=nc, activation=act)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


# ------------------------------------------------------------------------ G3 whole network
def gen_net():
    out = {}
    for name, c in gi.NET_CASES.items():
        sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
        m = ref_net(c["nc"], c["act"], sd).eval()
        x = onet.synth_input(c["xseed"], c["batch"], c["size"])
        taps = {}
        hooks = []
        mods = dict(m.named_modules())
        for k in gi.TAP_KEYS:
            hooks.append(mods[k].register_forward_hook(lambda _m, _i, o, k=k: taps.__setitem__(k, o.detach())))
        with torch.no_grad():
            preds = m(x)
        for h in hooks:
            h.remove()
        for i, p in enumerate(preds):
            p = p.contiguous()
            if c["full"]:
                out[f"{name}/p{i}"] = p.numpy()
            else:
                out[f"{name}/p{i}_sample"] = p.reshape(-1)[::gi.SAMPLE_STRIDE].numpy().copy()
            out[f"{name}/p{i}_sums"] = sums(p)
            print(name, i, tuple(p.shape), float(p.abs().mean()), float(p.abs().max()))
        for k, t in taps.items():
            out[f"{name}/tap/{k}"] = t.contiguous().reshape(-1)[::gi.TAP_STRIDE].numpy().copy()
            out[f"{name}/tapsums/{k}"] = sums(t)
    np.savez_compressed(os.path.join(OUT, "net_fwd.npz"), **out)


# ------------------------------------------------------------------- G1/G2 block goldens
def gen_blocks():
    out = {}
    for i, (cin, cout, k, s, bn, h) in enumerate(gi.BLOCK_CONFIGS):
        p = gi.block_params(i, cin, cout, k, bn)
        x = torch.from_numpy(gi.block_input(i, cin, h))
        for act in (("leaky_relu", "mish") if bn else ("leaky_relu",)):
            blk = ref_model.CNNBlock(cin, cout, batch_norm_act=bn, activation=act, kernel_size=k,
                                     stride=s, padding=1 if k == 3 else 0)
            blk.conv.weight.data.copy_(torch.from_numpy(p["w"]))
            if bn:
                blk.batch_norm.weight.data.copy_(torch.from_numpy(p["gamma"]))
                blk.batch_norm.bias.data.copy_(torch.from_numpy(p["beta"]))
                blk.batch_norm.running_mean.data.copy_(torch.from_numpy(p["mean"]))
                blk.batch_norm.running_var.data.copy_(torch.from_numpy(p["var"]))
            else:
                blk.conv.bias.data.copy_(torch.from_numpy(p["bias"]))
            blk.eval()
            with torch.no_grad():
                y = blk(x)
            tag = f"cfg{i}/{act}"
            out[f"{tag}/eval"] = y.reshape(-1)[::gi.BLOCK_STRIDE].numpy().copy()
            out[f"{tag}/eval_sums"] = sums(y)
            if bn:
                blk.train()
                xg = x.clone().requires_grad_(True)
                y = blk(xg)
                # a fixed upstream gradient pins backward (dgrad / wgrad / BN backward) too
                gy = torch.from_numpy(np.random.Generator(np.random.PCG64(3000 + i)).standard_normal(
                    tuple(y.shape), dtype=np.float32))
                y.backward(gy)
                out[f"{tag}/train"] = y.detach().reshape(-1)[::gi.BLOCK_STRIDE].numpy().copy()
                out[f"{tag}/train_sums"] = sums(y.detach())
                out[f"{tag}/new_mean"] = blk.batch_norm.running_mean.numpy().copy()
                out[f"{tag}/new_var"] = blk.batch_norm.running_var.numpy().copy()
                out[f"{tag}/dx"] = xg.grad.reshape(-1)[::gi.BLOCK_STRIDE].numpy().copy()
                out[f"{tag}/dx_sums"] = sums(xg.grad)
                out[f"{tag}/dw"] = blk.conv.weight.grad.reshape(-1)[::gi.BLOCK_DW_STRIDE].numpy().copy()
                out[f"{tag}/dw_sums"] = sums(blk.conv.weight.grad)
                out[f"{tag}/dgamma"] = blk.batch_norm.weight.grad.numpy().copy()
                out[f"{tag}/dbeta"] = blk.batch_norm.bias.grad.numpy().copy()
        print("block", i, (cin, cout, k, s, bn, h))
    # G2: residual stage and head block
    rb = ref_model.ResidualBlock(64, num_blocks=2).eval()
    sp = ref_model.ScalePredictionBlock(128, num_classes=2).eval()
    g = torch.Generator().manual_seed(77)
    for mod in (rb, sp):
        for prm in mod.parameters():
            prm.data.copy_(torch.randn(prm.shape, generator=g) * (0.08 if prm.dim() > 1 else 0.3) + (1.0 if prm.dim() == 1 else 0))
        for nm, buf in mod.named_buffers():
            if nm.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif nm.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
    xr = torch.randn(2, 64, 12, 12, generator=g)
    xs = torch.randn(2, 128, 6, 6, generator=g)
    with torch.no_grad():
        out["res64x2/x"] = xr.numpy()
        out["res64x2/y"] = rb(xr).numpy()
        out["head128/x"] = xs.numpy()
        out["head128/y"] = sp(xs).contiguous().numpy()
    for nm, t in list(rb.state_dict().items()):
        out["res64x2/sd/" + nm] = t.numpy()
    for nm, t in list(sp.state_dict().items()):
        out["head128/sd/" + nm] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)


# ------------------------------------------------------------------------ G4 loader map
def gen_loader():
    out = {}
    sd = onet.synth_state_dict(21, 3, 80, gain=1.0)
    stream = onet.darknet_stream(sd, 3, 80)
    assert stream.size == 62001757, stream.size
    with tempfile.TemporaryDirectory() as td:
        for fname in ("yolov3.weights", "darknet53.conv.74"):
            path = os.path.join(td, fname)
            with open(path, "wb") as f:
                np.array([0, 2, 0, 32013312, 0], np.int32).tofile(f)
                stream.tofile(f)
            m = ref_model.YOLOv3(num_classes=80, weights_path=path)
            before = {k: v.clone() for k, v in m.state_dict().items()}
            m.load_weights()
            after = m.state_dict()
            keys, offs, cnts, loaded = [], [], [], []
            off = 0
            # walk in Darknet order (bn: beta,gamma,mean,var then W; head: bias then W)
            for cv in onet.conv_list(3, 80):
                p = cv["prefix"]
                names = ([p + ".batch_norm." + n for n in ("bias", "weight", "running_mean", "running_var")]
                         if cv["bn"] else [p + ".conv.bias"]) + [p + ".conv.weight"]
                for nm in names:
                    t = after[nm]
                    n = t.numel()
                    is_loaded = bool(torch.equal(t.reshape(-1), torch.from_numpy(stream[off:off + n])))
                    changed = not torch.equal(t, before[nm])
                    assert is_loaded == changed or n == 0, (nm, is_loaded, changed)
                    keys.append(nm); offs.append(off); cnts.append(n); loaded.append(is_loaded)
                    off += n
            assert off == stream.size
            tag = "full" if fname == "yolov3.weights" else "conv74"
            out[f"{tag}/keys"] = np.array(keys)
            out[f"{tag}/offsets"] = np.array(offs, np.int64)
            out[f"{tag}/counts"] = np.array(cnts, np.int64)
            out[f"{tag}/loaded"] = np.array(loaded, bool)
            print(fname, "loaded tensors:", int(np.sum(loaded)), "of", len(loaded),
                  "last loaded:", [k for k, l in zip(keys, loaded) if l][-1])
    out["state_dict_keys"] = np.array(list(ref_model.YOLOv3(num_classes=80).state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "loader.npz"), **out)


# --------------------------------------------------------------------------- G5 decode
def gen_decode():
    out = {}
    for name in gi.DECODE_CASES:
        pred, anchors = gi.decode_input(name)
        p = torch.from_numpy(pred.copy())
        boxes = ref_utils.cells_to_boxes(p, torch.from_numpy(anchors), gi.DECODE_CASES[name]["g"], is_pred=True)
        out[f"{name}/boxes"] = np.asarray(boxes, np.float32)
        out[f"{name}/mutated"] = p.numpy()          # the in-place side effect (utils.py:106-110)
        # fp16 / bf16 inputs (autocast outputs): values differ, order of ops the same
        for dt, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
            ph = torch.from_numpy(pred.copy()).to(dt)
            bh = ref_utils.cells_to_boxes(ph, torch.from_numpy(anchors), gi.DECODE_CASES[name]["g"], is_pred=True)
            out[f"{name}/boxes_{tag}"] = np.asarray(bh, np.float32)
    # targets path (is_pred=False)
    t = torch.from_numpy(gi.synth_targets(2, 96, 2, gi.TRAIN_CASE["anchors"], 55)[2])
    tb = ref_utils.cells_to_boxes(t, torch.zeros(3, 2), 12, is_pred=False)
    out["targets_g12/boxes"] = np.asarray(tb, np.float32)
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **out)


# ------------------------------------------------------------------------------ G6 NMS
def rows_to_indices(kept_rows, boxes):
    """Map the reference's kept BOXES back to indices of the input list (first unused
    identical row, scanning in input order — identical rows are interchangeable)."""
    kept = np.asarray(kept_rows, np.float32).reshape(-1, 6)
    used = np.zeros(len(boxes), bool)
    idx = []
    view = boxes.view(np.uint32)
    for r in kept.view(np.uint32):
        cand = np.nonzero((view == r).all(1) & ~used)[0]
        assert cand.size, "kept box not found in the input"
        idx.append(int(cand[0])); used[cand[0]] = True
    return np.asarray(idx, np.int64)


def gen_nms():
    import time
    out = {}
    for name in gi.NMS_CASES:
        boxes, iou_thr, obj_thr, fmt = gi.nms_boxes(name)
        t = time.time()
        kept = ref_utils.non_max_suppression(boxes.tolist(), iou_thr, obj_thr, fmt)
        dt = time.time() - t
        idx = rows_to_indices(kept, boxes)
        out[f"{name}/keep"] = idx
        print("nms", name, "n=", len(boxes), "kept=", len(idx), f"{dt:.2f}s")
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **out)


# ----------------------------------------------------------------------- G7 train step
def gen_train():
    c = gi.TRAIN_CASE
    out = {}
    for tag, act in (("leaky", "leaky_relu"), ("mish", "mish")):
        sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
        m = ref_net(c["nc"], act, sd).train()
        x = onet.synth_input(c["xseed"], c["batch"], c["size"])
        tg = [torch.from_numpy(t) for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
        grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
        sa = torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)
        lf = ref_loss.YOLOLoss()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
        opt.zero_grad()
        preds = m(x)
        out[f"{tag}/pred_sums"] = np.stack([sums(p.detach()) for p in preds])
        parts = []
        for i in range(3):
            parts.append(torch.stack(lf(preds[i], tg[i].clone(), sa[i])))
        parts = torch.stack(parts)                      # (3 scales, 4 parts)
        total = parts.sum()
        total.backward()
        out[f"{tag}/loss_parts"] = parts.detach().numpy()
        out[f"{tag}/loss"] = np.array(float(total))
        g = {k: p.grad for k, p in m.named_parameters()}
        for k in ("layers.0.conv.weight", "layers.10.layers.3.1.conv.weight", "layers.15.pred_block.1.conv.bias",
                  "layers.22.pred_block.1.conv.bias", "layers.29.pred_block.1.conv.bias",
                  "layers.0.batch_norm.weight", "layers.0.batch_norm.bias", "layers.6.layers.7.1.batch_norm.weight",
                  "layers.18.conv.weight", "layers.29.pred_block.1.conv.weight"):
            out[f"{tag}/grad/{k}"] = g[k].reshape(-1)[::gi.TRAIN_GRAD_STRIDE].numpy().copy() if g[k].numel() > 4096 else g[k].numpy().copy()
            out[f"{tag}/gradsums/{k}"] = sums(g[k])
        out[f"{tag}/gradnorm_all"] = np.array([float(p.grad.double().norm()) for p in m.parameters()])
        out[f"{tag}/rm0"] = m.state_dict()["layers.0.batch_norm.running_mean"].numpy().copy()
        out[f"{tag}/rv0"] = m.state_dict()["layers.0.batch_norm.running_var"].numpy().copy()
        opt.step()
        out[f"{tag}/w0_after_sgd"] = m.state_dict()["layers.0.conv.weight"].numpy().copy()
        print("train", tag, parts.detach().numpy().round(4).tolist(), float(total))
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)


TRAIN_GRAD_KEYS = ("layers.0.conv.weight", "layers.10.layers.3.1.conv.weight", "layers.15.pred_block.1.conv.bias",
                   "layers.22.pred_block.1.conv.bias", "layers.29.pred_block.1.conv.bias",
                   "layers.0.batch_norm.weight", "layers.0.batch_norm.bias", "layers.6.layers.7.1.batch_norm.weight",
                   "layers.18.conv.weight", "layers.29.pred_block.1.conv.weight")


def gen_train_fp64():
    """The SAME fine-tune step as gen_train, run by the imported reference in float64 (model.double(), double inputs):
    the yardstick for the LeakyReLU gradients. LeakyReLU's derivative jumps at 0, so two fp32 implementations disagree on
    single entries by much more than rounding; what can be asked of an fp32 implementation is that it is about as close to
    the fp64 result as the reference's own fp32 run is (tests: |ours - fp64| <= 3 |reference fp32 - fp64| + 1e-4 max|g|)."""
    c = gi.TRAIN_CASE
    out = {}
    for tag, act in (("leaky", "leaky_relu"), ("mish", "mish")):
        sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
        m = ref_net(c["nc"], act, sd).double().train()
        x = onet.synth_input(c["xseed"], c["batch"], c["size"]).double()
        tg = [torch.from_numpy(t).double() for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
        grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
        sa = (torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)).double()
        lf = ref_loss.YOLOLoss()
        preds = m(x)
        parts = torch.stack([torch.stack(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
        parts.sum().backward()
        out[f"{tag}/loss_parts"] = parts.detach().numpy()
        g = {k: p.grad for k, p in m.named_parameters()}
        for k in TRAIN_GRAD_KEYS:
            out[f"{tag}/grad/{k}"] = g[k].reshape(-1)[::gi.TRAIN_GRAD_STRIDE].numpy().copy() if g[k].numel() > 4096 else g[k].numpy().copy()
        out[f"{tag}/gradnorm_all"] = np.array([float(p.grad.norm()) for p in m.parameters()])
        print("train fp64", tag, parts.detach().numpy().round(6).tolist())
    np.savez_compressed(os.path.join(OUT, "train_step_fp64.npz"), **out)


# -------------------------------------------------------------------------- G8 the KATs
def gen_kat():
    out = {}
    b = torch.tensor([0.5, 0.5, 0.25, 0.25])
    out["iou_self"] = ref_utils.calc_iou(b, b).numpy()            # utils_test.py:16-20 (0.99998…, not 1.0)
    pb = [[0, 0.5, 0.5, 0.25, 0.25, 0.9, 0], [0, 0.5, 0.5, 0.1, 0.1, 0.6, 0]]
    out["map_identical"] = np.array(float(ref_utils.calc_mAP(pb, [r[:] for r in pb])))   # utils_test.py:22-32
    z = torch.zeros((5, 3, 3, 3, 8))
    out["c2b_zeros"] = np.asarray(ref_utils.cells_to_boxes(z, torch.tensor([[0.28, 0.22], [0.38, 0.48], [0.9, 0.78]]), 3),
                                  np.float32)                     # utils_test.py:34-40
    rng = np.random.Generator(np.random.PCG64(99))
    a = rng.random((200, 4), dtype=np.float32)
    c = rng.random((200, 4), dtype=np.float32)
    out["iou_center"] = ref_utils.calc_iou(torch.from_numpy(a), torch.from_numpy(c), "center").numpy()
    out["iou_corners"] = ref_utils.calc_iou(torch.from_numpy(a), torch.from_numpy(c), "corners").numpy()
    out["iou_aligned"] = ref_utils.iou_aligned(torch.from_numpy(a[:, 2:]), torch.from_numpy(c[:, 2:])).numpy()
    for name, c in gi.MAP_CASES.items():                           # calc_mAP of the reference itself on seeded box lists
        pb, tb = gi.map_boxes(name)
        out[f"map_{name}"] = np.array(float(ref_utils.calc_mAP([r[:] for r in pb], [r[:] for r in tb], 0.5, "center", c["nc"])))
        out[f"map_{name}_iou75"] = np.array(float(ref_utils.calc_mAP([r[:] for r in pb], [r[:] for r in tb], 0.75, "center", c["nc"])))
    # check_model_accuracy (utils.py:334-381) of the reference on a stub model that replays seeded predictions
    import config as ref_config
    ref_config.DEVICE = "cpu"
    batches = gi.accuracy_batches()

    class Stub:
        def __init__(self):
            self.k = -1

        def eval(self):
            pass

        def train(self):
            pass

        def __call__(self, x):
            self.k += 1
            return [torch.from_numpy(p.copy()) for p in batches[self.k][2]]
    loader = [(torch.from_numpy(x), [torch.from_numpy(t.copy()) for t in tg]) for x, tg, _ in batches]
    acc = ref_utils.check_model_accuracy(Stub(), loader, gi.ACC_CASE["thr"])
    out["accuracy"] = np.array([float(a) for a in acc], np.float32)
    # get_eval_boxes (utils.py:276-332) of the reference on a stub model replaying seeded predictions
    ebatches = gi.eval_batches()

    class EStub(Stub):
        def __call__(self, x):
            self.k += 1
            return [torch.from_numpy(p.copy()) for p in ebatches[self.k][2]]
    eloader = [(torch.from_numpy(x), [torch.from_numpy(t.copy()) for t in tg]) for x, tg, _ in ebatches]
    ec = gi.EVAL_CASE
    pb, tb = ref_utils.get_eval_boxes(eloader, EStub(), ec["iou_thr"], ec["anchors"], ec["obj_thr"], "center", "cpu")
    out["eval_pred_boxes"] = np.asarray(pb, np.float64).reshape(-1, 7)
    out["eval_true_boxes"] = np.asarray(tb, np.float64).reshape(-1, 7)
    out["eval_map"] = np.array(float(ref_utils.calc_mAP(pb, tb, 0.5, "center", ec["nc"])))
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **out)


def gen_targets():
    """Targets from the reference's OWN YOLODataset.__getitem__ (dataset.py:119-161): the dataset object is built on
    a one-line csv + an empty label file in a temp dir, and only its I/O hooks (image / label loading, augmentation)
    are replaced so that the seeded box list reaches the assignment loop unchanged."""
    import tempfile
    import dataset as ref_dataset
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "split.csv"), "w") as f:
            f.write("img.jpg,label.txt\nimg2.jpg,missing.txt\n")
        open(os.path.join(tmp, "label.txt"), "w").close()
        for name, c in gi.TARGET_CASES.items():
            S = c["size"]
            ds = ref_dataset.YOLODataset(os.path.join(tmp, "split.csv"), tmp, tmp, [list(map(list, a)) for a in c["anchors"]], batch_size=1,
                                         image_size=S, grid_sizes=[S // 32, S // 16, S // 8], num_classes=c["nc"])
            ds.load_image = lambda idx: np.zeros((S, S, 3), np.uint8)
            for b, boxes in enumerate(gi.target_boxes(name)):
                ds.load_boxes = lambda label_path, idx, _b=boxes: [list(r) for r in _b]
                ds.apply_augmentations = lambda img, bx, idx: (img, bx)
                _, t = ds[0]
                for s_i, tt in enumerate(t):
                    out[f"{name}/img{b}/scale{s_i}"] = tt.numpy()
    np.savez_compressed(os.path.join(OUT, "targets.npz"), **out)


# ------------------------------------------------------------- checkpoint format
def gen_checkpoint():
    import yolo_for_turbines_amd as yt
    out = {}
    m, opt = gi.checkpoint_setup(ref_model.YOLOv3)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ref.pth.tar")
        ref_utils.save_checkpoint(m, opt, filename=path)
        ck = torch.load(path)
        out["top_keys"] = np.array(list(ck.keys()))
        out["param_names"] = np.array([k for k, _ in m.named_parameters()])
        out["state_dict_keys"] = np.array(list(ck["state_dict"].keys()))
        out["state_dict_sums"] = np.array([float(v.double().sum()) for v in ck["state_dict"].values()])
        o = ck["optimizer"]
        out["opt_top_keys"] = np.array(list(o.keys()))
        out["opt_state_ids"] = np.array(list(o["state"].keys()), np.int64)
        out["opt_state_entry_keys"] = np.array(sorted(o["state"][0].keys()))
        out["momentum_sums"] = np.array([float(o["state"][i]["momentum_buffer"].double().sum()) for i in o["state"]])
        g = o["param_groups"][0]
        out["group_keys"] = np.array(sorted(g.keys()))
        out["group_params"] = np.array(g["params"], np.int64)
        out["group_scalars"] = np.array([g["lr"], g["momentum"], g["dampening"], g["weight_decay"]], np.float64)
        # cross-load, here where both exist: the reference's file into this package's model + optimizer ...
        m2 = yt.YOLOv3(num_classes=2)
        opt2 = torch.optim.SGD(m2.parameters(), lr=0.5, momentum=0.9, weight_decay=5e-4)
        yt.load_checkpoint(m2, opt2, lr=0.125, filename=path)
        for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert k == k2 and torch.equal(a, b), k
        for i, (pa, pb) in enumerate(zip(m.parameters(), m2.parameters())):
            assert torch.equal(opt.state[pa]["momentum_buffer"], opt2.state[pb]["momentum_buffer"]), i
        assert all(gr["lr"] == 0.125 for gr in opt2.param_groups)
        # ... and this package's file into the reference's (its loader prefixes config.MODEL_FOLDER)
        mine = os.path.join(td, "mine.pth.tar")
        yt.save_checkpoint(m2, opt2, filename=mine)
        m3 = ref_model.YOLOv3(num_classes=2)
        opt3 = torch.optim.SGD(m3.parameters(), lr=0.5, momentum=0.9, weight_decay=5e-4)
        import config as ref_config
        ref_config.MODEL_FOLDER = td
        ref_utils.load_checkpoint(m3, opt3, lr=0.25, filename="mine.pth.tar")
        for (k, a), (k3, c) in zip(m.state_dict().items(), m3.state_dict().items()):
            assert k == k3 and torch.equal(a, c), k
        for pa, pc in zip(m.parameters(), m3.parameters()):
            assert torch.equal(opt.state[pa]["momentum_buffer"], opt3.state[pc]["momentum_buffer"])
        out["cross_load_ok"] = np.array([1, 1], np.int64)       # [reference file -> this package, this package's file -> reference]
    np.savez_compressed(os.path.join(OUT, "checkpoint.npz"), **out)


# ------------------------------------------------- multi-step trajectory (train.py:41-82, several iterations)
TRAJ_STEPS, TRAJ_OPT, TRAJ_SCHED, TRAJ_WEIGHT_KEYS = gi.TRAJ_STEPS, gi.TRAJ_OPT, gi.TRAJ_SCHED, gi.TRAJ_WEIGHT_KEYS


def gen_train_traj():
    """THREE consecutive iterations of the reference's loop body (train.py:41-82: zero_grad, forward, three per-scale losses,
    backward, optimizer.step, scheduler.step) on one seeded batch, run by the imported reference in fp32: steps 2+ exercise
    what a single step cannot - momentum buffers, running-statistic accumulation, num_batches_tracked, the LR schedule and
    re-packing after optimizer.step()."""
    c = gi.TRAIN_CASE
    out = {}
    for tag, act in (("leaky", "leaky_relu"), ("mish", "mish")):
        sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
        m = ref_net(c["nc"], act, sd).train()
        x = onet.synth_input(c["xseed"], c["batch"], c["size"])
        tg = [torch.from_numpy(t) for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
        grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
        sa = torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)
        lf = ref_loss.YOLOLoss()
        opt = torch.optim.SGD(m.parameters(), **TRAJ_OPT)
        sched = torch.optim.lr_scheduler.LinearLR(opt, **TRAJ_SCHED)
        parts_all, lrs = [], []
        for step in range(TRAJ_STEPS):
            opt.zero_grad()
            preds = m(x)
            parts = torch.stack([torch.stack(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
            parts.sum().backward()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
            parts_all.append(parts.detach().numpy())
        out[f"{tag}/loss_parts"] = np.stack(parts_all)                       # (steps, 3 scales, 4 parts)
        out[f"{tag}/lrs"] = np.array(lrs, np.float64)
        out[f"{tag}/gradnorm_step3"] = np.array([float(p.grad.double().norm()) for p in m.parameters()])
        st = m.state_dict()
        out[f"{tag}/rm0"] = st["layers.0.batch_norm.running_mean"].numpy().copy()
        out[f"{tag}/rv0"] = st["layers.0.batch_norm.running_var"].numpy().copy()
        out[f"{tag}/nbt0"] = np.array(int(st["layers.0.batch_norm.num_batches_tracked"]))
        out[f"{tag}/rv_last"] = st["layers.28.batch_norm.running_var"].numpy().copy()
        for k in TRAJ_WEIGHT_KEYS:
            out[f"{tag}/w/{k}"] = st[k].reshape(-1)[::7].numpy().copy()
            out[f"{tag}/wsums/{k}"] = sums(st[k])
        out[f"{tag}/param_norms"] = np.array([float(p.detach().double().norm()) for p in m.parameters()])
        out[f"{tag}/momentum_norms"] = np.array([float(opt.state[p]["momentum_buffer"].double().norm()) for p in m.parameters()])
        print("traj", tag, [float(p.sum()) for p in parts_all], lrs)
        # How well-conditioned is this trajectory? The SAME reference code on inputs perturbed by 1e-6 (relative, three seeds):
        # the spread of the summed loss per step is what any other fp32 implementation (different accumulation order in 75
        # convolutions) must be allowed. With Mish it stays ~1e-5; with LeakyReLU the deepest BatchNorm layers see 36 values
        # per channel at this size and a handful of flipped branches moves step 2 by ~0.5 % and step 3 by more.
        pert = []
        for seed in range(3):
            m2 = ref_net(c["nc"], act, onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)).train()
            gen = torch.Generator().manual_seed(100 + seed)
            x2 = x * (1 + gi.TRAJ_PERTURB * torch.randn(x.shape, generator=gen))
            opt2 = torch.optim.SGD(m2.parameters(), **TRAJ_OPT)
            sched2 = torch.optim.lr_scheduler.LinearLR(opt2, **TRAJ_SCHED)
            tot = []
            for step in range(TRAJ_STEPS):
                opt2.zero_grad()
                pr = m2(x2)
                l = sum(sum(lf(pr[i], tg[i].clone(), sa[i])) for i in range(3))
                l.backward()
                opt2.step()
                sched2.step()
                tot.append(float(l))
            pert.append(tot)
        out[f"{tag}/perturbed_totals"] = np.array(pert)
        print("   perturbed totals", np.array(pert).round(4).tolist())
    np.savez_compressed(os.path.join(OUT, "train_traj.npz"), **out)


# ---------------------------------- the reference's UNCHANGED loader on this package's module
def gen_loader_bind():
    """SURVEY §8a row 7 asks that the reference's loader (`model.py:227-337`) drops in unchanged. Here its four functions -
    the very code objects of the imported reference, with only the three class names its isinstance tests look up
    (CNNBlock / ResidualBlock / ScalePredictionBlock) resolving to this package's classes, i.e. what `from model import`
    becoming `from yolo_for_turbines_amd import` does - are run on a CPU-constructed `yolo_for_turbines_amd.YOLOv3`, and
    every parameter and buffer is compared with (a) the reference model filled by its own loader and (b) this package's
    restated loader. Full file, the `.conv.74` cutoff, and freeze=True."""
    import types
    import yolo_for_turbines_amd as yt
    out = {}
    names = ("load_weights", "load_CNNBlock", "load_block_weights", "load_layer_weights")
    glb = dict(ref_model.__dict__)
    glb.update(CNNBlock=yt.CNNBlock, ResidualBlock=yt.ResidualBlock, ScalePredictionBlock=yt.ScalePredictionBlock)
    bound = {n: types.FunctionType(getattr(ref_model.YOLOv3, n).__code__, glb, n) for n in names}
    sd = onet.synth_state_dict(21, 3, 80, gain=1.0)
    stream = onet.darknet_stream(sd, 3, 80)
    with tempfile.TemporaryDirectory() as td:
        for fname, freeze in (("yolov3.weights", False), ("darknet53.conv.74", True)):
            path = os.path.join(td, fname)
            with open(path, "wb") as f:
                np.array([0, 2, 0, 32013312, 0], np.int32).tofile(f)
                stream.tofile(f)
            torch.manual_seed(5)
            ref = ref_model.YOLOv3(num_classes=80, weights_path=path, freeze=freeze)
            init = {k: v.clone() for k, v in ref.state_dict().items()}
            ref.load_weights()
            mine_bound = yt.YOLOv3(num_classes=80, weights_path=path, freeze=freeze)
            mine_bound.load_state_dict(init)                    # same starting point for the tensors the cutoff leaves alone
            for n in names:                                      # instance attributes shadow the package's own methods
                setattr(mine_bound, n, types.MethodType(bound[n], mine_bound))
            mine_bound.load_weights()
            mine_own = yt.YOLOv3(num_classes=80, weights_path=path, freeze=freeze)
            mine_own.load_state_dict(init)
            mine_own.load_weights()
            a, b, c = ref.state_dict(), mine_bound.state_dict(), mine_own.state_dict()
            assert list(a) == list(b) == list(c)
            for k in a:
                assert torch.equal(a[k], b[k]), ("bound", fname, k)
                assert torch.equal(a[k], c[k]), ("restated", fname, k)
            ra = {k: p.requires_grad for k, p in ref.named_parameters()}
            rb = {k: p.requires_grad for k, p in mine_bound.named_parameters()}
            rc = {k: p.requires_grad for k, p in mine_own.named_parameters()}
            assert ra == rb == rc, fname
            assert (ref.param_idx, ref.layer_id) == (mine_bound.param_idx, mine_bound.layer_id) == (mine_own.param_idx, mine_own.layer_id)
            tag = "full" if fname == "yolov3.weights" else "conv74_freeze"
            out[f"{tag}/keys"] = np.array(list(a))
            out[f"{tag}/sums"] = np.array([float(v.double().sum()) for v in a.values()])
            out[f"{tag}/abs_sums"] = np.array([float(v.double().abs().sum()) for v in a.values()])
            out[f"{tag}/init_sums"] = np.array([float(v.double().sum()) for v in init.values()])
            out[f"{tag}/requires_grad"] = np.array([ra[k] for k, _ in ref.named_parameters()], bool)
            out[f"{tag}/counters"] = np.array([ref.param_idx, ref.layer_id], np.int64)
            out[f"{tag}/bound_equal"] = np.array(1)
            out[f"{tag}/restated_equal"] = np.array(1)
            print("loader_bind", fname, "frozen:", int(np.sum(~out[f'{tag}/requires_grad'])), "counters", ref.param_idx, ref.layer_id)
    np.savez_compressed(os.path.join(OUT, "loader_bind.npz"), **out)


# ----------------------------------------------------------------------- in_channels != 3 (model.py:151)
NET_IN1 = gi.NET_IN1


def gen_net_in1():
    """Whole-network eval forward of the reference built with in_channels=1 (greyscale input): the first block is then
    not the 3-channel stem the kernels special-case."""
    c = NET_IN1
    sd = onet.synth_state_dict(c["wseed"], c["in_channels"], c["nc"], gain=gi.NET_GAIN)
    m = ref_model.YOLOv3(in_channels=c["in_channels"], num_classes=c["nc"], activation=c["act"])
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = onet.synth_input(c["xseed"], c["batch"], c["size"], c["in_channels"])
    with torch.no_grad():
        preds = m(x)
    out = {f"p{i}": p.contiguous().numpy() for i, p in enumerate(preds)}
    np.savez_compressed(os.path.join(OUT, "net_in1.npz"), **out)
    print("net_in1", [tuple(p.shape) for p in preds])


if __name__ == "__main__":
    which = sys.argv[1:] or ["net", "blocks", "loader", "decode", "nms", "train", "train_fp64", "kat", "targets", "checkpoint", "train_traj", "loader_bind",
                             "net_in1"]
    for w in which:
        print("==", w)
        globals()["gen_" + w]()
