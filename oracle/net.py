"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU restatement of the reference network arithmetic (Darknet-53 + 3-scale head):
`/root/reference/code/model.py:20-45` (topology table), `:80-86` (Conv -> BN -> act),
`:115-121` (residual stage), `:145-148` (head reshape/permute), `:172-193` (walk,
route stack, upsample + concat order, NaN guards).

It is a *functional* program over a plain ``dict`` of tensors that uses the
reference's ``state_dict`` key names, written independently of the reference's
module classes: one flat op list is expanded from a compact stage description,
then interpreted with ``torch.nn.functional`` CPU ops (fp32; the third-party
arithmetic the reference itself dispatches to — SURVEY.md §8c "Third-party
arithmetic").

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference in the
build container (``tests/gen_golden.py``); ``tests/test_oracle_golden.py`` checks
this restatement against them.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default used at model.py:61
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.1    # model.py:64
ANCHORS_PER_SCALE = 3

# (kind, arg) — restates model.py:20-45. "c": (cout, k, stride); "r": n residual units;
# "s": scale head; "u": upsample + route concat.
_STAGES = (
    ("c", (32, 3, 1)), ("c", (64, 3, 2)), ("r", 1), ("c", (128, 3, 2)), ("r", 2),
    ("c", (256, 3, 2)), ("r", 8), ("c", (512, 3, 2)), ("r", 8), ("c", (1024, 3, 2)), ("r", 4),
    ("c", (512, 1, 1)), ("c", (1024, 3, 1)), ("s", None),
    ("c", (256, 1, 1)), ("u", None), ("c", (256, 1, 1)), ("c", (512, 3, 1)), ("s", None),
    ("c", (128, 1, 1)), ("u", None), ("c", (128, 1, 1)), ("c", (256, 3, 1)), ("s", None),
)


def program(in_channels: int = 3, num_classes: int = 80):
    """Flat op list. Each conv op: dict(prefix, cin, cout, k, stride, bn).

    Top-level index bookkeeping follows model.py:195-225: a tuple appends one module,
    "B" appends one ResidualBlock, "S" appends three modules (non-residual block, 1x1
    CNNBlock, ScalePredictionBlock) and halves the channel count, "U" appends
    nn.Upsample and triples it.
    """
    ops = []
    idx = 0
    c = in_channels

    def conv(prefix, cin, cout, k, stride=1, bn=True):
        return dict(op="conv", prefix=prefix, cin=cin, cout=cout, k=k, stride=stride, bn=bn)

    for kind, arg in _STAGES:
        if kind == "c":
            cout, k, s = arg
            ops.append(conv(f"layers.{idx}", c, cout, k, s))
            c = cout
            idx += 1
        elif kind == "r":
            units = []
            for j in range(arg):
                units.append((conv(f"layers.{idx}.layers.{j}.0", c, c // 2, 1),
                              conv(f"layers.{idx}.layers.{j}.1", c // 2, c, 3)))
            ops.append(dict(op="res", units=units, skip=True, route=(arg == 8)))
            idx += 1
        elif kind == "s":
            ops.append(dict(op="res", skip=False, route=False, units=[
                (conv(f"layers.{idx}.layers.0.0", c, c // 2, 1),
                 conv(f"layers.{idx}.layers.0.1", c // 2, c, 3))]))
            ops.append(conv(f"layers.{idx + 1}", c, c // 2, 1))
            h = c // 2
            ops.append(dict(op="head", convs=(
                conv(f"layers.{idx + 2}.pred_block.0", h, 2 * h, 3),
                conv(f"layers.{idx + 2}.pred_block.1", 2 * h, ANCHORS_PER_SCALE * (5 + num_classes), 1,
                     bn=False))))
            c = h
            idx += 3
        elif kind == "u":
            ops.append(dict(op="up"))
            c = c * 3
            idx += 1
    return ops


def conv_list(in_channels: int = 3, num_classes: int = 80):
    """The 75 convolutions in module (= Darknet file) order."""
    out = []
    for op in program(in_channels, num_classes):
        if op["op"] == "conv":
            out.append(op)
        elif op["op"] == "res":
            for a, b in op["units"]:
                out += [a, b]
        elif op["op"] == "head":
            out += list(op["convs"])
    return out


def state_dict_spec(in_channels: int = 3, num_classes: int = 80):
    """Ordered (key, shape) pairs exactly as the reference's ``state_dict()`` lists them."""
    spec = []
    for cv in conv_list(in_channels, num_classes):
        p = cv["prefix"]
        spec.append((p + ".conv.weight", (cv["cout"], cv["cin"], cv["k"], cv["k"])))
        if cv["bn"]:
            for nm in ("weight", "bias", "running_mean", "running_var"):
                spec.append((p + ".batch_norm." + nm, (cv["cout"],)))
            spec.append((p + ".batch_norm.num_batches_tracked", ()))
        else:
            spec.append((p + ".conv.bias", (cv["cout"],)))
    return spec


def activation(x, name):
    if name == "leaky_relu":
        return F.leaky_relu(x, LEAKY_SLOPE)
    if name == "mish":
        return F.mish(x)
    raise ValueError(f"Unsupported activation: {name}")   # model.py:68


def cnn_block(sd, cv, x, act, training=False, new_stats=None, leaky_masks=None):
    """model.py:80-86. ``new_stats`` (dict) receives updated running stats in training.

    ``leaky_masks`` (test aid, LeakyReLU only): {prefix: bool tensor (B,C,H,W)} — the branch (u > 0) to take per element
    instead of this run's own sign test. LeakyReLU's derivative jumps at 0, so two correct implementations that disagree
    on the sign of a |u| ~ 1e-6 element differ by O(|dy|) there; evaluating the float64 oracle on the branches the
    implementation under test took isolates everything ELSE (values and gradients then agree to rounding)."""
    p = cv["prefix"]
    pad = 1 if cv["k"] == 3 else 0                        # model.py:201
    if not cv["bn"]:
        return F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=cv["stride"], padding=pad)
    y = F.conv2d(x, sd[p + ".conv.weight"], None, stride=cv["stride"], padding=pad)
    if training:
        rm = sd[p + ".batch_norm.running_mean"].clone()
        rv = sd[p + ".batch_norm.running_var"].clone()
        y = F.batch_norm(y, rm, rv, sd[p + ".batch_norm.weight"], sd[p + ".batch_norm.bias"],
                         True, BN_MOMENTUM, BN_EPS)
        if new_stats is not None:
            new_stats[p + ".batch_norm.running_mean"] = rm
            new_stats[p + ".batch_norm.running_var"] = rv
    else:
        y = F.batch_norm(y, sd[p + ".batch_norm.running_mean"], sd[p + ".batch_norm.running_var"],
                         sd[p + ".batch_norm.weight"], sd[p + ".batch_norm.bias"], False, BN_MOMENTUM, BN_EPS)
    if leaky_masks is not None and act == "leaky_relu" and p in leaky_masks:
        return torch.where(leaky_masks[p], y, LEAKY_SLOPE * y)
    return activation(y, act)


def forward(sd, x, num_classes=80, act="leaky_relu", training=False, new_stats=None, taps=None, leaky_masks=None):
    """model.py:172-193. Returns [P(S/32), P(S/16), P(S/8)], each (B,3,g,g,5+nc).

    ``taps`` (dict) optionally collects intermediate activations keyed by conv prefix.
    """
    assert torch.sum(torch.isnan(x)) == 0                 # model.py:175
    preds, routes = [], []

    def run(cv, t):
        y = cnn_block(sd, cv, t, act, training, new_stats, leaky_masks)
        if taps is not None:
            taps[cv["prefix"]] = y
        return y

    for op in program(x.shape[1], num_classes):
        if op["op"] == "head":
            c0, c1 = op["convs"]
            y = run(c1, run(c0, x))
            b, _, g, _ = y.shape
            preds.append(y.reshape(b, ANCHORS_PER_SCALE, 5 + num_classes, g, g).permute(0, 1, 3, 4, 2))
            continue
        if op["op"] == "conv":
            x = run(op, x)
        elif op["op"] == "res":
            for a, b in op["units"]:
                y = run(b, run(a, x))
                x = x + y if op["skip"] else y
        elif op["op"] == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        if torch.sum(torch.isnan(x)) > 0:                 # model.py:183-184
            raise ValueError("Nan in layer")
        if op["op"] == "res" and op["route"]:
            routes.append(x)
        elif op["op"] == "up":
            x = torch.cat([x, routes.pop()], dim=1)       # upsampled first (model.py:190)
    return preds


# ----------------------------------------------------------------------------------------
# Seeded synthetic parameters (there is no pretrained file offline: SURVEY.md §8c).
# ----------------------------------------------------------------------------------------
def synth_state_dict(seed=0, in_channels=3, num_classes=80, gain=1.0):
    """Deterministic parameters from numpy PCG64. Conv W ~ N(0, gain/fan_in) keeps
    activations O(1) through 75 layers so an absolute 1e-3 tolerance is a real test."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for cv in conv_list(in_channels, num_classes):
        p = cv["prefix"]
        fan_in = cv["cin"] * cv["k"] * cv["k"]
        w = rng.standard_normal((cv["cout"], cv["cin"], cv["k"], cv["k"]), dtype=np.float32)
        sd[p + ".conv.weight"] = torch.from_numpy(w * np.float32(np.sqrt(gain / fan_in)))
        if cv["bn"]:
            c = cv["cout"]
            sd[p + ".batch_norm.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
            sd[p + ".batch_norm.bias"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
            sd[p + ".batch_norm.running_mean"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
            sd[p + ".batch_norm.running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
            sd[p + ".batch_norm.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        else:
            sd[p + ".conv.bias"] = torch.from_numpy((0.1 * rng.standard_normal(cv["cout"])).astype(np.float32))
    return sd


def darknet_stream(sd, in_channels=3, num_classes=80):
    """Serialise ``sd`` as the fp32 stream of a Darknet .weights file (without the
    20-byte header): per BN block beta, gamma, mean, var then W; per bare conv bias
    then W (model.py:293-328 read order)."""
    parts = []
    for cv in conv_list(in_channels, num_classes):
        p = cv["prefix"]
        if cv["bn"]:
            for nm in ("bias", "weight", "running_mean", "running_var"):
                parts.append(sd[p + ".batch_norm." + nm].numpy().ravel())
        else:
            parts.append(sd[p + ".conv.bias"].numpy().ravel())
        parts.append(sd[p + ".conv.weight"].numpy().ravel())
    return np.concatenate(parts).astype(np.float32)


def synth_input(seed, batch, size, in_channels=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random((batch, in_channels, size, size), dtype=np.float32))
