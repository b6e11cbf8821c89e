"""Drop-in mirror of the reference's model interface (`/root/reference/code/model.py`).

Same class names, constructor arguments, ``nn.Module`` tree and ``state_dict`` keys (438
entries for nc=80), so checkpoints, optimizers and the Darknet ``.weights`` format keep
working — but ``forward`` never runs ``torch.nn`` arithmetic: it hands the tensors to the
MI355X engine (``engine.py`` -> ``libyolo_mi355x.so``).  The ``nn.Conv2d`` / ``nn.BatchNorm2d``
children exist only as parameter holders in the reference's OIHW fp32 layout.

There is no CPU path: a CPU tensor or a missing HIP library raises.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import engine

# Architecture table, restating `/root/reference/code/model.py:20-45`:
# (filters, kernel, stride) | ["B", repeats] | "S" scale head | "U" upsample + route concat.
layer_config = [
    (32, 3, 1), (64, 3, 2), ["B", 1], (128, 3, 2), ["B", 2], (256, 3, 2), ["B", 8], (512, 3, 2), ["B", 8],
    (1024, 3, 2), ["B", 4],
    (512, 1, 1), (1024, 3, 1), "S", (256, 1, 1), "U", (256, 1, 1), (512, 3, 1), "S",
    (128, 1, 1), "U", (128, 1, 1), (256, 3, 1), "S",
]


def _make_activation(name):
    if name == "leaky_relu":
        return nn.LeakyReLU(0.1)
    if name == "mish":
        return nn.Mish()
    raise ValueError(f"Unsupported activation: {name}")          # model.py:68


class CNNBlock(nn.Module):
    """Conv2d -> BatchNorm2d -> LeakyReLU(0.1)/Mish, or a bare Conv2d with bias
    (reference: model.py:47-86). Executed as ONE fused HIP kernel."""

    def __init__(self, in_channels, out_channels, batch_norm_act=True, activation="leaky_relu", **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, bias=not batch_norm_act, **kwargs)
        self.batch_norm = nn.BatchNorm2d(out_channels) if batch_norm_act else None
        self.activation = _make_activation(activation) if batch_norm_act else None
        self.batch_norm_act = batch_norm_act
        self._pack_gen = 0

    def set_layers(self, layers):                                 # model.py:74-78
        # The reference's loader (model.py:238-250) fills the tensors through `.data.copy_` - which PyTorch's version
        # counters do not see - and then hands the layers back through this method: count it, so that packed weights made
        # before such a load are re-made (engine.PackedBlock.stamp_of includes this counter).
        self._pack_gen += 1
        self.conv = layers[0]
        if self.batch_norm_act:
            self.batch_norm = layers[1]
            self.activation = layers[2]

    def forward(self, x):
        return engine.run_module_nchw(self, x)


class ResidualBlock(nn.Module):
    """num_blocks x [1x1 C->C/2, 3x3 C/2->C] with optional skip (reference: model.py:88-121).
    The skip add is fused into the 3x3 kernel's epilogue."""

    def __init__(self, in_channels, activation="leaky_relu", use_residual=True, num_blocks=1):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(num_blocks):
            self.layers.append(nn.Sequential(
                CNNBlock(in_channels, in_channels // 2, activation=activation, kernel_size=1),
                CNNBlock(in_channels // 2, in_channels, activation=activation, kernel_size=3, padding=1)))
        self.use_residual = use_residual
        self.num_blocks = num_blocks

    def set_layers(self, layers):
        self.layers = layers

    def forward(self, x):
        return engine.run_module_nchw(self, x)


class ScalePredictionBlock(nn.Module):
    """3x3 C->2C (BN, act) + 1x1 2C->3*(5+nc) (bias); output (B,3,g,g,5+nc)
    (reference: model.py:123-148). The reshape + permute is the 1x1 kernel's store pattern."""

    def __init__(self, in_channels, num_classes, activation="leaky_relu", anchors_per_scale=3):
        super().__init__()
        self.pred_block = nn.Sequential(
            CNNBlock(in_channels, in_channels * 2, activation=activation, kernel_size=3, padding=1),
            CNNBlock(2 * in_channels, (num_classes + 5) * anchors_per_scale, activation=activation,
                     batch_norm_act=False, kernel_size=1))
        self.num_classes = num_classes
        self.anchors_per_scale = anchors_per_scale

    def set_layers(self, layers):
        self.pred_block = layers

    def forward(self, x):
        return engine.run_module_nchw(self, x)


class YOLOv3(nn.Module):
    """Same constructor and call surface as the reference (model.py:150-193).

    ``forward(x)``: x (B,3,S,S) float tensor on the GPU, S a multiple of 32 -> list of three
    writable tensors (B,3,g,g,5+nc), g = S/32, S/16, S/8. Raises AssertionError on NaN input and
    ValueError("Nan in layer") on a NaN activation, like model.py:175,183-184 — checked once per
    forward through a device-side sticky flag instead of 27 host syncs.
    """

    def __init__(self, in_channels=3, num_classes=80, activation="leaky_relu", weights_path=None, freeze=False):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.activation = activation
        self.layers = self._create_model_layers()
        self.param_idx = 0
        self.layer_id = 0
        self.weights_path = None
        self.freeze = freeze
        self._engine = engine.ModelState()
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._engine.invalidate())
        if weights_path:
            self.weights_path = weights_path
            with open(weights_path, "rb") as f:
                np.fromfile(f, dtype=np.int32, count=5)            # 20-byte header, unused (model.py:165)
                self.weights = np.fromfile(f, dtype=np.float32)
            self.cutoff = None
            file_name = os.path.basename(str(weights_path))
            if ".conv" in file_name:
                self.cutoff = int(file_name.split(".")[-1])         # e.g. darknet53.conv.74 -> 74

    # ------------------------------------------------------------------ structure
    def _create_model_layers(self):
        layers = nn.ModuleList()
        c = self.in_channels
        act = self.activation
        for block in layer_config:
            if isinstance(block, tuple):
                cout, k, s = block
                layers.append(CNNBlock(c, cout, activation=act, kernel_size=k, stride=s, padding=1 if k == 3 else 0))
                c = cout
            elif isinstance(block, list):
                layers.append(ResidualBlock(c, activation=act, num_blocks=block[1]))
            elif block == "S":
                layers += [ResidualBlock(c, activation=act, use_residual=False, num_blocks=1),
                           CNNBlock(c, c // 2, activation=act, kernel_size=1),
                           ScalePredictionBlock(c // 2, num_classes=self.num_classes, activation=act)]
                c = c // 2
            elif block == "U":
                layers.append(nn.Upsample(scale_factor=2))
                c = c * 3
        return layers

    # -------------------------------------------------------------------- forward
    def forward(self, x):
        return self._engine.forward(self, x)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if "_engine" in self.__dict__:
            self._engine.invalidate(drop_plans=True)               # .to(device) / .float() move the parameters
        return out

    # ------------------------------------------------------- Darknet weights loader
    def _cnn_blocks_in_file_order(self):
        """CNNBlocks in module order, with a marker for each nn.Upsample (it advances the
        reference's ``layer_id`` counter without consuming floats, model.py:234-235,336)."""
        for top in self.layers:
            if isinstance(top, CNNBlock):
                yield top
            elif isinstance(top, (ResidualBlock, ScalePredictionBlock)):
                for m in top.modules():
                    if isinstance(m, CNNBlock):
                        yield m
            else:
                yield None

    def load_weights(self):
        """Fill the parameters from the Darknet fp32 stream read by ``__init__``.

        Format and quirks follow the reference loader (model.py:227-337): per BN block
        beta, gamma, running_mean, running_var then the conv weight (OIHW); per bare conv
        bias then weight; a counter that advances once per BatchNorm, per Conv AND per
        Upsample is compared with the ``.conv.N`` cutoff, beyond which tensors are skipped
        (the stream position still advances); ``freeze`` clears requires_grad on what was
        loaded. The counters persist on the instance like the reference's.
        """
        if self.weights_path is None:
            raise AttributeError("YOLOv3 was constructed without weights_path")
        w = self.weights

        def take(t):
            n = t.numel()
            active = self.cutoff is None or self.layer_id < self.cutoff
            if active:
                chunk = w[self.param_idx:self.param_idx + n]
                if chunk.size != n:
                    raise ValueError(f"weights file too short at float {self.param_idx}")
                t.data.copy_(torch.from_numpy(chunk).view_as(t))
                if self.freeze:
                    t.requires_grad = False
            self.param_idx += n

        for blk in self._cnn_blocks_in_file_order():
            if blk is None:
                self.layer_id += 1
                continue
            if blk.batch_norm_act:
                bn = blk.batch_norm
                for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                    take(t)
                self.layer_id += 1
                take(blk.conv.weight)
                self.layer_id += 1
            else:
                take(blk.conv.bias)
                take(blk.conv.weight)
                self.layer_id += 1
        self._engine.invalidate()
        print(f"Weights from {self.weights_path} loaded successfully.")

    def save_weights(self, path, header=(0, 2, 0, 0, 0)):
        """Inverse of ``load_weights``: write the parameters as a Darknet .weights file."""
        parts = []
        for blk in self._cnn_blocks_in_file_order():
            if blk is None:
                continue
            if blk.batch_norm_act:
                bn = blk.batch_norm
                parts += [bn.bias, bn.weight, bn.running_mean, bn.running_var]
            else:
                parts.append(blk.conv.bias)
            parts.append(blk.conv.weight)
        with open(path, "wb") as f:
            np.asarray(header, np.int32).tofile(f)
            for t in parts:
                t.detach().float().cpu().contiguous().numpy().ravel().tofile(f)
