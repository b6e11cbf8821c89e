"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/yolo_mi355x.h declares; host logic (program builder, loader, state_dict) behaves like
the reference.  No compute calls (there is no GPU in the build container)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import net as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from yolo_for_turbines_amd import _lib
    return _lib


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "yolo_mi355x.h")).read()
    declared = set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"yolo_conv_desc", "yolo_conv_op"}
    assert declared, "no declarations parsed"
    lib = built.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(built.EXPORTS) == declared
    assert lib.yolo_version() >= 100


def test_struct_layout_matches_header(built):
    import ctypes as C
    assert C.sizeof(built.ConvDesc) == 18 * 4
    assert C.sizeof(built.ConvOp) == 18 * 4 + 8 * 8
    assert C.sizeof(built.Call) == 8 + 8 * built.CALL_MAX_ARGS and C.sizeof(built.Reloc) == 24      # yolo_call / yolo_reloc
    hdr = open(os.path.join(ROOT, "include", "yolo_mi355x.h")).read()
    enum = re.search(r"enum \{ (YOLO_FN_FILL_ZERO = 1[^}]*)\}", hdr).group(1)
    names = [t.split("=")[0].strip() for t in enum.split(",")]
    assert [n[len("YOLO_FN_"):].lower() for n in names] == [k[len("yolo_"):] for k in built.FN_IDS]   # same order = same ids
    assert list(built.FN_IDS.values()) == list(range(1, len(names) + 1))
    assert built.lib().yolo_packed_weight_elems(255, 1024, 1) == 2 * 256 * 1024      # row-major + fragment-order copy
    assert built.lib().yolo_packed_weight_elems(32, 3, 3) == 128 * 64
    assert built.lib().yolo_packed_weight_elems(32, 3, 2) == 0
    # 3x3 with cin % 4 == 0: + the Winograd-domain filters [16][cin/4][cout_pad64][4]
    assert built.lib().yolo_packed_weight_elems(256, 128, 3) == 2 * 256 * 1152 + 16 * 128 * 256


def test_argument_errors_are_reported(built):
    lib = built.lib()
    d = built.ConvDesc(n=1, h=8, w=8, cin=5, cout=8, ksize=3, stride=1, x_ld=8, y_ld=8)
    rc = lib.yolo_conv_pick_tile(d)
    assert rc < 0 and b"cin" in lib.yolo_last_error()
    d = built.ConvDesc(n=1, h=8, w=8, cin=32, cout=8, ksize=5, stride=1, x_ld=32, y_ld=8)
    assert lib.yolo_conv_pick_tile(d) < 0
    assert lib.yolo_nms_workspace_bytes(2, 1000) > 2 * 1000 * (4 + 32 + 16 * 8)


def test_state_dict_keys_and_shapes_match_reference(golden):
    import yolo_for_turbines_amd as yt
    m = yt.YOLOv3(num_classes=80)
    keys = list(m.state_dict().keys())
    assert keys == list(golden("loader")["state_dict_keys"])
    spec = dict(onet.state_dict_spec(3, 80))
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(spec[k]), k
    assert sum(p.numel() for p in m.parameters()) == 61949149
    assert sum(p.numel() for p in yt.YOLOv3(num_classes=2).parameters()) == 61529119
    with pytest.raises(ValueError, match="Unsupported activation"):
        yt.YOLOv3(activation="relu")
    assert isinstance(m.layers[17], torch.nn.Upsample) and isinstance(m.layers[24], torch.nn.Upsample)
    assert len(m.layers) == 30


@pytest.mark.parametrize("fname,tag", [("yolov3.weights", "full"), ("darknet53.conv.74", "conv74")])
def test_darknet_loader_offsets_and_cutoff(tmp_path, golden, fname, tag):
    """Own loader vs the map recorded from the reference loader (same synthetic stream)."""
    import yolo_for_turbines_amd as yt
    g = golden("loader")
    sd = onet.synth_state_dict(21, 3, 80, gain=1.0)
    stream = onet.darknet_stream(sd, 3, 80)
    path = tmp_path / fname
    with open(path, "wb") as f:
        np.array([0, 2, 0, 32013312, 0], np.int32).tofile(f)
        stream.tofile(f)
    assert os.path.getsize(path) == 248007048
    m = yt.YOLOv3(num_classes=80, weights_path=str(path), freeze=(tag == "conv74"))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_weights()
    after = m.state_dict()
    for key, off, cnt, loaded in zip(g[f"{tag}/keys"], g[f"{tag}/offsets"], g[f"{tag}/counts"], g[f"{tag}/loaded"]):
        key = str(key)
        if loaded:
            assert torch.equal(after[key].reshape(-1), torch.from_numpy(stream[off:off + cnt])), key
        else:
            assert torch.equal(after[key], before[key]), key
    assert m.param_idx == 62001757
    # ... and against what the reference's UNCHANGED loader functions (model.py:227-337) left in a
    # yolo_for_turbines_amd.YOLOv3 when gen_golden.py `loader_bind` ran them on it (same stream, same starting state):
    # per-tensor sums, requires_grad flags (freeze), and the loader's counters
    gb = golden("loader_bind")
    bt = "full" if tag == "full" else "conv74_freeze"
    assert int(gb[f"{bt}/bound_equal"]) == 1 and int(gb[f"{bt}/restated_equal"]) == 1
    assert [str(k) for k in gb[f"{bt}/keys"]] == list(after)
    for key, s_loaded, a_loaded, s_init in zip(after, gb[f"{bt}/sums"], gb[f"{bt}/abs_sums"], gb[f"{bt}/init_sums"]):
        t = after[key].double()
        if not torch.equal(after[key], before[key]):         # a loaded tensor: the values of the stream
            assert abs(float(t.sum()) - s_loaded) <= 1e-9 * max(1.0, a_loaded), key
            assert abs(float(t.abs().sum()) - a_loaded) <= 1e-9 * max(1.0, a_loaded), key
        else:                                                # left alone by the cutoff - and the reference left it alone too
            assert s_loaded == s_init or key.endswith("num_batches_tracked"), key
    assert [p.requires_grad for p in m.parameters()] == [bool(v) for v in gb[f"{bt}/requires_grad"]]
    assert (m.param_idx, m.layer_id) == tuple(int(v) for v in gb[f"{bt}/counters"])
    if tag == "conv74":
        named = dict(m.named_parameters())
        assert not named["layers.0.conv.weight"].requires_grad
        assert named["layers.8.layers.5.0.conv.weight"].requires_grad
    # round trip through the writer
    out = tmp_path / "out.weights"
    if tag == "full":
        m.save_weights(str(out))
        assert np.array_equal(np.fromfile(out, np.float32, offset=20), stream)


def test_program_structure():
    import yolo_for_turbines_amd as yt
    from yolo_for_turbines_amd import engine, _lib as L
    m = yt.YOLOv3(num_classes=80)
    p = engine.build_network_program(m, 2, 416)
    assert len(p.ops) == 75 and p.n_pred == 3
    ups = [op for op in p.ops if op["out_mode"] == L.OUT_UPSAMPLE2X]
    assert [(op["y"].ld, op["y"].off, op["y"].C) for op in ups] == [(768, 0, 256), (384, 0, 128)]
    routes = [op for op in p.ops if op["y"] is not None and op["y"].off > 0]
    assert [(op["y"].ld, op["y"].off, op["y"].C, op["y"].H) for op in routes] == [(384, 128, 256, 52), (768, 256, 512, 26)]
    heads = [op for op in p.ops if op["out_mode"] == L.OUT_HEAD]
    assert [op["Ho"] for op in heads] == [13, 26, 52]
    n_res = sum(1 for op in p.ops if op["flags"] & L.FLAG_RESIDUAL)
    assert n_res == 23
    flops = sum(2 * op["Ho"] * op["Wo"] * op["block"].conv.out_channels * op["block"].conv.in_channels * op["k"] ** 2
                for op in p.ops)
    assert abs(flops / 1e9 - 65.864) < 0.01


def test_cpu_tensor_and_missing_gpu_fail_loudly():
    import yolo_for_turbines_amd as yt
    m = yt.YOLOv3(num_classes=2).eval()
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="MI355X only"):
        yt.decode_boxes(torch.zeros(1, 3, 2, 2, 7), torch.ones(3, 2), 2)
    with pytest.raises(RuntimeError, match="MI355X only"):
        yt.nms_indices(torch.zeros(4, 6), 0.5, 0.5)


def test_checkpoint_format_matches_reference(golden, tmp_path):
    """save_checkpoint / load_checkpoint (utils.py:383-416): the file this package writes for a deterministic model +
    SGD-momentum state has the reference's structure, key order, parameter order and per-tensor sums
    (tests/golden/checkpoint.npz, written where the reference's own save_checkpoint ran; the generator also loaded each
    side's file with the other side's loader). Round trip restores everything and forces the learning rate."""
    import torch
    import yolo_for_turbines_amd as yt
    from tests import golden_inputs as gi
    g = golden("checkpoint")
    assert list(g["cross_load_ok"]) == [1, 1]
    m, opt = gi.checkpoint_setup(yt.YOLOv3)
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    path = tmp_path / "ck.pth.tar"
    yt.save_checkpoint(m, opt, filename=str(path))
    ck = torch.load(str(path))
    assert list(ck.keys()) == list(g["top_keys"])
    assert list(ck["state_dict"].keys()) == list(g["state_dict_keys"])
    np.testing.assert_allclose([float(v.double().sum()) for v in ck["state_dict"].values()], g["state_dict_sums"], rtol=1e-12, atol=0)
    o = ck["optimizer"]
    assert list(o.keys()) == list(g["opt_top_keys"])
    assert list(o["state"].keys()) == list(g["opt_state_ids"])
    assert sorted(o["state"][0].keys()) == list(g["opt_state_entry_keys"])
    np.testing.assert_allclose([float(o["state"][i]["momentum_buffer"].double().sum()) for i in o["state"]], g["momentum_sums"],
                               rtol=1e-12, atol=0)
    grp = o["param_groups"][0]
    assert sorted(grp.keys()) == list(g["group_keys"]) and grp["params"] == list(g["group_params"])
    assert [grp["lr"], grp["momentum"], grp["dampening"], grp["weight_decay"]] == list(g["group_scalars"])
    m2 = yt.YOLOv3(num_classes=2)
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.5, momentum=0.9, weight_decay=5e-4)
    yt.load_checkpoint(m2, opt2, lr=0.125, filename="ck.pth.tar", model_folder=str(tmp_path))
    assert all(gr["lr"] == 0.125 for gr in opt2.param_groups)
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    for pa, pb in zip(m.parameters(), m2.parameters()):
        assert torch.equal(opt.state[pa]["momentum_buffer"], opt2.state[pb]["momentum_buffer"])
