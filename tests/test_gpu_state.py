"""GPU tests (``-m gpu``) of the host-side state around the kernels: packed-weight / BatchNorm-fold freshness across
train <-> eval switches and HIP-graph replays, the lifetime of the buffers a captured graph points at, and the guard
against two train-mode forwards sharing one set of saved activations. Every check compares with a FRESH model loaded
from the live ``state_dict`` (nothing cached), i.e. with what the reference's stateless forward would compute."""
import pytest
import torch

from oracle import net as onet
from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu
NC, S, B = 2, 96, 2


@pytest.fixture(scope="module")
def yt():
    import yolo_for_turbines_amd as pkg
    from yolo_for_turbines_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return pkg


def _case(seed):
    anchors = gi.TRAIN_CASE["anchors"]
    sd = onet.synth_state_dict(seed, 3, NC, gain=gi.NET_GAIN)
    x = onet.synth_input(seed + 1, B, S).cuda()
    tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(B, S, NC, anchors, seed + 2)]
    sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).cuda()
    return sd, x, tg, sa


def _fresh_eval(yt, model, x, autocast):
    """Outputs of a brand-new model carrying ``model``'s current state_dict."""
    m = yt.YOLOv3(num_classes=NC, activation=model.activation)
    m.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()})
    m = m.cuda().eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        return m(x)


def _eval(model, x, autocast):
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        return model(x)


@pytest.mark.parametrize("autocast", [False, True])
def test_eval_after_training_with_frozen_blocks_uses_the_live_running_stats(yt, autocast):
    """freeze=True (model.py:306-309) leaves the backbone's weights untouched, but its BatchNorm layers still run in train
    mode and keep updating running_mean / running_var through the kernels' raw pointers. An eval forward after such
    steps must fold the CURRENT statistics, although no parameter of those blocks changed."""
    sd, x, tg, sa = _case(301)
    m = yt.YOLOv3(num_classes=NC, activation="mish")
    m.load_state_dict(sd)
    m = m.cuda()
    for layer in list(m.layers)[:9]:
        for p in layer.parameters():
            p.requires_grad_(False)
    first = _eval(m, x, autocast)                                   # packs weights, folds the initial statistics
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    lf = yt.FusedYOLOLoss()
    m.train()
    rm_before = m.state_dict()["layers.0.batch_norm.running_mean"].clone()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            po = m(x)
        sum(sum(lf(po[i], tg[i], sa[i])) for i in range(3)).backward()
        opt.step()
    assert not torch.equal(rm_before, m.state_dict()["layers.0.batch_norm.running_mean"])
    got = _eval(m, x, autocast)
    want = _fresh_eval(yt, m, x, autocast)
    for a, b, c in zip(got, want, first):
        assert torch.equal(a, b)                                    # same kernels, same inputs: bit-identical
        assert not torch.equal(a, c)                                # and really different from the stale fold


@pytest.mark.parametrize("ac", [None, torch.bfloat16])
def test_eval_after_graph_replays_sees_the_updated_parameters(yt, ac):
    """A HIP-graph replay rewrites every parameter and BatchNorm statistic without dispatching a torch op (no version
    counter moves): eval -> replay -> eval must still use the new values."""
    sd, x, tg, sa = _case(311)
    m = yt.YOLOv3(num_classes=NC, activation="mish")
    m.load_state_dict(sd)
    m = m.cuda().train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    step = yt.GraphedTrainStep(m, opt, sa, x, tg, autocast_dtype=ac)
    e0 = _eval(m, x, ac is not None)                                # eval cache is now warm
    m.train()
    step(x, tg)
    step(x, tg)
    got = _eval(m, x, ac is not None)
    want = _fresh_eval(yt, m, x, ac is not None)
    for a, b, c in zip(got, want, e0):
        assert torch.equal(a, b)
        assert not torch.equal(a, c)
    m.train()
    step(x, tg)                                                     # and the graph still replays after an eval in between
    assert all(torch.isfinite(p).all() for p in m.parameters())


def test_captured_graph_owns_its_train_plan(yt):
    """The graph bakes in pointers to the train plan's buffers. Multi-scale training builds plans of other sizes between
    replays (LRU of 2): the captured size must survive that, and replays must keep matching eager steps bit for bit.
    After model.to()/.float() (parameters may move, plans are dropped) a replay must refuse instead of scribbling."""
    sd, x, tg, sa = _case(321)
    anchors = gi.TRAIN_CASE["anchors"]

    def make():
        m = yt.YOLOv3(num_classes=NC, activation="mish")
        m.load_state_dict(sd)
        m = m.cuda().train()
        return m, torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    lf = yt.FusedYOLOLoss()

    def eager(m, o, xx, tt, ss):
        o.zero_grad(set_to_none=True)
        po = m(xx)
        sum(sum(lf(po[i], tt[i], ss[i])) for i in range(3)).backward()
        o.step()
    others = []
    for S2 in (64, 128, 160):
        x2 = onet.synth_input(330 + S2, B, S2).cuda()
        t2 = [torch.from_numpy(t).cuda() for t in gi.synth_targets(B, S2, NC, anchors, 331 + S2)]
        s2 = (torch.tensor(anchors) * torch.tensor([S2 // 32, S2 // 16, S2 // 8]).view(3, 1, 1)).cuda()
        others.append((x2, t2, s2))
    m1, o1 = make()
    m2, o2 = make()
    step = yt.GraphedTrainStep(m2, o2, sa, x, tg)                   # 3 warm-up steps at size S on m2
    for _ in range(3):
        eager(m1, o1, x, tg, sa)
    plan = step._plan
    for xx, tt, ss in others:                                       # three other sizes through both models: evicts everything unpinned
        eager(m1, o1, xx, tt, ss)
        eager(m2, o2, xx, tt, ss)
    assert plan in m2._engine._plans.values() and plan.pinned
    assert sum(1 for k in m2._engine._plans if k[0] == "train") <= 1 + m2._engine.max_train_plans
    eager(m1, o1, x, tg, sa)
    step(x, tg)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    m2.float()                                                      # nn.Module._apply: the engine drops its plans
    with pytest.raises(RuntimeError, match="dropped its plans"):
        step(x, tg)


def test_second_train_forward_before_backward_is_refused(yt):
    """One set of saved activations per (batch, size, dtype): a backward whose forward has been overwritten by a newer
    train-mode forward of the same shape must raise, not return gradients of the wrong batch."""
    sd, x, tg, sa = _case(341)
    m = yt.YOLOv3(num_classes=NC, activation="leaky_relu")
    m.load_state_dict(sd)
    m = m.cuda().train()
    lf = yt.FusedYOLOLoss()
    p1 = m(x)
    l1 = sum(sum(lf(p1[i], tg[i], sa[i])) for i in range(3))
    p2 = m(x * 0.5)
    l2 = sum(sum(lf(p2[i], tg[i], sa[i])) for i in range(3))
    with pytest.raises(RuntimeError, match="newer train-mode forward"):
        l1.backward()
    l2.backward()                                                   # the newest forward's backward is fine
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("ac", [False, True, "fp16_scaler"])
def test_training_with_the_fused_sgd_step_follows_torch_sgd(yt, ac):
    """Three fine-tune steps with yt.SGD (parameters written through raw pointers by one HIP launch) and three with
    torch.optim.SGD from the same start: every parameter and running statistic bit-equal after each step. Catches an update
    the engine does not notice (it re-packs a weight when its version counter moves), not only the update arithmetic.
    fp32, bf16 autocast, and fp16 autocast stepped through torch.amp.GradScaler as train.py:67-69 does."""
    sd, x, tg, sa = _case(311)
    lf = yt.FusedYOLOLoss()

    def run(opt_cls):
        m = yt.YOLOv3(num_classes=NC, activation="leaky_relu")
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        m = m.cuda().train()
        opt = opt_cls(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
        scaler = torch.amp.GradScaler(init_scale=256.0) if ac == "fp16_scaler" else None     # train.py:39,67-69
        snaps = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16 if scaler else torch.bfloat16, enabled=bool(ac)):
                preds = m(x)
                loss = sum(sum(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3))
            if scaler:
                scaler.scale(loss).backward()
                scaler.step(opt)                                   # unscale, inf check, optimizer.step()
                scaler.update()
            else:
                loss.backward()
                opt.step()
            snaps.append({k: v.detach().clone() for k, v in m.state_dict().items()})
        return snaps

    a, b = run(yt.SGD), run(torch.optim.SGD)
    for step in range(3):
        for k in a[step]:
            assert torch.equal(a[step][k], b[step][k]), (step, k)
    assert not torch.equal(a[2]["layers.0.conv.weight"], a[0]["layers.0.conv.weight"])


@pytest.mark.parametrize("ac", [None, torch.bfloat16])
def test_graph_replays_follow_a_per_step_lr_schedule(yt, ac):
    """The reference steps a LinearLR warm-up after EVERY batch (train.py:71-74,187-189). A replayed graph must train at
    the scheduler's current learning rate, not at the capture-time one: yt.SGD reads its hyper-parameters from device
    memory when the kernel runs. Replays under LinearLR == eager steps under LinearLR, bit for bit; and an optimizer that
    bakes its floats into the capture (torch.optim.SGD) makes the replay raise once the LR has moved."""
    sd, x, tg, sa = _case(351)
    lf = yt.FusedYOLOLoss()

    def make(opt_cls):
        m = yt.YOLOv3(num_classes=NC, activation="mish")
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        m = m.cuda().train()
        opt = opt_cls(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
        sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, total_iters=8)       # train.py:187-189
        return m, opt, sched

    def eager(m, opt):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=ac or torch.bfloat16, enabled=ac is not None):
            po = m(x)
            loss = sum(sum(lf(po[i], tg[i], sa[i])) for i in range(3))
        loss.backward()
        opt.step()

    m1, o1, s1 = make(yt.SGD)
    m2, o2, s2 = make(yt.SGD)
    step = yt.GraphedTrainStep(m2, o2, sa, x, tg, autocast_dtype=ac, warmup=3)   # 3 warm-up steps at the initial LR
    for _ in range(3):
        eager(m1, o1)
    lrs = []
    for _ in range(4):
        s1.step()
        s2.step()
        lrs.append(o2.param_groups[0]["lr"])
        eager(m1, o1)
        step(x, tg)
        for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(a, b), (k, lrs)
    assert len(set(lrs)) == 4                                       # the schedule really moved between replays
    # momentum / weight decay changed by hand between replays are picked up as well
    for o in (o1, o2):
        o.param_groups[0]["momentum"] = 0.5
        o.param_groups[0]["weight_decay"] = 1e-3
    eager(m1, o1)
    step(x, tg)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # torch.optim.SGD captures Python floats: refuse to replay at a stale learning rate
    m3, o3, s3 = make(torch.optim.SGD)
    step3 = yt.GraphedTrainStep(m3, o3, sa, x, tg, autocast_dtype=ac)
    step3(x, tg)
    s3.step()
    with pytest.raises(RuntimeError, match="hyper-parameters changed since the capture"):
        step3(x, tg)


@pytest.mark.parametrize("autocast", [False, True])
def test_head_bias_changed_alone_is_seen_by_the_next_train_forward(yt, autocast):
    """The heads have no BatchNorm: in training too the kernel gets (scale, shift) = (1, conv bias) from the folded cache.
    A bias that changes on its own (objectness-prior init after a first forward, frozen weights with a trainable bias, a
    partial load_state_dict) must reach the next train-mode forward although no conv weight moved."""
    sd, x, tg, sa = _case(361)
    m = yt.YOLOv3(num_classes=NC, activation="leaky_relu")
    m.load_state_dict(sd)
    m = m.cuda().train()

    def fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            return [p.detach().float().clone() for p in m(x)]
    a = fwd()
    heads = [m.layers[i].pred_block[1].conv for i in (15, 22, 29)]
    with torch.no_grad():
        for h in heads:
            h.bias.add_(1.0)
    b = fwd()
    for p, q in zip(a, b):
        d = q - p
        assert float((d - 1.0).abs().max()) < (2e-2 if autocast else 1e-5)
    # replacing the Parameter OBJECT (same values elsewhere in memory) is seen too, in eval (fast walk) and in training
    m.eval()
    with torch.no_grad():
        e0 = [p.clone() for p in m(x)]
        e0b = [p.clone() for p in m(x)]                              # second call: the fast freshness walk is armed
    for p, q in zip(e0, e0b):
        assert torch.equal(p, q)
    heads[0].bias = torch.nn.Parameter(heads[0].bias.detach().clone() - 1.0)
    with torch.no_grad():
        e1 = m(x)
    assert float((e1[0] - e0[0] + 1.0).abs().max()) < 1e-5
    assert torch.equal(e1[1], e0[1])


def test_rccl_all_reduce_inside_a_graph_capture_one_rank():
    """RCCL collectives inside a HIP-graph capture (opt-in: GraphedTrainStep(allow_data_parallel=True)): with a 1-rank RCCL
    group in a child process, replays of the captured data-parallel step (synchronous bucket all-reduces on the capturing
    stream, communicator warmed up eagerly, thread-local capture errors) equal eager data-parallel steps bit for bit. Run
    ONCE; the multi-rank case needs the driver's 8-GPU node."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YOLO_FORCE_DIST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29671", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "workers", "dp_graph_check.py")], env=env, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "DP_GRAPH_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
