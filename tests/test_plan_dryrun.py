"""CPU dry run of the engine's launch plan: the C-ABI calls of one forward / backward are RECORDED instead of launched (the
engine is built on CPU tensors, ``native.call`` and the HIP stream / event objects are stand-ins), so the control flow of
``UNetEngine.forward`` / ``backward`` -- which kernel entry, which stream, in which order -- is checked without a GPU."""
import pytest
import torch

from lifelong_nnunet_amd import engine as eng_mod
from lifelong_nnunet_amd import native as nat
from lifelong_nnunet_amd.engine import ConvBlock, UpBlock
from lifelong_nnunet_amd.network import Generic_UNet


class _FakeStream:
    def __init__(self, name):
        self.name = name
        self.cuda_stream = 0

    def wait_event(self, ev):
        _REC.append(("wait_event", self.name, ev.stream))

    def wait_stream(self, other):
        _REC.append(("wait_stream", self.name, other.name))


class _FakeEvent:
    def __init__(self, *a, **k):
        self.stream = None

    def record(self, stream=None):
        self.stream = (stream or _CUR[-1]).name


class _StreamCtx:
    def __init__(self, s):
        self.s = s

    def __enter__(self):
        _CUR.append(self.s)

    def __exit__(self, *a):
        _CUR.pop()


_REC = []
_CUR = [_FakeStream("main")]
_NSTREAMS = [0]


def install_dry_run(setattr_fn):
    """Replace the launch layer by recorders through ``setattr_fn(obj, name, value)`` (monkeypatch.setattr in the fixture below, plain
    setattr in the spawned workers of tests/test_dp_trainers_gloo.py)."""
    _REC.clear()
    del _CUR[1:]
    _NSTREAMS[0] = 0

    def new_stream(device=None):
        _NSTREAMS[0] += 1
        return _FakeStream(f"side{_NSTREAMS[0]}")

    def fake_call(name, *args):
        _REC.append(("call", name, _CUR[-1].name, args))

    setattr_fn(eng_mod, "_SIDE_STREAMS", {})
    setattr_fn(nat, "call", fake_call)
    setattr_fn(nat, "call_plain", lambda name, *a: _REC.append(("plain", name, a)))
    setattr_fn(torch.cuda, "current_stream", lambda *a, **k: _CUR[-1])
    setattr_fn(torch.cuda, "Stream", new_stream)
    setattr_fn(torch.cuda, "Event", _FakeEvent)
    setattr_fn(torch.cuda, "stream", lambda s: _StreamCtx(s))
    return _REC


@pytest.fixture
def dry(monkeypatch):
    return install_dry_run(monkeypatch.setattr)


def _engine(num_pool=3, patch=(16, 16, 16)):
    net = Generic_UNet(1, 8, 3, num_pool, patch_size=patch, batch_size=2, device='cpu')
    eng = net.engine_for(torch.zeros((2, 1) + patch))
    return net, eng


def _calls(rec, prefix=None):
    return [r for r in rec if r[0] == "call" and (prefix is None or r[1].startswith(prefix))]


def test_forward_plan_of_a_training_step(dry):
    """One fused normalise + head pass per decoder level whose channel count allows it (every level here), each writing its block's
    normalised tensor (the next level's transposed convolution / LwF's old heads read it); explicit head weights take the unfused
    pair; body=False re-evaluates heads on the stored activations without touching the body."""
    net, eng = _engine()
    x = torch.zeros(2, 1, 16, 16, 16)
    logits = eng.forward(x)
    assert len(logits) == 3
    segf = _calls(dry, "lnn_instnorm_lrelu_seg_fwd")
    assert len(segf) == 3 and all(c[3][1] is not None for c in segf)
    nconv = sum(isinstance(i, ConvBlock) for i in eng.order)
    # (blocks of <= lnn_instnorm_small_volume() voxels per sample that feed no fused head: conv + statistics + normalise in one call)
    small = _calls(dry, "lnn_conv3d_fwd_in_lrelu")
    nsmall = sum(isinstance(i, ConvBlock) and i.z.V <= eng.small_v and id(i) not in eng._seg_after for i in eng.order)
    assert len(small) == nsmall > 0
    assert len(_calls(dry, "lnn_instnorm_lrelu_fwd")) == nconv - 3 - nsmall and not _calls(dry, "lnn_seg1x1_fwd")
    dry.clear()
    w = [torch.zeros(3, s.cin, 1, 1, 1) for s in eng.segs]
    eng.forward(x, seg_weights=w, body=False)
    assert [c[1] for c in _calls(dry)] == ["lnn_seg1x1_fwd"] * 3
    dry.clear()
    eng.forward(x, seg_weights=w, body=True)
    assert not _calls(dry, "lnn_instnorm_lrelu_seg_fwd") and len(_calls(dry, "lnn_seg1x1_fwd")) == 3
    assert len(_calls(dry, "lnn_instnorm_lrelu_fwd")) + len(_calls(dry, "lnn_conv3d_fwd_in_lrelu")) == nconv


def _wgrads(rec):
    return [c for c in _calls(rec) if "wgrad" in c[1] and "unpack" not in c[1]]


def test_backward_launches_every_weight_gradient_once(dry):
    net, eng = _engine()
    x = torch.zeros(2, 1, 16, 16, 16)
    logits = eng.forward(x)
    dry.clear()
    dls = [None] + [torch.zeros_like(l) for l in logits[1:]]
    eng.backward(dls)
    wg = _wgrads(dry)
    layers = [i for i in eng.order if isinstance(i, (ConvBlock, UpBlock))]
    pan = lambda c: c[3][[j for j, a in enumerate(c[3]) if isinstance(a, eng_mod._Ptr)][0]]
    assert len(wg) == len(layers)
    assert len({id(pan(c).t) for c in wg}) == 1                                  # all into the one panel arena
    assert sorted(pan(c).off for c in wg) == sorted(i.panel for i in layers)     # every layer's panel exactly once
    assert all(c[2] != "main" for c in wg)                  # weight gradients never run on the main stream of the default plan
    # nothing but weight gradients on the side streams; the main stream joins every side stream before the batched unpack
    side_calls = [c for c in _calls(dry) if c[2] != "main"]
    assert all("wgrad" in c[1] for c in side_calls)
    joins = {r[2] for r in dry if r[0] == "wait_stream" and r[1] == "main"}
    assert joins == {c[2] for c in wg}
    last_join = max(i for i, r in enumerate(dry) if r[0] == "wait_stream")
    unpack = [i for i, r in enumerate(dry) if r[0] == "call" and r[1] == "lnn_unpack_wgrad_batched"]
    assert len(unpack) == 1 and unpack[0] > last_join
    # ONE side stream; the first layer's weight gradient (dy rebuilt from y and dL/dz inside it) is the last launch of backward
    first = [c for c in wg if c[1] == "lnn_conv3d_wgrad_c1_in_bwd"]
    assert len(first) == 1 and wg[-1] is first[0] and {c[2] for c in wg} == {"side1"}
    # every side launch waits for an event recorded on main when its dL/dy (and, for the first layer, the sums in ws) were enqueued
    waits = [r for r in dry if r[0] == "wait_event"]
    assert len(waits) == len(wg) and all(r[2] == "main" for r in waits)


def test_backward_without_overlap_runs_on_one_stream(dry):
    net, eng = _engine()
    eng.overlap_wgrad = False
    x = torch.zeros(2, 1, 16, 16, 16)
    logits = eng.forward(x)
    dry.clear()
    eng.backward([None] + [torch.zeros_like(l) for l in logits[1:]])
    assert {c[2] for c in _calls(dry)} == {"main"} and not [r for r in dry if r[0].startswith("wait")]


def test_data_parallel_backward_reports_watermarks_down_to_zero(dry):
    """With a ``progress`` callback (parallel.GradAllReducer.progress) every layer's panel is folded into the gradient arena on the
    side stream right behind its weight gradient, the watermarks never move up, and the LAST call hands over offset 0 from inside
    backward -- no bucket is left for finish().  All weight gradients (the first layer's too) keep ONE side stream here: the
    exchange is launched from it."""
    net, eng = _engine()
    x = torch.zeros(2, 1, 16, 16, 16)
    logits = eng.forward(x)
    dry.clear()
    marks = []
    eng.backward([None] + [torch.zeros_like(l) for l in logits[1:]], progress=lambda wm, side: marks.append((wm, side.name, len(dry))))
    assert marks[-1][0] == 0 and all(a[0] >= b[0] for a, b in zip(marks, marks[1:])) and {m[1] for m in marks} == {"side1"}
    assert marks[0][0] <= eng.arena.size
    calls = _calls(dry)
    assert {c[2] for c in calls if "wgrad" in c[1]} == {"side1"}              # no second side stream
    unp = [c for c in calls if c[1] == "lnn_unpack_wgrad"]
    assert len(unp) == sum(isinstance(i, (ConvBlock, UpBlock)) for i in eng.order) and all(c[2] == "side1" for c in unp)
    assert not [c for c in calls if c[1] == "lnn_unpack_wgrad_batched"]
    # the final watermark is reported after the first layer's weight gradient and its unpack were enqueued
    last_wgrad = max(i for i, r in enumerate(dry) if r[0] == "call" and r[1] == "lnn_unpack_wgrad")
    assert marks[-1][2] > last_wgrad


def test_deferred_loss_fetch_keeps_the_grad_scaler_semantics(dry, monkeypatch):
    """``defer_loss_fetch`` (what the epoch loop uses): run_iteration returns a DeferredLoss, the found-inf flag of iteration i is
    consumed when iteration i + 1 asks for the loss scale -- the scale sequence is the one of the synchronising loop
    (x 0.5 right after an overflowed iteration), and last_grad_norm / last_found_inf resolve on access."""
    from lifelong_nnunet_amd import get_trainer_class
    from lifelong_nnunet_amd.optim import DeferredLoss
    plans = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1}
    overflow_at = {2}
    scales = {}
    for mode in ("eager", "deferred"):
        tr = get_trainer_class("sequential")("seg_outputs", "A", plans=dict(plans), device="cpu")
        tr.initialize(True, num_epochs=1)
        tr.defer_loss_fetch = mode == "deferred"
        it = [0]
        ctrl = tr.optimizer._ctrl_buf

        def fake_call(name, *args, it=it, ctrl=ctrl):
            _REC.append(("call", name, _CUR[-1].name, args))
            if name == "lnn_gradnorm_sumsq":          # the norm pass leaves {sum g^2, #non-finite}: an overflow in iteration 2
                ctrl[0] = 4.0
                ctrl[1] = 1.0 if it[0] in overflow_at else 0.0
        monkeypatch.setattr(nat, "call", fake_call)
        seen = []
        for i in range(5):
            it[0] = i
            seen.append(tr.amp_grad_scaler.get_scale() if mode == "eager" else None)
            l = tr.run_iteration(tr.tr_gen, True)
            if mode == "deferred":
                assert isinstance(l, DeferredLoss) and tr._pending_step is not None
                # the scale this iteration USED is visible through the inverse it handed to the optimiser
                seen[-1] = 1.0 / tr.last_inv_scale
                if i == 3:
                    assert tr.last_found_inf is False and tr.last_grad_norm == 2.0 and tr._pending_step is None    # resolves on access
                float(l)
            else:
                assert not isinstance(l, DeferredLoss)
        scales[mode] = seen + [tr.amp_grad_scaler.get_scale() if mode == "eager" else None]
        if mode == "deferred":
            tr._finish_pending_step()
            scales[mode][-1] = tr.amp_grad_scaler.get_scale()
    assert scales["eager"] == scales["deferred"] == [65536.0, 65536.0, 65536.0, 32768.0, 32768.0, 32768.0]


def test_trainer_iterations_with_a_nested_split(dry):
    """``--split_at tu.1`` through the trainer: the iteration's ``update_after_iteration`` puts the module into the partition the
    reference arrives at (every tensor body, the head = ``tu.1`` at its construction-time value; multihead.py:_reference_resplit),
    the next task takes over the last head, and the step itself keeps launching the same kernels."""
    from lifelong_nnunet_amd import get_trainer_class
    plans = {"patch_size": (16, 16, 16), "batch_size": 2, "num_pool": 2, "base_num_features": 8, "num_classes": 3, "num_input_channels": 1}
    tr = get_trainer_class("sequential")("tu.1", "A", plans=dict(plans), device="cpu")
    tr.initialize(True, num_epochs=1)
    mh = tr.mh_network
    names = [n for n, _ in mh.model.named_parameters()]
    assert [n for n, _ in mh.heads["A"].named_parameters()] == ["tu.1.weight", "seg_outputs.0.weight", "seg_outputs.1.weight"]
    init = mh.state_init["tu.1.weight"].clone()
    calls = []
    for i in range(2):
        dry.clear()
        tr.run_iteration(tr.tr_gen, True)
        calls.append([c[1] for c in _calls(dry)])
        assert [n for n, _ in mh.body.named_parameters()] == names
        assert [n for n, _ in mh.heads["A"].named_parameters()] == ["tu.1.weight"]
        assert torch.equal(dict(mh.heads["A"].named_parameters())["tu.1.weight"], init)
    assert calls[0] == calls[1] and "lnn_sgd_nesterov_step_clipped" in calls[0]
    mh.add_new_task("B", use_init=False)
    mh.assemble_model("B")
    assert torch.equal(dict(mh.model.named_parameters())["tu.1.weight"], init)
    with pytest.raises(RuntimeError):
        mh.add_new_task("C", use_init=True)
