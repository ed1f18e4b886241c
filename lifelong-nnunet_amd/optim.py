"""Optimiser side of the iteration (multihead/nnUNetTrainerMultiHead.py:294-301, 626-631) on the flat arenas.

The reference runs, per iteration, ``scaler.scale(l).backward(); scaler.unscale_(opt);
clip_grad_norm_(params, 12); scaler.step(opt); scaler.update()`` -- about six foreach passes over the
31 M parameters.  Here: ONE norm pass (sum of squares + non-finite count, fp64 accumulation) and ONE
fused pass (unscale x clip coefficient, weight decay, Nesterov momentum, update); the clip
coefficient and the inf-skip decision are taken on the device, so nothing synchronises before the
step.  Only parameters with ``requires_grad`` are updated (MH.py:299), as contiguous arena ranges.
"""
from __future__ import annotations

import torch

from . import native as nat


def _ranges(net):
    """Contiguous arena ranges [lo, hi) of the parameters that require grad."""
    # torch.optim.SGD skips parameters whose .grad is None (no weight decay, no momentum): that is the case of
    # the zero-weight deep-supervision head, which the engine reports in net.params_without_grad
    # ... unless a penalty node (EWC / RW) wrote a gradient into its slot this iteration (losses._EWCPenaltyFunction)
    skip = set(getattr(net, "params_without_grad", ())) - set(getattr(net, "penalty_grad_names", ()))
    spans = sorted((p._lnn_slot.offset, p._lnn_slot.offset + (p._lnn_slot.numel + 3) // 4 * 4)
                   for n, p in net._named if p.requires_grad and n not in skip)
    out = []
    for lo, hi in spans:
        if out and out[-1][1] == lo:
            out[-1][1] = hi
        else:
            out.append([lo, hi])
    return [(lo, hi) for lo, hi in out]


class _Off:
    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.off


class GradScaler:
    """torch.cuda.amp.GradScaler semantics (init 65536, x2 after 2000 clean steps, x0.5 + skip on inf/nan)."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0
        self.enabled = enabled

    def scale(self, loss):
        return loss * self._scale if self.enabled else loss

    def backward(self, loss):
        """``self.scale(loss).backward()`` without the two multiply launches: the scale rides in as the seed gradient."""
        if not self.enabled:
            loss.backward()
            return
        t = getattr(self, "_seed", None)
        if t is None or t.device != loss.device or self._seed_value != self._scale:
            self._seed = t = torch.full((), self._scale, dtype=loss.dtype, device=loss.device)
            self._seed_value = self._scale
        loss.backward(gradient=t)

    def get_scale(self):
        return self._scale

    def update(self, found_inf: bool):
        if not self.enabled:
            return
        if found_inf:
            self._scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker = 0

    def state_dict(self):
        """Same keys as ``torch.cuda.amp.GradScaler.state_dict()`` (checkpoint interop, MH.py:1164-1197 -> upstream
        ``NetworkTrainer.save_checkpoint``)."""
        if not self.enabled:
            return {}
        return {"scale": self._scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, d):
        if not d:
            return
        self._scale, self._growth_tracker = float(d["scale"]), int(d["_growth_tracker"])
        self.growth_factor = d.get("growth_factor", self.growth_factor)
        self.backoff_factor = d.get("backoff_factor", self.backoff_factor)
        self.growth_interval = d.get("growth_interval", self.growth_interval)


class AsyncCtrl:
    """{sum g^2, #non-finite, loss} of ONE enqueued step on its way to the host: a copy into pinned memory + an event, both on the
    stream of the step.  ``get()`` waits for the event only (not for the device) and caches the three numbers."""

    def __init__(self, host, event):
        self._host, self._event, self._vals = host, event, None

    def ready(self):
        return self._vals is not None or self._event is None or self._event.query()

    def get(self):
        if self._vals is None:
            if self._event is not None:
                self._event.synchronize()
            self._vals = self._host.numpy().copy()
            self._host = self._event = None
        return self._vals


class DeferredLoss:
    """The loss of an iteration whose device-to-host copy is still in flight (``run_iteration`` of a trainer with
    ``defer_loss_fetch``: what its epoch loop uses -- the reference collects the per-iteration losses and takes their mean at the
    end of the epoch, MH.py:655).  Converts like a number; the first conversion waits for the copy."""
    __slots__ = ("_ctrl", "_v")

    def __init__(self, ctrl):
        self._ctrl, self._v = ctrl, None

    def value(self):
        if self._v is None:
            import numpy as np
            self._v = np.float32(self._ctrl.get()[2])
            self._ctrl = None
        return self._v

    def __float__(self):
        return float(self.value())

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        return np.asarray(self.value(), dtype=dtype)

    def item(self):
        return float(self.value())

    def __repr__(self):
        return repr(self.value())

    def __format__(self, spec):
        return format(float(self.value()), spec)

    # a subclass's run_iteration may compare or do arithmetic on what the base class returns (the reference returns a numpy scalar
    # there, MH.py:655): every numeric protocol entry resolves the copy and delegates to that scalar
    def __bool__(self):
        return bool(self.value())

    def __neg__(self):
        return -self.value()

    def __abs__(self):
        return abs(self.value())

    def __hash__(self):
        return hash(float(self.value()))


def _deferred_binop(name):
    def op(self, other):
        other = other.value() if isinstance(other, DeferredLoss) else other
        return getattr(self.value(), name)(other)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow"):
    setattr(DeferredLoss, f"__{_n}__", _deferred_binop(f"__{_n}__"))
    setattr(DeferredLoss, f"__r{_n}__", _deferred_binop(f"__r{_n}__"))
for _n in ("lt", "le", "gt", "ge", "eq", "ne"):
    setattr(DeferredLoss, f"__{_n}__", _deferred_binop(f"__{_n}__"))
del _n


class FusedSGD:
    """SGD(momentum=0.99, nesterov=True) over the flat arena with ``torch.optim.SGD``'s surface
    (``param_groups[0]['lr']``, ``zero_grad``, ``step``, ``state_dict``)."""

    def __init__(self, net, lr, weight_decay=0.0, momentum=0.99, nesterov=True):
        assert nesterov, "the reference uses nesterov=True (MH.py:300)"
        self.net = net
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "momentum": momentum, "nesterov": True,
                              "params": [p for _, p in net._named if p.requires_grad]}]
        # ctrl[0:2] = {sum g^2, #non-finite}; the rest is the scratch of the deterministic two-stage reduction
        self._ctrl_buf = torch.zeros(nat.query("lnn_flat_reduce_ws_doubles"), dtype=torch.float64, device=net.arena.theta.device)
        self.ctrl = self._ctrl_buf[:2]
        self._ctrl_valid = False
        self._ever_stepped = set()       # names the optimiser has stepped at least once (torch creates their momentum_buffer then)
        self._pending_stepped = None     # names of the last enqueued step, counted once its found-inf flag is known to be 0
        self._async = None               # AsyncCtrl of the last enqueued step, if its control block was fetched without a sync
        self._host_ring, self._ring_i = None, 0

    def zero_grad(self, set_to_none=False):
        self.net.arena.grad.zero_()
        self.net.penalty_grad_names = set()
        self.net.bind_grads()

    def grad_norm_pass(self, inv_scale=1.0):
        """sum of squares of the UNSCALED gradient + non-finite count -> self.ctrl (device)."""
        a = self.net.arena
        first = 1
        self._resolve_pending()          # (a host sync only if nobody fetched the previous step's control block)
        for lo, hi in _ranges(self.net):
            nat.call("lnn_gradnorm_sumsq", _Off(a.grad, lo), hi - lo, float(inv_scale), self._ctrl_buf, first)
            first = 0
        self._ctrl_valid = True

    def step(self, inv_scale=1.0, max_norm=0.0):
        """Fused unscale + clip (``max_norm`` > 0, needs grad_norm_pass) + Nesterov update; skipped on
        the device if a non-finite gradient was counted."""
        a, g = self.net.arena, self.param_groups[0]
        if not self._ctrl_valid:
            self._resolve_pending()
            self.ctrl.zero_()
            max_norm = 0.0
        for lo, hi in _ranges(self.net):
            nat.call("lnn_sgd_nesterov_step_clipped", _Off(a.theta, lo), _Off(a.momentum, lo), _Off(a.grad, lo), hi - lo,
                     float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), float(inv_scale), float(max_norm),
                     self.ctrl)
        self._ctrl_valid = False
        skip = set(getattr(self.net, "params_without_grad", ())) - set(getattr(self.net, "penalty_grad_names", ()))
        # GradScaler.step does not call optimizer.step() on overflow, so torch creates no momentum_buffer then: the names count
        # as stepped only once the device-side found-inf flag of THIS step has reached the host (fetch_with_loss / read_ctrl)
        self._pending_stepped = [n for n, p in self.net._named if p.requires_grad and n not in skip]
        self.net.mark_params_changed()

    def _resolve_pending(self, found_inf=None):
        if self._pending_stepped is None:
            return
        if found_inf is None:
            if self._async is not None:               # the step's control block is already on its way to the host
                found_inf = bool(self._async.get()[1] > 0)
            else:
                found_inf = bool(self.ctrl[1].item() > 0)
        self._async = None
        if not found_inf:
            self._ever_stepped.update(self._pending_stepped)
        self._pending_stepped = None

    def fetch_with_loss(self, loss):
        """numpy [sum g^2, #non-finite, loss]: the loss is parked next to the control block (slot 2 is scratch of the norm
        reduction, free once the step has been enqueued) so that ONE device-to-host copy carries all three."""
        self._ctrl_buf[2:3].copy_(loss.detach().reshape(1))
        arr = self._ctrl_buf[:3].cpu().numpy()
        self._resolve_pending(bool(arr[1] > 0))
        return arr

    def fetch_with_loss_async(self, loss):
        """``fetch_with_loss`` without the host synchronisation: the three numbers are copied into pinned host memory on the step's
        stream and an event marks their arrival; returns the AsyncCtrl.  The control block is rewritten by the NEXT step's norm
        pass, which the stream orders behind this copy."""
        dev = self._ctrl_buf.device
        if dev.type != "cuda":
            arr = self.fetch_with_loss(loss)
            h = AsyncCtrl(torch.from_numpy(arr.copy()), None)
            return h
        if self._host_ring is None:
            self._host_ring = [torch.empty(3, dtype=torch.float64).pin_memory() for _ in range(4)]
            self._ring_handles = [None] * len(self._host_ring)
        self._ctrl_buf[2:3].copy_(loss.detach().reshape(1))
        host = self._host_ring[self._ring_i]
        # the slot's previous user (four steps ago) takes its numbers with it before the slot is overwritten: a handle that a caller
        # kept unresolved for more than four steps still returns ITS step's values
        prev = self._ring_handles[self._ring_i]
        if prev is not None:
            prev.get()
        host.copy_(self._ctrl_buf[:3], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._async = h = AsyncCtrl(host, ev)
        self._ring_handles[self._ring_i] = h
        self._ring_i = (self._ring_i + 1) % len(self._host_ring)
        return h

    def read_ctrl(self):
        """(total_norm, found_inf) -- ONE host sync; call after the loss has been fetched anyway."""
        c = self.ctrl.cpu()
        self._resolve_pending(bool(c[1] > 0))
        return float(c[0]) ** 0.5, bool(c[1] > 0)

    def _trainable(self):
        """Parameters in ``torch.optim.SGD(self.network.parameters(), ...)`` order (MH.py:294-301: those that require grad)."""
        return [(n, p) for n, p in self.net._named if p.requires_grad]

    def state_dict(self):
        """``torch.optim.SGD.state_dict()`` layout, so checkpoints interoperate with the reference's trainers
        (upstream ``NetworkTrainer.save_checkpoint`` / ``load_checkpoint_ram``): ``state[i]['momentum_buffer']`` per
        parameter index, one param group.  A parameter the optimiser never stepped (e.g. the zero-weight deep-supervision
        head as long as no penalty term gave it a gradient) has no state entry, as in torch."""
        g = self.param_groups[0]
        named = self._trainable()
        self._resolve_pending()
        state = {}
        for i, (n, p) in enumerate(named):
            s = p._lnn_slot
            mom = self.net.arena.momentum[s.offset:s.offset + s.numel]
            if n not in self._ever_stepped and not bool(mom.any()):      # (a momentum loaded from a checkpoint counts as well)
                continue
            state[i] = {"momentum_buffer": mom.view(s.shape).clone()}
        group = {"lr": g["lr"], "momentum": g["momentum"], "dampening": 0, "weight_decay": g["weight_decay"],
                 "nesterov": True, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(named)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, d):
        # a step that was enqueued but whose found-inf flag nobody fetched belongs to the state being replaced
        self._pending_stepped, self._async = None, None
        if "param_groups" not in d:                      # round-1 private format {"momentum": arena, "lr": float}
            self.net.arena.momentum.copy_(d["momentum"])
            self.param_groups[0]["lr"] = d["lr"]
            return
        named = self._trainable()
        grp = d["param_groups"][0]
        assert len(grp["params"]) == len(named), "optimizer state was saved for a different set of trainable parameters"
        for k in ("lr", "momentum", "weight_decay"):
            self.param_groups[0][k] = grp[k]
        self.net.arena.momentum.zero_()
        self._ever_stepped = set()
        for pos, idx in enumerate(grp["params"]):
            st = d["state"].get(idx)
            if st is None or st.get("momentum_buffer") is None:
                continue
            self._ever_stepped.add(named[pos][0])
            s = named[pos][1]._lnn_slot
            self.net.arena.momentum[s.offset:s.offset + s.numel].copy_(
                st["momentum_buffer"].to(self.net.arena.momentum.device, torch.float32).reshape(-1))


def clip_grad_norm_(net, max_norm, optimizer: FusedSGD, inv_scale=1.0):
    """Deferred ``torch.nn.utils.clip_grad_norm_``: computes the norm now, the scaling is folded into
    the following ``optimizer.step(max_norm=...)``."""
    optimizer.grad_norm_pass(inv_scale)
    return max_norm
