"""Helpers for the -m gpu parity tests: layout conversion between the oracle's NCDHW fp32 tensors and
the library's channels-last fp16 buffers, and thin wrappers that call the C-ABI through ctypes."""
import torch

from lifelong_nnunet_amd import native as nat

DEV = "cuda:0"


def to_cl_h(x_ncdhw, ld=None, offset=0):
    """NCDHW fp32 (cpu) -> NDHWC fp16 (gpu) view inside a buffer with channel stride ``ld``."""
    n, c = x_ncdhw.shape[:2]
    sp = x_ncdhw.shape[2:]
    ld = ld or c
    buf = torch.zeros((n,) + tuple(sp) + (ld,), dtype=torch.float16, device=DEV)
    buf[..., offset:offset + c] = x_ncdhw.permute(0, 2, 3, 4, 1).to(DEV).half()
    return buf, buf[..., offset:]  # second = pointer-offset view (data_ptr at channel `offset`)


def from_cl_h(buf, c, offset=0):
    """NDHWC fp16 gpu buffer -> NCDHW fp32 cpu."""
    return buf[..., offset:offset + c].float().permute(0, 4, 1, 2, 3).contiguous().cpu()


def q16(x):
    """Round to fp16 and back (what the GPU stores)."""
    return x.half().float()


class View:
    """Pointer-offset handle: data_ptr() of a channel-offset view without making it contiguous."""

    def __init__(self, t, elem_off):
        self.t, self.off = t, elem_off

    def data_ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()


def pack(w, ntaps, M, KC, sm, skc, st):
    n = nat.query("lnn_packed_weight_elems", ntaps, M, KC)
    dst = torch.empty(n, dtype=torch.float16, device=DEV)
    nat.call("lnn_pack_weights", w, dst, ntaps, M, KC, sm, skc, st)
    return dst


def pack_conv_fwd(w):      # w (K,C,3,3,3) fp32 gpu
    K, C = w.shape[:2]
    return pack(w, 27, K, C, C * 27, 27, 1)


def pack_conv_dgrad(w):
    K, C = w.shape[:2]
    return pack(w, 27, C, K, 27, C * 27, 1)


def pack_convT_fwd(w):     # w (Cin,Cout,2,2,2)
    C, K = w.shape[:2]
    return pack(w, 8, K, C, 8, K * 8, 1)


def pack_convT_dgrad(w):
    C, K = w.shape[:2]
    return pack(w, 8, C, K, K * 8, 8, 1)


def rel_l2(a, b):
    return float((a - b).double().norm() / (b.double().norm() + 1e-30))


def rel_err(a, b):
    """max of (max-abs error / max-abs reference) and the relative L2 error: the first catches a bulk offset only if it is
    as large as the tolerance times the PEAK, the second lets a handful of wrong border voxels hide in a big tensor --
    a result has to pass both."""
    inf = float((a - b).abs().max() / (b.abs().max() + 1e-12))
    l2 = rel_l2(a, b)
    import os
    log = os.environ.get("LNN_RELERR_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{inf:.3e} {l2:.3e} {tuple(a.shape)}\n")
    return max(inf, l2)
