#!/usr/bin/env python
"""CPU check of the tap tables / parity classes of the generic-geometry kernels (lnn_gen_geometry in csrc/igemm_gen.hip,
restated line by line) before a GPU is involved: the sum  OUT[so*l + par] += W[slot] . IN[si*l + d]  over the table must equal
torch's conv3d / its data gradient / conv_transpose3d / its data gradient, and the weight-gradient form must equal autograd.
    python tools/gen_emulate.py"""
import itertools

import torch
import torch.nn.functional as F


def geometry(kind, k, st):
    kz, ky, kx = k
    classes, taps = [], []
    if kind in (0, 3):
        so, si = (1, 1, 1), tuple(st)
        classes.append(((0, 0, 0), len(taps)))
        for a in range(kz):
            for b in range(ky):
                for c in range(kx):
                    pz, py, px = (kz // 2, ky // 2, kx // 2) if kind == 0 else (0, 0, 0)
                    taps.append((a - pz, b - py, c - px, (a * ky + b) * kx + c))
    else:
        so, si = tuple(st), (1, 1, 1)
        for pz in range(st[0]):
            for py in range(st[1]):
                for px in range(st[2]):
                    classes.append(((pz, py, px), len(taps)))
                    for a in range(kz):
                        for b in range(ky):
                            for c in range(kx):
                                if kind == 2:
                                    if (a, b, c) == (pz, py, px):
                                        taps.append((0, 0, 0, (a * ky + b) * kx + c))
                                    continue
                                nz, ny, nx = pz + kz // 2 - a, py + ky // 2 - b, px + kx // 2 - c
                                # C++ '%' truncates towards zero: a negative numerator that is not a multiple stays non-zero
                                if nz % st[0] or ny % st[1] or nx % st[2]:
                                    continue
                                taps.append((nz // st[0], ny // st[1], nx // st[2], (a * ky + b) * kx + c))
    classes.append((None, len(taps)))
    return so, si, classes, taps


def run(inp, wslots, out_dims, L, so, si, classes, taps):
    """inp (N, Cin, D, H, W); wslots[slot] = (M, Cin) matrix; returns (N, M, *out_dims)."""
    N, _, Di, Hi, Wi = inp.shape
    M = wslots[0].shape[0]
    out = torch.zeros((N, M) + tuple(out_dims), dtype=inp.dtype)
    for ci in range(len(classes) - 1):
        par, t0 = classes[ci]
        t1 = classes[ci + 1][1]
        for lz, ly, lx in itertools.product(range(L[0]), range(L[1]), range(L[2])):
            oz, oy, ox = lz * so[0] + par[0], ly * so[1] + par[1], lx * so[2] + par[2]
            if oz >= out_dims[0] or oy >= out_dims[1] or ox >= out_dims[2]:
                continue
            for dz, dy, dx, slot in taps[t0:t1]:
                iz, iy, ix = lz * si[0] + dz, ly * si[1] + dy, lx * si[2] + dx
                if 0 <= iz < Di and 0 <= iy < Hi and 0 <= ix < Wi:
                    out[:, :, oz, oy, ox] += inp[:, :, iz, iy, ix] @ wslots[slot].T
    return out


def main():
    torch.manual_seed(0)
    dt = torch.float64
    worst = 0.0
    for k in [(3, 3, 3), (1, 3, 3), (3, 1, 3), (1, 1, 1), (3, 3, 1)]:
        for st in [(1, 1, 1), (2, 2, 2), (1, 2, 2), (2, 1, 2), (2, 2, 1)]:
            for dims in [(4, 5, 6), (3, 4, 4)]:
                N, C, K = 2, 3, 4
                x = torch.randn((N, C) + dims, dtype=dt, requires_grad=True)
                w = torch.randn((K, C) + k, dtype=dt, requires_grad=True)
                pad = tuple(kk // 2 for kk in k)
                y = F.conv3d(x, w, None, stride=st, padding=pad)
                dy = torch.randn_like(y)
                y.backward(dy)
                od = tuple(y.shape[2:])
                assert od == tuple((d - 1) // s + 1 for d, s in zip(dims, st))
                ntap = k[0] * k[1] * k[2]
                wf = [w.detach().reshape(K, C, ntap)[:, :, t] for t in range(ntap)]               # forward panel: rows K, contraction C
                wd = [w.detach().reshape(K, C, ntap)[:, :, t].T.contiguous() for t in range(ntap)]  # dgrad panel: rows C, contraction K
                so, si, cl, tp = geometry(0, k, st)
                worst = max(worst, float((run(x.detach(), wf, od, od, so, si, cl, tp) - y.detach()).abs().max()))
                so, si, cl, tp = geometry(1, k, st)
                L = tuple(-(-d // s) for d, s in zip(dims, st))
                worst = max(worst, float((run(dy, wd, dims, L, so, si, cl, tp) - x.grad).abs().max()))
                assert len(tp) == ntap and sorted(t[3] for t in tp) == list(range(ntap))
                # weight gradient: dW[slot][k][c] = sum_l dy[l, k] x[si*l + d, c] with the FORWARD table
                so, si, cl, tp = geometry(0, k, st)
                dw = torch.zeros((ntap, K, C), dtype=dt)
                for lz, ly, lx in itertools.product(*map(range, od)):
                    for dz, dy_, dx, slot in tp:
                        iz, iy, ix = lz * si[0] + dz, ly * si[1] + dy_, lx * si[2] + dx
                        if 0 <= iz < dims[0] and 0 <= iy < dims[1] and 0 <= ix < dims[2]:
                            dw[slot] += dy[:, :, lz, ly, lx].T @ x.detach()[:, :, iz, iy, ix]
                worst = max(worst, float((dw.permute(1, 2, 0).reshape(w.shape) - w.grad).abs().max()))
    for st in [(2, 2, 2), (1, 2, 2), (2, 1, 2), (2, 2, 1), (1, 1, 2)]:
        for dims in [(3, 4, 5), (2, 3, 3)]:
            N, C, K = 2, 3, 4
            x = torch.randn((N, C) + dims, dtype=dt, requires_grad=True)
            w = torch.randn((C, K) + st, dtype=dt, requires_grad=True)
            y = F.conv_transpose3d(x, w, None, stride=st)
            dy = torch.randn_like(y)
            y.backward(dy)
            od = tuple(d * s for d, s in zip(dims, st))
            assert tuple(y.shape[2:]) == od
            ntap = st[0] * st[1] * st[2]
            wf = [w.detach().reshape(C, K, ntap)[:, :, t].T.contiguous() for t in range(ntap)]     # rows K, contraction C
            wd = [w.detach().reshape(C, K, ntap)[:, :, t] for t in range(ntap)]                    # rows C, contraction K
            so, si, cl, tp = geometry(2, st, st)
            worst = max(worst, float((run(x.detach(), wf, od, dims, so, si, cl, tp) - y.detach()).abs().max()))
            so, si, cl, tp = geometry(3, st, st)
            worst = max(worst, float((run(dy, wd, dims, dims, so, si, cl, tp) - x.grad).abs().max()))
            dw = torch.zeros((ntap, C, K), dtype=dt)          # P = x (rows c), Q = dy gathered (cols k)
            for lz, ly, lx in itertools.product(*map(range, dims)):
                for dz, dy_, dx, slot in tp:
                    dw[slot] += x.detach()[:, :, lz, ly, lx].T @ dy[:, :, lz * st[0] + dz, ly * st[1] + dy_, lx * st[2] + dx]
            worst = max(worst, float((dw.permute(1, 2, 0).reshape(w.shape) - w.grad).abs().max()))
    print("worst abs error over all geometries:", worst)
    assert worst < 1e-9


if __name__ == "__main__":
    main()
