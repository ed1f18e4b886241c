#!/usr/bin/env python
"""Lane-level numpy emulation of csrc/igemm_conv_v9.hip (index arithmetic only, no timing): LDS ring layout and XOR key,
the DMA's lane -> source mapping, B-fragment reads, the three rolling accumulators, the weight-row rotation, the
partial-sum exchange and the v_permlane32_swap epilogue.  Runs on the CPU; compares with torch conv3d.  Used to
debug the kernel's addressing without a GPU:  python tools/v9_emulate.py"""
import itertools
import sys

import numpy as np
import torch
import torch.nn.functional as F


def lane_voxel(v):
    if v < 4: return 0, v
    if v < 12: return 2, v - 4
    if v < 16: return 0, v - 8
    if v < 20: return 3, v - 16
    if v < 28: return 1, v - 20
    return 3, v - 24


class Cfg:
    def __init__(s, NCK, NMB, NF):
        s.NCK, s.NMB, s.NF = NCK, NMB, NF
        s.NG = NCK // 2
        s.NFX = 2 if NF >= 4 else 1
        s.NFY = NF // s.NFX
        s.FY, s.FX = 4 * s.NFY, 8 * s.NFX
        s.PY, s.PX = s.FY + 2, s.FX + 2
        s.PXS = (s.PX + 3) // 4 * 4
        s.GRAW = s.PY * s.PXS * 64
        s.DPW = (s.NG * ((s.GRAW + 1023) // 1024) + 7) // 8
        s.GSLAB = s.DPW * 8 // s.NG * 1024
        s.PLANE = s.NG * s.GSLAB
        s.D = 3 if NCK == 8 else 4
        s.R = s.D + 1
        s.NFIN = min(NCK, 4)
        s.QN = 4 // s.NFIN
        s.EXB = NF * NMB * s.NFIN * (NCK - 1) * s.QN * 1024


def emulate(cfg, x, w, bias, S=1, flip=False, k133=False):
    """x (N,C,D,H,W), w (M,C,3,3,3) -> y (N,M,D,H,W) the way the kernel computes it (fp32 math on fp16-rounded data).
    k133 (the KY = 1, GS instances): w is (M,C,1,3,3); the column walk runs along H, the footprint spans (D, W): the tensors stay in
    their dense NDHWC order and every access goes through the element strides the host sets up (lnn_conv_k133_on_v9)."""
    K = cfg
    N, C, Dz, H, W = x.shape
    M = w.shape[0]
    assert C == 16 * K.NCK and M % (32 * K.NMB) == 0
    xflat = x.permute(0, 2, 3, 4, 1).contiguous().numpy().reshape(-1)            # NDHWC, flat
    yflat = np.full(N * Dz * H * W * M, np.nan, dtype=np.float32)
    if k133:
        # walk = H, footprint rows = D, footprint columns = W: strides of {walk, row, column}, of a sample
        gs_in, gs_in_n = (W * C, H * W * C, C), Dz * H * W * C
        gs_out, gs_out_n = (W * M, H * W * M, M), Dz * H * W * M
        Dz, H = H, Dz                                                              # extents of the walk / footprint-row axes
    else:
        gs_in, gs_in_n = (H * W * C, W * C, C), Dz * H * W * C
        gs_out, gs_out_n = (H * W * M, W * M, M), Dz * H * W * M
    wn = w.numpy()
    tiles_y, tiles_x = -(-H // K.FY), -(-W // K.FX)
    L = -(-Dz // S)
    mgroups = M // (32 * K.NMB)
    lds = np.zeros(K.R * K.PLANE // 2, dtype=np.float32)           # one value per fp16 slot
    exch = np.zeros((2, K.EXB // 4), dtype=np.float32)
    for mg, n, zs, fyb, fxb in itertools.product(range(mgroups), range(N), range(S), range(tiles_y), range(tiles_x)):
        y0, x0 = fyb * K.FY, fxb * K.FX
        zs0, zs1 = zs * L, min(zs * L + L, Dz)
        T = zs1 - zs0 + 2
        waves = []
        for wave in range(8):
            ck, mb, f = wave % K.NCK, (wave // K.NCK) % K.NMB, wave // (K.NCK * K.NMB)
            fxi, fyi, gi = f % K.NFX, f // K.NFX, f * K.NMB + mb
            m0 = 32 * (mg * K.NMB + mb)
            st = dict(ck=ck, mb=mb, f=f, fxi=fxi, fyi=fyi, gi=gi, m0=m0, acc=np.zeros((3, 32, 32), np.float32),
                      own=np.zeros((K.QN * 8, 32), np.float32))
            # A[tap][row rho][k 0..15]: row rho <-> channel m0 + ((rho + 8 QN ck) & 31)
            A = np.zeros((27, 32, 16), np.float32)
            for tl in range(27):
                if k133 and (tl % 9) // 3 != 1:
                    continue                                # 9 resident fragments: the centre row of the in-plane shifts
                slot = 26 - tl if flip else tl
                dz, dy, dx = slot // 9, (slot // 3) % 3, slot % 3
                for rho in range(32):
                    ch = m0 + ((rho + (8 * K.QN * ck if ck < K.NFIN else 0)) & 31)
                    # k133: the walk's plane shift is ky, its column shift kx (panel slot dz * 3 + dx of a 9-tap panel)
                    A[tl, rho] = wn[ch, 16 * ck:16 * ck + 16, 0, dz, dx] if k133 else wn[ch, 16 * ck:16 * ck + 16, dz, dy, dx]
            st["A"] = A
            waves.append(st)

        def dma(tp, slot):
            zi = zs0 - 1 + tp
            zok = tp < T and 0 <= zi < Dz
            for wave in range(8):
                for k in range(K.DPW):
                    j = wave * K.DPW + k
                    gg = (j * 1024) // K.GSLAB
                    for lane in range(64):
                        cc = j * 64 + lane - gg * (K.GSLAB // 16)
                        pos, pc = cc >> 2, cc & 3
                        py, pxs = pos // K.PXS, pos % K.PXS
                        key = ((pxs >> 2) & 1) | ((py & 1) << 1)
                        piece = pc ^ key
                        iy, ix = y0 - 1 + py, x0 - 1 + pxs
                        ok = zok and py < K.PY and pxs < K.PX and 0 <= iy < H and 0 <= ix < W
                        if k133 and not (1 <= py < K.PY - 1):
                            ok = False                      # the footprint's halo rows are never read: not fetched
                        dst = (slot * K.PLANE + j * 1024 + lane * 16) // 2
                        if ok:
                            c0 = 32 * gg + piece * 8
                            src = n * gs_in_n + zi * gs_in[0] + iy * gs_in[1] + ix * gs_in[2] + c0
                            lds[dst:dst + 8] = xflat[src:src + 8]
                        else:
                            lds[dst:dst + 8] = 0.0

        def bfrag(st, slot, i):
            dy, dx = i // 3, i % 3
            B = np.zeros((16, 32), np.float32)
            for lane in range(64):
                hk, v = lane >> 5, lane & 31
                vr, vx = lane_voxel(v)
                py0, px = 4 * st["fyi"] + vr, 8 * st["fxi"] + vx + dx
                par = dy & 1
                key = ((px >> 2) & 1) | (((py0 + par) & 1) << 1)
                ck = st["ck"]
                lb = (ck >> 1) * K.GSLAB + (py0 * K.PXS + px) * 64 + (((((ck & 1) << 1) | hk) ^ key) << 4)
                a = (slot * K.PLANE + lb + dy * K.PXS * 64) // 2
                B[8 * hk:8 * hk + 8, v] = lds[a:a + 8]
            return B

        def finalize(tprev):
            o = zs0 + tprev - 2
            ov = 2 <= tprev < T
            for st in waves:
                ck, gi = st["ck"], st["gi"]
                if ck >= K.NFIN:
                    continue
                fin = st["own"].copy()                      # rows: quad-in-Q * 8 + 4 hk + i  (MFMA rows of quads [0, QN))
                rb = (gi * K.NFIN + ck) * (K.NCK - 1) * K.QN * 1024
                for s_ in range((K.NCK - 1) * K.QN):
                    base = (rb + s_ * 1024) // 4
                    pv = exch[tprev & 1][base:base + 256].reshape(64, 4)
                    for lane in range(64):
                        hk, v = lane >> 5, lane & 31
                        qi = s_ % K.QN
                        fin[qi * 8 + 4 * hk:qi * 8 + 4 * hk + 4, v] += pv[lane]
                if not ov:
                    continue
                for lane in range(64):
                    hk, v = lane >> 5, lane & 31
                    vr, vx = lane_voxel(v)
                    oy, ox = y0 + 4 * st["fyi"] + vr, x0 + 8 * st["fxi"] + vx
                    if oy >= H or ox >= W:
                        continue
                    if K.NCK == 2:
                        # v_permlane32_swap(X = fin quad0 reg i, Y = quad1 reg i): lanes < 32 get (X_lo, X_hi), lanes >= 32 (Y_lo, Y_hi)
                        if hk == 0:
                            lo = fin[0:4, v]; hi = fin[4:8, v]            # quad 0 of lane v (hk 0) and of lane v+32 (hk 1)
                        else:
                            lo = fin[8:12, v]; hi = fin[12:16, v]
                        ch = st["m0"] + 16 * ck + 8 * hk
                        vals = np.concatenate([lo, hi])
                        dsto = n * gs_out_n + o * gs_out[0] + oy * gs_out[1] + ox * gs_out[2] + ch
                        yflat[dsto:dsto + 8] = vals + bias[ch:ch + 8]
                    else:
                        ch = st["m0"] + 8 * ck + 4 * hk
                        dsto = n * gs_out_n + o * gs_out[0] + oy * gs_out[1] + ox * gs_out[2] + ch
                        yflat[dsto:dsto + 4] = fin[4 * hk:4 * hk + 4, v] + bias[ch:ch + 4]

        T3 = (T + 1 + 2) // 3 * 3
        for tp in range(K.D):
            dma(tp, tp % K.R)
        dtp = K.D
        for t in range(T3):
            U = t % 3
            # fin_load/fin_store of the previous step happen before this step's publish (program order)
            finalize(t - 1)
            dma(dtp, dtp % K.R); dtp += 1
            slot = t % K.R
            i0, ni = (3, 3) if k133 else (0, 9)
            for st in waves:
                for j in range(ni):
                    i = i0 + j
                    B = bfrag(st, slot, i)
                    for dz in range(3):
                        a = (U + 1 - dz + 3) % 3
                        if j == 0 and dz == 0:
                            st["acc"][a][:] = 0
                        st["acc"][a] += st["A"][dz * 9 + i] @ B
            # publish
            for st in waves:
                c = (U + 2) % 3
                ck, gi = st["ck"], st["gi"]
                st["own"] = st["acc"][c][:8 * K.QN].copy()
                fin_w = ck < K.NFIN
                for a in range(4):
                    if a < K.QN and fin_w:
                        continue
                    Q = (a + K.QN * ck) & 3 if fin_w else a
                    f_ = Q // K.QN
                    wb = ((gi * K.NFIN + f_) * (K.NCK - 1) + (ck - f_ - 1 + K.NCK) % K.NCK) * K.QN * 1024 + (Q % K.QN) * 1024
                    base = wb // 4
                    blk = exch[t & 1][base:base + 256].reshape(64, 4)
                    for lane in range(64):
                        hk, v = lane >> 5, lane & 31
                        blk[lane] = st["acc"][c][8 * a + 4 * hk:8 * a + 4 * hk + 4, v]
    dd, hh = (H, Dz) if k133 else (Dz, H)              # back to the tensor's own (D, H)
    return torch.from_numpy(yflat.reshape(N, dd, hh, W, M)).permute(0, 4, 1, 2, 3)


def main():
    torch.manual_seed(0)
    cases = [((2, 1, 4), (1, 32, 32, 5, 9, 18)), ((4, 1, 2), (1, 64, 32, 4, 9, 9)), ((2, 2, 2), (1, 32, 64, 5, 6, 10)),
             ((4, 2, 1), (1, 64, 64, 4, 5, 9)), ((2, 1, 4), (1, 32, 96, 7, 8, 16)), ((8, 1, 1), (1, 128, 64, 4, 5, 9))]
    bad = 0
    for cfg, (N, C, M, D, H, W) in cases:
        x = torch.randn(N, C, D, H, W).half().float()
        w = (torch.randn(M, C, 3, 3, 3) * 0.1).half().float()
        b = torch.randn(M)
        ref = F.conv3d(x, w, b, padding=1)
        for S in (1, 2):
            got = emulate(Cfg(*cfg), x, w, b.numpy(), S=S)
            err = float((got - ref).abs().max())
            nan = int(torch.isnan(got).sum())
            print(f"cfg {cfg} {C}->{M} @{D}x{H}x{W} S={S}: max err {err:.2e}  unwritten {nan}")
            bad += err > 1e-3 or nan > 0
    # [1,3,3] convolutions: the walk along H with permuted axes (KY = 1, GS instances), forward and flipped (data gradient) taps
    for cfg, (N, C, M, D, H, W) in [((2, 1, 4), (1, 32, 32, 5, 7, 18)), ((4, 2, 1), (2, 64, 64, 3, 9, 9)), ((8, 1, 1), (1, 128, 32, 6, 5, 10))]:
        x = torch.randn(N, C, D, H, W).half().float()
        w = (torch.randn(M, C, 1, 3, 3) * 0.1).half().float()
        b = torch.randn(M)
        for flip in (False, True):
            wr = w.flip(3, 4) if flip else w
            ref = F.conv3d(x, wr, b, padding=(0, 1, 1))
            got = emulate(Cfg(*cfg), x, w, b.numpy(), S=2, flip=flip, k133=True)
            err = float((got - ref).abs().max())
            nan = int(torch.isnan(got).sum())
            print(f"cfg {cfg} k133 {C}->{M} @{D}x{H}x{W} flip={int(flip)}: max err {err:.2e}  unwritten {nan}")
            bad += err > 1e-3 or nan > 0
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
