#!/usr/bin/env python
"""Lane-level CPU emulation of csrc/igemm_conv_mt.hip's ADDRESSING (no GPU needed): the halo image the direct-to-LDS loads build
(position order, 16-byte half swizzle, zero fill outside the volume), the B-fragment read addresses per (sub-tile, tap, column
tile), the weight-fragment source rows (rotation by 8 kq), the accumulator-quad ownership of the tap quarters and the output
addresses.  Every formula is the kernel's, restated with numpy over the 64 lanes; the result is compared with F.conv3d.

    python tools/mt_emulate.py            # a few small shapes, forward and data gradient, cat input, split-K parts
"""
import itertools
import sys

import numpy as np
import torch
import torch.nn.functional as F

NW, NH, NIT = 8, 4, 7
HALO = NW * NH * 1024


def geometry(Lh, Lw):
    """(efficiency, WN, TY, TX) as csrc/igemm_conv_mt.hip:mt_geometry."""
    best = None
    prev = None
    for nx in range(1, 17):
        tx = -(-Lw // nx)
        if tx == prev:
            continue
        prev = tx
        for wn in (5, 4):
            ty = min(Lh, (32 * wn) // tx)
            while ty >= 1 and 4 * (ty + 2) * (tx + 2) > NW * NH * 32:
                ty -= 1
            if ty < 1:
                continue
            e = Lh * Lw / (-(-Lh // ty) * -(-Lw // tx) * 32 * wn)
            if best is None or e > best[0] + 1e-9:
                best = (e, wn, ty, tx)
    return best


def panel(w_mct, Mpad, KCpad):
    """w_mct: (M, C, 27) -> blocked panel [Mpad/32][KCpad/16][27][32][16] (igemm_common.h:lnn_panel_off)."""
    M, C, T = w_mct.shape
    p = np.zeros((Mpad // 32, KCpad // 16, T, 32, 16), np.float32)
    for m in range(M):
        for c in range(C):
            p[m // 32, c // 16, :, m % 32, c % 16] = w_mct[m, c]
    return p.reshape(-1)


def emulate(x, x2, csplit, wp, flip, M, ksplit=1):
    """x: (N, D, H, W, ld) float (channels-last); wp: flat panel; returns y (N, D, H, W, M) fp32 (sum over parts)."""
    N, D, H, W, ld = x.shape
    C = wp_C[0]
    Mpad = -(-M // 32) * 32
    KCpad = -(-C // 16) * 16
    _, WN, TY, TX = geometry(H, W)
    PY, PX = TY + 2, TX + 2
    PYX = PY * PX
    P = 4 * PYX
    mblk = -(-Mpad // 64)
    zb, yb, xb = -(-D // 2), -(-H // TY), -(-W // TX)
    nchunks = C // 16
    cpp = nchunks // ksplit
    kc16 = KCpad // 16
    rb_stride = kc16 * 27 * 512
    y = np.zeros((N, D, H, W, M), np.float64)
    lanes = np.arange(64)
    hk, v = lanes >> 5, lanes & 31
    xs = [x.reshape(N, -1), None if x2 is None else x2.reshape(N, -1)]
    for part, mb, n, zbi, ybi, xbi in itertools.product(range(ksplit), range(mblk), range(N), range(zb), range(yb), range(xb)):
        z0, y0, x0, m0 = zbi * 2, ybi * TY, xbi * TX, mb * 64
        acc = np.zeros((NW, 2, WN, 32, 32))          # [wave][rb][ct][mfma row][col]
        for c in range(cpp):
            chunk = part * cpp + c
            c0 = chunk * 16
            part2 = c0 >= csplit
            src = xs[1 if part2 else 0][n]
            coff = c0 - csplit if part2 else c0
            # ---- halo image: lds[byte // 2] halves
            lds = np.zeros(HALO // 2)
            for wave in range(NW):
                for k in range(NH):
                    pos = (wave * NH + k) * 32 + (lanes >> 1)
                    pz, rem = pos // PYX, pos % PYX
                    py, px = rem // PX, rem % PX
                    iz, iy, ix = z0 - 1 + pz, y0 - 1 + py, x0 - 1 + px
                    ok = (pos < P) & (iz >= 0) & (iz < D) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
                    half = (lanes & 1) ^ ((pos >> 3) & 1)
                    voff = (((iz * H + iy) * W + ix) * ld + half * 8)          # halves
                    base = ((wave * NH + k) * 1024 + lanes * 16) // 2
                    for l in range(64):
                        if ok[l]:
                            lds[base[l]:base[l] + 8] = src[coff + voff[l]: coff + voff[l] + 8]
            for wave in range(NW):
                sub, kq = wave >> 2, wave & 3
                nv = TY * TX
                for it in range(NIT):
                    tap = kq * NIT + it
                    if tap >= 27:
                        continue
                    slot = 26 - tap if flip else tap
                    dz, dy, dx = tap // 9, (tap // 3) % 3, tap % 3
                    toffb = (((dz - 1) * PY + (dy - 1)) * PX + (dx - 1)) * 32
                    for rb in range(2):
                        if m0 + rb * 32 >= Mpad:
                            continue
                        f0 = (m0 >> 5) * rb_stride + (chunk * 27 + slot) * 512 + rb * rb_stride
                        avoff = (((lanes & 31) + 8 * kq) & 31) * 32 + hk * 16          # bytes
                        A = np.zeros((32, 16))
                        for l in range(64):
                            A[l & 31, 8 * hk[l]: 8 * hk[l] + 8] = wp[f0 + avoff[l] // 2: f0 + avoff[l] // 2 + 8]
                        for ct in range(WN):
                            vv = ct * 32 + v
                            vc = np.minimum(vv, nv - 1)
                            yy, xx = vc // TX, vc % TX
                            lb = (((sub + 1) * PY + (yy + 1)) * PX + (xx + 1)) * 32 + hk * 16
                            bb = lb + toffb
                            addr = bb ^ ((bb >> 4) & 16)
                            B = np.zeros((16, 32))
                            for l in range(64):
                                B[8 * hk[l]: 8 * hk[l] + 8, v[l]] = lds[addr[l] // 2: addr[l] // 2 + 8]
                            acc[wave, rb, ct] += A @ B
        # ---- reduction: mfma row rho of wave kq holds channel 32 rb + ((rho + 8 kq) & 31)
        for sub in range(2):
            for rb in range(2):
                for ct in range(WN):
                    tot = np.zeros((32, 32))          # [channel in block][col]
                    for kq in range(4):
                        ch = (np.arange(32) + 8 * kq) & 31
                        tot[ch] += acc[sub * 4 + kq, rb, ct]
                    for col in range(32):
                        vv = ct * 32 + col
                        if vv >= TY * TX:
                            continue
                        yy, xx = vv // TX, vv % TX
                        oz, oy, ox = z0 + sub, y0 + yy, x0 + xx
                        if oz >= D or oy >= H or ox >= W:
                            continue
                        for chn in range(32):
                            m = m0 + rb * 32 + chn
                            if m < M:
                                y[n, oz, oy, ox, m] += tot[chn, col]
    return y


wp_C = [0]


def check(N, C, K, D, H, W, dgrad=False, cat=False, ksplit=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C, D, H, W), generator=g)
    w = torch.randn((K, C, 3, 3, 3), generator=g) * 0.1
    if not dgrad:
        ref = F.conv3d(x, w, None, padding=1)
        wm = w.reshape(K, C, 27).numpy()                      # fwd panel: rows = K, contraction = C
        M, Cc, src = K, C, x
    else:
        dy = torch.randn((N, K, D, H, W), generator=g)
        ref = F.conv_transpose3d(dy, w, None, padding=1)      # = data gradient of the stride-1 convolution
        wm = w.reshape(K, C, 27).permute(1, 0, 2).numpy()     # dgrad panel: rows = C, contraction = K
        M, Cc, src = C, K, dy
    wp_C[0] = Cc
    Mpad, KCpad = -(-M // 32) * 32, -(-Cc // 16) * 16
    wp = panel(wm, Mpad, KCpad)
    xcl = src.permute(0, 2, 3, 4, 1).contiguous().numpy()
    if cat:
        ca = Cc // 2
        xa, xb = np.ascontiguousarray(xcl[..., :ca]), np.ascontiguousarray(xcl[..., ca:])
        got = emulate(xa, xb, ca, wp, dgrad, M, ksplit)
    else:
        got = emulate(xcl, None, 1 << 30, wp, dgrad, M, ksplit)
    refcl = ref.permute(0, 2, 3, 4, 1).numpy()
    err = np.abs(got - refcl).max() / np.abs(refcl).max()
    print(f"N={N} C={C} K={K} {D}x{H}x{W} dgrad={int(dgrad)} cat={int(cat)} ksplit={ksplit} geometry={geometry(H, W)}: rel err {err:.2e}")
    return err


if __name__ == "__main__":
    errs = [check(1, 16, 64, 3, 5, 7),
            check(1, 32, 96, 2, 9, 10, ksplit=2),
            check(2, 32, 32, 5, 12, 10, dgrad=True),
            check(1, 32, 64, 4, 8, 20, cat=True),
            check(1, 16, 40, 3, 24, 20),
            check(1, 16, 32, 2, 10, 64),                    # x bands: 64 columns -> 4 x 16
            check(1, 16, 32, 3, 7, 45, dgrad=True)]         # ragged x bands
    sys.exit(0 if max(errs) < 1e-5 else 1)
