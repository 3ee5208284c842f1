"""numpy model of fft_tile.hpp's per-thread index algebra (test infrastructure).

Mirrors the device code formula by formula -- LDS addressing, butterfly->(j, c) mapping, stage
twiddle index, Stockham autosort position, inter-pass twiddle, the three store modes -- vectorised
over the threads of a work-group, so that indexing mistakes are caught on CPU before a GPU run.
"""
import numpy as np


def row_pitch(L):
    return L + (L // 16 if L >= 16 else 1)


def lds_addr(L, W, ROW, idx, c):
    if ROW:
        return c * row_pitch(L) + idx + (idx >> 4)
    return idx * W + c


def tile_kernel(inp, out, L, W, NT, ROW, TR, radices, total, logMS, logS, ostride_in, ostride_out,
                tw_lo=None, tw_hi=None, tw_shift=0, has_tw=False, inverse=False, scale=1.0, tile=0):
    """One work-group.  inp/out are flat complex128 arrays."""
    P = L * W
    PPT = P // NT
    assert PPT * NT == P and PPT % 2 == 0
    if ROW:
        lds_n = W * row_pitch(L)
    else:
        lds_n = max(P, W * (L + 1)) if TR else P
    lds = np.full(lds_n, np.nan + 0j)
    tid = np.arange(NT)
    col0 = tile * W
    MSmask = (1 << logMS) - 1
    twL = np.exp(-2j * np.pi * np.arange(L) / L)
    # load
    for it in range(PPT // 2):
        e = (it * NT + tid) * 2
        for d in (0, 1):
            if ROW:
                c, r = e // L, e % L + d
                rr = col0 + c
                valid = rr < total
                g = rr * ostride_in + r
                a = lds_addr(L, W, ROW, r, c)
            else:
                r, c = e // W, e % W + d
                cc = col0 + c
                valid = cc < total
                o, rem = cc >> logMS, cc & MSmask
                g = o * ostride_in + (r << logMS) + rem
                a = lds_addr(L, W, ROW, r, c)
            vals = np.where(valid, inp[np.where(valid, g, 0)], 0)
            if inverse:
                vals = np.conj(vals)
            lds[a] = vals
    # stages
    Ns = 1
    for si, R in enumerate(radices):
        NB, LR = PPT // R, L // R
        last = si == len(radices) - 1
        regs = {}
        for b in range(NB):
            bid = b * NT + tid
            if ROW:
                j, c = bid % LR, bid // LR
            else:
                c, j = bid % W, bid // W
            v = np.stack([lds[lds_addr(L, W, ROW, j + k * LR, c)] for k in range(R)])  # [R, NT]
            if Ns > 1:
                ai = (j & (Ns - 1)) * (L // (Ns * R))
                for k in range(1, R):
                    v[k] = v[k] * twL[k * ai]
            F = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R)
            v = F @ v
            regs[b] = (j, c, v)
        lds[:] = np.nan  # all reads happened before the barrier
        for b in range(NB):
            j, c, v = regs[b]
            idxD = (j & ~(Ns - 1)) * R + (j & (Ns - 1))
            for k in range(R):
                idx = idxD + k * Ns
                val = v[k]
                if last and not ROW and has_tw:
                    rem = (col0 + c) & MSmask
                    l = rem >> logS
                    ee = l * idx
                    val = val * tw_lo[ee & ((1 << tw_shift) - 1)] * tw_hi[ee >> tw_shift]
                if last and TR:
                    lds[c * (L + 1) + idx] = val
                else:
                    lds[lds_addr(L, W, ROW, idx, c)] = val
        Ns *= R
    # store
    for it in range(PPT // 2):
        e = (it * NT + tid) * 2
        for d in (0, 1):
            if ROW:
                c, r = e // L, e % L + d
                rr = col0 + c
                valid = rr < total
                g = rr * ostride_out + r
                val = lds[lds_addr(L, W, ROW, r, c)]
            elif TR:
                c, q = e // L, e % L + d
                cc = col0 + c
                valid = cc < total
                o, rem = cc >> logMS, cc & MSmask
                g = o * ostride_out + rem * L + q
                val = lds[c * (L + 1) + q]
            else:
                q, c = e // W, e % W + d
                cc = col0 + c
                valid = cc < total
                o, rem = cc >> logMS, cc & MSmask
                l, jp = rem >> logS, rem & ((1 << logS) - 1)
                g = o * ostride_out + (((l * L) + q) << logS) + jp
                val = lds[lds_addr(L, W, ROW, q, c)]
            val = val * scale
            if inverse:
                val = np.conj(val)
            out[g[valid]] = val[valid]


def run_pass(inp, kind_row, L, M, S, outer, ostride, W, NT, radices, inverse=False, scale=1.0):
    out = np.full(inp.shape, np.nan + 0j)
    if kind_row:
        total, logMS, logS = outer, 0, 0
        TR = False
        tw = {}
    else:
        total = outer * M * S
        logMS, logS = int(np.log2(M * S)), int(np.log2(S))
        TR = (S == 1)
        tw = {}
        if M > 1:
            n = L * M
            sh = (int(np.log2(n)) + 1) // 2
            tw = dict(tw_lo=np.exp(-2j * np.pi * np.arange(1 << sh) / n),
                      tw_hi=np.exp(-2j * np.pi * (np.arange(n >> sh) << sh) / n), tw_shift=sh, has_tw=True)
    ntiles = -(-total // W)
    for t in range(ntiles):
        tile_kernel(inp, out, L, W, NT, kind_row, TR, radices, total, logMS, logS, ostride, ostride,
                    inverse=inverse, scale=scale, tile=t, **tw)
    return out
