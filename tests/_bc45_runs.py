"""Run-length form of FindClosestUNORM (BC4BC5.cpp:314-337) as csrc/bc4_bc5.hip stores and evaluates it -- a numpy model for
the tests.  F[r0, r1, v] = index the search picks for texel code v under endpoints (r0, r1)."""
import numpy as np


def runs_table(F):
    """(table uint32 [65536][4], max number of runs).  Entry layout = csrc/bc4_bc5.hip bc45_build_index_table:
    x, y: 256 - start of run j (j = 1..7), one byte each; z: run indices, one nibble each; w: number of runs."""
    F = F.reshape(65536, 256)
    table = np.zeros((65536, 4), dtype=np.uint32)
    change = np.ones((65536, 256), dtype=bool)
    change[:, 1:] = F[:, 1:] != F[:, :-1]
    nruns = change.sum(axis=1)
    for p in range(65536):
        starts = np.flatnonzero(change[p])
        c = [0, 0]
        order = 0
        for j, v in enumerate(starts[:8]):
            if j >= 1:
                c[(j - 1) >> 2] |= (256 - int(v)) << (8 * ((j - 1) & 3))
            order |= int(F[p, v]) << (4 * j)
        table[p] = (c[0], c[1], order, len(starts))
    return table, int(nruns.max())


def evaluate(table):
    """The kernel's evaluation of the table for all 65536 pairs x 256 texel codes, two texels per dword exactly as the kernel
    does it (add the replicated byte, keep the carry bits 8 and 24, sum, shift the run-index word by 4 x count)."""
    v = np.arange(256, dtype=np.uint32)
    pairs = (v[None, :] | (v[::-1][None, :] << 16)).astype(np.uint32)            # texel v in the low half, 255 - v in the high half
    s = np.zeros((65536, 256), dtype=np.uint32)
    for j in range(7):
        cj = (table[:, j >> 2] >> (8 * (j & 3))) & 0xff
        cj = (cj | (cj << 16)).astype(np.uint32)
        s += (pairs + cj[:, None]) & np.uint32(0x01000100)
    order = table[:, 2][:, None]
    lo = (order >> ((s >> 6) & 0x1c)) & 7
    hi = (order >> ((s >> 22) & 0x1c)) & 7
    return lo.astype(np.uint8), hi[:, ::-1].astype(np.uint8)                      # both indexed by texel code
