"""Run-length form of FindClosestUNORM (BC4BC5.cpp:314-337) as csrc/bc4_bc5.hip stores and evaluates it -- a numpy model for
the tests.  F[r0, r1, v] = index the search picks for texel code v under endpoints (r0, r1)."""
import numpy as np


def runs_table(F):
    """(table uint32 [65536][4], max number of runs).  Entry layout = csrc/bc4_bc5.hip bc45_build_index_table:
    x, y: 256 - start of run j (j = 1..7), one byte each, the number of runs in the top byte of y; z, w: run indices, one byte each
    (z: runs 0..3, w: runs 4..7)."""
    F = F.reshape(65536, 256)
    table = np.zeros((65536, 4), dtype=np.uint32)
    change = np.ones((65536, 256), dtype=bool)
    change[:, 1:] = F[:, 1:] != F[:, :-1]
    nruns = change.sum(axis=1)
    for p in range(65536):
        starts = np.flatnonzero(change[p])
        c = [0, 0]
        order = [0, 0]
        for j, v in enumerate(starts[:8]):
            if j >= 1:
                c[(j - 1) >> 2] |= (256 - int(v)) << (8 * ((j - 1) & 3))
            order[j >> 2] |= int(F[p, v]) << (8 * (j & 3))
        table[p] = (c[0], c[1] | (min(len(starts), 255) << 24), order[0], order[1])
    return table, int(nruns.max())


def evaluate(table):
    """The kernel's evaluation of the table for all 65536 pairs x 256 texel codes, exactly as the kernel does it: two texels per
    dword (add the replicated byte, keep the carry bits 8 and 24, sum), the counts of four texels gathered into selector bytes, the
    run indices looked up bytewise, four 3-bit indices squeezed together."""
    v = np.arange(256, dtype=np.uint32)
    # four "texels" per step: v, 255 - v, v ^ 0x55, v ^ 0xaa -- every code appears in every position
    t4 = [v, 255 - v, v ^ 0x55, v ^ 0xAA]
    pa = (t4[0][None, :] | (t4[1][None, :] << 16)).astype(np.uint32)
    pb = (t4[2][None, :] | (t4[3][None, :] << 16)).astype(np.uint32)
    sa = np.zeros((65536, 256), dtype=np.uint32)
    sb = np.zeros((65536, 256), dtype=np.uint32)
    for j in range(7):
        cj = (table[:, j >> 2] >> (8 * (j & 3))) & 0xff
        cj = (cj | (cj << 16)).astype(np.uint32)
        sa += (pa + cj[:, None]) & np.uint32(0x01000100)
        sb += (pb + cj[:, None]) & np.uint32(0x01000100)
    counts = [(sa >> 8) & 0xff, (sa >> 24) & 0xff, (sb >> 8) & 0xff, (sb >> 24) & 0xff]       # v_perm 0x07050301
    tab = np.concatenate([((table[:, 2][:, None] >> (8 * np.arange(4))) & 0xff), ((table[:, 3][:, None] >> (8 * np.arange(4))) & 0xff)], axis=1)
    idx = [np.take_along_axis(tab, c.astype(np.int64), axis=1).astype(np.uint32) for c in counts]       # v_perm lookup
    word = idx[0] | (idx[1] << 8) | (idx[2] << 16) | (idx[3] << 24)
    two = (word | (word >> 5)) & np.uint32(0x003f003f)
    four = (two | (two >> 10)) & np.uint32(0x0fff)
    out = []
    for k in range(4):
        got = ((four >> (3 * k)) & 7).astype(np.uint8)
        inv = np.argsort(t4[k])                                   # back to "indexed by texel code"
        out.append(got[:, inv])
    return out
