"""Multi-GPU sharding of one surface: block-row bands, one per rank, and the gather of the output bands.

The reference already shards the path by rows twice -- 256 Ki-pixel slices (IntelPlugin.cpp:851-879) and one
4-row-aligned band per pool thread (win32Threads.cpp:211-249), each band encoded by an independent
CompressBlocks* call writing at dst + row0*(width/4)*bytes_per_block (win32Threads.cpp:228-230).  Blocks never
interact (kernel.ispc:573-596, 2014-2028, 3118-3130), so the same rule shards across GPUs with no halo and no
data-path collective; the only exchange is the gather of the compressed bands to whoever needs the whole image.

One process per GPU; torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Band geometry comes from the C ABI (itwBandForPart) so host code in any language shards alike.
"""
from . import abi


def band_of(width, height, fmt, rank, world):
    """(first_texel_row, texel_rows, out_byte_offset, out_byte_len) of `rank`'s band."""
    y0, rows, off = abi.band_for_part(width, height, fmt, rank, world)
    if fmt in abi.KEEPS_PARTIAL_BLOCKS:                        # BC4 / BC5: partial last block row and column are encoded
        nbytes = ((rows + 3) // 4) * ((width + 3) // 4) * abi.BYTES_PER_BLOCK[fmt]
    else:
        nbytes = (rows // 4) * (width // 4) * abi.BYTES_PER_BLOCK[fmt]
    return y0, rows, off, nbytes


def encode_band(fmt, surface, settings, rank, world, encode=None):
    """Encode this rank's band of `surface` (the whole (H, W, 4) surface or a view of it).  `encode(fmt, band,
    settings)` defaults to the HIP path (abi.compress); tests inject a CPU encoder."""
    h, w = surface.shape[:2]
    y0, rows, off, nbytes = band_of(w, h, fmt, rank, world)
    band = surface[y0:y0 + rows]
    enc = encode or abi.compress
    out = enc(fmt, band, settings)
    assert out.numel() == nbytes
    return out, off, nbytes


def gather_bands(local, width, height, fmt, group=None):
    """All-gather the per-rank block streams into the whole-image stream on every rank.  Bands may differ by one
    block row when ranks do not divide height/4, so each rank contributes a slot of the maximum band size."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [band_of(width, height, fmt, r, world)[3] for r in range(world)]
    offs = [band_of(width, height, fmt, r, world)[2] for r in range(world)]
    slot = max(sizes)
    if all(s == slot for s in sizes):
        full = torch.empty(slot * world, dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full
    padded = torch.zeros(slot, dtype=torch.uint8, device=local.device)
    padded[:local.numel()] = local
    slots = torch.empty(slot * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(slots, padded, group=group)
    full = torch.empty(sum(sizes), dtype=torch.uint8, device=local.device)
    for r in range(world):
        full[offs[r]:offs[r] + sizes[r]] = slots[r * slot:r * slot + sizes[r]]
    return full


def encode_sharded(fmt, surface, settings=None, group=None, encode=None):
    """Whole-image block stream, computed by all ranks of `group` together.  Every rank passes the same surface
    geometry and holds (at least) its own band of texels."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local, _, _ = encode_band(fmt, surface, settings, rank, world, encode)
    h, w = surface.shape[:2]
    return gather_bands(local, w, h, fmt, group)


def sub_band_of(width, height, fmt, rank, world, k, pieces):
    """Round 5, the content-aware partition (include/itw_multigpu.h): the surface is cut into pieces * world sub-bands and sub-band j belongs
    to rank j % world -- `k`-th piece of `rank` = sub-band k * world + rank.  Same return value as band_of."""
    return band_of(width, height, fmt, k * world + rank, pieces * world)


class InterleavedPipeline:
    """BandPipeline for K interleaved sub-bands per rank: the whole-image buffer is K groups of `world` equal sub-bands; every step this
    rank encodes its K sub-bands in place (group k, slot rank) and each group is all-gathered on its own, in NCCL's in-place form, as soon
    as its sub-band is encoded -- K collectives of 1/K the size, each overlapped with the next sub-band's encode (and, across steps, with
    the next step's, as before).  K = 1 is BandPipeline.

    encode_into[k](out_piece) launches / performs the encode of this rank's k-th sub-band into `out_piece` (a uint8 view)."""

    def __init__(self, piece_bytes, world, rank, device, encode_into, group=None, depth=2):
        import torch
        self.world, self.rank, self.group, self.depth = world, rank, group, depth
        self.encode_into = list(encode_into)
        self.pieces = len(self.encode_into)
        self.full = [torch.empty(self.pieces * world * piece_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.groups = [[f[k * world * piece_bytes:(k + 1) * world * piece_bytes] for k in range(self.pieces)] for f in self.full]
        self.piece = [[g[rank * piece_bytes:(rank + 1) * piece_bytes] for g in gs] for gs in self.groups]
        self.band = [ps[0] for ps in self.piece]          # (K = 1: the rank's band, as BandPipeline exposes it)
        self.work = [[] for _ in range(depth)]
        self.steps = 0

    def step(self):
        """Encode + start the gathers; returns the index of the buffer that will hold this step's whole image."""
        b = self.steps % self.depth
        self.steps += 1
        for w in self.work[b]:                       # the gathers that last used this buffer must be done with it
            w.wait()
        self.work[b] = []
        for k in range(self.pieces):
            self.encode_into[k](self.piece[b][k])
            if self.world > 1:
                import torch.distributed as dist
                self.work[b].append(dist.all_gather_into_tensor(self.groups[b][k], self.piece[b][k], group=self.group, async_op=True))
        return b

    def drain(self):
        """Wait for every gather in flight (stream-level on GPUs: follow with a device synchronize to read on the host)."""
        for b in range(self.depth):
            for w in self.work[b]:
                w.wait()
            self.work[b] = []


class BandPipeline:
    """Steady state of a sharded encoder: every step this rank encodes its band and the bands are all-gathered, with
    the gather of step i overlapped with the encode of step i+1.  Two (depth) whole-image buffers alternate; the
    collective is launched asynchronously (RCCL runs it on its own stream after the encode it depends on) and is only
    waited for when its buffer comes round again, or in drain().  The band is encoded in place at its offset of the
    whole-image buffer, so the all-gather is NCCL's in-place form (send = recv + rank * count) and nothing is copied.

    encode_into(out_band) launches / performs the encode of this rank's band into `out_band` (a uint8 view)."""

    def __init__(self, band_bytes, world, rank, device, encode_into, group=None, depth=2):
        import torch
        self.world, self.rank, self.group, self.depth = world, rank, group, depth
        self.encode_into = encode_into
        self.full = [torch.empty(world * band_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.band = [f[rank * band_bytes:(rank + 1) * band_bytes] for f in self.full]
        self.work = [None] * depth
        self.steps = 0

    def step(self):
        """Encode + start the gather; returns the index of the buffer that will hold this step's whole image."""
        b = self.steps % self.depth
        self.steps += 1
        if self.work[b] is not None:                 # the gather that last used this buffer must be done with it
            self.work[b].wait()
            self.work[b] = None
        self.encode_into(self.band[b])
        if self.world > 1:
            import torch.distributed as dist
            self.work[b] = dist.all_gather_into_tensor(self.full[b], self.band[b], group=self.group, async_op=True)
        return b

    def drain(self):
        """Wait for every gather in flight (stream-level on GPUs: follow with a device synchronize to read on the host)."""
        for b in range(self.depth):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
