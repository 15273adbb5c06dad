"""Tile-parallel execution across the GPUs of one node (one process per GPU, torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm, "gloo" drives the CPU tests).

The reference is single-device (no distributed code at all), so this layer is new.
doCrop's tiles are independent (inference; the SE pooling of SEDN/lite is per tile), which gives an
embarrassingly parallel decomposition with ONE exchange step:

  ownership   (frame f, tile k) flattened round-robin:  owner = (f * n_tiles + k) % world
  stitcher    frame f is folded by rank f % world  -> with `world` frames in flight every rank computes
              n_tiles tiles and stitches exactly one frame (weak scaling, balanced)
  exchange    one all_to_all_single of fp32 tile results: every rank sends the tiles it computed to the
              frame's stitcher (point-to-point traffic, all 7 xGMI links of a GPU busy at once; a ring
              collective would be bound by one link)
  weights     broadcast once from rank 0 (flattened state dict)

Nothing is copied around the collective.  `TileExchange` lays ONE device buffer out once per plan,

    [ own | recv from rank 0 | recv from rank 1 | ... | send to rank 0 | send to rank 1 | ... ]

and hands the engine a (frame, tile) -> offset table (moe_run_plan_tiles): the net writes every tile it owns
straight into its slot of the send region (or of `own` when this rank is also the frame's stitcher), the
all-to-all moves send -> recv inside that buffer, and moe_stitch reads each frame's tiles through a second
offset table -- wherever in `own`/`recv` they landed.  Segment lists, split sizes and both tables are plain
index arithmetic, computed in __init__ and reused every step; they run unchanged under gloo on CPU
(tests/test_dist_cpu.py).
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class TileExchange(object):
    """Layout of the exchange buffer for `n_frames` frames of one tile plan on `world` ranks.

    tile_elems[k] = fp32 elements of tile k's result (all C planes)."""

    def __init__(self, tile_elems, n_frames, rank, world, group=None):
        self.sizes = [int(v) for v in tile_elems]
        self.n_tiles, self.n_frames = len(self.sizes), int(n_frames)
        self.rank, self.world, self.group = int(rank), int(world), group
        r, W, nt = self.rank, self.world, self.n_tiles
        self.mine = [f for f in range(self.n_frames) if self.stitcher(f) == r]
        # where this rank WRITES its tiles (engine table) and where it READS the tiles of the frames it stitches
        self.tile_dst = np.full((self.n_frames, nt), -1, np.int64)
        self.stitch_off = {f: np.full(nt, -1, np.int64) for f in self.mine}
        pos = 0
        for f in self.mine:                                   # own: computed here, stitched here
            for k in self.tiles_of(f, r):
                self.tile_dst[f, k] = self.stitch_off[f][k] = pos
                pos += self.sizes[k]
        self.own_elems = pos
        self.recv_split = []
        for src in range(W):                                  # recv: computed by src, stitched here (wire order = _segments)
            n0 = pos
            if src != r:
                for f, k in self._segments(src, r):
                    self.stitch_off[f][k] = pos
                    pos += self.sizes[k]
            self.recv_split.append(pos - n0)
        self.recv_elems = pos - self.own_elems
        self.send_split = []
        for dst in range(W):                                  # send: computed here, stitched by dst
            n0 = pos
            if dst != r:
                for f, k in self._segments(r, dst):
                    self.tile_dst[f, k] = pos
                    pos += self.sizes[k]
            self.send_split.append(pos - n0)
        self.send_elems = pos - self.own_elems - self.recv_elems
        self.total_elems = pos
        for f in self.mine:
            assert (self.stitch_off[f] >= 0).all()
        self._tile_dst_c = (ctypes.c_int64 * self.tile_dst.size)(*self.tile_dst.reshape(-1).tolist())
        self._stitch_c = {f: (ctypes.c_int64 * nt)(*v.tolist()) for f, v in self.stitch_off.items()}
        self._stitch_dev = None        # one device blob [len(mine)][n_tiles] of the tables above, uploaded on first use (stitch_tables)

    def stitch_tables(self, device):
        """{frame: device pointer of its n_tiles int64 stitch offsets}: ONE upload per layout, whatever the number of frames
        this rank stitches (moe_stitch_dev reads the table in place)."""
        if self._stitch_dev is None or self._stitch_dev[0].device != device:
            blob = torch.from_numpy(np.stack([self.stitch_off[f] for f in self.mine]) if self.mine else np.zeros((1, self.n_tiles), np.int64)).to(device)
            self._stitch_dev = (blob, {f: blob[i].data_ptr() for i, f in enumerate(self.mine)})
        return self._stitch_dev[1]

    # ---- index arithmetic ----------------------------------------------------------------------------
    def owner(self, frame, tile):
        return (frame * self.n_tiles + tile) % self.world

    def stitcher(self, frame):
        return frame % self.world

    def tiles_of(self, frame, rank):
        return [k for k in range(self.n_tiles) if self.owner(frame, k) == rank]

    def _segments(self, src, dst):
        """(frame, tile) pairs computed by `src` whose stitcher is `dst`, in wire order."""
        return [(f, k) for f in range(self.n_frames) if self.stitcher(f) == dst for k in self.tiles_of(f, src)]

    def signature(self):
        """Order-sensitive digest of everything the ranks must agree on (checked once per layout in run_frames)."""
        h = 1469598103934665603
        for v in [self.n_tiles, self.n_frames, self.world] + self.sizes:
            h = ((h ^ int(v)) * 1099511628211) % (1 << 61)
        return h

    # ---- the collective --------------------------------------------------------------------------------
    def exchange(self, buf):
        """buf: 1-D fp32 tensor of total_elems holding this rank's tiles at tile_dst.  After the call the `own` + `recv`
        regions hold every tile of the frames this rank stitches (read them through stitch_off)."""
        if self.world == 1 and not FORCE_COLLECTIVE:
            return self.mine
        a, b = self.own_elems, self.own_elems + self.recv_elems
        recv, send = buf[a:b], buf[b:b + self.send_elems]
        if send.is_cuda and dist.get_backend(self.group) == 'gloo':
            # test mode (several ranks sharing one GPU, MOE_DIST_BACKEND=gloo): stage through the host
            recv_h = torch.empty(self.recv_elems, dtype=send.dtype)
            dist.all_to_all_single(recv_h, send.cpu(), self.recv_split, self.send_split, group=self.group)
            recv.copy_(recv_h)
        else:
            dist.all_to_all_single(recv, send, self.recv_split, self.send_split, group=self.group)
        return self.mine


FORCE_COLLECTIVE = False      # tests: issue the all-to-all even on a world of one rank (exercises the RCCL path on a 1-GPU box)


def broadcast_state_dict(sd, src=0, device=None, group=None):
    """Rank `src` passes its state dict (name -> tensor); every rank returns an equal OrderedDict."""
    rank = dist.get_rank(group)
    meta = [[(k, tuple(v.shape)) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    total = sum(int(torch.Size(s).numel()) for _, s in meta[0])
    if dist.get_backend(group) == 'gloo':
        device = None
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        flat.copy_(torch.cat([v.reshape(-1).float() for v in sd.values()]))
    dist.broadcast(flat, src=src, group=group)
    out, pos = OrderedDict(), 0
    flat = flat.cpu()
    for k, s in meta[0]:
        n = int(torch.Size(s).numel())
        out[k] = flat[pos:pos + n].reshape(s).clone()
        pos += n
    return out


DIST_PLAN_CACHE = 8


def _agreed_plan(opt, shape, group, device):
    """One tile plan for all ranks.  With an explicit cropsize the plan is a pure function of the shape; with cropsize
    'auto' it depends on free memory, which differs between ranks -- the MINIMUM over the ranks is used so that every rank
    derives the same grid (different grids would desynchronise the all-to-all split sizes)."""
    from .config import config
    from .imageProcess import EngineModule, prepare
    key = ('dist',) + tuple(int(v) for v in shape[-3:])
    plans = opt.__dict__.setdefault('_dist_plans', {})      # kept apart from doCrop's LRU of plans (imageProcess._plan_for): an evicted plan
    plan = plans.pop(key, None)                             # on ONE rank would leave that rank alone in the collectives below
    if plan is not None:
        plans[key] = plan                                   # (re-inserted last = most recently used)
        return plan
    # bounded like PLAN_CACHE: every rank runs the same sequence of shapes (SPMD), so every rank evicts the same entry at the same call and the
    # re-planning collective below stays matched; a plan owns its device tables, varied frame shapes must not accumulate them
    while len(plans) >= DIST_PLAN_CACHE:
        plans.pop(next(iter(plans)))
    free = config.calcFreeMem()
    if dist.get_world_size(group) > 1:
        t = torch.tensor([float(free)], dtype=torch.float64, device=device if dist.get_backend(group) != 'gloo' else None)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        free = float(t.item())
    model = opt.modelCached
    if isinstance(model, EngineModule):
        free = min(free, model.max_tile_pixels() * key[1] * key[1] / opt.ramCoef)
    it = prepare(key[1:], free, opt, opt.padding, opt.scale, opt.align, opt.cropsize)[0]
    plans[key] = it.plan
    return it.plan


def run_frames(opt, frames, group=None, out_dtype=None, max_tiles_per_batch=0):
    """Tile-parallel doCrop over a list of equally-shaped (C,H,W) frames that every rank holds
    (broadcast them first).  Returns {frame index: stitched (C, sc*H, sc*W) tensor} for the frames
    this rank stitches."""
    from .imageProcess import _DT
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    model = opt.modelCached
    x0 = frames[0]
    dev = x0.device
    plan = _agreed_plan(opt, x0.shape, group, dev)
    C = x0.shape[0]
    cache = opt.__dict__.setdefault('_exchanges', {})
    ck = (tuple(int(v) for v in x0.shape[-3:]), len(frames), rank, world, C)
    ent = cache.get(ck)
    if ent is not None and ent[0] is not plan:               # the entry owns its plan: a layout is only ever used with the plan it was built from
        ent = None
    if ent is None:
        off = plan.tile_offsets(C) + [plan.pool_elems(C)]
        ex = TileExchange([off[k + 1] - off[k] for k in range(plan.n_tiles)], len(frames), rank, world, group)
        if world > 1:       # every rank must have derived the same layout
            sig = torch.tensor([ex.signature()], dtype=torch.int64, device=dev if dist.get_backend(group) != 'gloo' else None)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if int(lo.item()) != int(hi.item()):
                raise RuntimeError('run_frames: the ranks derived different tile plans (pass an explicit cropsize)')
        buf = torch.empty(ex.total_elems, dtype=torch.float32, device=dev)
        ent = cache[ck] = (plan, ex, buf)
        while len(cache) > 4:
            cache.pop(next(iter(cache)))
    _, ex, buf = ent
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    padded = [plan.padImage(x) for x in frames]
    # one launch set for ALL frames: same-shaped tiles of different frames share batches (a rank owns only n_tiles/world tiles
    # of each frame -- run frame by frame they would go out as many small, inefficient launches)
    esz = padded[0].element_size()
    gap = [(p.data_ptr() - padded[0].data_ptr()) // esz for p in padded]
    fstride = gap[1] if len(padded) > 1 else 0
    same = all(p.stride() == padded[0].stride() and p.dtype == padded[0].dtype for p in padded)
    if not (same and all(g == f * fstride for f, g in enumerate(gap)) and (len(padded) == 1 or fstride > 0)):
        stacked = torch.stack(padded)            # frames are not slices of one tensor: gather them once
        padded = list(stacked.unbind(0))
        fstride = stacked.stride(0)
    sC, sH, sW = padded[0].stride()
    _lib.check(L.moe_run_plan_tiles(model._h, plan._h, padded[0].data_ptr(), _DT[padded[0].dtype], int(fstride), sC, sH, sW,
                                    len(frames), ctypes.c_void_p(buf.data_ptr()), ex._tile_dst_c, int(max_tiles_per_batch), stream))
    for p in padded:
        p.record_stream(torch.cuda.current_stream(dev))
    mine = ex.exchange(buf)
    out = {}
    odt = out_dtype if out_dtype is not None else x0.dtype
    tables = ex.stitch_tables(dev)
    for f in mine:
        y = torch.empty((C, plan.outH, plan.outW), dtype=odt, device=dev)
        _lib.check(L.moe_stitch_dev(plan._h, dev.index or 0, buf.data_ptr(), ctypes.c_void_p(tables[f]), C, y.data_ptr(), _DT[odt], stream))
        out[f] = y
    return out
