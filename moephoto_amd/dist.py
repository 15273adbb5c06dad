"""Tile-parallel execution across the GPUs of one node (one process per GPU, torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm, "gloo" drives the CPU tests).

The reference is single-device (no distributed code at all), so this layer is new.
doCrop's tiles are independent (inference; the SE pooling of SEDN/lite is per tile), which gives an
embarrassingly parallel decomposition with ONE exchange step:

  ownership   (frame f, tile k) flattened round-robin:  owner = (f * n_tiles + k) % world
  stitcher    frame f is folded by rank f % world  -> with `world` frames in flight every rank computes
              n_tiles tiles and stitches exactly one frame (weak scaling, balanced)
  exchange    one all_to_all_single of packed fp32 tile results: every rank sends the tiles it computed
              to the frame's stitcher (point-to-point traffic, all 7 xGMI links of a GPU busy at once;
              a ring collective would be bound by one link)
  weights     broadcast once from rank 0 (flattened state dict)

Everything in `TileExchange` is index arithmetic + collectives on plain tensors, so it runs unchanged
under gloo on CPU (tests/test_dist_cpu.py); `run_frames` adds the engine calls.
"""
import ctypes
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import _lib


class TileExchange(object):
    def __init__(self, n_tiles, tile_off, pool_elems, rank, world, group=None):
        self.n_tiles, self.rank, self.world, self.group = int(n_tiles), int(rank), int(world), group
        self.off = [int(v) for v in tile_off] + [int(pool_elems)]
        self.pool_elems = int(pool_elems)

    def owner(self, frame, tile):
        return (frame * self.n_tiles + tile) % self.world

    def stitcher(self, frame):
        return frame % self.world

    def shard_of(self, frame, rank=None):
        """(shard_index, shard_count) such that tile k belongs to `rank` iff k % count == index."""
        r = self.rank if rank is None else rank
        return (r - frame * self.n_tiles) % self.world, self.world

    def tiles_of(self, frame, rank):
        idx, cnt = self.shard_of(frame, rank)
        return [k for k in range(self.n_tiles) if k % cnt == idx]

    def _segments(self, src, dst, frames):
        """(frame, tile) pairs computed by `src` whose stitcher is `dst`, in wire order."""
        return [(f, k) for f in frames if self.stitcher(f) == dst for k in self.tiles_of(f, src)]

    def exchange(self, pools):
        """pools: {frame: 1-D fp32 tensor of pool_elems} holding this rank's tiles.  After the call the
        pools of the frames this rank stitches are complete.  Returns those frames."""
        frames = sorted(pools.keys())
        mine = [f for f in frames if self.stitcher(f) == self.rank]
        if self.world == 1:
            return mine
        any_pool = pools[frames[0]]
        send_parts, send_split, recv_split = [], [], []
        for dst in range(self.world):
            seg = [] if dst == self.rank else self._segments(self.rank, dst, frames)
            send_split.append(sum(self.off[k + 1] - self.off[k] for _, k in seg))
            send_parts += [pools[f][self.off[k]:self.off[k + 1]] for f, k in seg]
        for src in range(self.world):
            seg = [] if src == self.rank else self._segments(src, self.rank, frames)
            recv_split.append(sum(self.off[k + 1] - self.off[k] for _, k in seg))
        send = torch.cat(send_parts) if send_parts else any_pool.new_empty(0)
        if send.is_cuda and dist.get_backend(self.group) == 'gloo':
            # test mode (several ranks sharing one GPU, MOE_DIST_BACKEND=gloo): stage through the host
            recv_h = torch.empty(sum(recv_split), dtype=send.dtype)
            dist.all_to_all_single(recv_h, send.cpu(), recv_split, send_split, group=self.group)
            recv = recv_h.to(send.device)
        else:
            recv = any_pool.new_empty(sum(recv_split))
            dist.all_to_all_single(recv, send, recv_split, send_split, group=self.group)
        pos = 0
        for src in range(self.world):
            if src == self.rank:
                continue
            for f, k in self._segments(src, self.rank, frames):
                n = self.off[k + 1] - self.off[k]
                pools[f][self.off[k]:self.off[k + 1]] = recv[pos:pos + n]
                pos += n
        return mine


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


def run_frames(opt, frames, group=None, out_dtype=None, max_tiles_per_batch=0):
    """Tile-parallel doCrop over a list of equally-shaped (C,H,W) frames that every rank holds
    (broadcast them first).  Returns {frame index: stitched (C, sc*H, sc*W) tensor} for the frames
    this rank stitches."""
    from .imageProcess import _plan_for, _DT
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    model = opt.modelCached
    x0 = frames[0]
    plan = _plan_for(opt, x0.shape)
    C = x0.shape[0]
    ex = TileExchange(plan.n_tiles, plan.tile_offsets(C), plan.pool_elems(C), rank, world, group)
    L = _lib.lib()
    dev = x0.device
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
    all_pools = torch.empty((len(frames), ex.pool_elems), dtype=torch.float32, device=dev)
    pools = {f: all_pools[f] for f in range(len(frames))}
    sC, sH, sW = padded[0].stride()
    _lib.check(L.moe_run_plan_frames(model._h, plan._h, padded[0].data_ptr(), _DT[padded[0].dtype], int(fstride), sC, sH, sW,
                                     len(frames), ctypes.c_void_p(all_pools.data_ptr()), int(ex.pool_elems), rank, world,
                                     int(max_tiles_per_batch), stream))
    mine = ex.exchange(pools)
    out = {}
    odt = out_dtype if out_dtype is not None else x0.dtype
    for f in mine:
        y = torch.empty((C, plan.outH, plan.outW), dtype=odt, device=dev)
        _lib.check(L.moe_stitch(plan._h, dev.index or 0, pools[f].data_ptr(), None, C, y.data_ptr(), _DT[odt], stream))
        out[f] = y
    return out
