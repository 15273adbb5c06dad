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

  bands       fewer frames than ranks (config 5: ONE 8K frame, a 3.19-GB canvas): stitcher(f) = f % world would funnel every tile into one rank and
              leave the others idle.  Then each frame's canvas is cut into `world` ROW BANDS along tile-row boundaries (band r = tile rows
              [r n / world, (r+1) n / world), output rows [S(first), S(last + 1)) with S(i) = first un-blended row of tile row i): a tile goes to the rank
              of its tile row, and only the pad_sc rows of the NEXT band's first tile row that are blended into this band's last rows travel twice
              (as strips: ~1 % of the bytes).  Every rank folds its own band (moe_stitch_band) and keeps it: the canvas stays sharded.

  wire        the tiles cross as fp32 (default), or -- wire='f16s' -- as [fp16 image | fp32 seam rows | fp32 seam columns] records (moe_wire_pack /
              moe_wire_unpack, include/moephoto_amd.h): the fold needs a tile's exact value only where a blend reads it, so an fp16 canvas comes out
              bit-identical at 0.5 + 0.5 x (seam share) of the bytes on the links.  Pack and unpack are one HBM pass each around the collective.

Nothing is copied around the collective (band mode: but the strips; wire='f16s': the pack / unpack passes).  `TileExchange` lays ONE device buffer out once per plan,

    [ own | recv from rank 0 | recv from rank 1 | ... | send to rank 0 | send to rank 1 | ... ]

and hands the engine a (frame, tile) -> offset table (moe_run_plan_tiles): the net writes every tile it owns
straight into its slot of the send region (or of `own` when this rank is also the frame's stitcher), the
all-to-all moves send -> recv inside that buffer, and moe_stitch reads each frame's tiles through a second
offset table -- wherever in `own`/`recv` they landed.  Segment lists, split sizes and both tables are plain
index arithmetic, computed in __init__ and reused every step; they run unchanged under gloo on CPU
(tests/test_dist_cpu.py).
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class TileExchange(object):
    """Layout of the exchange buffer for `n_frames` frames of one tile plan on `world` ranks.

    tile_elems[k] = fp32 elements of tile k's result (all C planes)."""

    def __init__(self, tile_elems, n_frames, rank, world, group=None, bands=None, wire=None):
        """bands: None (frame mode) or (step_w, rows, tile_dims, pad_sc) -- tile columns per tile row, the plan's row table (TilePlan.rows), per tile its
        (C, height, width) in output pixels, the blend length -- for the band-sharded layout.
        wire: None (fp32 tiles on the links) or (tile_dims, seams, pad_sc) -- per tile (C, height, width) and TilePlan.seams() -- for the 'f16s' records."""
        self.sizes = [int(v) for v in tile_elems]
        self.n_tiles, self.n_frames = len(self.sizes), int(n_frames)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.bands = bands is not None
        self.wire = wire is not None
        self._send_items, self._recv_items = [], []      # (peer, offset in the buffer, tile, is_strip) in wire order
        if self.bands:
            self._init_bands(*bands)
        else:
            self._init_frames()
        if self.wire:
            self._init_wire(*wire)

    def _init_frames(self):
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
                    self._recv_items.append((src, pos, k, False))
                    pos += self.sizes[k]
            self.recv_split.append(pos - n0)
        self.recv_elems = pos - self.own_elems
        self.send_split = []
        for dst in range(W):                                  # send: computed here, stitched by dst
            n0 = pos
            if dst != r:
                for f, k in self._segments(r, dst):
                    self.tile_dst[f, k] = pos
                    self._send_items.append((dst, pos, k, False))
                    pos += self.sizes[k]
            self.send_split.append(pos - n0)
        self.send_elems = pos - self.own_elems - self.recv_elems
        self.total_elems = pos
        for f in self.mine:
            assert (self.stitch_off[f] >= 0).all()
        self._tile_dst_c = (ctypes.c_int64 * self.tile_dst.size)(*self.tile_dst.reshape(-1).tolist())
        self._stitch_c = {f: (ctypes.c_int64 * nt)(*v.tolist()) for f, v in self.stitch_off.items()}
        self._stitch_dev = None        # one device blob [len(mine)][n_tiles] of the tables above, uploaded on first use (stitch_tables)

    # ---- band mode -------------------------------------------------------------------------------------
    def band_rows(self, rank):
        """tile rows [i0, i1) of rank's band"""
        return (rank * self.step_h) // self.world, ((rank + 1) * self.step_h) // self.world

    def _init_bands(self, step_w, rows, tile_dims, pad_sc):
        r, W, nt = self.rank, self.world, self.n_tiles
        self.step_w, self.step_h, self.rows_tab, self.pad_sc = int(step_w), nt // int(step_w), [tuple(int(v) for v in t) for t in rows], int(pad_sc)
        self.dims = [tuple(int(v) for v in d) for d in tile_dims]
        strip_elems = [d[0] * self.pad_sc * d[2] for d in self.dims]
        i0, i1 = self.band_rows(r)
        self.my_rows = (i0, i1)
        self.mine = list(range(self.n_frames)) if i1 > i0 else []       # every rank with a non-empty band folds that band of EVERY frame

        def wants(dst):
            """(tile, is_strip) pairs of one frame that rank dst folds, in wire order"""
            a, b = self.band_rows(dst)
            if b <= a:
                return []
            out = [(k, False) for k in range(a * self.step_w, b * self.step_w)]
            if b < self.step_h:
                out += [(k, True) for k in range(b * self.step_w, (b + 1) * self.step_w)]
            return out
        self._wants = wants
        whole_dst = lambda k: next(d for d in range(W) if self.band_rows(d)[0] <= k // self.step_w < self.band_rows(d)[1])
        self.tile_dst = np.full((self.n_frames, nt), -1, np.int64)
        self.stitch_off = {f: np.full(nt, 0, np.int64) for f in self.mine}      # (entries of tile rows outside the band are never read: moe_stitch_band)
        self.strip_copies = []               # (src offset, C, th, tw, first row of the strip inside the tile, dst offset): done between compute and exchange
        pos = 0
        strip_row0 = lambda k: self.rows_tab[k // self.step_w][0] - self.rows_tab[k // self.step_w][2]       # first written row - row of the tile's row 0
        # own region: tiles computed here and folded here (whole), then strips computed here and folded here
        for f in range(self.n_frames):
            for k, is_strip in wants(r):
                if self.owner(f, k) != r:
                    continue
                if not is_strip:
                    self.tile_dst[f, k] = self.stitch_off[f][k] = pos
                    pos += self.sizes[k]
        own_strips = []
        for f in range(self.n_frames):
            for k, is_strip in wants(r):
                if is_strip and self.owner(f, k) == r:
                    self.stitch_off[f][k] = pos
                    own_strips.append((f, k, pos))
                    pos += strip_elems[k]
        self.own_elems = pos
        self.recv_split = []
        for src in range(W):
            n0 = pos
            if src != r:
                for f in range(self.n_frames):
                    for k, is_strip in wants(r):
                        if self.owner(f, k) == src:
                            self.stitch_off[f][k] = pos
                            self._recv_items.append((src, pos, k, is_strip))
                            pos += strip_elems[k] if is_strip else self.sizes[k]
            self.recv_split.append(pos - n0)
        self.recv_elems = pos - self.own_elems
        self.send_split = []
        send_strips = []
        for dst in range(W):
            n0 = pos
            if dst != r:
                for f in range(self.n_frames):
                    for k, is_strip in wants(dst):
                        if self.owner(f, k) != r:
                            continue
                        self._send_items.append((dst, pos, k, is_strip))
                        if is_strip:
                            send_strips.append((f, k, pos))
                            pos += strip_elems[k]
                        else:
                            self.tile_dst[f, k] = pos
                            pos += self.sizes[k]
            self.send_split.append(pos - n0)
        self.send_elems = pos - self.own_elems - self.recv_elems
        # a tile this rank computed whose WHOLE destination is nobody's band cannot exist (every tile row belongs to one band); one whose whole copy
        # goes elsewhere but whose strip is wanted: its whole copy sits in the send region, the strip is cut from there
        for f, k, at in own_strips + send_strips:
            src = int(self.tile_dst[f, k])
            assert src >= 0
            C, th, tw = self.dims[k]
            self.strip_copies.append((src, C, th, tw, strip_row0(k), at))
        self.total_elems = pos
        assert all(self.tile_dst[f, k] >= 0 for f in range(self.n_frames) for k in range(nt) if self.owner(f, k) == r)
        self._tile_dst_c = (ctypes.c_int64 * self.tile_dst.size)(*self.tile_dst.reshape(-1).tolist())
        self._stitch_c = {f: (ctypes.c_int64 * nt)(*v.tolist()) for f, v in self.stitch_off.items()}
        self._stitch_dev = None

    def cut_strips(self, buf):
        """band mode, between compute and exchange: the pad_sc rows of a tile that the band above blends into its last rows, copied next to the whole tile"""
        for src, C, th, tw, r0, at in self.strip_copies:
            buf[at:at + C * self.pad_sc * tw].view(C, self.pad_sc, tw).copy_(buf[src:src + C * th * tw].view(C, th, tw)[:, r0:r0 + self.pad_sc])

    # ---- wire format -----------------------------------------------------------------------------------
    def _init_wire(self, tile_dims, seams, pad_sc):
        """records (moe_wire_rec) of everything this rank sends and receives, and the split sizes of the collective in 4-byte words"""
        dims = [tuple(int(v) for v in d) for d in tile_dims]

        def table(items):
            recs = np.zeros(len(items), WIRE_REC)
            split, pos = [0] * self.world, 0
            for n, (peer, off, k, is_strip) in enumerate(items):
                C, th, tw = dims[k]
                if is_strip:
                    r = (off, pos, C, int(pad_sc), tw, 0, int(pad_sc), int(pad_sc), int(pad_sc), 0, 0, 0, 0, 0)
                else:
                    r = (off, pos, C, th, tw) + tuple(int(v) for v in seams[k]) + (0,)
                recs[n] = r
                w = wire_words(r)
                pos += w
                split[peer] += w
            return recs, split, pos
        self.send_recs, self.wire_send_split, self.wire_send_words = table(self._send_items)
        self.recv_recs, self.wire_recv_split, self.wire_recv_words = table(self._recv_items)
        self._wire_dev = {}          # exchange buffer -> (send words, recv words, send records, recv records) on its device

    def _wire_state(self, buf):
        key = (buf.data_ptr(), str(buf.device))
        st = self._wire_dev.get(key)
        if st is None:
            dev = buf.device
            up = lambda recs: torch.from_numpy(recs.view(np.uint8).reshape(-1).copy() if len(recs) else np.zeros(0, np.uint8)).to(dev)
            st = self._wire_dev[key] = (torch.empty(self.wire_send_words, dtype=torch.int32, device=dev), torch.empty(self.wire_recv_words, dtype=torch.int32, device=dev),
                                        up(self.send_recs), up(self.recv_recs))
        return st

    def _codec(self, pack, buf, words, recs_np, recs_dev):
        if not len(recs_np):
            return
        if not buf.is_cuda:
            if CPU_CODEC is None:
                raise _lib.EngineError('the wire format packs on the GPU (moe_wire_pack); CPU tensors only under the tests\' codec')
            return CPU_CODEC(pack, buf, words, recs_np)
        L = _lib.lib()
        big = int(max(int(r['C']) * int(r['th']) * int(r['tw']) for r in recs_np))
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        fn = L.moe_wire_pack if pack else L.moe_wire_unpack
        _lib.check(fn(buf.data_ptr(), words.data_ptr(), recs_dev.data_ptr(), len(recs_np), big, stream))

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
        for v in [self.n_tiles, self.n_frames, self.world, int(self.bands) + 2 * int(self.wire)] + self.sizes:
            h = ((h ^ int(v)) * 1099511628211) % (1 << 61)
        return h

    # ---- the collective --------------------------------------------------------------------------------
    def exchange(self, buf, async_op=False):
        """buf: 1-D fp32 tensor of total_elems holding this rank's tiles at tile_dst.  After the call the `own` + `recv`
        regions hold every tile of the frames this rank stitches (read them through stitch_off).
        async_op: returns a handle whose wait() makes the CURRENT stream wait for the collective -- the all-to-all then runs on the backend's own
        stream behind the work enqueued so far, and kernels enqueued before wait() (the next group's convolutions) overlap it."""
        if self.bands:
            self.cut_strips(buf)
        if self.world == 1 and not FORCE_COLLECTIVE:
            return _Done() if async_op else self.mine
        a, b = self.own_elems, self.own_elems + self.recv_elems
        recv, send = buf[a:b], buf[b:b + self.send_elems]
        rsplit, ssplit = self.recv_split, self.send_split
        after = None
        if self.wire:
            wsend, wrecv, srecs, rrecs = self._wire_state(buf)
            self._codec(True, buf, wsend, self.send_recs, srecs)
            recv, send, rsplit, ssplit = wrecv, wsend, self.wire_recv_split, self.wire_send_split
            after = lambda: self._codec(False, buf, wrecv, self.recv_recs, rrecs)
        if send.is_cuda and dist.get_backend(self.group) == 'gloo':
            # test mode (several ranks sharing one GPU, MOE_DIST_BACKEND=gloo): stage through the host
            recv_h = torch.empty(recv.numel(), dtype=send.dtype)
            _all_to_all(recv_h, send.cpu(), rsplit, ssplit, self.group, False)
            recv.copy_(recv_h)
            if after:
                after()
            return _Done() if async_op else self.mine
        work = _all_to_all(recv, send, rsplit, ssplit, self.group, async_op)
        if async_op:
            return _Then(work, after) if after else work
        if after:
            after()
        return self.mine


# ---- the exchange primitive ------------------------------------------------------------------------------------------------------------------------------------
# SURVEY 8(e) names grouped ncclSend / ncclRecv for the gather of tile results; the default here is ONE all_to_all_single (RCCL builds the same grouped send / recv
# underneath).  No run of this code has seen more than one rank on RCCL (one GPU per box in the builder's reach: the first multi-rank RCCL run is the driver's), so the
# collective has a second form with the same buffer layout -- one isend + one irecv per peer, batched (batch_isend_irecv = ncclGroupStart / ncclGroupEnd) -- that is taken
# when EXCHANGE_MODE says so (MOE_DIST_EXCHANGE=p2p) or when the all-to-all raises; bench.py reports which form ran (`exchange_mode`) and why (`EXCHANGE_FALLBACK`).
EXCHANGE_MODE = os.environ.get('MOE_DIST_EXCHANGE', 'all_to_all')
EXCHANGE_FALLBACK = None


class _Works(object):
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True

    def is_completed(self):
        return all(w.is_completed() for w in self.works)


def _p2p_exchange(recv, send, rsplit, ssplit, group, async_op):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops, ro, so = [], 0, 0
    roff, soff = [], []
    for r in range(world):
        roff.append(ro); soff.append(so)
        ro += int(rsplit[r]); so += int(ssplit[r])
    for r in range(world):
        peer = r if group is None else dist.get_global_rank(group, r)
        if r == rank:
            if rsplit[r]:
                recv[roff[r]:roff[r] + int(rsplit[r])].copy_(send[soff[r]:soff[r] + int(ssplit[r])])
            continue
        if rsplit[r]:
            ops.append(dist.P2POp(dist.irecv, recv[roff[r]:roff[r] + int(rsplit[r])], peer, group))
        if ssplit[r]:
            ops.append(dist.P2POp(dist.isend, send[soff[r]:soff[r] + int(ssplit[r])], peer, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    w = _Works(works)
    if async_op:
        return w
    w.wait()
    return None


def _all_to_all(recv, send, rsplit, ssplit, group, async_op):
    global EXCHANGE_MODE, EXCHANGE_FALLBACK
    if EXCHANGE_MODE != 'p2p':
        try:
            return dist.all_to_all_single(recv, send, rsplit, ssplit, group=group, async_op=async_op)
        except RuntimeError as e:       # (a communicator error at the first multi-rank contact: keep the job alive on the other form and say so)
            EXCHANGE_MODE, EXCHANGE_FALLBACK = 'p2p', 'all_to_all_single raised: {}'.format(str(e).splitlines()[0][:200])
    return _p2p_exchange(recv, send, rsplit, ssplit, group, async_op)


class _Then(object):
    """an asynchronous collective followed by work on the waiting stream (the unpack pass of the wire format)"""
    def __init__(self, work, after):
        self.work, self.after = work, after

    def wait(self):
        r = self.work.wait()
        self.after()
        return r


class _Done(object):
    def wait(self):
        return True


CPU_CODEC = None             # tests: codec(pack, buf, words, records) on CPU tensors (tests/wire_codec.py); the product packs with moe_wire_pack on the GPU
WIRE_REC = np.dtype([('tile_off', '<i8'), ('wire_off', '<i8'), ('C', '<i4'), ('th', '<i4'), ('tw', '<i4'), ('ra0', '<i4'), ('ra1', '<i4'), ('rb0', '<i4'), ('rb1', '<i4'),
                     ('ca0', '<i4'), ('ca1', '<i4'), ('cb0', '<i4'), ('cb1', '<i4'), ('reserved', '<i4')])       # = moe_wire_rec (include/moephoto_amd.h)


def wire_words(rec):
    """4-byte words of one record (= moe_wire_words): fp16 image of all values + fp32 seam rows + fp32 seam columns; a record that is all seam rows is fp32 alone"""
    _, _, C, th, tw, ra0, ra1, rb0, rb1, ca0, ca1, cb0, cb1 = [int(v) for v in tuple(rec)[:13]]
    nR, nC, n = (ra1 - ra0) + (rb1 - rb0), (ca1 - ca0) + (cb1 - cb0), C * th * tw
    return n if nR >= th else (n + 1) // 2 + C * nR * tw + C * th * nC


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


def agree_arithmetic(model, group=None):
    """Every rank runs the arithmetic rank 0 settled on.  moe_net_finalize(MOE_PREC_AUTO) calibrates per process (moe_net_calibrate: the count of split-operand ARSBs, or the
    exact mode); the measurement is deterministic for equal weights and kernels, but ranks that disagreed -- another driver, another device generation in one job -- would
    compute the tiles of ONE frame with different bits.  Rank 0's resolved precision and block count are broadcast and imposed (set_precision / set_exact_blocks);
    returns (precision, blocks).  Called once per finalized model by run_frames / run_frames_overlapped (a collective: every rank calls it at the same place)."""
    key = (id(group), getattr(model, '_finalized_key', None))
    if getattr(model, '_dist_agreed', None) == key:
        return model._dist_agreed_value
    mine = [model.resolved_precision(), model.exact_blocks()]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        box = [mine if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        want = box[0]
        if want[0] != mine[0]:
            model.set_precision(want[0])
        if want[0] == 'mixed' and model.exact_blocks() != want[1]:
            model.set_exact_blocks(want[1])
        mine = [model.resolved_precision(), model.exact_blocks()]
    model._dist_agreed, model._dist_agreed_value = (id(group), getattr(model, '_finalized_key', None)), tuple(mine)
    return tuple(mine)


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


def run_frames_overlapped(opt, frames, group=None, out_dtype=None, max_tiles_per_batch=0, wire=None, probe=None):
    """run_frames for a BATCH of frames (config 4: 64 of them), in groups of `world` frames with the exchange of a group overlapping the convolutions of the
    next: per group every rank computes its share of the group's tiles into one of TWO exchange buffers, starts the all-to-all asynchronously (it runs on
    the backend's stream), enqueues the next group's convolutions, and only then lets its stream wait for the previous group's exchange and folds the frame
    it stitches.  The result is the same dict as run_frames(opt, frames): same layout, same arithmetic per group.
    probe: a list (tests): per group whose exchange ran behind another group's convolutions, one dict {'group', 'exchange_done_while_computing'} -- the host polls the
    collective's handle after the NEXT group's kernels are enqueued and notes whether it completed while an event recorded behind those kernels was still pending, i.e.
    whether transfer and compute really overlapped in time (an asynchronous backend only: RCCL; under gloo the exchange is staged through the host and blocks)."""
    from .imageProcess import _DT
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if len(frames) <= world:
        return run_frames(opt, frames, group, out_dtype, max_tiles_per_batch, bands=False, wire=wire)
    model = opt.modelCached
    x0 = frames[0]
    dev = x0.device
    C = x0.shape[0]
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    odt = out_dtype if out_dtype is not None else x0.dtype
    out = {}
    pending = None           # (handle, ex, buf, first frame index of the group)

    def finish(p):
        work, ex, buf, f0 = p
        work.wait()
        tables = ex.stitch_tables(dev)
        for f in ex.mine:
            y = torch.empty((C, ex._plan.outH, ex._plan.outW), dtype=odt, device=dev)
            _lib.check(L.moe_stitch_dev(ex._plan._h, dev.index or 0, buf.data_ptr(), ctypes.c_void_p(tables[f]), C, y.data_ptr(), _DT[odt], stream))
            out[f0 + f] = y
    for gi, f0 in enumerate(range(0, len(frames), world)):
        grp = frames[f0:f0 + world]
        plan, ex, bufs, padded, fstride = _layout(opt, grp, group, dev, rank, world, C, False, nbuf=2, wire=wire)
        ex._plan = plan
        buf = bufs[gi & 1]
        sC, sH, sW = padded[0].stride()
        _lib.check(L.moe_run_plan_tiles(model._h, plan._h, padded[0].data_ptr(), _DT[padded[0].dtype], int(fstride), sC, sH, sW,
                                        len(grp), ctypes.c_void_p(buf.data_ptr()), ex._tile_dst_c, int(max_tiles_per_batch), stream))
        for p_ in padded:
            p_.record_stream(torch.cuda.current_stream(dev))
        work = ex.exchange(buf, async_op=True)
        if pending is not None:
            if probe is not None:
                inner = getattr(pending[0], 'work', pending[0])
                if hasattr(inner, 'is_completed'):
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))      # behind THIS group's convolutions (and the start of its exchange)
                    seen = None
                    while seen is None:
                        if inner.is_completed():
                            seen = not ev.query()                    # the previous group's exchange is done; are this group's kernels still running?
                        elif ev.query():
                            seen = False                             # the kernels finished first: the exchange did not hide behind them
                    probe.append({'group': gi - 1, 'exchange_done_while_computing': bool(seen)})
            finish(pending)      # (its buffer is the OTHER one; the convolutions above are already enqueued in front of this wait)
        pending = (work, ex, buf, f0)
    finish(pending)
    return out


def _layout(opt, frames, group, dev, rank, world, C, bands, nbuf=1, wire=None):
    """(plan, exchange layout, its buffer(s), the padded frames, their stride): cached on the Option per (shape, frame count, rank, world, C, mode)."""
    x0 = frames[0]
    if hasattr(opt.modelCached, 'exact_blocks'):
        agree_arithmetic(opt.modelCached, group)
    plan = _agreed_plan(opt, x0.shape, group, dev)
    cache = opt.__dict__.setdefault('_exchanges', {})
    if wire not in (None, 'f32', 'f16s'):
        raise ValueError("wire: None | 'f32' | 'f16s'")
    wire = wire == 'f16s'
    ck = (tuple(int(v) for v in x0.shape[-3:]), len(frames), rank, world, C, bands, wire)
    ent = cache.get(ck)
    if ent is not None and ent[0] is not plan:               # the entry owns its plan: a layout is only ever used with the plan it was built from
        ent = None
    if ent is None:
        off = plan.tile_offsets(C) + [plan.pool_elems(C)]
        dims = [(C, (t[1] - t[0]) * plan.sc, (t[3] - t[2]) * plan.sc) for t in plan.tiles]
        bspec = (plan.stepW, plan.rows, dims, plan.padSc) if bands else None
        wspec = (dims, plan.seams(), plan.padSc) if wire else None
        ex = TileExchange([off[k + 1] - off[k] for k in range(plan.n_tiles)], len(frames), rank, world, group, bands=bspec, wire=wspec)
        if world > 1:       # every rank must have derived the same layout
            sig = torch.tensor([ex.signature()], dtype=torch.int64, device=dev if dist.get_backend(group) != 'gloo' else None)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if int(lo.item()) != int(hi.item()):
                raise RuntimeError('run_frames: the ranks derived different tile plans (pass an explicit cropsize)')
        ent = cache[ck] = (plan, ex, [])
        while len(cache) > 4:
            cache.pop(next(iter(cache)))
    _, ex, bufs = ent
    while len(bufs) < nbuf:
        bufs.append(torch.empty(ex.total_elems, dtype=torch.float32, device=dev))
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
    return plan, ex, bufs, padded, fstride


def run_frame_bands(opt, frames, group=None, out_dtype=None, max_tiles_per_batch=0, wire=None):
    """run_frames in BAND mode: {frame index: (first output row, band tensor (C, rows, sc*W))} for EVERY frame -- this rank's row band of each canvas, which stays
    sharded over the ranks (gather_bands concatenates them).  The form for jobs with fewer frames than ranks (BASELINE config 5: one 8K frame, a 3.19-GB canvas):
    every rank folds its own band instead of one rank folding everything.  A rank whose band is empty (more ranks than tile rows) gets (0, a (C, 0, sc*W) tensor)."""
    return run_frames(opt, frames, group, out_dtype, max_tiles_per_batch, bands=True, wire=wire)


def run_frames(opt, frames, group=None, out_dtype=None, max_tiles_per_batch=0, bands=False, wire=None):
    """Tile-parallel doCrop over a list of equally-shaped (C,H,W) frames that every rank holds
    (broadcast them first).  Returns {frame index: stitched (C, sc*H, sc*W) tensor} for the frames
    this rank stitches (frame f is folded by rank f mod N; a rank that stitches nothing gets {}: use out.get(f)).  bands=True (opt-in: the return type differs -- see
    run_frame_bands) returns {frame index: (first output row, band tensor (C, rows, sc*W))} for every frame instead.
    wire: None / 'f32' -- fp32 tiles on the links; 'f16s' -- fp16 + fp32 seams (module docstring): for fp16 (or narrower) canvases, where the result is
    bit-identical; an fp32 canvas would carry fp16-rounded values outside the seams."""
    from .imageProcess import _DT
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    model = opt.modelCached
    x0 = frames[0]
    dev = x0.device
    C = x0.shape[0]
    bands = bool(bands)
    plan, ex, bufs, padded, fstride = _layout(opt, frames, group, dev, rank, world, C, bands, wire=wire)
    buf = bufs[0]
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    sC, sH, sW = padded[0].stride()
    _lib.check(L.moe_run_plan_tiles(model._h, plan._h, padded[0].data_ptr(), _DT[padded[0].dtype], int(fstride), sC, sH, sW,
                                    len(frames), ctypes.c_void_p(buf.data_ptr()), ex._tile_dst_c, int(max_tiles_per_batch), stream))
    for p in padded:
        p.record_stream(torch.cuda.current_stream(dev))
    mine = ex.exchange(buf)
    out = {}
    odt = out_dtype if out_dtype is not None else x0.dtype
    tables = ex.stitch_tables(dev)
    if ex.bands:
        i0, i1 = ex.my_rows
        y0 = plan.rows[i0][1] if i1 > i0 else 0
        y1 = (plan.rows[i1][1] if i1 < plan.stepH else plan.outH) if i1 > i0 else 0
        for f in mine:
            y = torch.empty((C, y1 - y0, plan.outW), dtype=odt, device=dev)
            _lib.check(L.moe_stitch_band(plan._h, dev.index or 0, buf.data_ptr(), ctypes.c_void_p(tables[f]), C, y.data_ptr(), _DT[odt], i0, i1, 1, stream))
            out[f] = (y0, y)
        if i1 <= i0:         # more ranks than tile rows: this rank's band is empty -- every frame still has an entry (a (C, 0, W) tensor on this rank's device)
            for f in range(len(frames)):
                out[f] = (0, torch.empty((C, 0, plan.outW), dtype=odt, device=dev))
        return out
    for f in mine:
        y = torch.empty((C, plan.outH, plan.outW), dtype=odt, device=dev)
        _lib.check(L.moe_stitch_dev(plan._h, dev.index or 0, buf.data_ptr(), ctypes.c_void_p(tables[f]), C, y.data_ptr(), _DT[odt], stream))
        out[f] = y
    return out


def gather_bands(band, group=None):
    """(first row, band tensor) of every rank -> the whole canvas on every rank (tests, or a caller that wants the image in one place after all;
    the sharded bands are the product: a 32K canvas is 3.19 GB)."""
    world = dist.get_world_size(group)
    y0, y = band if band is not None else (0, None)
    if y is not None and y.shape[-2] == 0:
        y = None                                             # an empty band (more ranks than tile rows) takes no part as a source
    metas = [None] * world
    dist.all_gather_object(metas, None if y is None else (int(y0), tuple(y.shape), str(y.dtype)), group=group)
    # receive buffers live where this rank's collectives run: on its HIP device for RCCL (a CPU tensor in an RCCL broadcast raises on this rank while the others
    # sit in the collective), on the host for gloo -- also on a rank that holds no band of its own
    if y is not None:
        rdev = y.device
    elif dist.get_backend(group) != 'gloo' and torch.cuda.is_available():
        rdev = torch.device('cuda', torch.cuda.current_device())
    else:
        rdev = None
    parts = []
    for r, m in enumerate(metas):
        if m is None:
            continue
        t = y if r == dist.get_rank(group) else torch.empty(m[1], dtype=getattr(torch, m[2].split('.')[-1]), device=rdev)
        if dist.get_backend(group) == 'gloo' and t.is_cuda:
            h = t.cpu()
            dist.broadcast(h, src=r, group=group)
            t = h.to(t.device)
        else:
            dist.broadcast(t, src=r, group=group)
        parts.append((m[0], t))
    parts.sort(key=lambda p: p[0])
    return torch.cat([p[1] for p in parts], dim=-2)
