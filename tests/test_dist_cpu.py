"""World-size-2 (and 3) CPU runs of the tile-parallel exchange (moephoto_amd/dist.py) under gloo:
ownership, the all-to-all of packed tile results, weight broadcast.  No engine calls."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ['MOE_ROOT']); sys.path.insert(0, os.path.join(os.environ['MOE_ROOT'], 'tests'))
import numpy as np, torch, torch.distributed as dist
from collections import OrderedDict
from moephoto_amd.dist import TileExchange, broadcast_state_dict
from moephoto_amd import dist as mdist_mod
from oracle import planner as oplanner, stitch as ostitch
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
shape, sc, pad, crop, frames = (2, 70, 90), 2, 5, 40, 5
pl = oplanner.prepare(shape, 1 << 40, 1e-3, pad, sc, 8, crop)
C = shape[0]
sizes = [C * (t[1] - t[0]) * (t[3] - t[2]) * sc * sc for t in pl.tiles]
nt = len(pl.tiles)
ex = TileExchange(sizes, frames, rank, world)
# every (frame, tile) has exactly one owner; the engine table marks exactly the owned tiles; regions do not overlap
for f in range(frames):
    owners = [[r for r in range(world) if k in ex.tiles_of(f, r)] for k in range(nt)]
    assert all(len(o) == 1 and o[0] == ex.owner(f, k) for k, o in enumerate(owners))
    assert [k for k in range(nt) if ex.tile_dst[f, k] >= 0] == ex.tiles_of(f, rank)
spans = sorted((int(ex.tile_dst[f, k]), int(ex.tile_dst[f, k]) + sizes[k]) for f in range(frames) for k in range(nt) if ex.tile_dst[f, k] >= 0)
assert all(a1 >= b0 for (_, b0), (a1, _) in zip(spans, spans[1:])) and spans[-1][1] <= ex.total_elems
assert ex.own_elems + ex.recv_elems + ex.send_elems == ex.total_elems
sigs = [None] * world
dist.all_gather_object(sigs, ex.signature())
assert len(set(sigs)) == 1
def tile_value(f, k):
    t = pl.tiles[k]
    return np.random.default_rng(1000 * f + k).random((C, (t[1] - t[0]) * sc, (t[3] - t[2]) * sc), dtype=np.float32)
buf = torch.full((ex.total_elems,), float('nan'))
for f in range(frames):                      # what moe_run_plan_tiles does with ex.tile_dst
    for k in ex.tiles_of(f, rank):
        at = int(ex.tile_dst[f, k])
        buf[at:at + sizes[k]] = torch.from_numpy(tile_value(f, k).reshape(-1))
for rep in range(4):                         # the layout is reused every step; reps 2, 3: the grouped isend / irecv form of the exchange (what an all_to_all_single
    if rep == 2:                             # failure falls back to, MOE_DIST_EXCHANGE=p2p) -- same buffer layout, same result; rep 3 asynchronously
        buf[ex.own_elems:ex.own_elems + ex.recv_elems] = float('nan')
        mdist_mod.EXCHANGE_MODE = 'p2p'
    if rep == 3:
        buf[ex.own_elems:ex.own_elems + ex.recv_elems] = float('nan')
        h = ex.exchange(buf, async_op=True)
        h.wait()
        mine = ex.mine
    else:
        mine = ex.exchange(buf)
    if rep == 3:
        mdist_mod.EXCHANGE_MODE = 'all_to_all'
    assert mine == [f for f in range(frames) if f % world == rank]
    for f in mine:                           # what moe_stitch does with ex.stitch_off[f]
        tiles = [buf[int(ex.stitch_off[f][k]):int(ex.stitch_off[f][k]) + sizes[k]].numpy().reshape(tile_value(f, k).shape) for k in range(nt)]
        assert not any(np.isnan(t).any() for t in tiles)
        want = ostitch.fold_stitch([tile_value(f, k) for k in range(nt)], pl, sc)
        assert np.array_equal(ostitch.fold_stitch(tiles, pl, sc), want)
# ---- agree_arithmetic: rank 0's calibration result is imposed on every rank (a stub with the module's methods: ranks that calibrated differently) ------------
class _M(object):
    def __init__(self, prec, blocks): self.p, self.b, self._finalized_key = prec, blocks, (0, 'auto')
    def resolved_precision(self): return self.p
    def exact_blocks(self): return self.b if self.p == 'mixed' else 0
    def set_precision(self, p): self.p = p; return self
    def set_exact_blocks(self, b): self.b = b; return self
for r0, rest, want in ((('mixed', 2), ('mixed', 1), ('mixed', 2)), (('fp16x3', 0), ('mixed', 4), ('fp16x3', 0)), (('mixed', 5), ('fp16x3', 0), ('mixed', 5))):
    mm = _M(*(r0 if rank == 0 else rest))
    assert mdist_mod.agree_arithmetic(mm) == want and (mm.resolved_precision(), mm.exact_blocks()) == want, (rank, mm.p, mm.b)
    assert mdist_mod.agree_arithmetic(mm) == want          # (cached: no second collective)
# ---- wire format 'f16s' (fp16 values + fp32 seam rows / columns on the links): same layout, fp16 canvases bit-identical ---------------------------
import wire_codec
from moephoto_amd import dist as mdist
from moephoto_amd.imageProcess import TilePlan
mdist.CPU_CODEC = wire_codec.codec           # (the product packs with moe_wire_pack on the GPU; tests/test_gpu_parity.py holds that kernel against this codec)
seams = TilePlan(shape, 1 << 40, 1e-3, pad, sc, 8, crop).seams()
dims = [(C, (t[1] - t[0]) * sc, (t[3] - t[2]) * sc) for t in pl.tiles]
wspec = (dims, seams, pl.pad_sc)
exw = TileExchange(sizes, frames, rank, world, wire=wspec)
assert np.array_equal(exw.tile_dst, ex.tile_dst) and exw.total_elems == ex.total_elems and exw.signature() != ex.signature()
assert sum(exw.wire_send_split) == exw.wire_send_words and sum(exw.wire_recv_split) == exw.wire_recv_words
if exw.send_elems:
    assert exw.wire_send_words < 0.9 * exw.send_elems, (exw.wire_send_words, exw.send_elems)
bufw = torch.full((exw.total_elems,), float('nan'))
for f in range(frames):
    for k in exw.tiles_of(f, rank):
        at = int(exw.tile_dst[f, k])
        bufw[at:at + sizes[k]] = torch.from_numpy(tile_value(f, k).reshape(-1))
rounded = 0
for f in exw.exchange(bufw):
    tiles = [bufw[int(exw.stitch_off[f][k]):int(exw.stitch_off[f][k]) + sizes[k]].numpy().reshape(dims[k]) for k in range(nt)]
    rounded += sum(not np.array_equal(t, tile_value(f, k)) for k, t in enumerate(tiles))
    want = ostitch.fold_stitch([tile_value(f, k) for k in range(nt)], pl, sc)
    assert np.array_equal(ostitch.fold_stitch(tiles, pl, sc).astype(np.float16), want.astype(np.float16))
assert rounded > 0 or world == 1             # (the tiles that crossed really were fp16 outside their seams)
# ---- band-sharded stitch (fewer frames than ranks): every rank folds its own row band of the ONE canvas -------------------------------
rows_o = ostitch.axis_cover(pl.anchors_h, sc, pl.pad_sc, pl.out_shape[-2])          # (first written, first un-blended, origin) per tile row
ext = [ (pl.tiles[i * pl.step_w][1] - pl.tiles[i * pl.step_w][0]) * sc for i in range(len(rows_o)) ]
rows_tab = [(r_[0], r_[1], r_[2], e) for r_, e in zip(rows_o, ext)]
for nf, wire in ((1, None), (2, None), (1, wspec), (2, wspec)):
    exb = TileExchange(sizes, nf, rank, world, bands=(pl.step_w, rows_tab, dims, pl.pad_sc), wire=wire)
    same = (lambda a, b: np.array_equal(a, b)) if wire is None else (lambda a, b: np.array_equal(a.astype(np.float16), b.astype(np.float16)))
    sigs = [None] * world
    dist.all_gather_object(sigs, exb.signature())
    assert len(set(sigs)) == 1
    assert exb.own_elems + exb.recv_elems + exb.send_elems == exb.total_elems
    bufb = torch.full((exb.total_elems,), float('nan'))
    for f in range(nf):                      # the engine writes every owned tile ONCE, at tile_dst
        for k in exb.tiles_of(f, rank):
            at = int(exb.tile_dst[f, k])
            assert at >= 0
            bufb[at:at + sizes[k]] = torch.from_numpy(tile_value(f, k).reshape(-1))
    mine = exb.exchange(bufb)
    i0, i1 = exb.my_rows
    nrow = len(rows_tab)
    assert (i0, i1) == ((rank * nrow) // world, ((rank + 1) * nrow) // world)
    if i1 <= i0:
        assert mine == []
    else:
        assert mine == list(range(nf))
        y0, y1 = rows_tab[i0][1], (rows_tab[i1][1] if i1 < nrow else pl.out_shape[-2])
        for f in mine:                       # what moe_stitch_band does: whole tiles of the band's tile rows, the strip of the next tile row, nothing else
            tiles = []
            for k in range(nt):
                i = k // pl.step_w
                Cc, th, tw = dims[k]
                t = np.full((Cc, th, tw), np.nan, np.float32)
                at = int(exb.stitch_off[f][k])
                if i0 <= i < i1:
                    t = bufb[at:at + sizes[k]].numpy().reshape(Cc, th, tw)
                    assert same(t, tile_value(f, k))
                elif i == i1:
                    r0 = rows_tab[i][0] - rows_tab[i][2]
                    strip = bufb[at:at + Cc * pl.pad_sc * tw].numpy().reshape(Cc, pl.pad_sc, tw)
                    assert np.array_equal(strip, tile_value(f, k)[:, r0:r0 + pl.pad_sc])
                    t[:, r0:r0 + pl.pad_sc] = strip
                tiles.append(t)
            want = ostitch.fold_stitch([tile_value(f, k) for k in range(nt)], pl, sc)
            with np.errstate(invalid='ignore'):
                got = ostitch.fold_stitch(tiles, pl, sc)
            assert same(got[:, y0:y1], want[:, y0:y1]), (rank, f, y0, y1)      # the band's rows need nothing but its own tiles and that strip
    total = [None] * world
    dist.all_gather_object(total, (i0, i1))
    assert sorted(set(v for a, b in total for v in range(a, b))) == list(range(nrow))  # the bands tile the canvas
# gather_bands incl. EMPTY bands (more ranks than tile rows: 2 tile rows here, so rank(s) of a world of 3 hold a (C, 0, W) band; ADVICE r04: such a rank must
# still allocate its receive buffers where its collectives run and must not become a broadcast source)
from moephoto_amd.dist import gather_bands
want = ostitch.fold_stitch([tile_value(0, k) for k in range(nt)], pl, sc)
i0, i1 = ((rank * nrow) // world, ((rank + 1) * nrow) // world)
if i1 > i0:
    band = (rows_tab[i0][1], torch.from_numpy(want[:, rows_tab[i0][1]:(rows_tab[i1][1] if i1 < nrow else pl.out_shape[-2])].copy()))
else:
    band = (0, torch.empty((C, 0, want.shape[-1])))
assert world <= nrow or any(((r * nrow) // world) == (((r + 1) * nrow) // world) for r in range(world))      # (world 3: one band is empty)
whole = gather_bands(band)
assert np.array_equal(whole.numpy(), want)
sd = OrderedDict([('a.weight', torch.arange(12.).reshape(3, 4)), ('b', torch.tensor([2.5]))]) if rank == 0 else None
out = broadcast_state_dict(sd, src=0)
assert list(out.keys()) == ['a.weight', 'b'] and out['a.weight'].shape == (3, 4) and float(out['b']) == 2.5
assert torch.equal(out['a.weight'], torch.arange(12.).reshape(3, 4))
dist.barrier()
print('RANK_OK', rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [2, 3])
def test_tile_exchange_gloo(world, tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, MOE_ROOT=ROOT, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count('RANK_OK') == world
