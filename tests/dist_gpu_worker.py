#!/usr/bin/env python
"""Two ranks sharing ONE GPU (MOE_DIST_BACKEND=gloo MOE_FORCE_DEVICE=0): the tile-parallel path of dist.run_frames
(owner-sharded moe_run_plan_frames -> all-to-all of tile results -> moe_stitch) must reproduce the single-process doCrop bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (tests/ -> repo root)
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import imageProcess as ip, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.dist import run_frames  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402

BACKEND = os.environ.get('MOE_DIST_BACKEND', 'gloo')      # 'nccl' = RCCL with the ranks on ONE device (tests/test_gpu_fullsize.py: test_dist_two_ranks_on_one_gpu_over_rccl)
torch.cuda.set_device(0)
if BACKEND == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
else:
    dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
config.deviceId, config.fp16, config.crop_sr, config.modelRoot = 0, True, 64, gd.ZOO
path = '/tmp/moe_chk_a4_{}.pth'.format(rank)
save_state_dict_file(gd.synth_state_dict('a4', load_state_dict_file), path)
runSR.mode_switch['a4'] = (path, runSR.mode_switch['a4'][1])
opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 4, 'ensemble': 0})
frames = [torch.from_numpy(gd.natural_image(50 + f, (3, 150, 200))).cuda().half() for f in range(3)]
out = run_frames(opt, frames, out_dtype=torch.float16)
torch.cuda.synchronize()
assert sorted(out) == [f for f in range(3) if f % world == rank], sorted(out)
for f, y in out.items():
    want = ip.doCrop(opt, frames[f])
    assert torch.equal(y, want), (f, float((y.float() - want.float()).abs().max()))
# frames that are slices of one tensor take the no-copy path of run_frames
allf = torch.stack(frames)
out2 = run_frames(opt, list(allf.unbind(0)), out_dtype=torch.float16)
torch.cuda.synchronize()
for f, y in out2.items():
    assert torch.equal(y, out[f]), f
# a batch of frames in groups of `world` with the exchange of a group overlapping the next group's convolutions (two exchange buffers)
from moephoto_amd.dist import run_frames_overlapped  # noqa: E402
more = frames + [torch.from_numpy(gd.natural_image(60 + f, (3, 150, 200))).cuda().half() for f in range(4)]        # 7 frames: groups of `world`, a short last one
probe = []
out3 = run_frames_overlapped(opt, more, out_dtype=torch.float16, probe=probe)
torch.cuda.synchronize()
if BACKEND == 'nccl' and world > 1:      # the asynchronous all-to-all of a group really ran WHILE the next group's convolutions were running
    assert probe and all(p['exchange_done_while_computing'] for p in probe), probe
    os.write(1, 'RANK {} OVERLAP {}\n'.format(rank, probe).encode())
assert sorted(out3) == [f for f in range(len(more)) if (f % world) == rank], (sorted(out3), rank)
for f, y in out3.items():
    assert torch.equal(y, ip.doCrop(opt, more[f])), f
# band-sharded stitch: ONE frame, every rank folds its row band of the canvas; concatenated, the bands are doCrop's canvas bit for bit
from moephoto_amd.dist import gather_bands, run_frame_bands  # noqa: E402
for fr in (frames[1], torch.from_numpy(gd.natural_image(77, (3, 264, 136))).cuda().half()):      # (7 and 6 tile rows at crop 64)
    bands = run_frame_bands(opt, [fr], out_dtype=torch.float16)      # fewer frames than ranks: band mode (opt-in since round 5)
    torch.cuda.synchronize()
    assert sorted(bands) == [0] and isinstance(bands[0], tuple), bands
    whole = gather_bands(bands[0])
    want = ip.doCrop(opt, fr)
    assert whole.shape == want.shape, (whole.shape, want.shape)
    assert torch.equal(whole, want), float((whole.float() - want.float()).abs().max())
    y0, yb = bands[0]
    assert torch.equal(yb, want[:, y0:y0 + yb.shape[1]])
# wire = 'f16s': fp16 values + fp32 seam rows / columns on the links (moe_wire_pack / moe_wire_unpack around the collective): the fp16 canvases are the same bits,
# in the frame layout, the overlapped groups and the band layout
out4 = run_frames(opt, frames, out_dtype=torch.float16, wire='f16s')
torch.cuda.synchronize()
assert sorted(out4) == sorted(out)
for f, y in out4.items():
    assert torch.equal(y, out[f]), ('wire frames', f, float((y.float() - out[f].float()).abs().max()))
out5 = run_frames_overlapped(opt, more, out_dtype=torch.float16, wire='f16s')
torch.cuda.synchronize()
for f, y in out5.items():
    assert torch.equal(y, out3[f]), ('wire overlapped', f)
fr = torch.from_numpy(gd.natural_image(77, (3, 264, 136))).cuda().half()
whole = gather_bands(run_frame_bands(opt, [fr], out_dtype=torch.float16, wire='f16s')[0])
torch.cuda.synchronize()
assert torch.equal(whole, ip.doCrop(opt, fr)), 'wire bands'
ex = [e for k, e in opt._exchanges.items() if k[-1]][0][1]
if world > 1:
    assert 0 < ex.wire_send_words < ex.send_elems, (ex.wire_send_words, ex.send_elems)
dist.barrier()
os.write(1, 'RANK {} OK frames {}\n'.format(rank, sorted(out)).encode())      # (one write: the ranks share the pipe)
