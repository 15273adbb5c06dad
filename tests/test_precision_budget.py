"""The error budget behind MOE_PREC_MIXED, on the CPU (tests/emu_precision.py emulates the engine's roundings on top of an
fp32 forward).  Pins the defaults of exact_blocks_of() in moephoto_amd/csrc/engine.cpp: with them every ARSB net stays
within 1e-3 of the fp32 forward on white noise (the adversarial input), and plain fp16 operands do not."""
import re
import os

import numpy as np
import pytest
import torch

import emu_precision as emu
import golden_defs as gd
from moephoto_amd.weights import load_state_dict_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_defaults_match_engine_source():
    src = open(os.path.join(ROOT, 'moephoto_amd', 'csrc', 'engine.cpp')).read()
    body = src[src.index('int exact_blocks_of('):]
    got = {a.lower(): int(v) for a, v in re.findall(r'case MOE_ARCH_(NET2X|NET3X|NET4X|NETDN): return (\d+);', body)}
    assert got == emu.DEFAULT_EXACT


@pytest.mark.parametrize('key', ['a2', 'a3', 'a4', 'dn_lite5'])
def test_mixed_budget_on_noise(key):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    arch = gd.MODELS[key][0]
    sd = gd.state_dict_for(key, load_state_dict_file)
    x = gd.noise_image(5, (3, 1, 64, 64))
    with torch.no_grad():
        want = emu.forward(arch, sd, x)
    mixed = emu.max_err(arch, sd, x, 'mixed', emu.DEFAULT_EXACT[arch], want)
    plain = emu.max_err(arch, sd, x, 'fp16', want=want)
    assert mixed <= 8.5e-4, (key, mixed)          # the GPU tests assert 1e-3; keep a margin for summation-order noise
    assert mixed < plain
    if key in ('a2', 'dn_lite5'):
        assert plain > 1e-3                        # why 'fp16' is not the default for these nets


def test_emulation_equals_oracle_in_fp32():
    from oracle import nets as onets
    for key in ('a2', 'dn_lite10'):
        arch = gd.MODELS[key][0]
        sd = gd.state_dict_for(key, load_state_dict_file)
        x = gd.natural_image(3, (2, 24, 40))[:, None]
        with torch.no_grad():
            a = emu.forward(arch, sd, x)
        b = onets.forward(arch, sd, x)
        assert float((a - b).abs().max()) <= 2e-6
