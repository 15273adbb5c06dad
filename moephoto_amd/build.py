"""Builds libmoephoto_amd.so (HIP kernels + engine + C ABI) in-tree with hipcc for gfx950.

    python -m moephoto_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the working tree (gpurun snapshot)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['conv_mfma.hip', 'conv3x3_sp.hip', 'conv3x3_rw.hip', 'conv3x3_ps4.hip', 'conv3x3_ps9.hip', 'arsb32c.hip', 'conv64_x3.hip', 'conv64_q8.hip', 'conv64_sq.hip', 'arsb_sq.hip', 'conv64_s.hip', 'conv1x1.hip', 'conv1x1_f2.hip', 'misc_kernels.hip', 'blend.hip', 'engine.cpp', 'planner.cpp']
HEADERS = ['common.h', 'engine.h', os.path.join('..', '..', 'include', 'moephoto_amd.h')]
LIB = os.path.join(HERE, 'libmoephoto_amd.so')
ARCH = 'gfx950'
# packed fp32 VALU (v_pk_add_f32 / v_pk_fma_f32, formed by the SLP vectoriser) costs ~+11 cycles per instruction beside MFMAs
# (MI355X_MICROARCH.md, per-instruction constants): scalar fp32 in the epilogues that ride in an MFMA stream
# -amdgpu-mfma-vgpr-form: MFMA results in arch VGPRs (the weights occupy the AGPRs), so the epilogues read them without v_accvgpr_read
EXTRA_FLAGS = {'blend.hip': ['-ffp-contract=off'],      # three separately rounded operations per blend, as torch evaluates the reference's expression
               'arsb32c.hip': ['-fno-slp-vectorize', '-fno-honor-nans', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv64_x3.hip': ['-fno-slp-vectorize'], 'conv64_q8.hip': ['-fno-slp-vectorize', '-fno-honor-nans', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv64_sq.hip': ['-fno-slp-vectorize', '-fno-honor-nans', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'arsb_sq.hip': ['-fno-slp-vectorize', '-fno-honor-nans', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv64_s.hip': ['-fno-slp-vectorize', '-fno-honor-nans', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv1x1.hip': ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv1x1_f2.hip': ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'conv3x3_sp.hip': ['-fno-honor-nans'], 'conv3x3_rw.hip': ['-fno-honor-nans', '-fno-slp-vectorize'],
               'conv3x3_ps4.hip': ['-fno-honor-nans', '-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1'],
               'conv3x3_ps9.hip': ['-fno-honor-nans', '-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def source_digest():
    """sha256 over the kernel / engine sources and headers (sorted by name): names the binary a set of profiles belongs to -- the GPU box has no .git, and
    bench.py must not pair the PMC bytes of one tree with the timings of another (profiles/pmc_bench.json carries this digest)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(SOURCES + ['common.h', 'engine.h', 'rowtile.h'])
    for f in files + [os.path.join('..', '..', 'include', 'moephoto_amd.h')]:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), 'rb').read())
    return h.hexdigest()


def _compile_cmd(src, extra=()):
    """(command line, its digest) of one source: the digest is what an object's .o.cmd tag must equal for the object to be reused."""
    obj = os.path.join(HERE, '_obj', os.path.splitext(src)[0] + '.o')
    cmd = [hipcc(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
    # -cuid: hipcc derives a compilation-unit id from the PATHS of source and output by default and plants it in symbol names (__hip_cuid_<hash>): the same tree built in
    # another directory gave a different binary ("the build is not bit-reproducible").  A fixed id per source makes libmoephoto_amd.so a function of the sources
    # and flags alone: `git archive HEAD` built anywhere with this hipcc reproduces the shipped library byte for byte (profiles/r05/summary.md).
    cmd += ['-cuid=moe_' + os.path.splitext(src)[0]]
    cmd += EXTRA_FLAGS.get(src, []) + list(extra)
    rel = [c.replace(CSRC, '<csrc>').replace(HERE, '<pkg>') for c in cmd[1:]]      # (the tag must not depend on where the tree lies: the GPU box sees it under another path)
    return cmd, hashlib.sha256('\0'.join(rel).encode()).hexdigest()


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for src in SOURCES:                  # an object compiled by another command line (other flags, an experiment's -D): build_lib rebuilds it
        tag = os.path.join(HERE, '_obj', os.path.splitext(src)[0] + '.o.cmd')
        if not os.path.exists(tag) or open(tag).read().strip() != _compile_cmd(src)[1]:
            return True
    return os.environ.get('MOE_HIPCC_FLAGS') is not None or any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    """Objects are rebuilt one by one when their source (or any header) is newer, in parallel; --force rebuilds all of them."""
    if not force and not stale():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, '_obj'), exist_ok=True)
    keep = {os.path.splitext(src)[0] + e for src in SOURCES for e in ('.o', '.o.cmd')}
    for old in os.listdir(os.path.join(HERE, '_obj')):          # objects of sources that no longer exist must not travel to the GPU box
        if old not in keep:
            os.remove(os.path.join(HERE, '_obj', old))
    extra = os.environ.get('MOE_HIPCC_FLAGS', '').split()        # experiments only, e.g. -DMOE_NO_SGB
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + ['rowtile.h'])
    jobs = []
    for src in SOURCES:
        obj = os.path.join(HERE, '_obj', os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        cmd, want = _compile_cmd(src, extra)
        # an object is reused only when it is newer than its source and headers AND was compiled by this very command line: an object left behind by an experiment
        # (MOE_HIPCC_FLAGS=-DPS4_ABL ...: results wrong by design) or by other EXTRA_FLAGS must not be linked into a later plain build
        tag = obj + '.cmd'
        have = open(tag).read().strip() if os.path.exists(tag) else ''
        if not force and have == want and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        if os.path.exists(tag):
            os.remove(tag)
        if verbose:
            print(' '.join(cmd))
        jobs.append((src, subprocess.Popen(cmd), tag, want))
        running = [j[1] for j in jobs if j[1].poll() is None]
        if len(running) >= max(1, min(8, (os.cpu_count() or 2) // 2)):
            running[0].wait()
    bad = []
    for src, proc, tag, want in jobs:
        if proc.wait() != 0:
            bad.append(src)
        else:
            open(tag, 'w').write(want + '\n')
    if bad:
        raise subprocess.CalledProcessError(1, 'hipcc ' + ' '.join(bad))
    cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_probes():
    """tools/micro/mfma_power (measurement infrastructure, not product): the matrix pipe's rate under the package power cap; bench.py runs it for the
    `power_roofline` object.  Built next to the library so that it travels to the GPU box with the working tree."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, 'tools', 'micro', 'mfma_power.hip')
    out = os.path.join(root, 'tools', 'micro', 'bin', 'mfma_power')
    if os.path.exists(src) and (not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([hipcc(), '--offload-arch=' + ARCH, '-O3', src, '-o', out])
    # round 6: the Winograd probes of the U branch's last up-conv (run by tools/r06_*.sh, numbers in profiles/r06/)
    for name in ('wino_probe', 'wino1d_probe'):
        psrc = os.path.join(root, 'tools', 'micro', name + '.hip')
        pout = os.path.join(root, 'tools', 'micro', 'bin', name)
        if os.path.exists(psrc) and (not os.path.exists(pout) or os.path.getmtime(pout) < os.path.getmtime(psrc)):
            os.makedirs(os.path.dirname(pout), exist_ok=True)
            subprocess.check_call([hipcc(), '--offload-arch=' + ARCH, '-O3', '-w', '-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1', psrc, '-o', pout])
    return out


if __name__ == '__main__':
    build_probes()
    print(build_lib(force='--force' in sys.argv, verbose=True))
