"""Builds nfmc_amd/libnfmc_hip.so (the C-ABI library of include/nfmc_hip.h) with hipcc for gfx950.

    python -m nfmc_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
with the tree to the GPU box.  Translation units are compiled in parallel and linked once.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import time
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
# NFMC_BUILD_VARIANT=name NFMC_EXTRA_FLAGS="-D..." builds an experimental variant next to the product library
# (nfmc_amd/libnfmc_hip.<name>.so, objects in csrc/build_<name>/): A/B runs select it with NFMC_LIB
VARIANT = os.environ.get('NFMC_BUILD_VARIANT', '')
OBJ = os.path.join(CSRC, 'build' + ('_' + VARIANT if VARIANT else ''))
LIB = os.path.join(HERE, 'libnfmc_hip%s.so' % ('.' + VARIANT if VARIANT else ''))

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-I' + INCLUDE, '-I' + CSRC,
         '-Wno-unused-result'] + os.environ.get('NFMC_EXTRA_FLAGS', '').split()
# The matrix-core units keep their per-chain state as arrays of 16-byte tiles (f32x4 x[8], ...) that live across loops.
# LLVM's AMDGPU alloca promotion turns such an array into ONE <32 x float> value = one 1024-bit register tuple: 32
# contiguous registers, spilled and reloaded as a whole (measured: the gradient tile set was reloaded 8 x 16 bytes before
# every GEMM step that accumulated into one of its tiles).  With the promotion limited to 16 bytes SROA splits the arrays
# into independent 128-bit values instead.
UNIT_FLAGS = {'neutra_mfma.hip': ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=16'],
              'flow_mfma.hip': ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=16'],
              'mfma_wide.hip': ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=16'],
              'fit_mfma.hip': ['-mllvm', '-amdgpu-promote-alloca-to-vector-limit=16']}


# units that take a minute or more to compile go first, so that the pool finishes with the short ones
# (NFMC_BUILD_TIMES=1 prints the time of each)
SLOW_FIRST = ['flow_b_kernels.hip', 'neutra_kernels_r64.hip', 'neutra_kernels_r32.hip', 'neutra_kernels_r16.hip',
              'imh_parallel_rqs.hip', 'imh_parallel.hip', 'fit_kernels.hip', 'neutra_mfma.hip', 'fit_mfma.hip']


def sources():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    return [f for f in SLOW_FIRST if f in names] + [f for f in names if f not in SLOW_FIRST]


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, INCLUDE):
        for f in sorted(os.listdir(root)):
            p = os.path.join(root, f)
            if os.path.isfile(p) and f.endswith(('.hip', '.hpp', '.h')):
                h.update(f.encode())
                with open(p, 'rb') as fh:
                    h.update(fh.read())
    # the include paths enter the digest under their canonical names, not where this checkout happens to live: the same
    # sources give the same digest in /root/repo, on the GPU box's scratch path and in a `git archive` copy
    canon = [f.replace('-I' + INCLUDE, '-I/root/repo/include').replace('-I' + CSRC, '-I/root/repo/nfmc_amd/csrc') for f in FLAGS]
    h.update(' '.join(canon).encode())
    h.update(repr(sorted(UNIT_FLAGS.items())).encode())
    return h.hexdigest()


def _library_digest():
    """The digest baked into the built library (nfmc_build_digest), or None."""
    try:
        import ctypes
        fn = ctypes.CDLL(LIB).nfmc_build_digest
        fn.restype = ctypes.c_char_p
        return fn().decode()
    except Exception:   # noqa: BLE001
        return None


_INCLUDE = None


def _unit_digest(src):
    """sha256 of one translation unit with the local headers it includes (transitively) and its flags: an object whose
    stamp matches is reused, so touching one unit does not recompile the other twenty."""
    import re
    seen, todo = [], [os.path.join(CSRC, src)]
    while todo:
        path = todo.pop()
        if path in seen or not os.path.isfile(path):
            continue
        seen.append(path)
        with open(path, 'r', errors='replace') as fh:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), re.M):
                for root in (CSRC, INCLUDE):
                    todo.append(os.path.join(root, inc))
    h = hashlib.sha256()
    for path in sorted(seen):
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as fh:
            h.update(fh.read())
    canon = [f.replace('-I' + INCLUDE, '-I/root/repo/include').replace('-I' + CSRC, '-I/root/repo/nfmc_amd/csrc') for f in FLAGS]
    h.update(' '.join(canon + UNIT_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + '.o')
    stamp, dig = obj + '.digest', _unit_digest(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig and not os.environ.get('NFMC_BUILD_ALL'):
        return obj
    cmd = [HIPCC] + FLAGS + UNIT_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    with open(stamp, 'w') as fh:
        fh.write(dig)
    if os.environ.get('NFMC_BUILD_TIMES'):
        print('[nfmc_amd.build] %-28s %6.1f s' % (src, time.time() - t0), flush=True)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        os.environ['NFMC_BUILD_ALL'] = '1'   # --force recompiles every unit
    stamp = os.path.join(OBJ, 'digest.txt')
    dig = _digest()
    if not force and os.path.exists(LIB):
        # current when the stamp says so -- or, where the build directory did not travel (the GPU box gets the library but not
        # csrc/build/), when the library itself reports the digest of these sources
        if (os.path.exists(stamp) and open(stamp).read() == dig) or _library_digest() == dig:
            return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError('hipcc not found at %s; cannot build libnfmc_hip.so' % HIPCC)
    srcs = sources()
    if verbose:
        print('[nfmc_amd.build] compiling %d translation units for gfx950' % len(srcs), flush=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    # the library reports the digest of what it was built from (nfmc_build_digest): generated unit, not part of the digest
    dsrc = os.path.join(OBJ, 'build_digest.cpp')
    with open(dsrc, 'w') as fh:
        fh.write('extern "C" const char* nfmc_build_digest(void) { return "%s"; }\n' % dig)
    dobj = os.path.join(OBJ, 'build_digest.o')
    r = subprocess.run([HIPCC, '-O1', '-fPIC', '-c', dsrc, '-o', dobj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for build_digest.cpp:\n' + r.stderr[-2000:])
    objs.append(dobj)
    cmd = [HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    with open(stamp, 'w') as fh:
        fh.write(dig)
    if verbose:
        print('[nfmc_amd.build] wrote', LIB, flush=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
