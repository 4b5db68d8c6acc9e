"""Build recipe for libgem_hip.so (gfx950) -- in-tree, explicit hipcc, no JIT cache.

    python -m gem_amd.build            # builds gem_amd/libgem_hip.so if stale
    python -m gem_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the
GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libgem_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Wno-unused-value',
          '-Wno-unused-result'] + os.environ.get('GEM_HIP_EXTRA_CFLAGS', '').split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def deps():
    out = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    out.append(os.path.join(HERE, '..', 'include', 'gem_hip.h'))
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in deps())


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for s in sources():
        o = os.path.join(HERE, 'build', os.path.basename(s) + '.o')
        if not force and os.path.exists(o) and all(os.path.getmtime(o) > os.path.getmtime(p) for p in [s] + deps()[len(sources()):]):
            objs.append(o)
            continue
        cmd = [HIPCC] + CFLAGS + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + objs + ['-ldl']     # (multi.hip loads RCCL with dlopen)
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)          # (rename, never truncate: a gpurun snapshot or a process that has the old file mapped sees a whole library)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
