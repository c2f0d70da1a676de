"""In-tree build of libkrotov_hip.so with hipcc for gfx950 (no JIT cache)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'krotov_hip.hip')


def _deps():
    """Everything the library is compiled from: every header under csrc/ (globbed, so a new kernel file
    cannot be forgotten) and the public C header."""
    import glob

    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 'krotov_hip.h')]


OUT = os.path.join(HERE, 'libkrotov_hip.so')


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 if the library is stale."""
    srcs = [SRC] + _deps()
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        if verbose:
            print('krotov_amd.build: %s is up to date (reused)' % OUT, file=sys.stderr)
        return OUT
    cmd = [
        _hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
        '-I' + os.path.join(ROOT, 'include'), SRC, '-o', OUT,
    ]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    if verbose:
        print('krotov_amd.build: compiled %s' % OUT, file=sys.stderr)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
