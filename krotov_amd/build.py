"""In-tree build of libkrotov_hip.so with hipcc for gfx950 (no JIT cache).

The library is compiled as several translation units in parallel (csrc/kh_common.h: KH_TU_*): ``krotov_hip.hip`` with
``-DKH_TU=KH_TU_MAIN`` -- host code, dispatch, the small set-up kernels; the sweep-kernel templates are
``extern template`` there -- and ``kh_tu.hip`` once per kernel family with ``-DKH_TU=<family>``, which holds the
family's explicit instantiations (csrc/kh_instances.inc, written by scripts/gen_kernel_instances.py) and its
non-template kernels.  Same flags as the one-unit build (``hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared
-Iinclude krotov_amd/csrc/krotov_hip.hip``, which still works and gives the same library: scripts/ use it for their
-DKH_TIMING / stress builds); objects under ``krotov_amd/_obj/`` (git-ignored), each re-compiled only when a file the
compiler's own dependency list (-MD) names for it is newer: an edit of one kernel family rebuilds that family's units
and the main unit (11 s), not the library.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
SRC = os.path.join(CSRC, 'krotov_hip.hip')
TU_SRC = os.path.join(CSRC, 'kh_tu.hip')
OBJ = os.path.join(HERE, '_obj')
OUT = os.path.join(HERE, 'libkrotov_hip.so')

# family units of kh_tu.hip, the ones that take longest first (they start first)
FAMILY_UNITS = [
    'KH_TU_COOP_UPDATE_A', 'KH_TU_COOP_UPDATE_B', 'KH_TU_ELL_UPDATE_A', 'KH_TU_ELL_UPDATE_B', 'KH_TU_TILE',
    'KH_TU_COOP_STORE', 'KH_TU_TILEN', 'KH_TU_STREAM', 'KH_TU_ELL_STORE', 'KH_TU_Q2', 'KH_TU_ENS', 'KH_TU_MINI',
    'KH_TU_GENERIC', 'KH_TU_TILEX',
]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _deps():
    """Everything the library is compiled from: every file under csrc/ (globbed, so a new kernel file cannot be
    forgotten) and the public C header."""
    return sorted(glob.glob(os.path.join(CSRC, '*'))) + [os.path.join(ROOT, 'include', 'krotov_hip.h')]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _unit_deps(dep_file):
    """The repo files a unit was compiled from, read off the compiler's own dependency file (-MD)."""
    try:
        text = open(dep_file).read().replace('\\\n', ' ')
    except OSError:
        return None
    files = [w for w in text.split() if w.startswith(ROOT + os.sep) and not w.endswith(':')]
    return sorted(set(files)) or None


def _compile(unit, extra, verbose, force, obj_dir):
    src = SRC if unit == 'KH_TU_MAIN' else TU_SRC
    obj = os.path.join(obj_dir, unit.lower() + '.o')
    dep = obj[:-2] + '.d'
    if not force and os.path.exists(obj):
        deps = _unit_deps(dep)
        if deps and all(os.path.exists(d) and os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            if verbose:
                print('krotov_amd.build: %-22s up to date' % unit, file=sys.stderr)
            return obj
    cmd = [_hipcc()] + FLAGS + list(extra) + ['-I' + os.path.join(ROOT, 'include'), '-DKH_TU=' + unit, '-MD', '-MF', dep,
                                              '-c', src, '-o', obj]
    t0 = time.time()
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("%s\n%s" % (' '.join(cmd), proc.stderr[-4000:]))
    if verbose:
        print('krotov_amd.build: %-22s %5.1f s' % (unit, time.time() - t0), file=sys.stderr)
    return obj


def build(force=False, verbose=False, extra_flags=(), out=None, jobs=None):
    """Compile the HIP kernels + C ABI for gfx950 if the library is stale.  ``extra_flags`` (e.g. ``['-DKH_TIMING']``)
    and ``out`` build a variant next to the product library (its objects are rebuilt every time)."""
    out = OUT if out is None else out
    variant = bool(extra_flags) or out != OUT
    srcs = _deps()
    newest = max(os.path.getmtime(s) for s in srcs)
    if not force and not variant and os.path.exists(out) and os.path.getmtime(out) >= newest:
        if verbose:
            print('krotov_amd.build: %s is up to date (reused)' % out, file=sys.stderr)
        return out
    obj_dir = OBJ
    if variant:  # a variant build keeps its objects apart (other flags: never reusable for the product library)
        obj_dir = os.path.join(OBJ, 'variant_' + os.path.basename(out).replace('.', '_'))
    os.makedirs(obj_dir, exist_ok=True)
    units = ['KH_TU_MAIN'] + FAMILY_UNITS
    jobs = jobs or int(os.environ.get('KH_BUILD_JOBS', '0')) or min(len(units), os.cpu_count() or 4)
    t0 = time.time()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        # the family units first (longest first), the main unit (10 s) fills a gap
        futures = [pool.submit(_compile, u, extra_flags, verbose, force or variant, obj_dir)
                   for u in FAMILY_UNITS + ['KH_TU_MAIN']]
        objs = [f.result() for f in futures]
    cmd = [_hipcc(), '--offload-arch=gfx950', '-fPIC', '-shared'] + objs + ['-o', out]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    if verbose:
        print('krotov_amd.build: compiled %s in %.1f s (%d units, %d at a time)' % (out, time.time() - t0, len(units), jobs),
              file=sys.stderr)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
