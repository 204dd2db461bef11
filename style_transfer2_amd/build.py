"""Builds libst2_hip.so (hand-written gfx950 kernels + engine + C ABI) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc cross-compiles
for gfx950 without a GPU, so this also runs in the build container.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'lib', 'libst2_hip.so')
ARCH = 'gfx950'

# (source, extra flags)
SOURCES = (
    ('conv3x3_mfma.hip', ()),
    ('conv3x3_mfma_bf16.hip', ()),
    ('conv3x3_winograd.hip', ()),
    ('conv3x3_wino_split.hip', ()),
    ('gram.hip', ()),
    ('passes.hip', ('-ffp-contract=off',)),      # NumPy-like one-rounding-per-operation arithmetic
    ('lbfgs.hip', ('-ffp-contract=off',)),
    ('style16.hip', ()),
    ('gram16.hip', ()),
    ('conv3x3_dgrad_first.hip', ()),
    ('conv3x3_first_split.hip', ()),
    ('engine.cpp', ('-x', 'hip')),
    ('engine_objective.cpp', ('-x', 'hip')),
    ('engine_step.cpp', ('-x', 'hip')),
    ('engine_resample.cpp', ('-x', 'hip')),
    ('engine_tile.cpp', ('-x', 'hip')),
    ('engine_comm.cpp', ('-x', 'hip')),
)
HEADERS = ('st2_kernels.h', 'wave_reduce.h', 'engine.h', os.path.join('..', '..', 'include', 'st2.h'))


def _hipcc():
    path = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(path):
        raise RuntimeError('hipcc not found: the HIP library cannot be built')
    return path


def _experiment_flags(src):
    """ST2_WS_VAR=n: the what-bounds-the-loop variants of conv3x3_wino_split.hip (development only; results are wrong)."""
    v = os.environ.get('ST2_WS_VAR')
    return ['-DWS_VAR=' + v] if (v and src == 'conv3x3_wino_split.hip') else []


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers + [os.path.abspath(__file__)]):
            cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC',
                   '-Wall', '-Wno-unused-result'] + list(extra) + _experiment_flags(src) + ['-c', s, '-o', o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err:
                print(err, file=sys.stderr)
    if jobs or force or _stale(LIB, objs):
        # -z defs: an unresolved symbol (e.g. a kernel host stub the compiler dropped) fails the LINK, here,
        # instead of the first dlopen on the GPU box
        run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-Wl,-z,defs', '-o', LIB] + objs + ['-ldl'])
    return LIB


PROBES_DIR = os.path.join(os.path.dirname(HERE), 'tools', 'probes')
PROBES_LIB = os.path.join(PROBES_DIR, 'libst2_probes.so')


def build_probes(force=False):
    """Development probes + the isolated conv timing hook (tools/probes/): a library of their own, linked against
    libst2_hip.so for the product's conv launchers.  Nothing on the product path loads it."""
    lib = build_lib()
    srcs = [os.path.join(PROBES_DIR, 'probes.hip'), os.path.join(PROBES_DIR, 'wino_split_probe.hip')]
    deps = srcs + [os.path.join(PROBES_DIR, 'st2_probes.h'), os.path.join(CSRC, 'st2_kernels.h'), lib]
    if force or _stale(PROBES_LIB, deps):
        r = subprocess.run([_hipcc(), '-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC', '-shared', '-Wall', '-Wno-unused-result',
                            ] + srcs + ['-o', PROBES_LIB, '-L' + os.path.dirname(lib), '-lst2_hip',
                            '-Wl,-rpath,$ORIGIN/../../style_transfer2_amd/lib', '-Wl,-z,defs'], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on the probes library:\n' + r.stderr)
    return PROBES_LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
    if '--probes' in sys.argv:
        print(build_probes(force='--force' in sys.argv))
