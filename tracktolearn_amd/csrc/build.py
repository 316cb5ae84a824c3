"""Build libttl_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m tracktolearn_amd.csrc.build [--force]

The shared library is written next to the package (tracktolearn_amd/
libttl_hip.so) so that it travels with the source tree; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = [os.path.join(HERE, f) for f in
           ('ttl_hip.hip', 'ttl_state.hip', 'ttl_peaks.hip', 'ttl_resample.hip', 'ttl_order.hip', 'ttl_learner.hip',
            'ttl_oracle_net.hip')]
HEADERS = [os.path.join(ROOT, 'include', 'ttl_hip.h'),
           os.path.join(ROOT, 'include', 'ttl_learner.h'),
           os.path.join(HERE, 'ttl_internal.h')]
OUTPUT = os.path.join(PKG, 'libttl_hip.so')

FLAGS = [
    '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17',
    # the stopping decisions must round like NumPy/SciPy: never fuse a*b+c,
    # IEEE divide and sqrt, no fast-math
    '-ffp-contract=off', '-fno-fast-math',
    '-fhip-fp32-correctly-rounded-divide-sqrt',
    # only the TTL_API entry points of include/ttl_hip.h leave the library
    '-fvisibility=hidden', '-fvisibility-inlines-hidden',
    '-Wall', '-Wno-unused-function',
    # MFMA results the VALU consumes next (ttl_oracle_net.hip: every accumulator tile is
    # converted to fp16 operand fragments) go to architectural VGPRs, not to AGPRs that
    # would have to be copied out one v_accvgpr_read at a time
    '-mllvm', '-amdgpu-mfma-vgpr-form=1',
]


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'),
                 '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC=/path/to/hipcc)')


def up_to_date():
    if not os.path.exists(OUTPUT):
        return False
    t = os.path.getmtime(OUTPUT)
    deps = SOURCES + HEADERS + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUTPUT
    # link into a temporary name and rename: a process that has the old
    # library mapped keeps its (unlinked) file instead of seeing it truncated
    tmp = OUTPUT + f'.tmp{os.getpid()}'
    cmd = [find_hipcc()] + FLAGS + ['-I', os.path.join(ROOT, 'include'), '-I', HERE] + \
        SOURCES + ['-o', tmp]
    if verbose:
        print(' '.join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, OUTPUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUTPUT


EXAMPLE_SRC = os.path.join(ROOT, 'examples', 'ttl_track_c.c')
EXAMPLE_BIN = os.path.join(ROOT, 'examples', 'ttl_track_c')


def build_example(verbose=True):
    """examples/ttl_track_c: the C ABI driven from plain C99 (no Python, no
    torch, compiled by gcc), linked against the in-tree libttl_hip.so and the
    HIP runtime of the ROCm installation hipcc belongs to."""
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(find_hipcc())))
    cc = shutil.which('gcc') or shutil.which('cc')
    if not cc:
        raise RuntimeError('no C compiler (gcc) for examples/ttl_track_c')
    cmd = [cc, '-std=c99', '-O2', '-Wall', '-Wextra', '-D__HIP_PLATFORM_AMD__',
           '-I', os.path.join(rocm, 'include'), '-I', os.path.join(ROOT, 'include'),
           EXAMPLE_SRC, '-L', PKG, '-lttl_hip', '-L', os.path.join(rocm, 'lib'), '-lamdhip64',
           '-Wl,-rpath,$ORIGIN/../tracktolearn_amd', '-Wl,-rpath,' + os.path.join(rocm, 'lib'),
           '-lm', '-o', EXAMPLE_BIN]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return EXAMPLE_BIN


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUTPUT)
    if '--example' in sys.argv:
        print(build_example())
