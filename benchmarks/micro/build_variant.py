"""Build an experimental variant of libttl_hip.so next to the product library:

    python benchmarks/micro/build_variant.py NAME -DMACRO[=v] ...

-> benchmarks/micro/_exp/libttl_hip_NAME.so (same sources and flags + the macros);
scripts load it by pointing tracktolearn_amd._lib.LIB_PATH at it (TTL_EXP_LIB)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tracktolearn_amd.csrc import build as B  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, 'benchmarks', 'micro', '_exp')
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, f'libttl_hip_{name}.so')
cmd = [B.find_hipcc()] + B.FLAGS + extra + ['-I', os.path.join(ROOT, 'include'), '-I', B.HERE] + \
    B.SOURCES + ['-o', out]
subprocess.run(cmd, check=True)
print(out)
