#!/usr/bin/env python3
"""Per-kernel register / LDS / spill figures of the gfx950 code objects inside libsbc_hip.so (or any object file given).

    python tools/kernel_resources.py [lib-or-object] [name filter]

Reads the AMDGPU metadata note of every code object (llvm-readelf --notes): VGPRs (incl. AGPRs), SGPRs, static LDS, scratch (a
non-zero private segment means spills) and the waves per SIMD the register count allows (MI355X_MICROARCH.md: 512 / alloc)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_no_packed import code_objects  # noqa: E402

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FILT = '/usr/bin/c++filt'


def main(path, pat):
    blob = open(path, 'rb').read()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, (triple, co) in enumerate(code_objects(blob)):
            if 'gfx' not in triple:
                continue
            fn = os.path.join(tmp, 'co%d.o' % i)
            open(fn, 'wb').write(co)
            txt = subprocess.run([READELF, '--notes', fn], capture_output=True, text=True).stdout
            for blk in txt.split('- .agpr_count:')[1:]:
                def g(key):
                    m = re.search(r'\.%s:\s+(\S+)' % key, blk)
                    return m.group(1) if m else '?'
                name = g('name')
                rows.append((name, int(g('vgpr_count')), int(blk.split()[0]), int(g('sgpr_count')), int(g('group_segment_fixed_size')),
                             int(g('private_segment_fixed_size')), g('vgpr_spill_count'), int(g('max_flat_workgroup_size'))))
    names = subprocess.run([FILT], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print('%-5s %-5s %-5s %-7s %-8s %-6s %-5s %s' % ('vgpr', 'agpr', 'sgpr', 'lds', 'scratch', 'spill', 'w/SIMD', 'kernel'))
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        if pat and not re.search(pat, n):
            continue
        alloc = (r[1] + 7) // 8 * 8
        print('%-5d %-5d %-5d %-7d %-8d %-6s %-5d %s' % (r[1], r[2], r[3], r[4], r[5], r[6], min(8, 512 // max(alloc, 1)), n[:150]))


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'score_based_channels_amd', 'libsbc_hip.so'),
         sys.argv[2] if len(sys.argv) > 2 else None)
