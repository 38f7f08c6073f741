#!/usr/bin/env python3
"""Fail if the device code of libsbc_hip.so contains packed-fp32 vector arithmetic (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).

Why: a reproducibility precaution, not a measured hardware fault.  The round-2 library lost bit-reproducibility under concurrent
streams and regained it when it was built without packed-fp32 instructions; a stand-alone reproducer written in round 5
(tools/experiments/pk_fma_hazard.hip, profiles/r05_pk_fma_hazard.txt: 0 of 1200 concurrent launches differ) did NOT reproduce a
hardware hazard, so the claim is withdrawn as a statement about MI355X.  -fno-slp-vectorize stays because it
costs nothing measurable and keeps the one property that was observed; this script checks the build instead of trusting the flag: it
pulls every gfx950 code object out of the
clang offload bundles embedded in the shared library and disassembles it with llvm-objdump."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PACKED = re.compile(r'\bv_pk_(fma|add|mul)_f32\b')


def code_objects(blob):
    """(triple, bytes) of every entry of every offload bundle in `blob`."""
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from('<Q', blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size:
                yield triple, blob[pos + off:pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def main(lib):
    blob = open(lib, 'rb').read()
    found, n_obj, n_insn = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, (triple, co) in enumerate(code_objects(blob)):
            if 'gfx' not in triple:
                continue
            n_obj += 1
            fn = os.path.join(tmp, 'co%d.o' % i)
            open(fn, 'wb').write(co)
            dis = subprocess.run([OBJDUMP, '-d', '--no-show-raw-insn', fn], capture_output=True, text=True, check=True).stdout
            sym = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    sym = m.group(1)
                    continue
                if line.startswith('\t') or line.startswith(' '):
                    n_insn += 1
                    if PACKED.search(line):
                        found.append((sym, line.strip()))
    if not n_obj or n_insn < 1000:
        print('no device code found in %s (%d code objects, %d instructions)' % (lib, n_obj, n_insn))
        return 2
    if found:
        print('%d packed-fp32 instructions, e.g.' % len(found))
        for sym, line in found[:10]:
            print('  %s: %s' % (sym, line))
        return 1
    print('no packed-fp32 arithmetic in %d code objects, %d instructions' % (n_obj, n_insn))
    return 0


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'score_based_channels_amd', 'libsbc_hip.so')))
