#!/usr/bin/env python3
"""Write the MATLAB ``-v7.3`` fixtures under tests/golden/ with the REAL HDF5 library (build container only).

``score_based_channels_amd/mat73.py`` is a from-the-spec HDF5 parser; to pin it against files the HDF5 library itself
wrote, this script uses the ``h5py`` 3.3 / HDF5 1.10 of ``/opt/conda/bin/python3.9`` (the only HDF5 writer in the image;
the main interpreter has none) to produce files laid out the way MATLAB's ``save -v7.3`` does: 512-byte user block with
the MATLAB header text, old-style root group, one dataset per variable with reversed dimensions, complex numbers as the
compound ``{real, imag}``, chunked + deflate (MATLAB's default) or contiguous (``-nocompression``), ``MATLAB_class``
attribute.  The expected arrays are stored next to them as ``mat73_expected.npz``.

    python tests/gen_mat_fixture.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')
CONDA_PY = '/opt/conda/bin/python3.9'

WRITER = r'''
import sys, numpy as np, h5py
src, out_dir = sys.argv[1], sys.argv[2]
d = np.load(src)
HEADER = b'MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Fri Oct  2 2026 HDF5 schema 1.00 .'
def finish(fn):
    with open(fn, 'r+b') as f:                      # MATLAB's text header + version/endian bytes in the user block
        f.write(HEADER.ljust(116, b' ') + b'\x00' * 8 + b'\x00\x02' + b'IM')
def cplx(a, real_t):
    t = np.dtype([('real', real_t), ('imag', real_t)])
    o = np.empty(a.shape, t); o['real'] = a.real; o['imag'] = a.imag
    return o
def put(f, name, a, cls, **kw):
    ds = f.create_dataset(name, data=np.ascontiguousarray(a.T) if a.dtype.names is None else np.ascontiguousarray(a.T), **kw)
    ds.attrs['MATLAB_class'] = np.bytes_(cls)
    return ds
# 1) the reference data layout: output_h complex single [N, n_sym, Nr, Nt], deflate level 3, ragged chunk edges
fn = out_dir + '/CDL-C_Nt64_Nr16_ULA0.50_seed4321.mat'
with h5py.File(fn, 'w', userblock_size=512, libver='earliest') as f:
    put(f, 'output_h', cplx(d['output_h'], '<f4'), 'single', chunks=(24, 16, 1, 2), compression='gzip', compression_opts=3)
    put(f, 'spacing', d['spacing'], 'double')
finish(fn)
# 2) -nocompression: contiguous; complex double; plus shuffle + deflate on an integer array and a big-endian float
fn = out_dir + '/mat73_variants.mat'
with h5py.File(fn, 'w', userblock_size=512, libver='earliest') as f:
    put(f, 'zc', cplx(d['zc'], '<f8'), 'double')
    put(f, 'ints', d['ints'], 'int32', chunks=(7, 5), compression='gzip', shuffle=True)
    put(f, 'be', d['be'].astype('>f4'), 'single')
    put(f, 'scalar', d['scalar'], 'double')
    for i in range(40):                              # enough variables to split the group B-tree / symbol nodes
        put(f, 'filler_%02d' % i, np.arange(3.0) + i, 'double')
finish(fn)
print('written')
'''


def main():
    from score_based_channels_amd import synth
    rng = np.random.default_rng(73)
    exp = {
        'output_h': synth.generate_output_h('CDL-C', 3, 64, 16, 0.5, 4321, n_sym=2),          # [N, n_sym, Nr, Nt]
        'spacing': np.array([[0.5]]),
        'zc': (rng.standard_normal((5, 3, 4)) + 1j * rng.standard_normal((5, 3, 4))).astype(np.complex128),
        'ints': rng.integers(-1000, 1000, (11, 23)).astype(np.int32),
        'be': rng.standard_normal((6, 2)).astype(np.float32),
        'scalar': np.array([[3.25]]),
    }
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'src.npz')
        np.savez(src, **exp)
        subprocess.run([CONDA_PY, '-c', WRITER, src, GOLD], check=True)
    np.savez_compressed(os.path.join(GOLD, 'mat73_expected.npz'), **exp)
    for fn in ('CDL-C_Nt64_Nr16_ULA0.50_seed4321.mat', 'mat73_variants.mat'):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)), 'bytes')


if __name__ == '__main__':
    main()
