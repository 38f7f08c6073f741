"""Dependency-free reader for the one thing the loaders need from a MATLAB ``-v7.3`` file: a numeric (real or complex)
N-d array stored as an HDF5 dataset in the root group.

The reference reads ``./data/<profile>_Nt64_Nr16_ULA<sp>_seed<seed>.mat`` with ``hdf5storage.loadmat`` (``loaders.py:29``);
the files are written by ``save(..., '-v7.3')`` (``matlab/generate_data.m:36-38``), i.e. an HDF5 file behind a 512-byte
MATLAB user block.  Neither ``hdf5storage`` nor ``h5py`` is part of this image, so this module parses the subset of the
HDF5 file format (HDF5 File Format Specification, version 2.0/3.0) that MATLAB's HDF5 1.8 library emits for such a
variable:

* super block version 0 or 1 (also 2/3 when the root group still uses a version-1 object header), located at offset
  0, 512, 1024, ...;
* "old style" groups: symbol-table message -> version-1 B-tree (node type 0) + local heap + symbol nodes;
* version-1 object headers with continuation blocks; dataspace message v1/v2; datatype classes 0 (integer), 1 (IEEE
  float, either byte order) and 6 (compound, versions 1-3 -- MATLAB stores complex numbers as ``{real, imag}``);
* data layout message version 3: compact, contiguous, or chunked (version-1 B-tree, node type 1) with the ``deflate``
  and ``shuffle`` filters (MATLAB compresses v7.3 variables with deflate unless ``-nocompression`` is given).

Anything else (new-style groups, fractal heaps, v2 B-trees, other filters, references / cell arrays) raises
``Mat73Error`` with the name of the missing feature.  ``h5py`` remains the preferred route when it is installed
(``loaders.read_output_h``); tests/test_mat73.py checks this reader against files written by the real HDF5 library.

Array orientation: HDF5 stores the dimensions of a MATLAB array reversed (MATLAB is column-major); ``read_dataset``
returns the array in HDF5 (C) order, exactly what ``h5py`` would return, and the caller transposes.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class Mat73Error(ValueError):
    pass


class _File:
    def __init__(self, buf):
        self.buf = buf
        off = 0
        while True:                                   # the super block sits at 0 or at a power-of-two multiple of 512
            if buf[off:off + 8] == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(buf):
                raise Mat73Error('no HDF5 signature found: not a MATLAB v7.3 / HDF5 file')
        self.sb = off
        ver = buf[off + 8]
        if ver in (0, 1):
            self.so, self.sl = buf[off + 13], buf[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                              # base, free-space info, end of file, driver info
            # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self._u(p + self.so, self.so)
        elif ver in (2, 3):
            self.so, self.sl = buf[off + 9], buf[off + 10]
            p = off + 12
            self.base = self._u(p, self.so)
            self.root = self._u(p + 3 * self.so, self.so)
        else:
            raise Mat73Error('unsupported HDF5 super block version %d' % ver)
        if self.so != 8 or self.sl != 8:
            raise Mat73Error('only 8-byte offsets/lengths are supported (got %d/%d)' % (self.so, self.sl))

    # --- primitives (all addresses are relative to the base address = start of the super block for MATLAB files) ---
    def _u(self, p, n):
        return int.from_bytes(self.buf[p:p + n], 'little')

    def addr(self, a):
        return self.base + a

    # --- object headers ----------------------------------------------------------------------------
    def messages(self, address):
        """[(type, flags, bytes)] of a version-1 object header, following continuation messages."""
        p = self.addr(address)
        if self.buf[p:p + 4] == b'OHDR':
            raise Mat73Error('version-2 object headers (HDF5 "latest" format) are not supported')
        if self.buf[p] != 1:
            raise Mat73Error('unsupported object header version %d' % self.buf[p])
        nmsg = self._u(p + 2, 2)
        size = self._u(p + 8, 4)
        blocks = [(p + 16, size)]                         # 12-byte prefix padded to 16
        out = []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u(q, 2), self._u(q + 2, 2), self.buf[q + 4]
                body = self.buf[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:                         # object header continuation: offset, length
                    blocks.append((self.addr(self._u(q - msize, 8)), self._u(q - msize + 8, 8)))
                out.append((mtype, flags, body))
        return out

    # --- old-style groups ----------------------------------------------------------------------------
    def group_entries(self, address):
        """{name: object header address} of a group with a symbol-table message."""
        st = [b for t, _, b in self.messages(address) if t == 0x11]
        if not st:
            raise Mat73Error('group without a symbol table (new-style link messages / fractal heap) is not supported')
        btree, heap = struct.unpack('<QQ', st[0][:16])
        hp = self.addr(heap)
        if self.buf[hp:hp + 4] != b'HEAP':
            raise Mat73Error('local heap signature missing')
        heap_data = self.addr(self._u(hp + 24, 8))
        entries = {}

        def name_at(off):
            s = heap_data + off
            return self.buf[s:self.buf.index(b'\0', s)].decode('utf-8')

        def walk(node):
            p = self.addr(node)
            if self.buf[p:p + 4] == b'SNOD':
                n = self._u(p + 6, 2)
                for i in range(n):
                    e = p + 8 + i * 40
                    entries[name_at(self._u(e, 8))] = self._u(e + 8, 8)
                return
            if self.buf[p:p + 4] != b'TREE' or self.buf[p + 4] != 0:
                raise Mat73Error('unexpected group B-tree node')
            used = self._u(p + 6, 2)
            q = p + 24                                   # after signature, type, level, entries used, two siblings
            for i in range(used):
                walk(self._u(q + 8 + i * 16, 8))          # key (8), child (8), key, child, ...
        walk(btree)
        return entries

    # --- datasets --------------------------------------------------------------------------------------
    def _dtype(self, body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size = struct.unpack('<I', body[4:8])[0]
        order = '>' if bits0 & 1 else '<'
        if cls == 0:                                      # fixed point
            signed = bool(bits0 & 0x08)
            return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size)), 8 + 4
        if cls == 1:                                      # floating point (IEEE layouts only)
            if size not in (2, 4, 8):
                raise Mat73Error('unsupported float size %d' % size)
            return np.dtype('%sf%d' % (order, size)), 8 + 12
        if cls == 6:                                      # compound
            nmemb = body[1] | (body[2] << 8)
            p, fields = 8, []
            for _ in range(nmemb):
                e = body.index(b'\0', p)
                name = body[p:e].decode('ascii')
                if ver < 3:
                    p += ((e - p) // 8 + 1) * 8           # null-terminated, padded to a multiple of 8
                else:
                    p = e + 1
                if ver == 3:
                    nb = 1 if size < 256 else 2 if size < 65536 else 4
                    offset = int.from_bytes(body[p:p + nb], 'little')
                    p += nb
                else:
                    offset = struct.unpack('<I', body[p:p + 4])[0]
                    p += 4
                    if ver == 1:
                        if body[p] != 0:
                            raise Mat73Error('array members of compound types are not supported')
                        p += 1 + 3 + 4 + 4 + 16           # dimensionality, reserved, permutation, reserved, 4 sizes
                mt, used = self._dtype(body[p:])
                p += used
                fields.append((name, mt, offset))
            dt = np.dtype({'names': [f[0] for f in fields], 'formats': [f[1] for f in fields],
                           'offsets': [f[2] for f in fields], 'itemsize': size})
            return dt, p
        raise Mat73Error('unsupported datatype class %d (only integers, floats and {real, imag} compounds)' % cls)

    def dataset(self, address):
        msgs = self.messages(address)
        get = lambda t: [b for mt, _, b in msgs if mt == t]                  # noqa: E731
        space, dtype, layout, filters = get(0x01), get(0x03), get(0x08), get(0x0B)
        if not (space and dtype and layout):
            raise Mat73Error('object is not a simple dataset (a group, cell array or reference?)')
        s = space[0]
        sver, rank = s[0], s[1]
        p = 8 if sver == 1 else 4
        dims = [struct.unpack('<Q', s[p + 8 * i:p + 8 * i + 8])[0] for i in range(rank)]
        dt, _ = self._dtype(dtype[0])
        lay = layout[0]
        count = int(np.prod(dims)) if dims else 1
        if lay[0] in (1, 2):
            # HDF5 1.6-era layout (MATLAB R2006b-R2008): version, dimensionality, class, 5 reserved bytes, [address],
            # dimension sizes (4 bytes each; chunked: chunk sizes, the last one = element size), [compact: size + data]
            nd, cls = lay[1], lay[2]
            if cls == 1:
                a = struct.unpack('<Q', lay[8:16])[0]
                return (np.zeros(dims, dt) if a == UNDEF else
                        np.frombuffer(self.buf[self.addr(a):self.addr(a) + count * dt.itemsize], dt, count).reshape(dims))
            if cls == 2:
                btree = struct.unpack('<Q', lay[8:16])[0]
                cdims = [struct.unpack('<I', lay[16 + 4 * i:20 + 4 * i])[0] for i in range(nd)]
                return self._read_chunked(btree, dims, cdims[:-1], dt, self._filters(filters[0]) if filters else [])
            n = struct.unpack('<I', lay[8 + 4 * nd:12 + 4 * nd])[0]
            return np.frombuffer(lay[12 + 4 * nd:12 + 4 * nd + n], dt, count).reshape(dims)
        if lay[0] != 3:
            raise Mat73Error('data layout message version %d is not supported' % lay[0])
        if lay[1] == 0:                                   # compact
            n = struct.unpack('<H', lay[2:4])[0]
            raw = lay[4:4 + n]
        elif lay[1] == 1:                                 # contiguous
            a, n = struct.unpack('<QQ', lay[2:18])
            if a == UNDEF:
                return np.zeros(dims, dt)
            raw = self.buf[self.addr(a):self.addr(a) + n]
        elif lay[1] == 2:                                 # chunked
            nd = lay[2]
            btree = struct.unpack('<Q', lay[3:11])[0]
            cdims = [struct.unpack('<I', lay[11 + 4 * i:15 + 4 * i])[0] for i in range(nd)]   # last one = element size
            return self._read_chunked(btree, dims, cdims[:-1], dt, self._filters(filters[0]) if filters else [])
        else:
            raise Mat73Error('unknown layout class %d' % lay[1])
        return np.frombuffer(raw, dt, count).reshape(dims)

    @staticmethod
    def _filters(body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack('<H', body[p:p + 2])[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack('<H', body[p + 2:p + 4])[0]
                p += 4
            else:
                nlen = 0
                p += 2
            ncd = struct.unpack('<H', body[p + 2:p + 4])[0]
            p += 4 + (((nlen + 7) // 8) * 8 if ver == 1 else nlen)
            cd = [struct.unpack('<I', body[p + 4 * i:p + 4 * i + 4])[0] for i in range(ncd)]
            p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
            out.append((fid, cd))
        return out

    def _read_chunked(self, btree, dims, cdims, dt, filters):
        for fid, _ in filters:
            if fid not in (1, 2):
                raise Mat73Error('unsupported HDF5 filter id %d (only deflate = 1 and shuffle = 2)' % fid)
        out = np.zeros(dims, dt)
        rank = len(dims)
        if btree == UNDEF:
            return out
        csize = int(np.prod(cdims)) * dt.itemsize

        def walk(node):
            p = self.addr(node)
            if self.buf[p:p + 4] != b'TREE' or self.buf[p + 4] != 1:
                raise Mat73Error('unexpected chunk B-tree node')
            level, used = self.buf[p + 5], self._u(p + 6, 2)
            q = p + 24
            ksz = 8 + 8 * (rank + 1)                      # chunk size, filter mask, offsets (rank + 1)
            for i in range(used):
                k = q + i * (ksz + 8)
                nbytes, mask = struct.unpack('<II', self.buf[k:k + 8])
                offs = [struct.unpack('<Q', self.buf[k + 8 + 8 * j:k + 16 + 8 * j])[0] for j in range(rank)]
                child = self._u(k + ksz, 8)
                if level > 0:
                    walk(child)
                    continue
                raw = bytes(self.buf[self.addr(child):self.addr(child) + nbytes])
                for idx in range(len(filters) - 1, -1, -1):               # undo the pipeline in reverse order
                    if mask & (1 << idx):
                        continue
                    fid, cd = filters[idx]
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    else:                                                  # shuffle: byte planes -> elements
                        es = cd[0] if cd else dt.itemsize
                        n = len(raw) // es
                        raw = np.frombuffer(raw[:n * es], np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
                chunk = np.frombuffer(raw[:csize], dt).reshape(cdims)
                sel = tuple(slice(o, min(o + c, d)) for o, c, d in zip(offs, cdims, dims))
                out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
        walk(btree)
        return out


def list_variables(filename):
    """Names of the datasets / groups in the root group of a MATLAB v7.3 (HDF5) file."""
    with open(filename, 'rb') as f:
        h = _File(f.read())
    return sorted(h.group_entries(h.root))


def read_dataset(filename, name):
    """The root-group dataset ``name`` as a numpy array in HDF5 (C) dimension order.  MATLAB complex arrays come back as
    a structured array with fields ``real`` / ``imag`` (as from ``h5py``); see ``as_complex``."""
    with open(filename, 'rb') as f:
        h = _File(f.read())
    entries = h.group_entries(h.root)
    if name not in entries:
        raise KeyError('%r not in %s (variables: %s)' % (name, filename, ', '.join(sorted(entries))))
    return h.dataset(entries[name])


def as_complex(a):
    """``{real, imag}`` structured array -> complex ndarray (other arrays are returned unchanged)."""
    if a.dtype.names and set(a.dtype.names) >= {'real', 'imag'}:
        return a['real'] + 1j * a['imag']
    return a


def loadmat_variable(filename, name):
    """``hdf5storage.loadmat(filename)[name]`` for a numeric array: complex handling + MATLAB dimension order."""
    return np.transpose(as_complex(read_dataset(filename, name)))
