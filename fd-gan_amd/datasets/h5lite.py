"""Minimal HDF5 reader / writer for the sample files of the reference's data tier -- no h5py, no libhdf5.

What the reference writes (/root/reference/generate_testsample.py:31-38) and reads (datasets/pix2pix.py:62-77):

    f = h5py.File(root + str(i) + '.h5', 'w'); f.create_dataset('gt', data=gt); f.create_dataset('haze', data=haze)

i.e. a root group with two float32 HWC datasets, contiguous layout, no filters.  With h5py's defaults libhdf5 emits
a version-0 superblock, an "old style" root group (symbol-table message -> v1 B-tree -> symbol-table nodes, names in
a local heap) and version-1 object headers; with libver='latest' a version-2/3 superblock, version-2 object headers
("OHDR") and compact link messages.  The reader follows both; the writer emits the first form, byte layout per the
HDF5 File Format Specification (version 0 superblock / III.A B-trees / III.C symbol-table nodes / III.D local heaps /
IV.A.1 version-1 object headers, messages 0x0001 dataspace, 0x0003 datatype, 0x0005 fill value, 0x0008 layout).

Supported datasets: fixed-point and IEEE floating point of 1 / 2 / 4 / 8 bytes, either endianness, contiguous or compact
layout.  Chunked / filtered datasets raise NotImplementedError naming the dataset (the reference never writes them).

UNVERIFIED AGAINST libhdf5: neither h5py nor any libhdf5 tool exists in the build image, so every fixture under tests/golden/h5
was written by this module and the round trip reader <-> writer is all the tests can check; the byte layout was checked by
hand against the format specification only.  `datasets/pix2pix.py` prefers h5py whenever it is importable; before relying on
this module for real RESIDE files, read one h5py-written sample with it and open one file it wrote with h5py / h5dump.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5FormatError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.so = self.sl = 8      # size of offsets / lengths
        self.base = 0

    def u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    def addr(self, off):
        v = self.u(off, self.so)
        return None if v == (1 << (8 * self.so)) - 1 else v + self.base

    # ---- superblock --------------------------------------------------------------------------
    def root_header(self):
        b = self.b
        pos = 0
        while b[pos:pos + 8] != SIGNATURE:       # the superblock may sit at 0, 512, 1024, ... (user block)
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(b):
                raise H5FormatError("not an HDF5 file (signature not found)")
        ver = b[pos + 8]
        if ver in (0, 1):
            self.so, self.sl = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self.u(p, self.so)
            p += 4 * self.so                     # base, free-space info, end of file, driver info
            # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
            return self.addr(p + self.so)
        if ver in (2, 3):
            self.so, self.sl = b[pos + 9], b[pos + 10]
            p = pos + 12
            self.base = self.u(p, self.so)
            return self.addr(p + 3 * self.so)    # base, superblock extension, end of file, root object header
        raise H5FormatError("unsupported superblock version %d" % ver)

    # ---- object headers ----------------------------------------------------------------------
    def messages(self, addr):
        """-> list of (type, bytes) of the object header at `addr`, continuation blocks followed."""
        b = self.b
        out = []
        if b[addr:addr + 4] == b"OHDR":          # version 2
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16                          # access, modification, change, birth times
            if flags & 0x10:
                p += 4                           # max compact / min dense attributes
            nsz = 1 << (flags & 3)
            size0 = self.u(p, nsz)
            p += nsz
            blocks = [(p, p + size0)]
            track = bool(flags & 0x04)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], self.u(p + 1, 2), b[p + 3]
                    p += 4 + (2 if track else 0)
                    data = b[p:p + msize]
                    p += msize
                    if mtype == 0x10:            # continuation: offset, length of an "OCHK" block (+4 signature, -4 checksum)
                        caddr, clen = self.addr_from(data, 0), int.from_bytes(data[self.so:self.so + self.sl], "little")
                        blocks.append((caddr + 4, caddr + clen - 4))
                    elif mtype != 0:
                        out.append((mtype, bytes(data)))
            return out
        if b[addr] != 1:
            raise H5FormatError("object header version %d at %d" % (b[addr], addr))
        nmsg = self.u(addr + 2, 2)
        size0 = self.u(addr + 8, 4)
        blocks = [(addr + 16, addr + 16 + size0)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                mtype, msize = self.u(p, 2), self.u(p + 2, 2)
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x10:
                    caddr, clen = self.addr_from(data, 0), int.from_bytes(data[self.so:self.so + self.sl], "little")
                    blocks.append((caddr, caddr + clen))
                elif mtype != 0:
                    out.append((mtype, bytes(data)))
        return out

    def addr_from(self, data, off):
        v = int.from_bytes(data[off:off + self.so], "little")
        return None if v == (1 << (8 * self.so)) - 1 else v + self.base

    # ---- groups ------------------------------------------------------------------------------
    def links(self, header_addr):
        """name -> object header address of the group whose header is at header_addr."""
        out = {}
        for mtype, data in self.messages(header_addr):
            if mtype == 0x11:                    # symbol table: B-tree + local heap
                btree, heap = self.addr_from(data, 0), self.addr_from(data, self.so)
                if self.b[heap:heap + 4] != b"HEAP":
                    raise H5FormatError("local heap signature missing at %d" % heap)
                heap_data = self.addr(heap + 8 + 2 * self.sl)
                self._walk_btree(btree, heap_data, out)
            elif mtype == 0x06:                  # link message (new-style compact groups)
                name, target = self._link(data)
                if target is not None:
                    out[name] = target
            elif mtype == 0x02:                  # link info: dense storage when a fractal heap address is set
                flags = data[1]
                p = 2 + (8 if flags & 1 else 0)
                if self.addr_from(data, p) is not None:
                    raise NotImplementedError("group with dense link storage (more than 8 links, libver='latest')")
        return out

    def _link(self, data):
        flags = data[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = data[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        nlen = int.from_bytes(data[p:p + nsz], "little")
        p += nsz
        name = data[p:p + nlen].decode("utf-8")
        p += nlen
        return name, (self.addr_from(data, p) if ltype == 0 else None)

    def _walk_btree(self, addr, heap_data, out):
        b = self.b
        if b[addr:addr + 4] != b"TREE":
            raise H5FormatError("B-tree signature missing at %d" % addr)
        level, used = b[addr + 5], self.u(addr + 6, 2)
        p = addr + 8 + 2 * self.so               # skip sibling pointers
        for i in range(used):
            child = self.addr(p + self.sl + i * (self.sl + self.so))   # key0, child0, key1, child1, ...
            if level > 0:
                self._walk_btree(child, heap_data, out)
                continue
            if b[child:child + 4] != b"SNOD":
                raise H5FormatError("symbol-table node signature missing at %d" % child)
            nsym = self.u(child + 6, 2)
            for e in range(nsym):
                q = child + 8 + e * (2 * self.so + 24)
                noff = self.u(q, self.so)
                end = heap_data + noff
                while b[end] != 0:
                    end += 1
                out[bytes(b[heap_data + noff:end]).decode("utf-8")] = self.addr(q + self.so)

    # ---- datasets ----------------------------------------------------------------------------
    def dataset(self, header_addr, name="?"):
        shape = dtype = layout = None
        for mtype, data in self.messages(header_addr):
            if mtype == 0x01:
                ver, rank = data[0], data[1]
                p = 8 if ver == 1 else 4
                shape = tuple(int.from_bytes(data[p + i * self.sl:p + (i + 1) * self.sl], "little") for i in range(rank))
            elif mtype == 0x03:
                dtype = self._dtype(data, name)
            elif mtype == 0x08:
                layout = data
            elif mtype == 0x0B:
                raise NotImplementedError("dataset '%s' has a filter pipeline (compression); only plain contiguous "
                                          "datasets, as generate_testsample.py writes them, are supported" % name)
        if shape is None or dtype is None or layout is None:
            raise H5FormatError("'%s' is not a dataset (dataspace / datatype / layout message missing)" % name)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        ver = layout[0]
        if ver == 3 or ver == 4:
            cls = layout[1]
            if cls == 1:
                a = self.addr_from(layout, 2)
                if a is None:                    # never written: libhdf5 returns the fill value (zeros)
                    return np.zeros(shape, dtype)
                return np.frombuffer(self.b, dtype, n, a).reshape(shape).copy()
            if cls == 0:
                size = int.from_bytes(layout[2:4], "little")
                return np.frombuffer(layout[4:4 + size], dtype, n).reshape(shape).copy()
            raise NotImplementedError("dataset '%s' is chunked; only contiguous / compact layouts are supported" % name)
        if ver in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls == 1:
                a = self.addr_from(layout, 8)
                return np.frombuffer(self.b, dtype, n, a).reshape(shape).copy()
            raise NotImplementedError("dataset '%s': layout version %d class %d" % (name, ver, cls))
        raise H5FormatError("data layout message version %d" % ver)

    @staticmethod
    def _dtype(data, name):
        cls, bits0, size = data[0] & 0x0F, data[1], int.from_bytes(data[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 1 and size in (2, 4, 8):
            return np.dtype(order + "f%d" % size)
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(order + ("i" if bits0 & 0x08 else "u") + "%d" % size)
        raise NotImplementedError("dataset '%s': datatype class %d of %d bytes" % (name, cls, size))


class File:
    """Read-only, h5py-like: `with h5lite.File(path) as f: f['haze'][:]`, `f.keys()`, `'gt' in f`."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("h5lite.File is read-only; use h5lite.write(path, {...})")
        with open(path, "rb") as fh:
            self._r = _Reader(memoryview(fh.read()))
        self._links = self._r.links(self._r.root_header())
        self._cache = {}

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        return name in self._links

    def __getitem__(self, name):
        if name not in self._cache:
            if name not in self._links:
                raise KeyError("no object '%s' in the root group (have %s)" % (name, self.keys()))
            self._cache[name] = self._r.dataset(self._links[name], name)
        return self._cache[name]

    def close(self):
        self._r = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


# ------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


def _datatype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        # class 1 (floating point) version 1; bit field: little endian, mantissa normalisation 2 (implied msb), sign bit
        # location; properties: bit offset, precision, exponent location / size, mantissa location / size, exponent bias
        exp, man, bias = ((8, 23, 127) if dt.itemsize == 4 else (11, 52, 1023))
        body = struct.pack("<BBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize)
        body += struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, man, exp, 0, man, bias)
        return _msg(0x03, body, flags=1)
    if dt.kind in "iu":
        body = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        body += struct.pack("<HH", 0, 8 * dt.itemsize)
        return _msg(0x03, body, flags=1)
    raise NotImplementedError("h5lite.write: dtype %s" % dt)


def write(path, datasets):
    """datasets: {name: array}.  Writes a version-0-superblock HDF5 file with one contiguous dataset per entry in the
    root group -- the shape of file `h5py.File(path, 'w').create_dataset(name, data=array)` produces."""
    names = sorted(datasets)                     # symbol-table entries are ordered by name
    if not names or len(names) > 8:
        raise ValueError("h5lite.write: 1..8 datasets (one symbol-table node)")
    arrays = {}
    for k in names:
        a = np.asarray(datasets[k])
        arrays[k] = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))
    # local heap data segment: "" at offset 0, then the names, each null-terminated and padded to 8 bytes
    heap = bytearray(8)
    name_off = {}
    for k in names:
        name_off[k] = len(heap)
        heap += _pad8(k.encode("utf-8") + b"\0")
    heap_free = len(heap)
    heap += struct.pack("<QQ", 1, 32)            # one free block: next = 1 (H5HL_FREE_NULL), size 32
    heap += b"\0" * 16
    # layout of the file
    SUPER, ROOT_HDR = 0, 96
    root_hdr_size = 16 + 8 + 16                  # prefix + one symbol-table message
    BTREE = ROOT_HDR + root_hdr_size
    btree_size = 8 + 16 + (2 * 16 + 1) * 8 + 2 * 16 * 8
    HEAP = BTREE + btree_size
    HEAP_DATA = HEAP + 32
    SNOD = HEAP_DATA + len(heap)
    snod_size = 8 + 8 * 40
    pos = SNOD + snod_size
    hdr_addr, hdr_bytes, data_addr = {}, {}, {}
    for k in names:
        a = arrays[k]
        dims = a.shape
        dataspace = struct.pack("<BBB5x", 1, len(dims), 1) + b"".join(struct.pack("<Q", d) for d in dims) * 2
        fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)     # version 2, allocate late, write if set, defined, size 0
        hdr_addr[k] = pos
        head = [_msg(0x01, dataspace), _datatype_msg(a.dtype), _msg(0x05, fill, flags=1)]
        hdr_len = 16 + sum(len(m) for m in head) + 8 + 24
        data_addr[k] = (pos + hdr_len + 7) // 8 * 8
        layout = struct.pack("<BBQQ", 3, 1, data_addr[k], a.nbytes)
        hdr = _object_header(head + [_msg(0x08, layout)])
        assert len(hdr) == hdr_len, (len(hdr), hdr_len)
        hdr_bytes[k] = hdr
        pos = data_addr[k] + (a.nbytes + 7) // 8 * 8
    eof = pos
    out = bytearray(eof)
    # superblock
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, ROOT_HDR, 1, 0) + struct.pack("<QQ", BTREE, HEAP)
    out[SUPER:SUPER + len(sb)] = sb
    out[ROOT_HDR:ROOT_HDR + root_hdr_size] = _object_header([_msg(0x11, struct.pack("<QQ", BTREE, HEAP))])
    # B-tree: one leaf entry -> the symbol-table node; key[0] = "" (offset 0), key[1] = the largest name
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, SNOD, name_off[names[-1]])
    out[BTREE:BTREE + len(bt)] = bt
    out[HEAP:HEAP + 32] = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), heap_free, HEAP_DATA)
    out[HEAP_DATA:HEAP_DATA + len(heap)] = heap
    sn = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for k in names:
        sn += struct.pack("<QQII16x", name_off[k], hdr_addr[k], 0, 0)
    out[SNOD:SNOD + len(sn)] = sn
    for k in names:
        out[hdr_addr[k]:hdr_addr[k] + len(hdr_bytes[k])] = hdr_bytes[k]
        raw = arrays[k].tobytes()
        out[data_addr[k]:data_addr[k] + len(raw)] = raw
    with open(path, "wb") as fh:
        fh.write(out)
    return path
