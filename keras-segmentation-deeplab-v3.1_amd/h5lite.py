"""Self-contained reader/writer for the HDF5 subset that Keras 2.2.x `save_weights` files use
(reference: model.load_weights(weights_path, by_name=True) deeplabv3p.py:465; utils.py:207,:229; the bonlime
release files named at deeplabv3p.py:42-43).  Needed because h5py is not installed next to the GPU runtime.

Subset (verified there with h5dump): superblock v0/v1, version-1 object headers (+continuation
blocks), old-style groups (symbol-table message -> v1 B-tree + local heap + SNOD nodes), contiguous or compact
little-endian datasets (f32/f64/i32/i64), attributes holding fixed-length strings, variable-length strings
(global heap; what h5py >= 3 writes for a Python list) or numbers.  Not supported (clear error): v2 object
headers / new-style groups (libver='latest'), chunked or filtered datasets.

The writer emits the classic layout (superblock v0, symbol-table groups, fixed-length NULLPAD string attributes,
contiguous IEEE_F32LE datasets), i.e. what Keras 2.2.4 + h5py 2.x produced.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


# ======================================================================================
# reader
# ======================================================================================


class Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        if self.b[:8] != SIG:
            raise H5Error("%s is not an HDF5 file" % path)
        ver = self.b[8]
        if ver not in (0, 1):
            raise H5Error("HDF5 superblock version %d (libver='latest' files) is not supported; re-save with "
                          "h5py's default libver or convert to .npz" % ver)
        so, sl = self.b[13], self.b[14]
        if (so, sl) != (8, 8):
            raise H5Error("only 8-byte offsets/lengths are supported")
        self.leaf_k, self.int_k = struct.unpack_from("<HH", self.b, 16)
        p = 24 if ver == 0 else 28  # v1 adds indexed-storage K + reserved
        self.base = struct.unpack_from("<Q", self.b, p)[0]
        root_entry = p + 32
        self.root = self._entry(root_entry)

    # -- primitives ------------------------------------------------------------------
    def _u(self, off, fmt):
        return struct.unpack_from("<" + fmt, self.b, off)

    def _entry(self, off):
        name_off, ohdr, cache, _ = self._u(off, "QQII")
        scratch = self.b[off + 24:off + 40]
        return dict(name_off=name_off, ohdr=ohdr + self.base, cache=cache, scratch=scratch)

    def _messages(self, addr):
        """yield (type, flags, bytes) of a version-1 object header, following continuation blocks"""
        if self.b[addr:addr + 4] == b"OHDR":
            raise H5Error("version-2 object headers (libver='latest') are not supported")
        ver, _, nmsg, _, hsize = self._u(addr, "BBHII")
        if ver != 1:
            raise H5Error("object header version %d is not supported" % ver)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u(p, "HHB")
                data = self.b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:  # continuation
                    caddr, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((caddr + self.base, clen))
                out.append((mtype, flags, data))
        return out

    # -- groups ----------------------------------------------------------------------
    def _heap_name(self, heap_addr, off):
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        data_addr = self._u(heap_addr + 24, "Q")[0] + self.base
        s = data_addr + off
        e = self.b.index(b"\0", s)
        return self.b[s:e].decode("utf-8")

    def _walk_btree(self, addr, heap, out):
        if self.b[addr:addr + 4] != b"TREE":
            raise H5Error("bad B-tree signature")
        ntype, level, used = self._u(addr + 4, "BBH")
        if ntype != 0:
            raise H5Error("unexpected B-tree node type %d" % ntype)
        p = addr + 24
        for i in range(used):
            child = self._u(p + 8, "Q")[0] + self.base
            p += 16
            if level > 0:
                self._walk_btree(child, heap, out)
            else:
                if self.b[child:child + 4] != b"SNOD":
                    raise H5Error("bad symbol-table node signature")
                nsym = self._u(child + 6, "H")[0]
                for k in range(nsym):
                    e = self._entry(child + 8 + 40 * k)
                    out[self._heap_name(heap, e["name_off"])] = e["ohdr"]

    def children(self, ohdr):
        """{name: object header address} of a group (empty dict for a dataset)"""
        for mtype, _, data in self._messages(ohdr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", data, 0)
                out = {}
                self._walk_btree(btree + self.base, heap + self.base, out)
                return out
            if mtype in (0x0002, 0x0006):
                raise H5Error("new-style (link message) groups are not supported")
        return {}

    def resolve(self, path):
        node = self.root["ohdr"]
        for part in [p for p in path.split("/") if p]:
            ch = self.children(node)
            if part not in ch:
                raise KeyError(path)
            node = ch[part]
        return node

    # -- datatypes / dataspaces ---------------------------------------------------------
    @staticmethod
    def _dtype(data):
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", data, 0)
        cls = cv & 0x0F
        if cls == 0:
            return ("int", np.dtype("<%s%d" % ("i" if (b0 & 0x08) else "u", size)), size)
        if cls == 1:
            return ("float", np.dtype("<f%d" % size), size)
        if cls == 3:
            return ("str", None, size)
        if cls == 9:
            return ("vlen", None, size)  # (type: bits 0-3 of b0: 1 = string)
        raise H5Error("datatype class %d is not supported" % cls)

    @staticmethod
    def _dspace(data):
        ver, rank, flags = struct.unpack_from("<BBB", data, 0)
        p = 8 if ver == 1 else 4
        dims = struct.unpack_from("<%dQ" % rank, data, p) if rank else ()
        return tuple(int(d) for d in dims)

    def _gheap_obj(self, addr, index):
        addr += self.base
        if self.b[addr:addr + 4] != b"GCOL":
            raise H5Error("bad global heap signature")
        size = self._u(addr + 8, "Q")[0]
        p = addr + 16
        while p < addr + size:
            idx, _, _, osize = self._u(p, "HHIQ")
            if idx == 0:
                break
            if idx == index:
                return self.b[p + 16:p + 16 + osize]
            p += 16 + (osize + 7) // 8 * 8
        raise H5Error("global heap object %d not found" % index)

    def _decode(self, kind, npdt, size, dims, raw):
        n = int(np.prod(dims)) if dims else 1
        if kind in ("int", "float"):
            a = np.frombuffer(raw[:n * size], dtype=npdt).reshape(dims)
            return a.copy()
        if kind == "str":
            vals = [raw[i * size:(i + 1) * size].split(b"\0")[0] for i in range(n)]
        else:  # vlen string: (length u32, heap address u64, index u32)
            vals = []
            for i in range(n):
                ln, ga, gi = struct.unpack_from("<IQI", raw, 16 * i)
                vals.append(self._gheap_obj(ga, gi)[:ln] if ln else b"")
        return np.array(vals, dtype=object) if dims else vals[0]

    def attrs(self, ohdr):
        out = {}
        for mtype, _, data in self._messages(ohdr):
            if mtype != 0x000C:
                continue
            ver = data[0]
            if ver == 1:
                _, _, nsz, tsz, ssz = struct.unpack_from("<BBHHH", data, 0)
                p, pad = 8, lambda x: (x + 7) // 8 * 8
            elif ver in (2, 3):
                _, _, nsz, tsz, ssz = struct.unpack_from("<BBHHH", data, 0)
                p, pad = (8 if ver == 2 else 9), lambda x: x
            else:
                raise H5Error("attribute message version %d is not supported" % ver)
            name = data[p:p + nsz].split(b"\0")[0].decode()
            p += pad(nsz)
            kind, npdt, size = self._dtype(data[p:p + tsz])
            p += pad(tsz)
            dims = self._dspace(data[p:p + ssz])
            p += pad(ssz)
            out[name] = self._decode(kind, npdt, size, dims, data[p:])
        return out

    def dataset(self, ohdr):
        kind = dims = layout = None
        for mtype, _, data in self._messages(ohdr):
            if mtype == 0x0001:
                dims = self._dspace(data)
            elif mtype == 0x0003:
                kind, npdt, size = self._dtype(data)
            elif mtype == 0x0008:
                layout = data
            elif mtype == 0x000B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if layout is None or kind is None:
            raise H5Error("object is not a dataset")
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise H5Error("data layout version %d is not supported" % ver)
        n = int(np.prod(dims)) if dims else 1
        if cls == 1:  # contiguous
            addr, _ = struct.unpack_from("<QQ", layout, 2)
            raw = b"" if addr == UNDEF else self.b[addr + self.base:addr + self.base + n * size]
            if addr == UNDEF:
                return np.zeros(dims, npdt)
        elif cls == 0:  # compact
            csz = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + csz]
        else:
            raise H5Error("chunked datasets are not supported (Keras save_weights writes contiguous ones)")
        return self._decode(kind, npdt, size, dims, raw)


def read_keras_weights(path):
    """-> (layer_names, {layer: [(weight_name, float32 array), ...]}) of a Keras weights (or full-model) file"""
    r = Reader(path)
    root = r.root["ohdr"]
    top = r.children(root)
    if "model_weights" in top:  # model.save(): weights nested one level down
        root = top["model_weights"]
        top = r.children(root)
    from .h5io import attr_list   # (`layer_names` / `weight_names` whole, or in Keras' pieces name0, name1, ...)
    names = attr_list(r.attrs(root), "layer_names")
    if not names:
        raise H5Error("no layer_names attribute: not a Keras weights file")
    per = {}
    for n in names:
        g = top[n]
        items = []
        for w in attr_list(r.attrs(g), "weight_names"):
            node = g
            for part in w.split("/"):
                node = r.children(node)[part]
            items.append((w, np.asarray(r.dataset(node), np.float32)))
        per[n] = items
    return names, per


# ======================================================================================
# writer
# ======================================================================================


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


def _dt_f32():
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_str(n):
    return struct.pack("<BBBBI", 0x13, 0x01, 0x00, 0x00, n)  # fixed length, NULLPAD, ASCII


def _ds(dims):
    return struct.pack("<BBBBI", 1, len(dims), 0, 0, 0) + b"".join(struct.pack("<Q", d) for d in dims)


def _attr(name, dtype, dspace, data):
    nm = name.encode() + b"\0"
    return _msg(0x000C, struct.pack("<BBHHH", 1, 0, len(nm), len(dtype), len(dspace)) + _pad8(nm) + _pad8(dtype) +
                _pad8(dspace) + data)


def _attr_strings(name, strings):
    if not strings:
        return _attr(name, _dt_str(1), _ds((0,)), b"")
    n = max(len(s) for s in strings)
    return _attr(name, _dt_str(n), _ds((len(strings),)), b"".join(s.ljust(n, b"\0") for s in strings))


def _attr_scalar_string(name, value):
    """scalar (rank-0) fixed-length string, what `f.attrs['backend'] = b'tensorflow'` produces"""
    return _attr(name, _dt_str(max(len(value), 1)), struct.pack("<BBBBI", 1, 0, 0, 0, 0), value)


class Writer:
    """Builds the file image in memory; objects are appended and addressed by their byte offset."""
    LEAF_K, INT_K = 4, 16

    def __init__(self):
        self.buf = bytearray(96)  # superblock placeholder

    def _alloc(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def _ohdr(self, msgs):
        body = b"".join(msgs)
        return self._alloc(struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\0" * 4 + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        data_addr = self._alloc(arr.tobytes()) if arr.size else UNDEF
        msgs = [_msg(0x0001, _ds(arr.shape)), _msg(0x0003, _dt_f32(), flags=1),
                _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),
                _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes))]
        return self._ohdr(msgs)

    def group(self, children, attr_msgs=()):
        """children: {name: (object header address, is_group, btree, heap)}"""
        names = sorted(children, key=lambda s: s.encode())
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\0")
        heap_seg = self._alloc(bytes(heap_data))
        # free-list head 1 == H5HL_FREE_NULL ("no free block") in libhdf5
        heap = self._alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), 1, heap_seg))
        # leaves
        cap = 2 * self.LEAF_K
        level_nodes = []  # (address, largest name offset)
        for i in range(0, max(len(names), 1), cap):
            chunk = names[i:i + cap]
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
            for n in chunk:
                ohdr, is_group, bt, hp = children[n]
                scratch = struct.pack("<QQ", bt, hp) if is_group else b"\0" * 16
                body += struct.pack("<QQII", offs[n], ohdr, 1 if is_group else 0, 0) + scratch
            body += b"\0" * (40 * (cap - len(chunk)))
            level_nodes.append((self._alloc(body), offs[chunk[-1]] if chunk else 0))
        level = 0
        node_size = 24 + (2 * self.INT_K + 1) * 8 + 2 * self.INT_K * 8
        while True:
            nxt = []
            for i in range(0, len(level_nodes), 2 * self.INT_K):
                kids = level_nodes[i:i + 2 * self.INT_K]
                body = b"TREE" + struct.pack("<BBHQQ", 0, level, len(kids), UNDEF, UNDEF) + struct.pack("<Q", 0)
                for addr, last in kids:
                    body += struct.pack("<QQ", addr, last)
                body += b"\0" * (node_size - len(body))
                nxt.append((self._alloc(body), kids[-1][1]))
            level_nodes, level = nxt, level + 1
            if len(level_nodes) == 1:
                break
        btree = level_nodes[0][0]
        ohdr = self._ohdr([_msg(0x0011, struct.pack("<QQ", btree, heap))] + list(attr_msgs))
        return ohdr, btree, heap

    def finish(self, root):
        ohdr, btree, heap = root
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INT_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, ohdr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_keras_weights(path, layers, attr_limit=None):
    """layers: [(layer_name, [(weight_name, array), ...]), ...] in model.layers order (Keras 2.2.x save_weights layout);
    string-array attributes beyond attr_limit bytes (default: Keras' HDF5_OBJECT_HEADER_LIMIT) are written in pieces"""
    from .h5io import attr_pieces
    w = Writer()
    top = {}
    for lname, weights in layers:
        tree = {}
        for wname, arr in weights:  # "conv/kernel:0" -> nested groups
            parts = wname.split("/")
            d = tree
            for part in parts[:-1]:
                d = d.setdefault(part, {})
            d[parts[-1]] = np.asarray(arr)

        def build(d):
            kids = {}
            for k, v in d.items():
                if isinstance(v, dict):
                    o, bt, hp = build(v)
                    kids[k] = (o, True, bt, hp)
                else:
                    kids[k] = (w.dataset(v), False, 0, 0)
            return w.group(kids)

        kids = {}
        for k, v in tree.items():
            if isinstance(v, dict):
                o, bt, hp = build(v)
                kids[k] = (o, True, bt, hp)
            else:
                kids[k] = (w.dataset(v), False, 0, 0)
        o, bt, hp = w.group(kids, [_attr_strings(an, piece) for an, piece in
                                   attr_pieces("weight_names", [n for n, _ in weights], attr_limit)] if weights else
                            [_attr_strings("weight_names", [])])
        top[lname] = (o, True, bt, hp)
    root = w.group(top, [_attr_strings(an, piece) for an, piece in attr_pieces("layer_names", [n for n, _ in layers], attr_limit)] +
                        [_attr_scalar_string("backend", b"tensorflow"),
                         _attr_scalar_string("keras_version", b"2.2.4")])
    with open(path, "wb") as f:
        f.write(w.finish(root))
