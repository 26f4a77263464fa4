"""Keras weight-file I/O for the model mirror (reference: model.load_weights(path, by_name=True)
deeplabv3p.py:465; positional load utils.py:207,:229; ModelCheckpoint(save_weights_only=True) notebook cell 5).

Layout [TF-semantics: Keras 2.2.x save_weights] — root attrs `layer_names`, `backend`, `keras_version`;
one group per layer with attr `weight_names`; datasets `/<layer>/<layer>/<var>:0`, float32, HWIO.
`.h5` files go through h5py when it is importable, otherwise through the package's own HDF5 subset
reader/writer (h5lite.py; validated against h5py/libhdf5 1.10.6 in both directions).  The same content can also be
read/written as `.npz` with keys "<layer>/<var>:0" plus "__layer_names__".
"""
import numpy as np


# Keras 2.2.4 engine/saving.py: an attribute larger than this is stored in pieces `name0`, `name1`, ... instead of `name`
# (save_attributes_to_hdf5_group / load_attributes_from_hdf5_group: HDF5 object headers hold at most 64 KB)
HDF5_OBJECT_HEADER_LIMIT = 64512


def attr_list(attrs, name):
    """the string-array attribute `name` of a Keras file — whole, or reassembled from its pieces name0, name1, ...;
    attrs: any mapping (h5py AttributeManager, h5lite's dict); [] if neither form exists"""
    dec = lambda x: x.decode() if isinstance(x, bytes) else str(x)
    if name in attrs:
        return [dec(x) for x in np.atleast_1d(attrs[name])]
    out, i = [], 0
    while "%s%d" % (name, i) in attrs:
        out += [dec(x) for x in np.atleast_1d(attrs["%s%d" % (name, i)])]
        i += 1
    return out


def attr_pieces(name, strings, limit=None):
    """[(attribute name, [bytes, ...]), ...] as Keras writes a string-array attribute: one attribute, or — if the
    fixed-length array would exceed the object-header limit — the smallest number of equal pieces that fit"""
    limit = HDF5_OBJECT_HEADER_LIMIT if limit is None else limit
    data = [x if isinstance(x, bytes) else str(x).encode() for x in strings]
    width = max([len(x) for x in data] or [1])
    if any(len(x) > limit for x in data):
        raise RuntimeError("a name in %r is longer than the HDF5 object header limit" % name)
    chunks = 1
    split = [data]
    while any(len(c) * width > limit for c in split):
        chunks += 1
        split = [list(a) for a in np.array_split(np.array(data, dtype=object), chunks)]
    if chunks == 1:
        return [(name, data)]
    return [("%s%d" % (name, i), c) for i, c in enumerate(split)]


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def _is_npz(path):
    return str(path).endswith(".npz")


def save_weights(model, path):
    layers = model.layers
    if _is_npz(path):
        d = {"__layer_names__": np.array([l.name for l in layers], dtype="S")}
        for l in layers:
            for n, w in zip(l.weights.keys(), l.get_weights()):
                d[n] = w
        np.savez(path, **d)
        return
    h5py = _h5py()
    if h5py is None:
        from . import h5lite
        h5lite.write_keras_weights(path, [(l.name, list(zip(l.weights.keys(), l.get_weights()))) for l in layers])
        return
    with h5py.File(path, "w", libver="earliest") as f:
        for an, piece in attr_pieces("layer_names", [l.name for l in layers]):
            f.attrs[an] = np.array(piece, dtype="S")
        f.attrs["backend"] = b"tensorflow"
        f.attrs["keras_version"] = b"2.2.4"
        for l in layers:
            g = f.create_group(l.name)
            names = list(l.weights.keys())
            if names:
                for an, piece in attr_pieces("weight_names", names):
                    g.attrs[an] = np.array(piece, dtype="S")
            else:
                g.attrs["weight_names"] = np.zeros((0,), "S1")
            for n, w in zip(names, l.get_weights()):
                g.create_dataset(n, data=np.asarray(w, np.float32))


def _read_file(path):
    """-> (layer_names, {layer: [(weight_name, array), ...]})"""
    if _is_npz(path):
        z = np.load(path)
        names = [n.decode() for n in z["__layer_names__"]]
        per = {n: [] for n in names}
        for k in z.files:
            if k == "__layer_names__":
                continue
            per.setdefault(k.split("/")[0], []).append((k, z[k]))
        return names, per
    h5py = _h5py()
    if h5py is None:
        from . import h5lite
        return h5lite.read_keras_weights(path)
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        names = attr_list(root.attrs, "layer_names")
        if not names:
            raise ValueError("%s: no layer_names attribute (whole or in pieces): not a Keras weights file" % path)
        per = {}
        for n in names:
            g = root[n]
            per[n] = [(w, np.asarray(g[w])) for w in attr_list(g.attrs, "weight_names")]
        return names, per


def load_weights(model, path, by_name=False):
    names, per = _read_file(path)
    if by_name:
        # Keras load_weights_from_hdf5_group_by_name: layers matched by name, others left untouched
        for l in model.layers:
            if l.weights and l.name in per and per[l.name]:
                arrs = [a for _, a in per[l.name]]
                if len(arrs) != len(l.weights):
                    raise ValueError("layer %s: file has %d arrays, model expects %d" % (l.name, len(arrs), len(l.weights)))
                l.set_weights(arrs)
        return
    # positional: weight-bearing layers zipped in order (utils.py:207)
    file_layers = [n for n in names if per.get(n)]
    model_layers = [l for l in model.layers if l.weights]
    if len(file_layers) != len(model_layers):
        raise ValueError("file holds %d weight-bearing layers, model has %d" % (len(file_layers), len(model_layers)))
    for n, l in zip(file_layers, model_layers):
        l.set_weights([a for _, a in per[n]])
