"""Binary PLY point-cloud files: the storage format of the reference's prepared datasets
(PointSegment/helper_ply.py:116-196 read_ply, :217-328 write_ply; written by utils/dataPrepareBraTS.py:102, read by
runBraTS.py:99-104 / runPancreas.py:103-108).  Own implementation over numpy structured dtypes; same call signatures,
same header layout and byte order choices, so files are interchangeable in both directions
(tests/test_host_logic.py checks a file written by the reference and byte-equality of this writer's output)."""
import sys

import numpy as np

_SCALARS = {"int8": "i1", "char": "i1", "uint8": "u1", "uchar": "u1", "int16": "i2", "short": "i2", "uint16": "u2",
            "ushort": "u2", "int32": "i4", "int": "i4", "uint32": "u4", "uint": "u4", "float32": "f4", "float": "f4",
            "float64": "f8", "double": "f8"}
_ENDIAN = {"binary_little_endian": "<", "binary_big_endian": ">"}


def _read_header(f):
    """-> (byte-order prefix, {element name: count}, [(vertex property name, numpy code)])"""
    if b"ply" not in f.readline():
        raise ValueError("not a PLY file: the first line must be 'ply'")
    fmt = f.readline().split()[1].decode()
    if fmt not in _ENDIAN:
        raise ValueError("only binary PLY files are supported (format is %r)" % fmt)
    order = _ENDIAN[fmt]
    counts, props, element = {}, [], None
    while True:
        line = f.readline()
        if not line:
            raise ValueError("PLY header is not terminated by end_header")
        tok = line.split()
        if not tok:
            continue
        if tok[0] == b"end_header":
            break
        if tok[0] == b"element":
            element = tok[1].decode()
            counts[element] = int(tok[2])
        elif tok[0] == b"property" and element == "vertex":
            props.append((tok[2].decode(), order + _SCALARS[tok[1].decode()]))
    return order, counts, props


def read_ply(filename, triangular_mesh=False):
    """Structured array with one named field per vertex property; with triangular_mesh=True a list
    [vertices, faces int32 [F,3]]."""
    with open(filename, "rb") as f:
        order, counts, props = _read_header(f)
        vertices = np.fromfile(f, dtype=props, count=counts.get("vertex", 0))
        if not triangular_mesh:
            return vertices
        face_t = [("k", order + "u1"), ("v1", order + "i4"), ("v2", order + "i4"), ("v3", order + "i4")]
        faces = np.fromfile(f, dtype=face_t, count=counts.get("face", 0))
        return [vertices, np.stack([faces["v1"], faces["v2"], faces["v3"]], axis=1)]


def _columns(field_list):
    if not isinstance(field_list, (list, tuple)):
        field_list = [field_list]
    cols = []
    for arr in field_list:
        arr = np.asarray(arr)
        if arr.ndim > 2:
            return None
        arr = arr.reshape(len(arr), -1)
        cols += [arr[:, j] for j in range(arr.shape[1])]
    return cols


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """Every 1-D array and every column of a 2-D array in `field_list` becomes one vertex property, named by
    `field_names` in order.  Returns True, or False (after printing why) when the fields do not line up."""
    cols = _columns(field_list)
    if cols is None:
        print("fields have more than 2 dimensions")
        return False
    if len({len(c) for c in cols}) > 1:
        print("wrong field dimensions")
        return False
    if len(cols) != len(field_names):
        print("wrong number of field names")
        return False
    if not filename.endswith(".ply"):
        filename += ".ply"
    n = len(cols[0]) if cols else 0
    head = ["ply", "format binary_%s_endian 1.0" % sys.byteorder, "element vertex %d" % n]
    head += ["property %s %s" % (c.dtype.name, name) for c, name in zip(cols, field_names)]
    if triangular_faces is not None:
        head += ["element face %d" % len(triangular_faces), "property list uchar int vertex_indices"]
    head.append("end_header")
    table = np.empty(n, dtype=[(name, c.dtype.str) for c, name in zip(cols, field_names)])
    for c, name in zip(cols, field_names):
        table[name] = c
    with open(filename, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        table.tofile(f)
        if triangular_faces is not None:
            tri = np.asarray(triangular_faces, dtype=np.int32)
            rec = np.empty(len(tri), dtype=[("k", "uint8"), ("0", "int32"), ("1", "int32"), ("2", "int32")])
            rec["k"] = 3
            rec["0"], rec["1"], rec["2"] = tri[:, 0], tri[:, 1], tri[:, 2]
            rec.tofile(f)
    return True
