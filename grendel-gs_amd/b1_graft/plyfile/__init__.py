"""Minimal `plyfile` stand-in for the two uses in the reference (scene/gaussian_model.py:418-769 save/load
of the model, scene/dataset_readers.py:32 SfM point clouds): read ascii / binary PLY with scalar
properties into numpy structured arrays, write binary_little_endian.  Off the hot path; numpy only."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
          "f8": "double"}


class PlyProperty:
    def __init__(self, name, dtype):
        self.name = name
        self.dtype = dtype


class PlyElement:
    def __init__(self, name, data):
        self.name = name
        self.data = data
        self.properties = [PlyProperty(n, data.dtype[n].str[1:]) for n in data.dtype.names]

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    @property
    def count(self):
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements=(), text=False):
        self.elements = list(elements)
        self.text = text

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            if f.readline().strip() != b"ply":
                raise ValueError("not a PLY file")
            fmt, elems = None, []
            while True:
                line = f.readline()
                if not line:
                    raise ValueError("unexpected end of PLY header")
                tok = line.decode("ascii").split()
                if not tok or tok[0] == "comment":
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    elems.append([tok[1], int(tok[2]), []])
                elif tok[0] == "property":
                    if tok[1] == "list":
                        raise NotImplementedError("list properties are not supported by this stand-in")
                    elems[-1][2].append((tok[2], _TYPES[tok[1]]))
                elif tok[0] == "end_header":
                    break
            out = []
            for name, count, props in elems:
                if fmt == "ascii":
                    dt = np.dtype([(n, "<" + t) for n, t in props])
                    arr = np.zeros(count, dtype=dt)
                    for i in range(count):
                        vals = f.readline().split()
                        for (n, _), v in zip(props, vals):
                            arr[n][i] = float(v)
                else:
                    end = "<" if fmt == "binary_little_endian" else ">"
                    dt = np.dtype([(n, end + t) for n, t in props])
                    arr = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count).copy()
                out.append(PlyElement(name, arr))
        return PlyData(out)

    def write(self, path):
        with open(path, "wb") as f:
            head = ["ply", "format binary_little_endian 1.0"]
            for e in self.elements:
                head.append(f"element {e.name} {len(e.data)}")
                for n in e.data.dtype.names:
                    head.append(f"property {_NAMES[e.data.dtype[n].str[1:]]} {n}")
            head.append("end_header")
            f.write(("\n".join(head) + "\n").encode("ascii"))
            for e in self.elements:
                le = e.data.astype(e.data.dtype.newbyteorder("<"), copy=False)
                f.write(le.tobytes())
