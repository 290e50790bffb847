"""LDUB container: named int32 / float64 arrays in one flat binary file.

Record layout: char name[32]; int32 dtype (0 = int32, 1 = float64); int64 count; payload.
Shared by the oracle (oracle/ldu_oracle.c, oracle/ref_driver.C), the golden
fixtures under tests/golden/ and the Python host utilities.
"""
import struct
import numpy as np


def write(path, arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype.kind in "iub":
                a = a.astype(np.int32)
                dt = 0
            else:
                a = a.astype(np.float64)
                dt = 1
            nm = name.encode()[:31]
            f.write(nm + b"\0" * (32 - len(nm)))
            f.write(struct.pack("<iq", dt, a.size))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    o = 0
    while o + 44 <= len(buf):
        name = buf[o:o + 32].split(b"\0", 1)[0].decode()
        dt, n = struct.unpack_from("<iq", buf, o + 32)
        o += 44
        if dt == 0:
            out[name] = np.frombuffer(buf, dtype=np.int32, count=n, offset=o).copy()
            o += 4 * n
        else:
            out[name] = np.frombuffer(buf, dtype=np.float64, count=n, offset=o).copy()
            o += 8 * n
    return out
