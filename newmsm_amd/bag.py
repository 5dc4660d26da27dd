"""The array container the compiled C++ host programs (tests/cpp/*.cpp, tools/cpp/registration_bench.cpp) read and write: a sequence of
`<name> <f8|i4> <count>\\n` headers, each followed by the raw little-endian values."""
import numpy as np


def write_bag(path, **arrays):
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            dt = "f8" if a.dtype.kind == "f" else "i4"
            a = a.astype(np.float64 if dt == "f8" else np.int32)
            f.write(("%s %s %d\n" % (name, dt, a.size)).encode())
            f.write(a.tobytes())


def read_bag(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            name, dt, n = line.decode().split()
            n = int(n)
            out[name] = np.frombuffer(f.read(n * (8 if dt == "f8" else 4)), dtype=np.float64 if dt == "f8" else np.int32)
    return out
