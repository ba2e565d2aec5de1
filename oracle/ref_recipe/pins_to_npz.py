"""ref_pins.bin (written by pbd_ref_dump, README.md) -> tests/golden/ref_pins_v1.npz."""
import struct
import sys

import numpy as np

DT = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64}


def main(src, dst):
    g = {}
    with open(src, "rb") as f:
        while True:
            hd = f.read(4)
            if len(hd) < 4:
                break
            name = f.read(struct.unpack("<I", hd)[0]).decode()
            dt, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack(f"<{nd}I", f.read(4 * nd))
            n = int(np.prod(dims)) if nd else 1
            g[name] = np.frombuffer(f.read(n * np.dtype(DT[dt]).itemsize), DT[dt]).reshape(dims).copy()
    np.savez_compressed(dst, **g)
    print("wrote", dst, len(g), "arrays; OpenCV", bytes(g["opencv_version"]).decode())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
