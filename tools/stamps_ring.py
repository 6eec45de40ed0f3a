"""Per-launch span / clock / section cycles of the patch kernel under the real bench load.
Usage: M355_S2C32_STAMPS=/tmp/ring.bin M355_S2C32_RING=64 python bench.py --serial --no-train --no-cpu-baseline; python tools/stamps_ring.py /tmp/ring.bin 64 <slots>"""
import sys
import numpy as np
path, ring = sys.argv[1], int(sys.argv[2])
raw = np.fromfile(path, dtype=np.uint64).reshape(ring, -1, 8)
for k in range(ring):
    r = raw[k]
    r = r[r[:, 7] != 0]
    if not len(r):
        continue
    a0 = (r[:, 7] & np.uint64(0xffffffff)).astype(np.int64); a1 = (r[:, 7] >> np.uint64(32)).astype(np.int64) & 0xffffffff
    cyc = r[:, :7].astype(np.int64).sum(1)
    nt = 25
    print(f"launch {k:3d}: span {(a1.max() - a0.min()) / 100:7.1f} us  entry skew max {(a0.max() - a0.min()) / 100:6.1f} us  clock {np.median(cyc / np.maximum(a1 - a0, 1)) / 10:5.2f} GHz  "
          f"prologue {int(np.median(r[:, 6]))}  per-tile sections {[int(np.median(r[:, j].astype(np.int64)) / nt) for j in range(6)]}")
