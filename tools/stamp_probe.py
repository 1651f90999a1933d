"""Where does a tile of the fused reorder spend its cycles?  Needs the diagnostic build
(tools/build_variant.sh stamps -DRSX_STAMPS) selected with RSX_LIB and RSX_STAMP_PASS=<pass>.
Prints, per phase boundary, the median / p90 over tiles of the cycles since the previous stamp, the whole
tile's residency in cycles and ns, and how many tiles were resident at once."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

m = entry.load_package()
lib = m.load_library()
lib.rsx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
lib.rsx_debug_stamps.restype = C.c_int
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
keys = np.random.default_rng(1).integers(0, 2**32, size=n, dtype=np.uint32)
names = ["start", "keys arrived", "P2 count", "barrier", "P3 scan+barrier", "P4 stage+barrier", "P5 reads+runs", "stores issued", "la atomics issued",
         "barrier", "flush issued", "all acked"]
with m.Engine("uint32", n) as e:
    e.upload(keys, None)
    for _ in range(3):
        e.sort()
    e.sync()
    tiles = e.geometry().num_tiles
    buf = np.zeros((tiles, 16), dtype=np.uint64)
    assert lib.rsx_debug_stamps(e._h, buf.ctypes.data, tiles) == 0
st = buf[:, :12].astype(np.int64)
d = np.diff(st, axis=1)
whole = st[:, 11] - st[:, 0]
ns = (buf[:, 15].astype(np.int64) - buf[:, 14].astype(np.int64)) * 10      # s_memrealtime ticks at 100 MHz
print(f"pass {os.environ.get('RSX_STAMP_PASS')}: {tiles} tiles; residency median {np.median(whole):.0f} cycles = {np.median(ns):.0f} ns (p90 {np.percentile(whole, 90):.0f} cycles), "
      f"clock {np.median(whole / np.maximum(ns, 1)):.2f} GHz")
for k in range(11):
    print(f"  {names[k]:>22} -> {names[k + 1]:<22} median {np.median(d[:, k]):7.0f}  p90 {np.percentile(d[:, k], 90):7.0f}  share {np.median(d[:, k]) / np.median(whole) * 100:5.1f} %")
t0, t1 = buf[:, 14].astype(np.int64), buf[:, 15].astype(np.int64)
span = (t1.max() - t0.min()) * 10
print(f"  launch span {span / 1e3:.1f} us; mean tiles in flight {ns.sum() / span:.0f} (= {ns.sum() / span / 256:.2f} per CU)")
