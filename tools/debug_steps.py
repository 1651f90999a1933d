"""Step-by-step bring-up of the C-ABI path with a sync after every launch (debug aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
m = entry.load_package()
def say(*a):
    print(*a, flush=True)
say("devices", m.device_count(), m.device_name(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dt = sys.argv[2] if len(sys.argv) > 2 else "uint32"
rng = np.random.default_rng(0)
info = np.iinfo(dt)
keys = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
e = m.Engine(dt, n)
say("created")
e.upload(keys); say("uploaded")
g = e.geometry(); say("tiles", g.num_tiles, "tile", g.tile_keys)
e.histogram(0); e.sync(); say("histogram ok")
_, t = e.download(hist_cap=16 * g.num_tiles); say("table sum", int(t.sum()), t[:8])
e.scan(); e.sync(); say("scan ok")
e.paste(); e.sync(); say("paste ok")
_, t = e.download(hist_cap=16 * g.num_tiles); say("scanned", t[:8], t[-4:])
e.reorder(0); e.sync(); say("reorder ok")
out = e.download()
u = keys.view(np.uint32 if keys.dtype.itemsize == 4 else np.uint64)
d = (u & 15).astype(np.int64)
say("pass0 match", bool(np.array_equal(out, keys[np.argsort(d, kind='stable')])))
e.upload(keys); e.sort(); e.sync(); say("sort ok")
out = e.download()
say("sorted match", bool(np.array_equal(out, np.sort(keys))))
