import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
m = entry.load_package()
n = 1 << 28
keys = np.random.default_rng(1).integers(0, 2**32, size=n, dtype=np.uint32)
with m.Engine("uint32", n) as e:
    e.set_option(m.OPT_PROFILE, 1)
    e.upload(keys, None); e.sort(); e.timings(reset=True)
    for _ in range(5):
        e.upload(keys, None); e.sort()
    a = e.timings(reset=True)
    e.upload(keys, None)
    for _ in range(6):
        e.sort()
    b = e.timings(reset=True)
    print("upload before every sort: reorder avg %.4f ms (min %.4f max %.4f), histogram %.4f" % (a.reorder.avg_ms, a.reorder.min_ms, a.reorder.max_ms, a.histogram.avg_ms))
    print("back-to-back sorts      : reorder avg %.4f ms (min %.4f max %.4f), histogram %.4f" % (b.reorder.avg_ms, b.reorder.min_ms, b.reorder.max_ms, b.histogram.avg_ms))
