#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$(pwd)/gpurun_out/r03j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL --output-format json -d $O/inst -- python3 $GRAFT_REPO_ROOT/tools/mode_probe.py > $O/inst.txt 2> $O/inst.err
echo "rocprof rc=$?"; cat $O/inst.txt; find $O/inst -type f | head; python3 - <<PY
import json, glob, collections, sys
f = glob.glob("$O/inst/**/*results.json", recursive=True)
print(f)
if not f: sys.exit(0)
d = json.load(open(f[0]))
top = d["rocprofiler-sdk-tool"][0]
print(list(top.keys()))
bufs = top.get("buffer_records", {}); cbs = top.get("callback_records", {})
print("buffer:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in bufs.items()}); print("callback:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in cbs.items()})
cc = cbs.get("counter_collection", []) or bufs.get("counter_collection", [])
print(len(cc))
if cc:
    print(json.dumps(cc[0])[:3000])
    json.dump(cc[:4], open("$O/cc_sample.json", "w"))
for k in ("counters", "agents"):
    if k in top: print(k, json.dumps(top[k])[:600])
PY
du -sh $O/inst; rm -rf $O/inst/*/*.json.bak
