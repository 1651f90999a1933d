#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$(pwd)/gpurun_out/r03h; mkdir -p $O
export MODE_ENGINES=6 MODE_PAYLOAD=1 MODE_BITS=8 MODE_SORTS=10
for mode in 0 1 2; do for chunk in 0; do
  echo "== u32+payload 8-bit, RSX_ALLOC_MODE=$mode chunk=${chunk}MB"; RSX_ALLOC_MODE=$mode RSX_ALLOC_CHUNK_MB=$chunk python tools/mode_probe.py 2>&1 | grep -E "engine|rror" | tee -a $O/modes_alloc.txt
done; done
for chunk in 16 256; do
  echo "== u32+payload 8-bit, RSX_ALLOC_MODE=2 chunk=${chunk}MB"; RSX_ALLOC_MODE=2 RSX_ALLOC_CHUNK_MB=$chunk python tools/mode_probe.py 2>&1 | grep -E "engine|rror" | tee -a $O/modes_alloc.txt
done
echo "== u32 keys-only 4-bit headline shape, modes 0 / 2"; for mode in 0 2; do RSX_ALLOC_MODE=$mode MODE_PAYLOAD=0 MODE_BITS=4 MODE_SORTS=20 python tools/mode_probe.py 2>&1 | grep -E "engine|rror" | tee -a $O/modes_alloc_headline.txt; done
echo "== u64+payload 4-bit (config 3), modes 0 / 2"; for mode in 0 2; do RSX_ALLOC_MODE=$mode MODE_DTYPE=uint64 MODE_BITS=4 MODE_SORTS=6 MODE_ENGINES=5 python tools/mode_probe.py 2>&1 | grep -E "engine|rror" | tee -a $O/modes_alloc_c3.txt; done
# per-instance counters (json keeps the dimensions)
cd /tmp && export TMPDIR=/tmp
MODE_SORTS=3 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL --output-format json -d $O/inst -- python3 $GRAFT_REPO_ROOT/tools/mode_probe.py > $O/inst.txt 2> $O/inst.err
ls -la $O/inst/*/* | head; python3 - <<PY
import json, glob, collections
f = glob.glob("$O/inst/*/*results.json")
print(f)
if f:
    d = json.load(open(f[0]))
    top = d["rocprofiler-sdk-tool"][0]
    print(list(top.keys()))
    cc = top.get("callback_records", {}).get("counter_collection", [])
    print(len(cc))
    if cc:
        print(json.dumps(cc[0])[:1500])
PY
