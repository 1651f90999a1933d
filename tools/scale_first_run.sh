#!/bin/bash
# scale_first_run.sh [N=8] — what to run first on a node with N GPUs (never executed on this pool, which has one GPU per box).
# BASELINE config 4 (2^30 uint32 keys in total) through the default path and the three variants the budget names, plus the single-GPU line the
# scaling is measured against; one JSON line each into scale_first_run/, and a table of step time, Gkeys/s and fraction of N x the single-GPU rate.
N=${1:-8}
O=${2:-scale_first_run}
mkdir -p "$O"
cd "$(dirname "$0")/.."
python bench.py --no-cpu-baseline > "$O/n1.json" 2> "$O/n1.err" || { echo "the single-GPU line failed: $O/n1.err"; exit 1; }
run () {   # tag, env..., -- bench args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --gpus "$N" --no-cpu-baseline --steps 20 --warmup 3 "$@" > "$O/$tag.json" 2> "$O/$tag.err" || echo "  $tag failed: see $O/$tag.err"
}
run waves_4bit            RSX_STRATEGY=waves     --
run waves_8bit            RSX_STRATEGY=waves     -- --radix-bits 8
run peer_stores_4bit      RSX_STRATEGY=waves-p2p --
run peer_stores_8bit      RSX_STRATEGY=waves-p2p -- --radix-bits 8
run peer_stores_8bit_b5   RSX_STRATEGY=waves-p2p -- --radix-bits 8 --partition-bits 5
run peer_stores_8bit_b7   RSX_STRATEGY=waves-p2p -- --radix-bits 8 --partition-bits 7
python - "$O" "$N" <<'PY'
import glob, json, os, sys
out, n = sys.argv[1], int(sys.argv[2])
one = json.loads([l for l in open(os.path.join(out, "n1.json")) if l.startswith("{")][-1])
if one.get("value") is None:
    print("single GPU: rehearsal line (no value)")
else:
    print(f"single GPU: {one['value'] / 1e3:.1f} Gkeys/s ({one['ms_per_step']} ms per 2^28 keys)")
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    if f.endswith("n1.json"):
        continue
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        print(f"{os.path.basename(f):28s} no line")
        continue
    d = json.loads(lines[-1])
    if d.get("value") is None:          # a rehearsal line (RSX_BENCH_SHARED_GPU=1): checks the plumbing, carries no number
        print(f"{os.path.basename(f):28s} rehearsal, verified: {d['config']['verified'][:60]}  {d['config']['parallelism'][:90]}")
        continue
    print(f"{os.path.basename(f):28s} {d['ms_per_step']:8.3f} ms  {d['value'] / 1e3:7.1f} Gkeys/s  {d['value'] / (n * one['value']):.2f} of {n} x single  {d['config']['parallelism'][:90]}  {d.get('sharded_phases_ms')}")
PY
