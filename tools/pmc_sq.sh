#!/bin/bash
# SQ-side PMC of one bench configuration: two counter passes, per-kernel means -> stdout.
#   tools/pmc_sq.sh <tag> [bench args...]      (env vars such as RSX_LOOKAHEAD pass through)
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
cd /tmp
rocprofv3 --pmc $A --output-format csv -d $R/gpurun_out/sq_${TAG}_a -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-events "$@" > /dev/null 2> $R/gpurun_out/sq_${TAG}_a.err
rocprofv3 --pmc $B --output-format csv -d $R/gpurun_out/sq_${TAG}_b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-events "$@" > /dev/null 2> $R/gpurun_out/sq_${TAG}_b.err
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/sq_${TAG}_[ab]/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('rsx::','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
print("== $TAG")
for k,c in agg.items():
    if 'reorder' not in k and 'histogram' not in k: continue
    m={n:sum(v)/len(v) for n,v in c.items()}
    wc=m.get('SQ_WAVE_CYCLES',1)
    print(k, "dispatches", len(next(iter(c.values()))))
    print("  wave_cycles %.3e busy %.3e | wait_any %.2f wait_inst_any %.2f active_any %.2f | active_valu %.2f active_lds %.2f wait_inst_lds %.2f active_sca %.2f (fractions of wave cycles)" % (
        wc, m.get('SQ_BUSY_CYCLES',0), m.get('SQ_WAIT_ANY',0)/wc, m.get('SQ_WAIT_INST_ANY',0)/wc, m.get('SQ_ACTIVE_INST_ANY',0)/wc,
        m.get('SQ_ACTIVE_INST_VALU',0)/wc, m.get('SQ_ACTIVE_INST_LDS',0)/wc, m.get('SQ_WAIT_INST_LDS',0)/wc, m.get('SQ_ACTIVE_INST_SCA',0)/wc))
    print("  insts: valu %.3e salu %.3e lds %.3e vmem_rd %.3e vmem_wr %.3e | lds_idx_active %.3e bank_conflict %.3e (%.1f%%)" % (
        m.get('SQ_INSTS_VALU',0), m.get('SQ_INSTS_SALU',0), m.get('SQ_INSTS_LDS',0), m.get('SQ_INSTS_VMEM_RD',0), m.get('SQ_INSTS_VMEM_WR',0),
        m.get('SQ_LDS_IDX_ACTIVE',0), m.get('SQ_LDS_BANK_CONFLICT',0), 100*m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1)))
PY
