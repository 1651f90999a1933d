#!/bin/bash
# isa_waits.sh KERNEL_SYMBOL_REGEX: device assembly of rsx_capi.hip, then loads / stores / barriers / vmcnt waits / labels of the first matching kernel in order
# (where does a prefetch get waited for?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-strict-aliasing -I$R/include -S --cuda-device-only $R/radix-sort_amd/csrc/rsx_capi.hip -o /tmp/capi.s 2>/dev/null
start=$(grep -n -E "^$1.*:" /tmp/capi.s | head -1 | cut -d: -f1)
[ -z "$start" ] && { echo "no kernel matches $1"; exit 1; }
tail -n +$start /tmp/capi.s | awk '{print} /s_endpgm/{exit}' > /tmp/kernel.s
grep -n "global_load\|s_waitcnt vmcnt\|s_barrier\|global_store\|s_cbranch\|^\.LBB\|s_endpgm\|s_branch\|global_atomic" /tmp/kernel.s | awk '{print $1,$2,$3}' | awk '{key=$2; if(key==prev){c++} else {if(prev!="")print last, (c>1?"x"c:""); c=1} prev=key; last=$0} END{print last, (c>1?"x"c:"")}'
