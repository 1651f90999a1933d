#!/bin/bash
# idle_ab.sh VARIANT_SO [rounds]: the scatter after an upload (idle clock) and back to back, product vs variant, interleaved
V=$1; R=${2:-3}
for i in $(seq 1 $R); do
  unset RSX_LIB; echo "product: $(python tools/idle_probe.py 2>/dev/null | grep 'upload before' | cut -c1-70)"
  export RSX_LIB=$V; echo "variant: $(python tools/idle_probe.py 2>/dev/null | grep 'upload before' | cut -c1-70)"
done
unset RSX_LIB
