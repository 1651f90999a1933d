#!/bin/bash
# round 4, call h: the whole GPU suite on the final build, then a soak
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout -k 10 200 python tools/soak.py 150 4 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak.txt
