#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 300 python tools/wino_debug.py > gpurun_out/r3_wino_dbg.log 2>&1; echo "rc=$?"
cat gpurun_out/r3_wino_dbg.log | cut -c1-900
