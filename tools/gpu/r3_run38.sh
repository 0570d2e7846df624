#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 200 ./tools/kloop_probe > gpurun_out/r3_kloop_xb.log 2>&1; echo "probe rc=$?"
grep -A1 "BK32 REG" gpurun_out/r3_kloop_xb.log
