#!/bin/bash
# split-bf16 k-loop probe (on-the-fly split while staging)
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 120 ./tools/sbloop_probe > gpurun_out/r3_sbloop.log 2>&1; echo "probe rc=$?"
cat gpurun_out/r3_sbloop.log
