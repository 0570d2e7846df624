#!/bin/bash
# de-phasing experiment: co-resident workgroups start their k loops 0..3 x stagger x 512 cycles apart
set -x
cd /root/repo; export TMPDIR=/tmp
timeout -k 10 500 python tools/bench_conv.py --bs 64 --ab NNL_IGEMM_STAGGER=0,1,2,4 > gpurun_out/r3_stagger_bs64.log 2>&1; echo "ab rc=$?"
tail -36 gpurun_out/r3_stagger_bs64.log
