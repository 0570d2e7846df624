#!/bin/bash
# Same-box A/B of two builds of libnnl_hip.so: tools/ab/libnnl_hip_base.so (built from the commit to compare against) vs the
# in-tree library.  Interleaves bench.py runs (no CPU baseline) and prints ms/step + the per-kind conv numbers.
# usage (GPU box): bash tools/ab_builds.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
  for which in base new; do
    if [ $which = base ]; then export NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_base.so; else unset NNL_LIB_PATH; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sweep --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['by_kind']
print('$which', 'ms/step %.3f' % d['ms_per_step'], 'conv %.2f ms @ %.1f TF' % (d['roofline']['conv_ms_per_step'], d['roofline']['achieved']), ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('conv_fwd','conv_dgrad','conv_wgrad')))
" || exit 1
  done
done
