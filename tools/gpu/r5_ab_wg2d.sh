# same-box A/B: tools/ab/libnnl_hip_base.so (previous commit) vs the in-tree library, headline and RetinaNet interleaved
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for r in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_base.so; else unset NNL_LIB_PATH; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-counters --no-sweep --configs retinanet 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['by_kind']
print('$which', 'ms/step %.3f' % d['ms_per_step'], 'conv %.2f ms' % d['roofline']['conv_ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('conv_fwd','conv_dgrad','conv_wgrad')), '| retinanet %.2f' % d['configs']['retinanet']['ms_per_step'])
" || exit 1
  done
done
