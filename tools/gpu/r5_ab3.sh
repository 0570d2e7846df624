cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5ab3
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_text.py tests/test_detection.py tests/test_vision_gpu.py tests/test_tabular.py tests/test_fcnet_fit_curves.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export NNL_LIB_PATH=$PWD/tools/ab/libnnl_hip_base.so; else unset NNL_LIB_PATH; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-counters --no-sweep --configs retinanet,lm 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['by_kind']
print('$which', 'ms/step %.3f' % d['ms_per_step'], 'conv %.2f ms' % d['roofline']['conv_ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('conv_fwd','conv_dgrad','conv_wgrad')), '| retinanet %.2f' % d['configs']['retinanet']['ms_per_step'], '| lm %.2f' % d['configs']['lm']['ms_per_step'])
" || exit 1
  done
done
