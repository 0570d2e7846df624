cd $GRAFT_REPO_ROOT
O=gpurun_out/r5pos
mkdir -p $O
for bs in 64 32; do
timeout -k 10 300 python tools/bench_conv.py --bs $bs --only l2_3x3,l3_3x3,l4_3x3 --ab "NNL_WINO2_POS=0,-1,1" > $O/pos5_bs$bs.log 2>&1; echo "rc=$?"; grep -v "s2 \|wgrad\|amdgpu" $O/pos5_bs$bs.log
done
for v in 0 -1; do
  NNL_WINO2_POS=$v timeout -k 10 300 python tools/bench_small_batch.py --bs 8,16,32,64 --steps 40 > $O/sb5_pos$v.log 2>&1; echo "pos=$v rc=$?"
  python - <<PY
import json
for l in open('$O/sb5_pos$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['bs'], 'graph', d.get('graph_ms'), 'eager', d['eager_ms'], 'kern', d['nnl_kernel_ms'], {k:v['ms_per_step'] for k,v in d['by_kind'].items() if k.startswith('conv')})
PY
done
