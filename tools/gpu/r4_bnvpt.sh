cd $GRAFT_REPO_ROOT
for cfg in "2 8192" "4 8192" "8 8192" "1 16384" "2 2048" "4 4096" "2 8192"; do set -- $cfg; echo "VPT=$1 MAXB=$2"; NNL_BN_VPT=$1 NNL_BN_MAXB=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sweep --configs none 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('by_kind',{}).get('elementwise') if isinstance(d['roofline'].get('by_kind'),dict) else '')"; done
