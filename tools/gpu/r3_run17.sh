# round 3, GPU call 17: SQ stall counters of the 64x64 taps kernel in isolation (l3 3x3 fwd = BK32, l2 3x3 fwd = BK16)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | head -c 6000; echo
for layer in l3_3x3 l2_3x3; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" ; do
  d=gpurun_out/r3p/pmc_sq_$layer; mkdir -p $d
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python3 tools/prof_one.py $layer fwd --iters 3 > /dev/null 2>$d/err.txt
  python - <<PY
import csv,collections,glob
f=glob.glob('$d/*counter_collection.csv')
if f:
    t=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f[0])):
        if 'igemm_taps' in r['Kernel_Name']:
            t[r['Counter_Name']]+=float(r['Counter_Value'])
    print('$layer', {k:int(v/3) for k,v in t.items()})
else:
    print('$layer', 'no csv for', '$set', open('$d/err.txt').read()[-300:])
PY
  rm -rf $d
done
done
