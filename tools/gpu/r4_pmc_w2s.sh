# round 4: counter passes on one layer of the spatially staged 2-D Winograd kernel (and the 2-D kernel it succeeds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L=${1:-l1}
O=gpurun_out/r4pmc_$L
mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
P4="TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P5="FETCH_SIZE"
P6="TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o p -- python3 tools/w2s_one.py $L 3 > /dev/null 2> $O/p$i.err || { echo "pass $i failed"; tail -3 $O/p$i.err; }
done
python tools/pmc_sum.py $O/summary.txt $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 --match wino2
find $O -name "*.csv" -size +2M -delete
