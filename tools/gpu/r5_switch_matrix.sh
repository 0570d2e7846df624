# round 5 (VERDICT r4 #9b): the GPU parity tests with every remaining numerics-relevant switch of the SHIPPED library flipped, one at a time.
# One call may run 20 minutes: sections conv1 / conv2 / lstm / heads (argument), appended to gpurun_out/r5sw/matrix_<section>.log
cd $GRAFT_REPO_ROOT
SEC=${1:-conv1}
O=gpurun_out/r5sw
mkdir -p $O
M=$O/matrix_$SEC.log
: > $M
run() {   # run "<ENV=V ...>" <pytest args...>
  local envs="$1"; shift
  local tag=$(echo "$envs" | tr ' =' '__')
  ( export $envs; timeout -k 10 500 python -m pytest "$@" -q -m gpu -p no:cacheprovider > $O/$tag.log 2>&1 ); local rc=$?
  echo "$envs :: $* :: rc=$rc :: $(grep -E 'passed|failed' $O/$tag.log | tail -1)" | tee -a $M
  grep -E "^FAILED" $O/$tag.log | head -5 | tee -a $M
}
CONV="tests/test_conv_gpu.py tests/test_vision_gpu.py tests/test_pool_gpu.py"
if [ $SEC = conv1 ]; then for e in NNL_CONV_WINO=0 NNL_CONV_WINO=2 NNL_CONV_WINO=3 NNL_CONV_WINO2=0 NNL_WGRAD_WINO=0 NNL_WGRAD_WINO=2 NNL_WGRAD_WINO2D=0 NNL_WGRAD_WINO2D=2 NNL_IGEMM_BALANCE=0; do
  run "$e" $CONV
done; fi
if [ $SEC = conv2 ]; then for e in NNL_WINO_BALANCE=0 NNL_WINO2_POS=0 NNL_WINO2_POS=1 NNL_IGEMM_KTAIL=0; do
  run "$e" $CONV
done; fi
# after the round's last kernel changes (igemm_wgrad2d.h / igemm_wgrad.h staging): the weight-gradient switches once more
if [ $SEC = wgrad ]; then for e in NNL_WGRAD_WINO2D=2 NNL_WGRAD_WINO2D=0 NNL_WGRAD_WINO=0; do
  run "$e" tests/test_conv_gpu.py tests/test_vision_gpu.py
done; fi
if [ $SEC = lstm ]; then for e in NNL_LSTM_PERSIST=0 NNL_LSTM_PERSIST=1 NNL_LSTM_PERSIST=3 NNL_LSTM_FUSED_BWD=1; do
  run "$e" tests/test_text.py
done; fi
HEADS="tests/test_collab_gpu.py tests/test_tabular.py tests/test_bn_gpu.py tests/test_optim_gpu.py tests/test_fcnet_fit_curves.py tests/test_step_loss_parity.py"
if [ $SEC = heads ]; then for e in NNL_SCATTER_ATOMIC=1 NNL_EMBDOT_SCAN=0 NNL_TAB_SCAN=0 NNL_IGEMM_KTAIL=0 NNL_BN_EPI_STATS=0 NNL_FUSED_OPTIM=0 NNL_DEFAULT_GRAPHS=0; do
  run "$e" $HEADS
done; fi
