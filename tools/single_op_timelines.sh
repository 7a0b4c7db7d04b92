cd /tmp && export TMPDIR=/tmp
for op in commit proof blob_proof verify_proof; do
  case $op in commit) need=k_msm_finalize;; proof) need=k_msm_finalize,k_quotient;; blob_proof) need=k_msm_finalize,k_quotient;; verify_proof) need=k_pairing;; esac
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/tlo_$op
  OP=$op rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/tlo_$op -o tl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/exp_single_call_timeline.py run > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/tlo_$op.log
  echo "==== $op: $(tail -1 $GRAFT_REPO_ROOT/gpurun_out/tlo_$op.log)"
  NEED=$need python3 $GRAFT_REPO_ROOT/tools/exp_single_call_timeline.py parse $GRAFT_REPO_ROOT/gpurun_out/tlo_$op
done
