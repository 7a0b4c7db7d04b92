#!/bin/bash
# Sets-in-flight sweep of the device-resident verify bench: G batches per launch set, D sets kept in flight by one
# host thread through kzg355_verify_blob_kzg_proof_batch_many_device_submit / kzg355_verify_collect.  usage: pipeline_sweep.sh OUTFILE
OUT=${1:-gpurun_out/pipeline_sweep.txt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
echo "bench.py --no-cpu-baseline --no-host-leg --no-latency --batches-per-step G --pipeline D --steps S --warmup 4 (device-resident; columns: G, D, blobs/s, ms per step, HIP-event ms per kernel family = average per launch while the sets overlap)" > $OUT
CFGS=${CFGS:-"64,1,64 64,8,64 256,1,48 256,4,48 256,8,48 512,1,48 512,4,48 1024,1,32 1024,2,32 1024,3,32 1024,4,32 1024,6,32 2048,1,24 2048,2,24 2048,3,24 4096,1,16 4096,2,16 8192,1,10 8192,2,10"}
for cfg in $CFGS; do
  set -- ${cfg//,/ }
  python bench.py --no-cpu-baseline --no-host-leg --no-latency --batches-per-step $1 --pipeline $2 --steps $3 --warmup 4 2>/dev/null | python tools/set_size_line.py $2 >> $OUT || echo "$1 $2 FAILED" >> $OUT
  tail -1 $OUT
done
