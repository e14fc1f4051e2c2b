#!/bin/bash
# Rehearsal of the driver's multi-process bench launch on a box with ONE GPU: all ranks share cuda:0
# and the exchanges are staged by gloo.  Checks the flow (rendezvous, per-rank shards, exchanges,
# rank-0 JSON line) with the HIP kernels; the throughput it prints is not a measurement.
set -u
export PMESH_AMD_SHARE_GPU=1 PMESH_AMD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
out=${1:-gpurun_out/rehearse}
mkdir -p "$out"
port=29600
for n in 2 4 8; do
  for mesh in 64 512; do
    port=$((port + 1))
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $port bench.py --gpus $n --steps 5 --warmup 2 --mesh $mesh --no-cpu-baseline \
        > "$out/n${n}_mesh${mesh}.json" 2> "$out/n${n}_mesh${mesh}.err"
    echo "n=$n mesh=$mesh rc=$?"; tail -c 600 "$out/n${n}_mesh${mesh}.json"; echo
  done
done
# the pencil mesh (2 x 4 at n = 8, pipelined transposes) with PCS on the clustered set, per-particle mass
for n in 4 8; do
  port=$((port + 1))
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus $n --steps 3 --warmup 1 --mesh 256 --decomp pencil --window pcs \
      --data clustered --double 1 --mass array --no-cpu-baseline \
      > "$out/n${n}_pencil.json" 2> "$out/n${n}_pencil.err"
  echo "n=$n pencil rc=$?"; tail -c 400 "$out/n${n}_pencil.json"; echo
done
