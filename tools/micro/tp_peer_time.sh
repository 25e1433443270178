# Usage (GPU box, repo root): bash tools/micro/tp_peer_time.sh   -- the peer all-reduce at 2, 4 and 8 ranks sharing one GPU
set -e
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
P=29800
for W in 2 4 8; do
  P=$((P+1))
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $P tools/micro/tp_peer_time.py 32768 131072 524288 1048576 2>&1 | grep -E "world|Error|error|assert" || true
done
