set -e
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
P=29800
for W in 2 4; do for G in 64 128 256; do
  P=$((P+1))
  NVLLM_TP_PEER_WGS=$G timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $P tools/micro/tp_peer_time.py 131072 524288 2>&1 | grep -E "world|Error|error|assert" || true
done; done
