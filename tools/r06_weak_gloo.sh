# bench.py with 8 gloo ranks time-slicing ONE GPU (a functional check of the N > 1 paths, not a measurement of xGMI):
#   $1 = cams per rank, $2 = edges per rank; --scaling weak (the graph grown with the rank count) and the default strong run with its weak leg
cd "$(dirname "$0")/.."
export GSFM_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 GSFM_BENCH_PEER=0
C=${1:-12500}; E=${2:-1250000}; N=${3:-8}
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus $N --scaling weak --cams $C --edges $E --steps 2 --warmup 1
