set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BMI_BENCH_SHARE_DEVICE=1 BMI_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --batch 2048 --inverse-sharded --inverse-sizes 2 --shard-threshold 48 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_2rank.log | tail -3 | cut -c1-2500
