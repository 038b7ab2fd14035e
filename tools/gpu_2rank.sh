# two ranks sharing cuda:0 (gloo stands in for RCCL): the N > 1 bench path and the sharded encrypted inverse with the
# DEFAULT sharding (levels re-packed for 2 x 256 per round, every level wider than a round split)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BMI_BENCH_SHARE_DEVICE=1 BMI_BENCH_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 2048 --inverse-sharded --inverse-sizes 2,3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_2rank.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n_gpus', d['n_gpus'], 'world', d['world_size_seen'], 'value', round(d['value']), 'verified', d['config']['verified_decrypt'])
for k,v in d['config']['encrypted_inverse_wall_clock_sharded'].items(): print(k, 'ranks', v['ranks'], 'sharded_levels', v['sharded_levels'], 'of', v['depth'], 'evaluate_s', v['evaluate_s'], 'matches', v['matches_plaintext_circuit'])
"
