set -e
show() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'rccl_ranks', d['config'].get('rccl_ranks'), 'recall', d['config'].get('recall_at_10'))" $1; }
VDB_BENCH_EXCHANGE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r03_rehearsal_one_rank_rccl.json 2> gpurun_out/r03_rehearsal_one_rank_rccl.log
show gpurun_out/r03_rehearsal_one_rank_rccl.json
VDB_BENCH_EXCHANGE=1 VDB_SHARD_EXCHANGE=torch timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --ef 144 > gpurun_out/reh_torch.json 2> gpurun_out/reh_torch.log
show gpurun_out/reh_torch.json
