# Rehearsal of bench.py's N > 1 path on ONE GPU (the ranks share it; gloo, because two ranks on one GPU cannot form an RCCL
# communicator -- GENIE_BENCH_BACKEND is a rehearsal switch only).  usage (on the GPU box): bash tools/rehearse_multi.sh <outdir>
O=$1; mkdir -p $O
export GENIE_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 5 --warmup 1 > $O/rehearse_cfg1_x2.json 2> $O/rehearse_cfg1_x2.err || exit 1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 4 --config 4 --reads 16000000 --steps 2 --warmup 1 > $O/rehearse_cfg4_x4.json 2> $O/rehearse_cfg4_x4.err || exit 1
tail -n 1 $O/rehearse_cfg1_x2.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg1 x2', j['n_gpus'], j['value']/1e9, j['config']['backend'], j['config']['world_size'], j['config']['broadcast_ms']); [print('  other', o['baseline_configs_index'], o['n_gpus'], o['scaling'], o['reads_per_step_all_gpus'], o['value']/1e9, o['broadcast_ms']) for o in j['other_configs']]"
tail -n 1 $O/rehearse_cfg4_x4.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg4 x4', j['n_gpus'], j['value']/1e9, j['config']['backend'], j['config']['world_size'], j['config']['broadcast_ms'], j['config']['reads_per_step_all_gpus'])"
