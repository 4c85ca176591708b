#!/bin/bash
# Rehearsal of the paths the round-end driver can hit: smoke(), bench with a 1-rank RCCL group (collectives inside the captured
# step), 2 ranks on one GPU with the peer exchange disabled (gloo all-reduce, eager fallback).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 && tail -1 gpurun_out/smoke.log
MBPO_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_pg1.json 2> gpurun_out/bench_pg1.err && python -c "import json;d=json.load(open('gpurun_out/bench_pg1.json'));print('force-pg', d['value'], d['config']['hipgraph'], d['config']['grad_exchange'], d['params_finite'])"
MBPO_P2P_ALLREDUCE=0 MBPO_BENCH_SHARE_GPU=1 MBPO_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err && python -c "import json;d=json.load(open('gpurun_out/bench_gloo2.json'));print('gloo-2', d['value'], d['config']['hipgraph'], d['config']['grad_exchange'], d['params_finite'])"
# Round 4: the generic kernels behind the specialised ones, the one-launch-per-problem layered path, the BPTT recompute path —
# each selected by its environment switch and run through the same tests.
MBPO_BPTT_ZSTORE_MAX_MB=0 timeout -k 10 400 python -m pytest tests/test_gpu_bptt.py -x -q -m gpu > gpurun_out/fallback_bptt.log 2>&1 && tail -1 gpurun_out/fallback_bptt.log
MBPO_LAYERED_GROUP=0 timeout -k 10 400 python -m pytest tests/test_gpu_sac.py tests/test_gpu_ppo.py -x -q -m gpu -k "layered or widths" > gpurun_out/fallback_layered.log 2>&1 && tail -1 gpurun_out/fallback_layered.log
MBPO_SAC_LEAN=0 MBPO_PPO_LEAN=0 MBPO_ENS_LEAN=0 MBPO_ROLLOUT_LEAN=0 timeout -k 10 600 python -m pytest tests/test_gpu_trainer_parity.py tests/test_gpu_rollout.py tests/test_gpu_sac.py tests/test_gpu_ppo.py -x -q -m gpu > gpurun_out/fallback_generic.log 2>&1 && tail -1 gpurun_out/fallback_generic.log
