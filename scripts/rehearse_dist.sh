set -e
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/dist.log 2>&1 && tail -3 gpurun_out/dist.log &&
MBPO_BENCH_SHARE_GPU=1 MBPO_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 2 > gpurun_out/bench2.json 2> gpurun_out/bench2.err && cat gpurun_out/bench2.json &&
timeout -k 10 300 python bench.py --steps 30 --warmup 3 > gpurun_out/bench1.json 2> gpurun_out/bench1.err && cat gpurun_out/bench1.json
