cd /root/repo
mkdir -p gpurun_out
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 > gpurun_out/r04_bench_nccl_1rank.json 2> gpurun_out/r04_bench_nccl_1rank.err
tail -c 600 gpurun_out/r04_bench_nccl_1rank.json; tail -3 gpurun_out/r04_bench_nccl_1rank.err
timeout 900 python -m pytest tests/test_multi_rank_gpu.py -x -q -k "dry_run" 2>&1 | tail -4
