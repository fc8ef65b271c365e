set -e
export RHJ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --tuples 200000000 --cpu-sample 0 --no-extras > gpurun_out/r03_g2_200m.json 2> gpurun_out/r03_g2_200m.err
unset RHJ_BENCH_BACKEND
python bench.py --steps 5 --warmup 2 --tuples 200000000 --cpu-sample 0 --no-extras --no-auto > gpurun_out/r03_n1_200m.json 2> gpurun_out/r03_n1_200m.err
python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-extras > gpurun_out/r03_n1_1b.json 2> gpurun_out/r03_n1_1b.err
