# two ranks on ONE GPU over gloo: the multi-rank code paths of bench.py with the real kernels (no RCCL: one device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo --no-cpu-baseline > gpurun_out/r3o_video2.log 2>&1; echo rc=$?; tail -1 gpurun_out/r3o_video2.log | cut -c1-300
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --workload flowbatch --pairs-per-gpu 8 > gpurun_out/r3o_flow2.log 2>&1; echo rc=$?; tail -1 gpurun_out/r3o_flow2.log | cut -c1-300
