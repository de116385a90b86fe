set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the launch the driver uses for N > 1 (torch.distributed.run, one rank per process), rehearsed on ONE GPU: gloo backend, both
# ranks on cuda:0
export ES_BENCH_BACKEND=gloo ES_BENCH_SHARE_GPU=1
for wl in config3 config4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl > gpurun_out/s24_torchrun_$wl.json 2> gpurun_out/s24_torchrun_$wl.err || { tail -20 gpurun_out/s24_torchrun_$wl.err; exit 1; }
tail -c 900 gpurun_out/s24_torchrun_$wl.json; echo
done
# one rank under torchrun with the RCCL backend (world size 1): the nccl code path of the exchange
unset ES_BENCH_BACKEND ES_BENCH_SHARE_GPU
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s24_torchrun_n1.json 2> gpurun_out/s24_torchrun_n1.err || { tail -20 gpurun_out/s24_torchrun_n1.err; exit 1; }
tail -c 300 gpurun_out/s24_torchrun_n1.json
