#!/bin/bash
# the N > 1 code path of bench.py on a one-GPU box: two ranks over gloo, both on device 0 (reduced pipelines so that two fit one GPU's memory; the two
# processes share the device's hardware queues, hence PP_PIPE_ALLOW_SHARED_QUEUES).  Not a performance figure.
O=gpurun_out/r4two; mkdir -p $O; export TMPDIR=/tmp
PP_BENCH_BACKEND=gloo PP_BENCH_DEVICE=0 PP_PIPE_ALLOW_SHARED_QUEUES=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
	bench.py --gpus 2 --steps 3 --warmup 1 --capacity 8192 --pipe-rows 1024 --no-cpu-baseline > $O/two_ranks.json 2> $O/two_ranks.err
echo "rc $?"; tail -c 1500 $O/two_ranks.json; tail -5 $O/two_ranks.err
