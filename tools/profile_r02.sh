#!/bin/bash
# Round-2 profiles on the GPU box: kernel statistics of the bench command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE,
# separate runs), sampler kernel statistics, training-step rates.  Outputs under gpurun_out/r02/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
echo "kernel stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 2
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $O/pmc_write.json 2> $O/pmc_write.err || exit 3
echo "write pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_samp -o samp -- python3 tools/bench_sampler.py --steps 5 > $O/sampler.json 2> $O/sampler.err || exit 4
echo "sampler stats done"
timeout -k 10 300 python3 bench_train.py > $O/bench_train_1gpu.json 2> $O/bench_train_1gpu.err || exit 5
timeout -k 10 300 python3 bench_train.py --gpus 2 --backend gloo --one-device > $O/bench_train_dp2_gloo_onedevice.json 2> $O/bench_train_dp2.err || exit 6
echo "train done"
ls -la $O $O/*/ | head -40
