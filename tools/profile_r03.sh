#!/bin/bash
# Round-3 profiles on the GPU box: kernel statistics of the bench command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE,
# separate runs), sampler kernel statistics and SQ counters.  Outputs under gpurun_out/r03/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || exit 1
echo "kernel stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 2
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $O/pmc_write.json 2> $O/pmc_write.err || exit 3
echo "write pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_samp -o samp -- python3 tools/bench_sampler.py --steps 5 > $O/sampler.json 2> $O/sampler.err || exit 4
echo "sampler stats done"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -o sq -- python3 tools/bench_sampler.py --steps 1 > $O/pmc_sq.json 2> $O/pmc_sq.err || exit 5
echo "sq pass done"
ls $O $O/*/ | head -40
