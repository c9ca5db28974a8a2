#!/bin/bash
# Round-2 profiles of the reference's own top_k (0.05 -> k = 50 000 at c3): bench line, kernel statistics of the same command.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 500 python3 bench.py --steps 2 --warmup 0 --topk 0.05 > $O/bench_c3_topk005.json 2> $O/bench_c3_topk005.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_topk -o topk -- python3 bench.py --steps 1 --warmup 0 --topk 0.05 --no-cpu > $O/bench_topk005_under_rocprof.json 2> $O/bench_topk005_under_rocprof.err || exit 2
echo "kernel stats done"
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $O/bench_c3_full_1gpu.json 2> $O/bench_c3_full_1gpu.err || exit 3
echo "default bench done"
