#!/bin/bash
# usage: nan_hunt_pytest.sh N [LIB]   (round-2 defect reproduction: old direct feed under pytest)
cd "$(dirname "$0")/.."
# (the library of round 2 is needed to see the defect: git checkout 058115a -- ddgani_amd && make -C ddgani_amd/csrc)
export DDG_DEBUG_NAN=1 DDG_GRAPH_FEED=${FEED:-direct}
[ -n "$2" ] && export DDGANI_LIB=$PWD/$2
mkdir -p gpurun_out; LOG=gpurun_out/nan_hunt_${TAG:-pytest}.log; : > $LOG
for i in $(seq 1 $1); do
  timeout -k 10 120 python -u -m pytest tests/test_gpu_e2e.py -x -q -s -k "throughput_mode" --timeout 100 --timeout-method thread 2>&1 | grep "XTJ1\|STOPDUMP\|RESLOSS\|ADAM\|EMBGRAD\|ADDEMB\|BNACT\|PROBE\|DEBUG_NAN G\|passed\|failed" | grep -v "import\|print" | cut -c1-400 | head -400 >> $LOG; echo "--- run $i" >> $LOG
  tail -3 $LOG | cut -c1-200 | tail -1
done
