#!/bin/bash
# Experiment behind DirectQueue::inflight_cap() (aql_queue.h): under `rocprofv3 --pmc` a queue that is too many dispatches ahead of the
# profiler stops being forwarded.  Round 2 bisected the number of dispatches between two drains with the 1024-packet ring: 8, 64, 256
# complete, 600 and no limit time out.  Hypothesis: counter collection turns every dispatch into ~4 packets of the hardware queue the
# intercepting queue wraps (same size as the ring the application asked for), so the wrapped queue overflows after ring / 4 dispatches,
# and the overflow path never resumes.  Test: the same run with a 4096-packet ring (build_variants/libbipymc_q4096.so): the threshold
# must move to ~1024.   usage (GPU box): bash tools/pmc_queue_threshold.sh
set -u
R=$PWD; O=$R/gpurun_out/pmc_threshold; mkdir -p $O
cd /tmp; export TMPDIR=/tmp BPM_QUEUE_TIMEOUT_S=20
run() {   # <label> <lib or ""> <inflight> <warm-up generations: 2 dispatches each, enqueued without a drain>
  local lib=$2
  ( [ -n "$lib" ] && export BPM_LIB_PATH=$lib; export BPM_QUEUE_INFLIGHT=$3
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVES -d $O/$1 -o $1 -- python $R/bench.py --steps 20 --warmup $4 --no-cpu-baseline --no-moments --no-other-configs --preheat 0 > $O/$1.log 2>&1
    rc=$?
    echo "$1: ring $( [ -n "$lib" ] && echo 4096 || echo 1024 ) packets, at most $3 dispatches between two drains, $4 warm-up generations -> rc=$rc, $(grep -c 'timeout waiting' $O/$1.log) timeout message(s), $(grep -c '"metric"' $O/$1.log) bench line(s), $(python $R/tools/rocpd_summary.py pmc $(find $O/$1 -name '*.db' | head -1) SQ_WAVES phase_fused 40 2>/dev/null | cut -c1-90)" ) >> $O/summary.txt
  find $O -name "*.db" -delete
}
: > $O/summary.txt
run ring1024_inflight600 "" 600 600
run ring1024_inflight1000 "" 1000 600
run ring1024_nolimit "" 0 600
cat $O/summary.txt
