#!/bin/bash
# The round's measurement set for bench.py (run on the GPU box: gpurun -- 'bash tools/profile_bench.sh r02'):
#   the driver's invocation un-profiled (x3) and the default invocation, then rocprofv3 passes of the driver's invocation:
#   kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters (separate passes, as MI355X_MICROARCH.md prescribes).
# Summaries land in gpurun_out/prof_<tag>/summary/ (only gpurun_out/ travels back from the GPU box): copy them to profiles/ afterwards
#   cp gpurun_out/prof_<tag>/summary/* profiles/
# the rocpd databases stay in gpurun_out/.
set -u
TAG=${1:-r03}
R=$PWD; O=$R/gpurun_out/prof_$TAG; P=$O/summary; mkdir -p $O $P
ARGS="--steps 20 --warmup 5"
for i in 1 2 3; do python bench.py $ARGS 2>/dev/null | tail -1 >> $O/bench_driver.jsonl; done
python bench.py 2>/dev/null | tail -1 > $O/bench_default.jsonl
cd /tmp; export TMPDIR=/tmp
echo "kernel trace" >> $O/progress.txt
# (the DEFAULT invocation's 1000 timed generations, and nothing else of the steady-state kernel in the process: no pre-heat, no posterior gates -- their
# samplers keep no history and their launches of the same instantiation are ~0.2 us shorter; rounds 2-4 traced the driver's 20 generations behind a
# 0.1 s pre-heat, i.e. mostly pre-heat launches)
# (--no-torch --burnin-gens 0: under the profiler a process that has torch's HIP runtime loaded, or has run the burn-in kernels, shows a mode of slow steady-state
# launches -- up to a third of them at 6-7.6 us -- that un-profiled runs do not have, profiles/r05_rocprof_torch_artefact.txt; the flags keep both out of the traced
# process, the timed region is the default invocation's 1000 steady-state generations; the trace of the unmodified command is kept beside it)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-other-configs --no-moments --preheat 0 --no-torch --burnin-gens 0 > $O/kt.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_torch -o kt -- python $R/bench.py --no-cpu-baseline --no-other-configs --no-moments --preheat 0 > $O/kt_torch.log 2>&1
# ... and the same loop in a process WITHOUT torch (tools/micro/nohist_probe.py): under the profiler a process on torch's HIP runtime shows a second mode (a fifth of
# the launches at 6-7.6 us) that un-profiled runs do not have (profiles/r05_rocprof_torch_artefact.txt); this trace is the one that agrees with bench.py's live figure
KEEP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_notorch -o kt -- python $R/tools/micro/nohist_probe.py > $O/kt_notorch.log 2>&1
# The counter passes run on the library's own queue like everything else, throttled to 64 dispatches between two drains
# (BPM_QUEUE_INFLIGHT): rocprofv3's counter collection serialises every dispatch behind packets of its own and stops forwarding the
# packets of a queue that has more than a few hundred dispatches outstanding (fine with 256, a drain timeout with 600 or without a
# limit; a HIP stream never gets that far ahead of the profiler because its launch calls block).  The library applies this bound by itself
# when ROCPROF_COUNTER_COLLECTION is in its environment; it is spelled out here.
export BPM_QUEUE_INFLIGHT=64 BPM_QUEUE_TIMEOUT_S=30
echo "pmc FETCH_SIZE" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f -o f -- python $R/bench.py $ARGS --no-cpu-baseline --no-moments --no-other-configs --preheat 0 > $O/f.log 2>&1
echo "pmc WRITE_SIZE" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w -o w -- python $R/bench.py $ARGS --no-cpu-baseline --no-moments --no-other-configs --preheat 0 > $O/w.log 2>&1
echo "pmc SQ" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $O/sq -o sq -- python $R/bench.py $ARGS --no-cpu-baseline --no-moments --no-other-configs --preheat 0 > $O/sq.log 2>&1
unset BPM_QUEUE_INFLIGHT
echo "summaries" >> $O/progress.txt
cd $R
K="phase_fused_kernel<1, 1, 64, 2, 3, 1>"
python tools/rocpd_summary.py stats $(find $O/kt -name "*.db" | head -1) > $P/${TAG}_kernel_stats_bench_default.csv
python tools/rocpd_summary.py stats $(find $O/kt_torch -name "*.db" | head -1) > $P/${TAG}_kernel_stats_bench_default_with_torch_in_the_process.csv
python tools/rocpd_summary.py stats $(find $O/kt_notorch -name "*.db" | head -1) > $P/${TAG}_kernel_stats_headline_loop_without_torch.csv
{
  echo "# rocprofv3 --pmc passes of: python bench.py $ARGS --no-cpu-baseline --no-moments --preheat 0   (kernel $K, the timed + untimed steady-state launches)"
  python tools/rocpd_summary.py pmc $(find $O/f -name "*.db" | head -1) FETCH_SIZE "$K" 40
  python tools/rocpd_summary.py pmc $(find $O/w -name "*.db" | head -1) WRITE_SIZE "$K" 40
  for k in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY; do
    python tools/rocpd_summary.py pmc $(find $O/sq -name "*.db" | head -1) $k "$K" 40
  done
} > $P/${TAG}_pmc_bench_driver.txt
cp $O/bench_driver.jsonl $P/${TAG}_bench_lines_driver_invocation.jsonl
cp $O/bench_default.jsonl $P/${TAG}_bench_line_default.json
# HBM bytes per launch for bench.py's roofline.traffic: FETCH_SIZE (KB) x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE (KB)
python - <<PY
import json, re
t = open("$P/${TAG}_pmc_bench_driver.txt").read()
f = float(re.search(r"FETCH_SIZE per .*?steady\(last \d+ dispatches\)-mean=([0-9.]+)", t).group(1))
w = float(re.search(r"WRITE_SIZE per .*?steady\(last \d+ dispatches\)-mean=([0-9.]+)", t).group(1))
import sys
sys.path.insert(0, "$R")
from bipymc_amd import _lib
json.dump({"hbm_bytes_per_launch": (2 * f + w) * 1024.0, "fetch_size_kb_uncorrected": f, "write_size_kb": w,
           "build_id": _lib.build_id(_lib.load()),      # the library the counters were taken on: bench.py nulls roofline.traffic for any other
           "source": "profiles/${TAG}_pmc_bench_driver.txt", "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)"},
          open("$P/traffic_cfg2.json", "w"), indent=1)
PY
find $O -name "*.db" -size +20M -delete
cat $P/${TAG}_bench_lines_driver_invocation.jsonl | cut -c1-250; cut -c1-250 $P/${TAG}_bench_line_default.json; head -5 $P/${TAG}_kernel_stats_bench_default.csv | cut -c1-200; cat $P/${TAG}_pmc_bench_driver.txt; cat $P/traffic_cfg2.json
