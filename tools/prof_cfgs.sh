#!/bin/bash
# Kernel traces and counter passes of the configurations beside the headline one (run on the GPU box: gpurun -- 'bash tools/prof_cfgs.sh r03').
# Summaries land in gpurun_out/prof_cfgs_<tag>/summary/: copy them to profiles/, then `python tools/rooflines.py <tag> > profiles/<tag>_rooflines.json`.
set -u
TAG=${1:-r03}
R=$PWD; O=$R/gpurun_out/prof_cfgs_$TAG; P=$O/summary; mkdir -p $O $P
cd /tmp; export TMPDIR=/tmp
for c in cfg3 cfg5 cfg5_burnin cfg5_local cfg2_burnin; do
  python $R/tools/profile_config.py $c 200 2>&1 | grep -v amdgpu >> $P/${TAG}_other_configs_plain.txt
  echo "kt $c" >> $O/progress.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt -- python $R/tools/profile_config.py $c 200 > $O/kt_$c.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find $O/kt_$c -name "*.db" | head -1) > $P/${TAG}_kernel_stats_$c.csv
done
export BPM_QUEUE_INFLIGHT=64 BPM_QUEUE_TIMEOUT_S=30      # (counter passes: at most 64 dispatches in flight, profiles/r03_pmc_queue_stall.txt)
for c in cfg3 cfg5; do
  echo "pmc $c" >> $O/progress.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_$c -o f -- python $R/tools/profile_config.py $c 100 > $O/f_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_$c -o w -- python $R/tools/profile_config.py $c 100 > $O/w_$c.log 2>&1
  { echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python tools/profile_config.py $c 100";
    python $R/tools/rocpd_summary.py pmc $(find $O/f_$c -name "*.db" | head -1) FETCH_SIZE phase_fused 100;
    python $R/tools/rocpd_summary.py pmc $(find $O/w_$c -name "*.db" | head -1) WRITE_SIZE phase_fused 100; } > $P/${TAG}_pmc_$c.txt
done
unset BPM_QUEUE_INFLIGHT
cd $R
cat $P/${TAG}_other_configs_plain.txt; for c in cfg3 cfg5 cfg5_burnin cfg5_local cfg2_burnin; do echo "== $c"; head -5 $P/${TAG}_kernel_stats_$c.csv | cut -c1-200; done; cat $P/${TAG}_pmc_cfg3.txt $P/${TAG}_pmc_cfg5.txt
# HBM-side bytes per update launch of cfg3 / cfg5 for bench.py's configs[].roofline.traffic, WITH the build id of the library the counters were taken on
python - <<PY
import json, re, sys
sys.path.insert(0, "$R")
from bipymc_amd import _lib
out = {"build_id": _lib.build_id(_lib.load()), "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)"}
for c in ("cfg3", "cfg5"):
    t = open("$P/${TAG}_pmc_%s.txt" % c).read()
    f = float(re.search(r"FETCH_SIZE per .*?steady\(last \d+ dispatches\)-mean=([0-9.]+)", t).group(1))
    w = float(re.search(r"WRITE_SIZE per .*?steady\(last \d+ dispatches\)-mean=([0-9.]+)", t).group(1))
    out[c] = {"hbm_bytes_per_launch": (2 * f + w) * 1024.0, "fetch_size_kb_uncorrected": f, "write_size_kb": w, "source": "profiles/${TAG}_pmc_%s.txt" % c}
json.dump(out, open("$P/traffic_configs.json", "w"), indent=1)
PY
find $O -name "*.db" -size +20M -delete
