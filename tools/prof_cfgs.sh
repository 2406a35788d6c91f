set -u
R=$PWD; O=$R/gpurun_out/r2i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in cfg3 cfg5 cfg5_burnin cfg5_local cfg2_burnin; do
  python $R/tools/profile_config.py $c 200 2>&1 | grep -v amdgpu >> $O/plain.txt
  echo "kt $c" >> $O/progress.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt -- python $R/tools/profile_config.py $c 200 > $O/kt_$c.log 2>&1
  python $R/tools/rocpd_summary.py stats $(find $O/kt_$c -name "*.db" | head -1) > $O/kt_$c.csv
done
export BPM_QUEUE_INFLIGHT=64 BPM_QUEUE_TIMEOUT_S=30      # (counter passes: at most 64 dispatches in flight, see tools/profile_bench.sh)
for c in cfg3 cfg5; do
  echo "pmc $c" >> $O/progress.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_$c -o f -- python $R/tools/profile_config.py $c 100 > $O/f_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_$c -o w -- python $R/tools/profile_config.py $c 100 > $O/w_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $O/sq_$c -o sq -- python $R/tools/profile_config.py $c 100 > $O/sq_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq2_$c -o sq2 -- python $R/tools/profile_config.py $c 100 > $O/sq2_$c.log 2>&1
  for k in FETCH_SIZE; do python $R/tools/rocpd_summary.py pmc $(find $O/f_$c -name "*.db" | head -1) $k phase_fused 100; done >> $O/pmc_$c.txt
  python $R/tools/rocpd_summary.py pmc $(find $O/w_$c -name "*.db" | head -1) WRITE_SIZE phase_fused 100 >> $O/pmc_$c.txt
  for k in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY; do python $R/tools/rocpd_summary.py pmc $(find $O/sq_$c -name "*.db" | head -1) $k phase_fused 100; done >> $O/pmc_$c.txt
  for k in SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python $R/tools/rocpd_summary.py pmc $(find $O/sq2_$c -name "*.db" | head -1) $k phase_fused 100; done >> $O/pmc_$c.txt
done
unset BPM_QUEUE_INFLIGHT
cd $R
cat $O/plain.txt; for c in cfg3 cfg5 cfg5_burnin cfg5_local cfg2_burnin; do echo "== $c"; head -8 $O/kt_$c.csv | cut -c1-200; done; cat $O/pmc_cfg3.txt $O/pmc_cfg5.txt
# keep the merge small
find $O -name "*.db" -size +20M -delete
