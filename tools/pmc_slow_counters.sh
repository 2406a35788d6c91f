set -u
R=$PWD; O=$R/gpurun_out/pmc_threshold; mkdir -p $O
cd /tmp; export TMPDIR=/tmp BPM_QUEUE_TIMEOUT_S=170 BPM_QUEUE_INFLIGHT=0
for c in FETCH_SIZE SQ_WAVES; do
t0=$(date +%s.%N)
timeout -k 10 240 rocprofv3 --pmc $c -d $O/slow_$c -o s -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-moments --no-other-configs --preheat 0 > $O/slow_$c.log 2>&1
rc=$?
t1=$(date +%s.%N)
echo "$c, no limit on dispatches in flight, wait limit 170 s: rc=$rc wall=$(echo "$t1 - $t0" | bc) s, $(grep -c 'timeout waiting' $O/slow_$c.log) timeout message(s), $(grep -c '"metric"' $O/slow_$c.log) bench line(s); $(grep -o '"burnin_updates_per_s": [0-9.e+]*' $O/slow_$c.log)" >> $O/summary_slow.txt
find $O -name "*.db" -delete
done
cat $O/summary_slow.txt
