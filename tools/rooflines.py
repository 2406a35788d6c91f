#!/usr/bin/env python3
"""Per-configuration `roofline` objects from the committed rocprofv3 summaries (profiles/<tag>_*): the same fields bench.py prints for
config 2, for the kernels of configs 3 and 5 and the burn-in variants.  achieved = algorithmic bytes per launch (SURVEY 8(d):
bytes per chain-update x updates per launch) / average kernel duration of the kernel trace; traffic = memory-side bytes per launch
from the --pmc passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, both in KB).

    python tools/rooflines.py r03 > profiles/r03_rooflines.json
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
PEAK = 8000.0
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"


def kernel_avg_us(csv_name, needle):
    with open(os.path.join(P, csv_name)) as f:
        for row in csv.DictReader(f):
            if needle in row["Name"]:
                return float(row["AverageUs"]), int(row["Calls"])
    raise KeyError((csv_name, needle))


def pmc(txt_name, counter):
    t = open(os.path.join(P, txt_name)).read()
    m = re.search(counter + r" per .*?steady\(last \d+ dispatches\)-mean=([0-9.]+)", t)
    return float(m.group(1))


def traffic(txt_name):
    try:
        return (2 * pmc(txt_name, "FETCH_SIZE") + pmc(txt_name, "WRITE_SIZE")) * 1024.0
    except (OSError, AttributeError):
        return None


def entry(config, kernel, csv_name, needle, bytes_per_unit, units, tr, note):
    us, calls = kernel_avg_us(csv_name, needle)
    ach = units * bytes_per_unit / (us * 1e-6) / 1e9
    return {"config": config, "kernel": kernel, "bound": "hbm", "bytes_per_unit": bytes_per_unit, "units_per_launch": units,
            "avg_launch_us": us, "launches": calls, "achieved": ach, "peak": PEAK, "unit": "GB/s", "frac": ach / PEAK,
            "traffic": tr, "traffic_over_algorithmic": (tr / (units * bytes_per_unit)) if tr else None,
            "traffic_frac_of_peak": (tr / (us * 1e-6) / 1e9 / PEAK) if tr else None,
            "source": "profiles/" + csv_name, "note": note}


T = TAG
out = [
    # (round 5 on: the trace of the DEFAULT invocation's 1000 timed generations, nothing else of this instantiation in the process; rounds 2-4: the driver's)
    entry("cfg2 DREAM d=100 N=8192 steady", "phase_fused_kernel<1,1,64,2,3,1>",
          T + ("_kernel_stats_bench_default.csv" if os.path.exists(os.path.join(P, T + "_kernel_stats_bench_default.csv")) else "_kernel_stats_bench_driver.csv"),
          "product::phase_fused_kernel<1, 1, 64, 2, 3, 1>" if os.path.exists(os.path.join(P, T + "_kernel_stats_bench_default.csv")) else "64, 2, 3, 1>", 7216, 4096,
          traffic(T + "_pmc_bench_driver.txt"), "kernel-trace duration (python bench.py --no-cpu-baseline --no-other-configs --no-moments --preheat 0 --no-torch --burnin-gens 0 under rocprofv3 --kernel-trace: torch's runtime and the burn-in kernels kept out of the traced process, profiles/r05_rocprof_torch_artefact.txt); bench.py reports the back-to-back launch period"),
    entry("cfg2 burn-in (CR adaptation)", "phase_fused_kernel<1,1,64,2,3,3>", T + "_kernel_stats_cfg2_burnin.csv", "64, 2, 3, 3>", 7216 + 3200, 4096, None,
          "Welford moments r/w add 32 d bytes per update; level 1 of the CR reduction inside the kernel (round 4); the fold of a generation's sums inside the NEXT generation's first launch (round 5: that launch 10.0-10.6 us, the other 8.3-8.7), no reduction dispatch"),
    entry("cfg3 DE-MC banana d=2 N=65536 snooker 0.1", "phase_fused_kernel<0,3,1,2,1,2>", T + "_kernel_stats_cfg3.csv", "<0, 3, 1, 2, 1, 2>", 97.6, 32768,
          traffic(T + "_pmc_cfg3.txt"), "latency bound: launch floor + dependent Infinity-Cache round trips; a 16-byte row is an eighth of a 128-byte line"),
    entry("cfg5 DREAM mixture d=8 N=262144 steady", "phase_fused_kernel<1,2,4,2,3,2>", T + "_kernel_stats_cfg5.csv", "<1, 2, 4, 2, 3, 2>", 592, 131072,
          traffic(T + "_pmc_cfg5.txt"), "bound by the rate of random 64-byte rows (5.1e10 rows/s out of a 16.8 MB table: the access pattern alone takes 18.0 us per launch, profiles/r04_row_gather_floor.txt); 7 of 8 rows per update are random"),
    entry("cfg5 one GPU's share N=32768 steady", "phase_fused_kernel<1,2,4,2,3,2>", T + "_kernel_stats_cfg5_local.csv", "<1, 2, 4, 2, 3, 2>", 592, 16384, None,
          "latency bound (1024 wavefronts)"),
    entry("cfg5 burn-in + outlier check N=262144", "phase_fused_kernel<1,2,4,2,3,4>", T + "_kernel_stats_cfg5_burnin.csv", "<1, 2, 4, 2, 3, 4>", 592 + 256, 131072, None,
          "level 1 of the CR reduction inside the kernel (round 4), + cr_mid_kernel + cr_final_kernel per generation, outlier check every 50 generations"),
]
# the same loop traced in a torch-free process (profile_bench.sh; profiles/r05_rocprof_torch_artefact.txt): the figure that agrees with bench.py's live one
nt = T + "_kernel_stats_headline_loop_without_torch.csv"
if os.path.exists(os.path.join(P, nt)):
    us, calls = kernel_avg_us(nt, "product::phase_fused_kernel<1, 1, 64, 2, 3, 1>")
    out[0]["torch_free_trace"] = {"avg_launch_us": us, "launches": calls, "frac": 4096 * 7216 / (us * 1e-6) / 1e9 / PEAK, "source": "profiles/" + nt,
                                  "note": "under rocprofv3 a process on torch's HIP runtime (bench.py imports torch first) shows a second mode of slow launches "
                                          "that un-profiled runs do not have; this trace has none"}
print(json.dumps(out, indent=1))
