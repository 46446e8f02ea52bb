#!/bin/bash
# Copies the newest outputs of tools/refresh_profiles.sh (merged back under gpurun_out/refresh/) into profiles/.
set -e
cd "$(dirname "$0")/.."
for d in kt1 kt2 pmc/fetch pmc/write; do
  ls -t gpurun_out/refresh/$d/runc/*agent_info.csv | tail -n +2 | while read f; do rm -f ${f%_agent_info.csv}_*; done
done
cp gpurun_out/refresh/bench_line.json profiles/r01_bench_line.json
cp gpurun_out/refresh/kt1/runc/*_kernel_stats.csv profiles/r01_bench_kernel_stats_1stream.csv
cp gpurun_out/refresh/kt2/runc/*_kernel_stats.csv profiles/r01_bench_kernel_stats_2streams.csv
python tools/pmc_summary.py gpurun_out/refresh/pmc profiles/r01_pmc_traffic.json 512 > /dev/null
python - <<'PY' > profiles/r01_bench_kernel_trace_summary.txt
import csv, glob, statistics, re
f = glob.glob('gpurun_out/refresh/kt1/runc/*_kernel_trace.csv')[0]
print("rocprofv3 --kernel-trace of `python3 bench.py --batch 512 --streams 1 --cpu-frames 0 --no-ba --no-gba --no-e2e` (tools/refresh_profiles.sh):")
print("per-launch kernel durations in microseconds, 512 stereo frames = 1024 images per launch")
print("(80 ms wake-up + 15 warm-up + 20 timed + 5 profiled steps; the HIP events of bench.py cover the LAST 5).  After idle the chip's clock")
print("ramps for ~50 ms, so the all-launch average of the kernel_stats CSV is above the last-5 average.")
d = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split()[-1]
    d.setdefault(k, []).append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, v in d.items():
    v = [x[1] for x in sorted(v)]
    if len(v) > 5:
        print("%-34s n=%d first=%.1f median=%.1f mean_all=%.1f mean_last5=%.1f" % (k, len(v), v[0], statistics.median(v), sum(v) / len(v), sum(v[-5:]) / 5))
m = re.search(r'"stage_ms_per_launch": \{[^}]*\}', open('gpurun_out/refresh/kt1.log').read())
print("HIP events of the same run (bench.py, ms per launch, last 5 steps):", m.group(0))
PY
cat profiles/r01_bench_kernel_trace_summary.txt
