#!/bin/bash
# Copies the outputs of tools/refresh_profiles.sh (merged back under gpurun_out/refresh/) into profiles/<round>_*.
#   bash tools/install_profiles.sh r03
set -e
cd "$(dirname "$0")/.."
R=${1:-r04}
O=gpurun_out/refresh
cp $O/bench_line.json profiles/${R}_bench_line.json
cp $(ls -t $O/kt1/*/*_kernel_stats.csv | head -1) profiles/${R}_bench_kernel_stats_1stream.csv
cp $(ls -t $O/kt2/*/*_kernel_stats.csv | head -1) profiles/${R}_bench_kernel_stats_2streams.csv
cp $(ls -t $O/lba/*/*_kernel_stats.csv | head -1) profiles/${R}_local_ba_kernel_stats.csv
cp $(ls -t $O/gba/*/*_kernel_stats.csv | head -1) profiles/${R}_global_ba_kernel_stats.csv
cp $(ls -t $O/bow/*/*_kernel_stats.csv | head -1) profiles/${R}_bow_orb_kernel_stats.csv
cp $O/${R}_e2e_kernel_stats.json $O/${R}_e2e_kernel_stats.csv profiles/
cp $O/${R}_stages_e2e_kernel_stats.json profiles/${R}_e2e_loop_closing_stages_kernel_stats.json
cp $O/${R}_stages_e2e_kernel_stats.csv profiles/${R}_e2e_loop_closing_stages_kernel_stats.csv
python tools/pmc_summary.py $O profiles/${R}_pmc_traffic.json 512
[ -f $O/${R}_ba_pmc_traffic.json ] && cp $O/${R}_ba_pmc_traffic.json profiles/${R}_ba_pmc_traffic.json
python tools/sq_summary.py $O profiles/${R}_matcher_sq_counters.json
python tools/frame_sq_summary.py $O profiles/${R}_frame_sq_counters.json
python - $R <<'PY'
import csv, glob, os, statistics, re, sys
R = sys.argv[1]
f = max(glob.glob('gpurun_out/refresh/kt1/*/*_kernel_trace.csv'), key=os.path.getmtime)
lines = ["rocprofv3 --kernel-trace of `python3 bench.py --streams 1 --batch 512 --passes 18 --steps 20 --warmup 5 --cpu-frames 0 --no-ba --no-gba --no-e2e --no-bow --stream-seconds 0` (tools/refresh_profiles.sh):",
         "per-launch kernel durations in microseconds, 512 stereo frames = 1024 images per launch; the HIP events of bench.py cover the LAST 5 passes."]
d = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split('<')[0].split()[-1]
    d.setdefault(k, []).append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, v in d.items():
    v = [x[1] for x in sorted(v)]
    if len(v) > 5:
        lines.append("%-34s n=%d first=%.1f median=%.1f mean_all=%.1f mean_last5=%.1f" % (k, len(v), v[0], statistics.median(v), sum(v) / len(v), sum(v[-5:]) / 5))
m = re.search(r'"stage_ms_per_launch": \{[^}]*\}', open('gpurun_out/refresh/kt1.log').read())
lines.append("HIP events of the same run (bench.py, ms per launch, last 5 passes): " + (m.group(0) if m else "?"))
open('profiles/%s_bench_kernel_trace_summary.txt' % R, 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
