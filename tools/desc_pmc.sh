# Texture-addresser / vector-L1 counters of the per-frame kernels on the stage probe's launch, ONE small counter set per
# pass, each pass under its own timeout (a pass with six TA/TCP counters once hung the profiler).
R=$PWD; O=$R/gpurun_out/descpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1)); echo "pass $i: $set" >> $O/progress.log
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/stage_probe.py "$@" > $O/p$i.log 2>&1 || echo "pass $i failed or timed out" >> $O/progress.log
done
python3 - <<PY
import csv, glob, collections
for kern in ("describe_tile", "describe_fast", "min_eig_response", "hamming_mx", "select_kernel"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    print(kern, {c: round(sum(d.values()) / len(d) / 1e6, 3) for c, d in sorted(acc.items())}, flush=True)
PY
cat $O/progress.log
