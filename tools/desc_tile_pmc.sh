# Where the shared-tile describe kernel spends its launch: kernel trace (bin / tile split), then SQ / LDS / addresser
# counters, one small set per pass under its own timeout.
R=$PWD; O=$R/gpurun_out/tilepmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/stage_probe.py "$@" > $O/kt.log 2>&1 || echo "kernel trace failed" >> $O/progress.log
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "TA_TA_BUSY_sum SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); echo "pass $i: $set" >> $O/progress.log
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/stage_probe.py "$@" > $O/p$i.log 2>&1 || echo "pass $i failed or timed out" >> $O/progress.log
done
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if any(k in row["Name"] for k in ("describe", "exact_bits")):
            print(row["Name"][:40], row["Calls"], row["AverageNs"], flush=True)
for kern in ("describe_tile", "describe_bin", "describe_fast"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    print(kern, {c: round(sum(d.values()) / len(d) / 1e6, 3) for c, d in sorted(acc.items())}, flush=True)
PY
cat $O/progress.log
