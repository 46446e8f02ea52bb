R=$GRAFT_REPO_ROOT
# kernel-time table of the local-BA probe under rocprofv3 (run on the GPU box): bash tools/lba_kt.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lbakt -- python3 $R/tools/local_ba_probe.py 7 > $R/gpurun_out/lbakt.log 2>&1
python3 - $R/gpurun_out/lbakt <<'PY'
import csv,sys,glob,os
f=max(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-50s calls %5s avg %9.1f us"%(r['Name'].split('(')[0][-50:], r['Calls'], float(r['AverageNs'])/1e3))
PY
