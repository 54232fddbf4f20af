cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT/tools"
out=$GRAFT_REPO_ROOT/gpurun_out/sc_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python small_conv_probe.py > $out/run.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sc_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('%-110s calls %6s avg_us %9.2f' % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3))
PY
