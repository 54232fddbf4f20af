cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/g2pmc
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/raw1 -o p -- tools/bin/gemm2_probe 2 1 > $out/run1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/raw2 -o p -- tools/bin/gemm2_probe 2 1 > $out/run2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw3 -o p -- tools/bin/gemm2_probe 2 1 > $out/run3.log 2>&1
find $out -name "*.csv" | head; 
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/g2pmc/raw[12]/**/*counter_collection.csv', recursive=True)):
    d=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k=(int(r['Dispatch_Id']), r['Kernel_Name'][:60])
        d.setdefault(k,{})
        d[k][r['Counter_Name']]=d[k].get(r['Counter_Name'],0)+float(r['Counter_Value'])
    for k,v in d.items():
        if 'gemm2' in k[1]: print(k, {a:int(b) for a,b in v.items()})
PY
cat $out/raw3/*/*kernel_stats.csv 2>/dev/null | head -8
