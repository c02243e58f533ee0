# instruction counts of one launch for several library variants (development aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default eko1 eko2; do
  if [ $v != default ]; then export BCOS_HIP_LIB=$R/b-cosification_amd/lib/variants/$v.so; fi
  d=$R/gpurun_out/insts_$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $d -- python3 $R/scripts/${PMC_SCRIPT:-pmc_fwd.py} > /dev/null 2>$d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv,sys
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'tapconv' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
w=agg.get('SQ_WAVES',[1])[-1]
for k,v in agg.items(): print(k, v[-1], round(v[-1]/w,1))
PY
done
