# cache-side counter passes over one forward launch (development aid): CIN=256 COUT=64 HH=56 bash scripts/_pmc_mem.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmcm}
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  d=$R/gpurun_out/${TAG}_$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/scripts/pmc_fwd.py > /dev/null 2>$d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "no output for $c"; tail -2 $d.err; continue; fi
  python3 - "$f" <<'PY'
import csv,sys
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'tapconv' in r['Kernel_Name'] or 'tappatch' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, v[-1])
PY
done
