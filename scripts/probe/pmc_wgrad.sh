#!/bin/bash
# SQ counter passes over one weight-gradient launch: SHAPE=H,Cin,Cout,k bash scripts/probe/pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  d=/tmp/pw_$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/scripts/probe/pmc_wgrad.py > /dev/null 2>$d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, v[-1])
PY
done
f=$(find /tmp/pw_SQ_BUSY* -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad' in r['Kernel_Name']: print('dur_us', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'grid', r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
PY
