#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/isz
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/isz -- python3 $R/scripts/exp_isz.py 2>&1 | grep " ms"
python3 - <<PY
import csv,glob,os,collections
f=sorted(glob.glob("$R/gpurun_out/isz/**/*counter_collection.csv",recursive=True),key=os.path.getmtime)[-1]
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    if "k_insert_size" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for k,v in acc.items(): print(k, round(v/cnt[k]))
f=sorted(glob.glob("$R/gpurun_out/isz/**/*kernel_trace.csv",recursive=True),key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "k_insert_size" in r["Kernel_Name"]:
        print("duration ms", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "grid", r.get("Grid_Size_X"), "vgpr", r.get("VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
