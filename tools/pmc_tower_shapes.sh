# PMC comparison of the fused tower on the two MFMA shapes: launch time, shader clock, matrix-pipe busy share.
# Run on the MI355X box: bash tools/pmc_tower_shapes.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 32 16; do
  export AZX_TOWER_SHAPE=$s
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_shape$s -- python3 $R/bench.py --workload resnet --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python3 - $R/gpurun_out/pmc_shape$s <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda:[0.0,0]); dur=[]
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tower" in row["Kernel_Name"] and float(row["Grid_Size"])>100000:
            agg[row["Counter_Name"]][0]+=float(row["Counter_Value"]); agg[row["Counter_Name"]][1]+=1
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tower" in row["Kernel_Name"] and float(row["Grid_Size_X"])>100000:
            dur.append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e6)
d={k:v/n for k,(v,n) in agg.items()}
ms=sum(dur)/len(dur)
print(sys.argv[1][-2:], "ms %.2f"%ms, "clock GHz %.2f"%(d["GRBM_GUI_ACTIVE"]/8/ms/1e6), "mfma busy frac %.2f"%(d["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/(d["GRBM_GUI_ACTIVE"]/8)), "insts_mfma %.3g"%d["SQ_INSTS_MFMA"])
PY
done
