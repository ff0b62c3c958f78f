cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/l2probe; mkdir -p $O
AZX_WIDE_STREAMS=1 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/a -- python3 $R/bench.py --workload resnet --board 13 --blocks 19 --chans 256 --sims 100 --games 512 --steps 1 --warmup 0 --no-cpu-baseline --no-replay-exchange > $O/a.json 2> $O/a.err; echo rc=$?
python3 - $O <<'P'
import csv,glob,sys,collections
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(sys.argv[1]+"/a/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv_wide" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k,v in agg.items(): print(k, v/n[k], "per launch over", n[k])
if agg.get("TCC_HIT_sum") is not None: print("L2 hit rate", agg["TCC_HIT_sum"]/(agg["TCC_HIT_sum"]+agg["TCC_MISS_sum"]))
P
