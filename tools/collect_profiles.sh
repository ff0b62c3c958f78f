#!/bin/bash
# Copies what a `tools/prof.sh ${TAG}` + bench + training-profile session left under gpurun_out/ into profiles/ (the files
# the judge reads; see profiles/README.md for the commands that produce them).  usage: tools/collect_profiles.sh <round tag, e.g. r6>
cd "$(dirname "$0")/.."
set -e
TAG=${1:?round tag}
for w in resnet tree config5; do for k in traffic counters; do cp gpurun_out/${TAG}/summary/${w}_pmc_${k}.json profiles/${TAG}_${w}_pmc_${k}.json; done; done
cp gpurun_out/${TAG}/summary/kernel_stats.csv profiles/${TAG}_selfplay_kernel_stats.csv
cp gpurun_out/${TAG}/summary/kernel_durations.json profiles/${TAG}_selfplay_kernel_durations.json
tail -1 gpurun_out/bench_${TAG}.json > profiles/${TAG}_selfplay_bench.json
cp gpurun_out/prof_train_${TAG}/kernel_stats.csv profiles/${TAG}_train_step_kernel_stats.csv
cp gpurun_out/prof_train_${TAG}wide/kernel_stats.csv profiles/${TAG}_train_wide_kernel_stats.csv
cp gpurun_out/prof_train_${TAG}wide/summary.json profiles/${TAG}_train_wide_summary.json
(echo "hand-written training step (split-f16 kernels), 6x64 on 11x11, batch 128: per-kernel time with the launches in line (AZX_TRAIN_GRAPH=0 AZX_TRAIN_FORK=0), 30 steps"; cat gpurun_out/prof_train_${TAG}.txt) > profiles/${TAG}_train_step_kernel_summary.txt
[ -f gpurun_out/prof_traintl_${TAG}.txt ] && (echo "one step as it ships (two streams) under rocprofv3 --kernel-trace.  NOTE: traced, the HOST needs ~0.7 ms to queue a step (0.25 ms untraced,"; echo "0.49 ms of device time per step untraced): the gaps between kernels below are the host's; read the durations and the two-stream order."; cat gpurun_out/prof_traintl_${TAG}.txt) > profiles/${TAG}_train_step_timeline.txt
[ -f gpurun_out/stamps_${TAG}.txt ] && grep -v "NCCL\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/stamps_${TAG}.txt > profiles/${TAG}_train_step_phase_stamps.txt
tail -1 gpurun_out/train_loop_${TAG}.json > profiles/${TAG}_train_loop_bench.json
cp gpurun_out/perf_floor.json profiles/${TAG}_perf_floor.json
python3 - $TAG <<'P'
import json, sys
TAG = sys.argv[1]
d = json.loads(open('profiles/%s_selfplay_bench.json' % TAG).read())
r = d['roofline']
print("headline %.4g sims/s, %.1f games/s, frac %.4f, traffic %s (%s)" % (d['value'], d['games_per_sec'], r['frac'], r['traffic'], r['traffic_source']))
print("tree %.4g sims/s frac %.3f hbm %.3f | config5 %.4g frac %.3f" % (d['tree']['value'], d['tree']['roofline']['frac'],
      d['tree']['roofline'].get('achieved_hbm_frac', 0), d['config5']['value'], d['config5']['roofline']['frac']))
t = d['train_step']
print({m: (round(t[m]['steps_per_sec'], 1), round(t[m]['ms_per_step'], 4)) for m in ('eager', 'eager_nosync', 'hip_graph', 'native')},
      "step only %.4f ms" % t['native']['step_only_ms'], "x%.2f vs graph" % t['speedup_native_vs_hip_graph'])
l = json.loads(open('profiles/%s_train_loop_bench.json' % TAG).read())
print([(x['mode'], x['step'], round(x['steps_per_sec'], 1)) for x in l['runs']], round(l['selfplay_share_of_loop_native'], 3))
print("perf floor:", json.load(open('profiles/%s_perf_floor.json' % TAG))['train_step'])
P
grep -h src_sha profiles/${TAG}_*_pmc_traffic.json | sort | uniq -c
