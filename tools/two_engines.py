#!/usr/bin/env python3
"""Experiment (tools only): the configs[2] pool split over E engine handles on ONE GPU, one host thread
each, so that one half's tree + heads kernels run beside the other half's tower launches.
    python tools/two_engines.py [E] [steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench
from azalea_amd.network import HexNetwork

E_N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
G = 4096 // E_N
args = argparse.Namespace(board=11, games=G, sims=400, batch=10, noise_scale=0.25, blocks=6, chans=64,
                          nodes_per_game=0, seed=0xBAD5EED5, desync=92, settle=242)
torch.manual_seed(0)
net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).eval().to("cuda:0")
sd = {k: v for k, v in net.state_dict().items() if v.dtype == torch.float32}
engines = []
for i in range(E_N):
    T = bench.make_engine("tree", args, i, E_N, 0)
    bench.settle_pool(T, args, i, E_N, 0)
    start = bench.pool_positions(T)
    T.close()
    E = bench.make_engine("resnet", args, i, E_N, 0)
    E.set_weights({k: (v.data_ptr(), v.numel()) for k, v in sd.items()}, on_device=True)
    E.reset(moves=start)
    engines.append(E)
out = [None] * E_N
def run(i, n):
    out[i] = engines[i].play_steps(n)
def go(n):
    th = [threading.Thread(target=run, args=(i, n)) for i in range(E_N)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
go(5)
dt = go(steps)
sel = sum(o["selects"] for o in out); games = sum(o["games"] for o in out)
print("engines %d x %d games: %.1f ms/step, %.3e sims/s, %.1f games/s; net ms/launch %s" % (
    E_N, G, 1e3 * dt / steps, sel / dt, games / dt, [round(1e3 * o["net_seconds"] / o["net_launches"], 3) for o in out]))
