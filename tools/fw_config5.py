"""Diagnostic: forward time of the BASELINE configs[4] network (13x13, 19x256) through azx_forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from azalea_amd import engine as eng
from azalea_amd.network import HexNetwork
torch.manual_seed(0)
n, blocks, chans = 13, 19, 256
net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans).eval()
E = eng.Engine(board_size=n, n_games=int(sys.argv[1]) if len(sys.argv) > 1 else 64, simulations=10, search_batch_size=10, evaluator=eng.EVAL_RESNET, num_blocks=blocks, base_chans=chans)
E.set_weights({k: v.detach().numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32})
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
boards = np.random.RandomState(0).randint(0, 3, (B, n, n)).astype(np.int32)
lm = np.zeros((B, n * n), np.int32)
for i in range(B):
    e = np.flatnonzero(boards[i].ravel() == 0) + 1
    lm[i, :len(e)] = e
E.forward(boards, lm)
t = time.perf_counter(); E.forward(boards, lm); dt = time.perf_counter() - t
print("forward of %d positions (13x13, 19x256): %.1f ms = %.2f ms/position = %.2f TFLOP/s" % (B, dt * 1e3, dt * 1e3 / B, B * 7.58e9 / dt / 1e12))
