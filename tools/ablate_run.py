"""Diagnostic (tools only): time the tower kernel of an ablated build: python tools/ablate_run.py <lib>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
from azalea_amd import engine as eng
from azalea_amd.network import HexNetwork
torch.manual_seed(0)
net = HexNetwork(11, 6, 64).eval()
E = eng.Engine(board_size=11, n_games=4096, simulations=400, search_batch_size=10, evaluator=eng.EVAL_RESNET)
E.set_weights({k: v.numpy() for k, v in net.state_dict().items()})
rng = np.random.RandomState(0)
B = 40960
boards = rng.randint(0, 3, (B, 11, 11)).astype(np.int32)
lm = np.tile(np.arange(1, 122, dtype=np.int32), (B, 1))
E.forward(boards[:4096], lm[:4096])
t0 = time.perf_counter(); E.forward(boards, lm); t1 = time.perf_counter()
print(sys.argv[1], "forward(40960) wall %.1f ms (incl. host copies)" % (1e3 * (t1 - t0)))
