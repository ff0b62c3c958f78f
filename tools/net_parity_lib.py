"""Diagnostic: the wide tower of an alternative library build against the product build on random 13x13
positions (19x256 would take long on the host; 2x256 exercises the same kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from azalea_amd import _lib
def run(libname):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), libname)
    _lib._LIB = None if hasattr(_lib, "_LIB") else None
    from azalea_amd import engine as eng
    from azalea_amd.network import HexNetwork
    torch.manual_seed(0)
    net = HexNetwork(board_size=13, num_blocks=2, base_chans=256).eval()
    E = eng.Engine(board_size=13, n_games=4, simulations=20, search_batch_size=10, evaluator=eng.EVAL_RESNET, num_blocks=2, base_chans=256)
    E.set_weights({k: v.numpy() for k, v in net.state_dict().items()})
    rng = np.random.RandomState(3)
    boards = rng.randint(0, 3, (64, 13, 13)).astype(np.int32)
    lm = np.tile(np.arange(1, 170, dtype=np.int32), (64, 1))
    v, lp = E.forward(boards, lm)
    E.close()
    return v, lp
v, lp = run(sys.argv[1])
np.save("/tmp/np_v.npy", v); np.save("/tmp/np_lp.npy", lp)
print(sys.argv[1], "value[:4]", v[:4], "logprob[0,:4]", lp[0, :4])
