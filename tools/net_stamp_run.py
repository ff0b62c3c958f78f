"""Diagnostic: per-region cycle shares of k_tower_f16x3 from a -DAZX_NET_STAMP build (tools only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("AZX_STAMP_LIB", "libazx_netstamp.so"))
from azalea_amd import engine as eng
import torch
from azalea_amd.network import HexNetwork
torch.manual_seed(0)
net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).eval()
E = eng.Engine(board_size=11, n_games=int(sys.argv[1]) if len(sys.argv) > 1 else 4096, simulations=400, search_batch_size=10, evaluator=eng.EVAL_RESNET, noise_scale=0.25)
E.set_weights({k: v.detach().numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32})
st = E.play_steps(1)
print("net ms/launch", 1e3 * st["net_seconds"] / max(1, st["net_launches"]), "launches", st["net_launches"])
E.close()
