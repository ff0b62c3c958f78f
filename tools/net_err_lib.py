"""Diagnostic (tools only): max |error| of a library build's 6x64 forward against the reference's recorded outputs --
golden G3 (the seeded 6x64 net with non-trivial BatchNorm statistics, ~256 real-game positions), G8 (the reference's
TRAINED checkpoint: saturated values, peaked policies) and G5r (every leaf of a reference game, 4k+ positions) -- and
against the torch module on 2048 random positions.  Usage: python tools/net_err_lib.py libazx_variant.so"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
from azalea_amd import engine as eng
from azalea_amd.network import HexNetwork
GOLDEN = os.path.join(ROOT, "tests", "golden")


def engine(n, blocks, chans, state):
    E = eng.Engine(board_size=n, n_games=64, simulations=10, search_batch_size=10, evaluator=eng.EVAL_RESNET,
                   num_blocks=blocks, base_chans=chans)
    E.set_weights(state)
    return E


out = ["%-18s" % sys.argv[1]]
for name, f in (("G3", "g3_forward_11_6x64.npz"), ("G8", "g8_checkpoint.npz")):
    z = np.load(os.path.join(GOLDEN, f))
    state = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    n, blocks, chans = (int(x) for x in z["cfg"]) if "cfg" in z.files else (11, 6, 64)
    E = engine(n, blocks, chans, state)
    v, lp = E.forward(z["board"], z["legal_moves"])
    legal = z["legal_moves"] > 0
    out.append("%s(%d) v %.1e lp %.1e" % (name, len(v), np.abs(v - z["value"]).max(), np.abs(lp - z["moves_logprob"])[legal].max()))
    if name == "G3":
        net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).eval()
        net.load_state_dict({k: torch.from_numpy(np.asarray(a)) for k, a in state.items()}, strict=False)
        rng = np.random.RandomState(1)
        B = 2048
        rb = rng.randint(0, 3, (B, 11, 11)).astype(np.int32)
        rb[rng.rand(B, 11, 11) < 0.5] = 0
        rl = np.zeros((B, 121), np.int32)
        for i in range(B):
            e = np.flatnonzero(rb[i].ravel() == 0) + 1
            rl[i, :len(e)] = e
        v2, lp2 = E.forward(rb, rl)
        with torch.no_grad():
            t = net(torch.tensor(rb), torch.tensor(rl))
        out.append("rand(%d) v %.1e lp %.1e" % (B, np.abs(v2 - t["value"].numpy()).max(), np.abs(lp2 - t["moves_logprob"].numpy())[rl > 0].max()))
    E.close()

from run_tape import RunTape
z = np.load(os.path.join(GOLDEN, "g5r_game_11_6x64.npz"))
w = np.load(os.path.join(GOLDEN, str(z["cfg_net"])))
tape = RunTape(z)
rows = np.arange(len(tape))
boards, lm = tape.inputs(rows, int(z["cfg_n"]))
E = engine(int(z["cfg_n"]), 6, 64, {k[2:]: w[k] for k in w.files if k.startswith("w:")})
v, lp = E.forward(boards, lm)
E.close()
worst = max(float(np.abs(lp[r, :int(tape.off[r + 1]) - int(tape.off[r])] - tape.logprob[int(tape.off[r]):int(tape.off[r + 1])]).max()) for r in rows)
out.append("G5r(%d) v %.1e lp %.1e" % (len(rows), np.abs(v - tape.value).max(), worst))
print("  ".join(out))
