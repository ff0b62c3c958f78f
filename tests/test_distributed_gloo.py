"""The N>1 path on CPU: world_size-2 gloo processes exchange replay rows (all-gather of
fixed-size records), sum metrics and broadcast weights exactly as the RCCL path does."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(rank, count, n=5):
    rng = np.random.RandomState(100 + rank)
    cells = n * n
    return dict(board=rng.randint(0, 3, (count, n, n)).astype(np.int32),
                color=rng.randint(0, 2, count).astype(np.int32),
                nlegal=rng.randint(1, cells, count).astype(np.int32),
                moves_prob=rng.rand(count, cells).astype(np.float32),
                reward=rng.choice([-1.0, 1.0], count).astype(np.float32),
                game_uid=(np.arange(count) + (rank << 40)).astype(np.int64))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import distributed as azd
    from azalea_amd.network import HexNetwork
    counts = [7, 12]
    got = azd.all_gather_rows(_rows(rank, counts[rank]), 5)
    want = {k: np.concatenate([_rows(r, counts[r])[k] for r in range(world)]) for k in got}
    ok = all(np.array_equal(got[k], want[k]) for k in got)
    m = azd.all_reduce_metrics({"games": 1 + rank, "reward": 0.5})
    ok = ok and m == {"games": 3.0, "reward": 1.0}
    ok = ok and [azd.shard_quota(25, r, 2) for r in range(2)] == [13, 12]
    torch.manual_seed(rank)
    net = HexNetwork(board_size=5, num_blocks=1, base_chans=8)
    azd.broadcast_weights(net, src=0)
    torch.manual_seed(0)
    ref = HexNetwork(board_size=5, num_blocks=1, base_chans=8)
    ok = ok and all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
    # Player.read sharded over ranks (random mover -> host loop): every rank ends with all rows
    from azalea_amd import AzaleaAgent, HexGame, Player
    agent = AzaleaAgent(lambda: HexGame(4))
    agent.seed(10 + rank)
    pl = Player(None, [agent])
    frame, metrics = pl.read(40)
    pl.stop()
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(frame)]))
    ok = ok and len(frame) >= 40 and sizes[0].item() == sizes[1].item()
    ok = ok and metrics["moves_per_game"] == len(frame)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_world2_gloo_all_gather_and_broadcast():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
