"""Policy/value resnet used by the engine: the PyTorch module only HOLDS the weights (and trains
them with stock PyTorch-ROCm); self-play inference runs in libazx_hip.so (net_kernels.hip).

State-dict layout and forward semantics follow azalea/network.py (Resblock :17-39, Network
:42-85, HexNetwork :120-152) so checkpoints (policy.py:112-130) are interchangeable and
policy_trainer.supervised_step (policy_trainer.py:123-142) can train this module unchanged.
"""
import torch
from torch import nn
from torch.nn import functional as F


def _conv(cin, cout, k):
    return nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=False)


class Resblock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN, identity shortcut, ReLU (network.py:17-39)."""

    def __init__(self, in_dim, dim):
        super().__init__()
        self.conv1 = _conv(in_dim, dim, 3)
        self.bn1 = nn.BatchNorm2d(dim)
        self.conv2 = _conv(dim, dim, 3)
        self.bn2 = nn.BatchNorm2d(dim)
        self.res_conv = _conv(in_dim, dim, 1) if dim != in_dim else None
        self.res_bn = nn.BatchNorm2d(dim) if dim != in_dim else None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.res_conv is not None:
            x = self.res_bn(self.res_conv(x))
        return F.relu(y + x)


class HexNetwork(nn.Module):
    """Embedding(3->4) + stem + `num_blocks` Resblocks + value and policy heads."""

    def __init__(self, board_size=11, num_blocks=6, base_chans=64):
        super().__init__()
        cells = board_size * board_size
        self.board_size = board_size
        self.conv1 = _conv(4, base_chans, 3)
        self.bn1 = nn.BatchNorm2d(base_chans)
        self.resblocks = nn.Sequential(*[Resblock(base_chans, base_chans) for _ in range(num_blocks)])
        self.value_conv1 = _conv(base_chans, 2, 1)
        self.value_bn1 = nn.BatchNorm2d(2)
        self.value_fc2 = nn.Linear(2 * cells, 64)
        self.value_fc3 = nn.Linear(64, 1)
        self.move_conv1 = _conv(base_chans, 4, 1)
        self.move_bn1 = nn.BatchNorm2d(4)
        self.encoder = nn.Embedding(3, 4)
        self.move_fc = nn.Linear(4 * cells, cells)

    @property
    def device(self):
        return self.conv1.weight.device

    def forward(self, board, legal_moves):
        """board [B,N,N] in {0,1,2}; legal_moves [B,K] 1-based tiles, 0 = padding."""
        x = self.encoder(board.long()).permute(0, 3, 1, 2).contiguous()
        return self.forward_embedded(x, legal_moves)

    def forward_embedded(self, x, legal_moves):
        """forward() behind the embedding: x [B,4,N,N] (policy_trainer.GraphedTrainStep embeds by masks)."""
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.resblocks(x)
        v = F.relu(self.value_bn1(self.value_conv1(x))).flatten(1)     # (c, h, w) order
        v = self.value_fc3(F.relu(self.value_fc2(v)))
        value = torch.tanh(v).squeeze(1)
        p = F.relu(self.move_bn1(self.move_conv1(x))).flatten(1)
        logit = self.move_fc(p)
        logit = torch.gather(logit, 1, (legal_moves - 1).clamp(min=0).long())
        logit = logit.masked_fill(legal_moves == 0, -99)
        return dict(value=value, moves_logprob=F.log_softmax(logit, dim=1))

    def run(self, batch, *, compute_loss=False):
        """Network.run (network.py:87-105): inference dict, or (dict, loss) for training."""
        out = self.forward(batch["board"], batch["legal_moves"])
        if not compute_loss:
            return {k: v.detach() for k, v in out.items()}
        value_loss = F.mse_loss(out["value"], batch["reward"])
        moves_loss = -(batch["moves_prob"] * out["moves_logprob"]).sum() / len(batch["moves_prob"])
        loss = value_loss.to(moves_loss.device) + moves_loss
        det = {k: v.detach() for k, v in out.items()}
        det.update(value_loss=value_loss.item(), moves_loss=moves_loss.item())
        return det, loss
