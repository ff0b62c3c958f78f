#!/usr/bin/env python3
"""Parameter-gradient error of the hand-written training step against float64 autograd, per tensor, for networks whose
gradients sit far outside the f16 range (the cases of tests/test_gpu_native_train.py::test_gradient_range_is_managed).
    python3 tools/diag_grad_range.py [tiny|huge|plain] [n blocks chans B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F

from azalea_amd.native_train import NativeTrainStep


def main():
    import test_gpu_native_train as T
    kind = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    n, blocks, chans, B = [int(v) for v in sys.argv[2:6]] if len(sys.argv) > 5 else (11, 3, 64, 16)

    def make(dtype):
        net = T._net(n, blocks, chans, seed=7)
        with torch.no_grad():
            if kind == "tiny":
                for blk in net.resblocks:
                    blk.conv1.weight.mul_(1e-6)
                    blk.conv2.weight.mul_(1e-6)
            elif kind == "huge":
                pass                   # (the rewards below)
        return net.to(dtype).train()
    batch = {k: v.to(T.DEV) for k, v in T._random_batch(n, B, 11).items()}
    if kind == "huge":
        batch["reward"] = batch["reward"] * 1e6
    grads = {}
    for dtype in (torch.float32, torch.float64):
        net = make(dtype)
        o = net.forward(batch["board"], batch["legal_moves"])
        loss = F.mse_loss(o["value"], batch["reward"].to(dtype)) - (batch["moves_prob"].to(dtype) * o["moves_logprob"]).sum() / B
        loss.backward()
        grads[dtype] = {name: p.grad.double().cpu().numpy().ravel() for name, p in net.named_parameters()}
    net = make(torch.float32)
    step = NativeTrainStep(net, torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.9), B, T.DEV)
    step.step(batch)
    torch.cuda.synchronize()
    print("%-28s %12s %12s %12s" % ("tensor", "|truth|", "native/|t|", "torch/|t|"))
    for name, truth in grads[torch.float64].items():
        got = step.debug("grad:" + name).astype(np.float64)
        nt = float(np.linalg.norm(truth)) or 1e-300
        print("%-28s %12.3e %12.3e %12.3e" % (name, nt, np.linalg.norm(got - truth) / nt,
                                               np.linalg.norm(grads[torch.float32][name] - truth) / nt))
    for l in range(2 * blocks + 1):
        print("max|g%d| = %.3e" % (l, np.abs(step.debug("g%d" % l)).max()))


if __name__ == "__main__":
    main()
