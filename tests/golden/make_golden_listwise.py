"""Golden vectors for the list-wise softmax CE loss FROM THE REFERENCE
(models/BaseImpressionModel.py:44-107, loss_n='softmaxCE'), build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_listwise.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference  # noqa: E402


def make_case(name, B, max_pos, max_neg, seed):
    torch, _, _ = _import_reference()
    from models.BaseImpressionModel import ImpressionModel
    rng = np.random.default_rng(seed)
    n = max_pos + max_neg
    pred = rng.normal(0, 2.0, size=(B, n)).astype(np.float32)
    target = np.full((B, n), -1, dtype=np.int64)
    for b in range(B):
        n_pos = rng.integers(1, max_pos + 1)
        n_neg = rng.integers(0, max_neg + 1) if b % 5 else 0   # some impressions have no negatives
        target[b, :n_pos] = 1
        target[b, max_pos:max_pos + n_neg] = 0
    stub = SimpleNamespace(loss_n="softmaxCE", train_max_pos_item=max_pos, device=torch.device("cpu"))
    p = torch.from_numpy(pred).requires_grad_(True)
    loss = ImpressionModel.loss(stub, {"prediction": p}, torch.from_numpy(target))
    loss.backward()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, pred=pred, target=target, max_pos=np.int64(max_pos), loss=np.float32(loss.item()),
                        gpred=p.grad.numpy())
    print("wrote", path)


if __name__ == "__main__":
    make_case("listwise_p20_n20", 64, 20, 20, 41)
    make_case("listwise_p3_n10", 33, 3, 10, 42)
    make_case("listwise_p1_n99", 17, 1, 99, 43)
