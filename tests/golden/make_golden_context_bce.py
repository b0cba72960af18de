"""Golden vectors for the point-wise BCE ranking loss FROM THE REFERENCE (ContextModel.loss, loss_n='BCE',
models/BaseContextModel.py:53-56), build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_context_bce.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference  # noqa: E402

if __name__ == "__main__":
    torch, _, _ = _import_reference()
    from models.BaseContextModel import ContextModel
    rng = np.random.default_rng(5)
    out = {}
    for i, (B, C) in enumerate(((40, 5), (7, 100), (300, 2), (1, 1))):
        pred = rng.normal(0, 2.5, size=(B, C)).astype(np.float32)
        p = torch.from_numpy(pred).requires_grad_(True)
        loss = ContextModel.loss(SimpleNamespace(loss_n="BCE"), {"prediction": p})
        loss.backward()
        out["%d/pred" % i], out["%d/loss" % i], out["%d/gpred" % i] = pred, np.float32(loss.item()), p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "context_bce_ranking.npz"), **out)
    print("wrote context_bce_ranking.npz")
