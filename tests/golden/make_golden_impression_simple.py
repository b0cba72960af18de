"""Golden vectors for ImpressionModel.loss with loss_n = 'BPRsimple' / 'BPRhardsimple' FROM THE REFERENCE
(models/BaseImpressionModel.py:82-83), build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_impression_simple.py
The reference returns this loss UNREDUCED (one value per list); its own training loop cannot call .backward() on it, so the
gradient stored here is autograd's for `loss.sum()` -- i.e. d row_b / d pred[b, :] for every row b, which is what the
kernel's closed form has to reproduce.  (A separate file: impression_losses_metrics.npz stays exactly what its own script
generates.)"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE_DIR)
sys.path.insert(0, os.path.dirname(HERE_DIR))
from make_golden import HERE, _import_reference  # noqa: E402
from make_golden_impression import lists  # noqa: E402


def main():
    torch, _, _ = _import_reference()
    from models.BaseImpressionModel import ImpressionModel
    out = {}
    rng = np.random.default_rng(91)
    for shape_id, (B, mp, mn) in enumerate(((48, 20, 20), (31, 3, 10), (9, 1, 70))):
        for name in ("BPRsimple", "BPRhardsimple"):
            pred, target = lists(rng, B, mp, mn, need_neg=True)
            stub = SimpleNamespace(loss_n=name, train_max_pos_item=mp, device=torch.device("cpu"))
            p = torch.from_numpy(pred).requires_grad_(True)
            rows = ImpressionModel.loss(stub, {"prediction": p}, torch.from_numpy(target))
            assert tuple(rows.shape) == (B,)
            rows.sum().backward()
            key = "loss/{}/{}/".format(shape_id, name)
            out[key + "pred"], out[key + "target"], out[key + "max_pos"] = pred, target, np.int64(mp)
            out[key + "rows"], out[key + "gpred"] = rows.detach().numpy().copy(), p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "impression_bpr_simple.npz"), **out)
    print("wrote impression_bpr_simple.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
