"""ImpressionRunner: training / evaluation over impression lists with a variable number of positives and
negatives per instance (mirror of the reference's helpers/ImpressionRunner.py:18-197: same signatures,
same metrics HR@k / NDCG@k / MAP@k).

`fit` builds the {1 positive slot, 0 negative slot, -1 padding} label matrix of :187-190 on the device and
hands it to `model.loss(out_dict, labels)`; the forward is the model's HIP gather-dot, the default
list-level BPR loss one HIP kernel.  `evaluate` ranks each list's positives and forms the three metrics on the
device (rc_list_metrics); `evaluate_method` keeps the reference's numpy signature for callers that hold predictions.
"""
from typing import Dict

import numpy as np
import torch

from helpers.BaseRunner import BaseRunner
from rechorus_amd import engine, graph as hgraph, nn as hnn
from models.BaseModel import BaseModel


def _ranked_labels(predictions, pos_num, neg_num, pos_num_max):
    """labels [N, n] in ranked order (1 = a ground-truth item), list lengths [N] (reference :84-106)"""
    n = predictions.shape[1]
    pred = predictions.astype(np.float64).copy()
    pred[:, :pos_num_max] -= 1e-6          # a positive that ties with a negative is ranked below it
    order = np.argsort(-pred, axis=1, kind='mergesort')
    pos = np.minimum(np.asarray(pos_num), pos_num_max)
    neg = np.minimum(np.asarray(neg_num), n - pos_num_max)
    labels = (np.arange(n)[None, :] < pos[:, None]).astype(np.int64)
    return np.take_along_axis(labels, order, axis=1), pos + neg


def _metrics_from_ranked(labels, length, topk):
    """HR / NDCG / MAP @k of 0-1 label rows already in ranked order, truncated to `length` (:18-69)"""
    n = labels.shape[1]
    labels = labels * (np.arange(n)[None, :] < length[:, None])
    n_pos = labels.sum(axis=1)
    disc = 1.0 / np.log2(np.arange(2, n + 2))
    ideal = -np.sort(-labels, axis=1)
    cum = np.cumsum(labels, axis=1)
    out = {}
    for k in topk:
        cap = np.clip(n_pos, 1, k)
        dcg, idcg = (labels[:, :k] * disc[:k]).sum(axis=1), (ideal[:, :k] * disc[:k]).sum(axis=1)
        out['NDCG@{}'.format(k)] = dcg / np.where(idcg == 0, 1, idcg)
        prec = np.where(np.arange(n)[None, :] < k, cum, 0) / np.arange(1, n + 1)
        out['MAP@{}'.format(k)] = (prec * labels).sum(axis=1) / cap
        out['HR@{}'.format(k)] = (labels[:, :k].sum(axis=1) > 0).astype(np.float64)
    return out


class ImpressionRunner(BaseRunner):
    @staticmethod
    def evaluate_method(predictions: np.ndarray, topk: list, metrics: list, test_all: bool, neg_num, pos_num_max,
                        pos_num=None, check_sort_idx=0, test_num_neg=0, ret_all=0) -> Dict[str, float]:
        """predictions [N, pos_num_max + max_neg]: columns [0, pos_num) positives, [pos_num_max,
        pos_num_max + neg_num) negatives, everything else -inf.  Returns NDCG / MAP / HR @k (all three,
        whatever `metrics` lists, like the reference)."""
        if test_all:
            return dict()
        if pos_num is None:
            pos_num = np.ones(len(predictions), dtype=np.int64)
        labels, length = _ranked_labels(predictions, pos_num, neg_num, pos_num_max)
        per_row = _metrics_from_ranked(labels, length, topk)
        ordered = {}
        for name in ('NDCG', 'MAP', 'HR'):
            for k in topk:
                key = '{}@{}'.format(name, k)
                ordered[key] = per_row[key] if ret_all else per_row[key].mean()
        return ordered

    def evaluate(self, data: BaseModel.Dataset, topks: list, metrics: list, check_sort_idx=0, all=0) -> Dict[str, float]:
        """reference :135-172.  On the GPU the predictions never leave the device: rc_list_metrics ranks every list's valid
        positives among its valid entries (the reference's -inf mask, 1e-6 tie shift and stable sort restated as counts) and
        forms NDCG / MAP / HR @k in float64 there; the host reads 3 * len(topks) means (all=1: the per-row values)."""
        model = data.model
        mp, mn = model.test_max_pos_item, model.test_max_neg_item
        if not model.test_all and torch.device(model.device).type == 'cuda' and engine.list_metrics_supported(mp + mn, mp, len(topks)):
            pred = self._predict_device(data).float().contiguous()
            if pred.shape[1] == mp + mn:
                dev = pred.device
                pos = torch.from_numpy(np.asarray(data.data['pos_num'], dtype=np.int64)).to(dev) if 'pos_num' in data.data else None
                neg = torch.from_numpy(np.asarray(data.data['neg_num'], dtype=np.int64)).to(dev)
                per_row, mean = engine.list_metrics(pred, pos, neg, mp, topks, want_mean=not all)
                vals = (per_row.permute(1, 2, 0) if all else mean).cpu().numpy()      # [3, K(, N)]
                return {'{}@{}'.format(name, k): (vals[m, j] if all else float(vals[m, j]))
                        for m, name in enumerate(('NDCG', 'MAP', 'HR')) for j, k in enumerate(topks)}
        predictions = self.predict(data)
        if model.test_all:
            rows, cols = list(), list()
            for i, u in enumerate(data.data['user_id']):
                clicked = [x[0] for x in data.corpus.user_his[u]]
                rows.extend([i] * len(clicked))
                cols.extend(clicked)
            predictions[rows, cols] = -np.inf
        pos_num = np.asarray(data.data['pos_num']) if 'pos_num' in data.data else np.ones(len(predictions), dtype=np.int64)
        neg_num = np.asarray(data.data['neg_num'])
        col = np.arange(predictions.shape[1])[None, :]
        keep = (col < np.minimum(pos_num, mp)[:, None]) | ((col >= mp) & (col < mp + np.minimum(neg_num, mn)[:, None]))
        predictions = np.where(keep, predictions, -np.inf)
        return self.evaluate_method(predictions, topks, metrics, model.test_all, neg_num, mp, pos_num, check_sort_idx,
                                    test_num_neg=data.neg_len, ret_all=all)

    @staticmethod
    def _labels(batch, n_cand, max_pos, device):
        """1 on real positive slots, 0 on real negative slots, -1 on padding (reference :187-190)"""
        col = torch.arange(n_cand, device=device)[None, :]
        pos = col < batch['pos_num'][:, None]
        neg = (col >= max_pos) & (col < max_pos + batch['neg_num'][:, None])
        return torch.where(pos, 1, torch.where(neg, 0, -1)).long()

    def fit(self, data: BaseModel.Dataset, epoch=-1) -> float:
        model = data.model
        if model.optimizer is None:
            model.optimizer = self._build_optimizer(model)
        model.train()
        losses = list()

        def list_loss(m, batch):  # forward + the runner-built {1, 0, -1} labels + the model's list-wise loss
            out = m(batch)
            return m.loss(out, self._labels(batch, out['prediction'].shape[1], m.train_max_pos_item, out['prediction'].device))
        graphable = (self.use_graph and hgraph.usable() and isinstance(model.optimizer, hnn.HipOptimizer)
                     and model.optimizer.capturable and torch.device(model.device).type == 'cuda')
        for batch in self._batches(data, self.batch_size, train=True):  # device pipeline, or DataLoader for custom datasets
            if graphable:  # hipGraph replay of the whole step, one graph per batch shape (rechorus_amd/graph.py)
                key = (id(model), hgraph.GraphedStep.signature(batch))
                step = self._graphed.get(key)
                if step is None:
                    step = self._graphed[key] = hgraph.GraphedStep(model, loss_of=list_loss)
                losses.append(step.run(batch))
                continue
            model.optimizer.zero_grad()
            out_dict = model(batch)
            pred = out_dict['prediction']
            labels = self._labels(batch, pred.shape[1], model.train_max_pos_item, pred.device)
            loss = model.loss(out_dict, labels)
            loss.backward()
            model.optimizer.step()
            losses.append(loss.detach().reshape(1))
        # one D2H copy per epoch; a NaN loss surfaces in BaseRunner.train (reference logs it per batch, :192)
        return float(torch.cat(losses).mean().item()) if losses else float('nan')
