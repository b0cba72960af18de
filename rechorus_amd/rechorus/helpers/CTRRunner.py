"""CTRRunner: BaseRunner for click-through-rate models (mirror of the reference's
helpers/CTRRunner.py:19-79).  Training is BaseRunner.fit unchanged (HIP gathers + HIP optimizer);
evaluation scores every (user, item, context) row once and reports ACC / AUC / F1_SCORE / LOG_LOSS
on the host (sklearn, as the reference does)."""
from typing import Dict

import numpy as np
import torch

from helpers.BaseRunner import BaseRunner
from models.BaseModel import BaseModel


class CTRRunner(BaseRunner):
    @staticmethod
    def evaluate_method(predictions: np.ndarray, labels: np.ndarray, metrics: list) -> Dict[str, float]:
        import sklearn.metrics as sk_metrics
        hard = (predictions > 0.5).astype(int)
        res = dict()
        for metric in metrics:
            if metric == 'ACC':
                res[metric] = (hard == labels.astype(int)).mean()
            elif metric == 'AUC':
                res[metric] = sk_metrics.roc_auc_score(labels, predictions)
            elif metric == 'F1_SCORE':
                res[metric] = sk_metrics.f1_score(labels, hard)
            elif metric == 'LOG_LOSS':
                p = np.clip(predictions, a_min=1e-7, a_max=1 - 1e-7)
                res[metric] = -(np.log(p) * labels + np.log(1 - p) * (1 - labels)).mean()
            else:
                raise ValueError('Undefined evaluation metric: {}.'.format(metric))
        return res

    def __init__(self, args):
        super().__init__(args)
        if not len(args.main_metric):
            self.main_metric = self.metrics[0]

    def evaluate(self, dataset: BaseModel.Dataset, topks: list, metrics: list) -> Dict[str, float]:
        predictions, labels = self.predict(dataset)
        return self.evaluate_method(predictions, labels, metrics)

    def predict(self, dataset: BaseModel.Dataset, save_prediction: bool = False):
        """(predictions [n], labels [n]) over the whole dataset; one D2H copy at the end"""
        model = dataset.model
        model.eval()
        model.phase = 'eval'
        preds, labels = list(), list()
        with torch.no_grad():
            for batch in self._batches(dataset, self.eval_batch_size, train=False):  # device pipeline or DataLoader
                out = model.inference(batch) if hasattr(model, 'inference') else model(batch)
                preds.append(out['prediction'].reshape(-1))
                labels.append(out['label'].reshape(-1))
        return torch.cat(preds).cpu().numpy(), torch.cat(labels).cpu().numpy()
