"""BaseRunner on the HIP engine (mirror of the reference's helpers/BaseRunner.py: same flags,
`train / fit / evaluate / predict / print_res / evaluate_method` signatures, same log lines).

What changed underneath `fit` (reference :174-208):
  * the optimizer is `HipOptimizer` (rc_dense_update: exact torch.optim SGD/Adam/Adagrad maths);
    models exposing `hip_train_step` can run a whole iteration as one fused C-ABI call with a
    row-wise update (--engine rowwise; 'auto' picks it for tables above 2^20 rows);
  * the per-batch loss stays on the GPU; one D2H copy per epoch instead of one sync per step;
  * the candidate-column shuffle (:187-202) is skipped for models that declare
    `candidate_permutation_equivariant` (a dot/MLP head scores each candidate independently);
  * standard General / Sequential datasets never touch the host after start-up (--device_pipeline,
    rechorus_amd/pipeline.py): negatives are sampled by rc_sample_negatives once per epoch, batches
    are assembled by rc_assemble_candidates / rc_gather_history; custom datasets keep the DataLoader.
And underneath `evaluate` (:217-252): predictions stay on the GPU, the rank of the ground truth is
rc_target_rank, and `--test_all` on dot-product heads is rc_full_catalogue_rank (fp32 MFMA over the
catalogue, the [N, n_items] matrix never exists); `predict` still returns the reference's ndarray.
"""
import gc
import inspect
import itertools
import logging
import os
from time import time
from typing import Dict, List

import numpy as np
import torch
from torch.utils.data import DataLoader

from models.BaseModel import BaseModel
from rechorus_amd import engine, graph as hgraph, nn as hnn, pipeline
from utils import utils


class BaseRunner(object):
    @staticmethod
    def parse_runner_args(parser):
        parser.add_argument('--epoch', type=int, default=200, help='Number of epochs.')
        parser.add_argument('--check_epoch', type=int, default=1, help='Check some tensors every check_epoch.')
        parser.add_argument('--test_epoch', type=int, default=-1, help='Print test results every test_epoch (-1 means no print).')
        parser.add_argument('--early_stop', type=int, default=10, help='The number of epochs when dev results drop continuously.')
        parser.add_argument('--lr', type=float, default=1e-3, help='Learning rate.')
        parser.add_argument('--l2', type=float, default=0, help='Weight decay in optimizer.')
        parser.add_argument('--batch_size', type=int, default=256, help='Batch size during training.')
        parser.add_argument('--eval_batch_size', type=int, default=256, help='Batch size during testing.')
        parser.add_argument('--optimizer', type=str, default='Adam', help='optimizer: SGD, Adam, Adagrad, Adadelta')
        parser.add_argument('--num_workers', type=int, default=5, help='Number of processors when prepare batches in DataLoader')
        parser.add_argument('--pin_memory', type=int, default=0, help='pin_memory in DataLoader')
        parser.add_argument('--topk', type=str, default='5,10,20,50', help='The number of items recommended to each user.')
        parser.add_argument('--metric', type=str, default='NDCG,HR', help='metrics: NDCG, HR')
        parser.add_argument('--main_metric', type=str, default='', help='Main metric to determine the best model.')
        # additive, engine-specific
        parser.add_argument('--engine', type=str, default='auto',
                            help='dense: exact reference optimizer semantics; rowwise: fused step, update touched rows only; auto')
        parser.add_argument('--graph', type=int, default=1,
                            help='1: replay the dense training step from a hipGraph when the model allows it; 0: eager')
        parser.add_argument('--device_pipeline', type=int, default=1,
                            help='1: sample negatives / assemble batches on the GPU for standard datasets; 0: DataLoader')
        return parser

    @staticmethod
    def evaluate_method(predictions: np.ndarray, topk: list, metrics: list) -> Dict[str, float]:
        """predictions [-1, n_candidates], column 0 = ground truth.  rank = #candidates scoring
        >= the target (ties count against it, reference :63); HR@k, NDCG@k."""
        rank = (predictions >= predictions[:, :1]).sum(axis=-1)
        res = dict()
        for k in topk:
            hit = rank <= k
            for metric in metrics:
                key = '{}@{}'.format(metric, k)
                if metric == 'HR':
                    res[key] = hit.mean()
                elif metric == 'NDCG':
                    res[key] = (hit / np.log2(rank + 1)).mean()
                else:
                    raise ValueError('Undefined evaluation metric: {}.'.format(metric))
        return res

    def __init__(self, args):
        self.train_models = args.train
        self.epoch, self.check_epoch, self.test_epoch = args.epoch, args.check_epoch, args.test_epoch
        self.early_stop = args.early_stop
        self.learning_rate, self.l2 = args.lr, args.l2
        self.batch_size, self.eval_batch_size = args.batch_size, args.eval_batch_size
        self.optimizer_name = args.optimizer
        self.num_workers, self.pin_memory = args.num_workers, args.pin_memory
        self.engine = getattr(args, 'engine', 'auto')
        self.device_pipeline = bool(getattr(args, 'device_pipeline', 1))
        self.use_graph = bool(getattr(args, 'graph', 1))
        self._graphed = {}
        self.topk = [int(x) for x in args.topk.split(',')]
        self.metrics = [m.strip().upper() for m in args.metric.split(',')]
        self.main_metric = args.main_metric if len(args.main_metric) else '{}@{}'.format(self.metrics[0], self.topk[0])
        self.main_topk = int(self.main_metric.split('@')[1]) if '@' in self.main_metric else 0
        self.time = None  # [start, last checkpoint]
        self.log_path = os.path.dirname(args.log_file)
        self.save_appendix = args.log_file.split('/')[-1].split('.')[0]

    def _check_time(self, start=False):
        now = time()
        if self.time is None or start:
            self.time = [now, now]
            return now
        last, self.time[1] = self.time[1], now
        return now - last

    def _build_optimizer(self, model):
        logging.info('Optimizer: ' + self.optimizer_name)
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu and self.optimizer_name in ('SGD', 'Adam', 'Adagrad', 'Adadelta'):  # every --optimizer the reference documents
            return hnn.HipOptimizer(model.customize_parameters(), self.optimizer_name,
                                    lr=self.learning_rate, weight_decay=self.l2, capturable=self.use_graph)
        # anything else (other torch.optim names, CPU debugging) keeps torch's implementation
        return getattr(torch.optim, self.optimizer_name)(
            model.customize_parameters(), lr=self.learning_rate, weight_decay=self.l2)

    def _use_rowwise(self, model) -> bool:
        """Row-wise ("lazy") updates through the model's fused `hip_train_step`, or dense torch.optim semantics.
        'rowwise' / 'dense' are explicit; 'auto' picks row-wise only for tables past 2^20 rows AND only when the
        model says its fused step handles this configuration (`hip_rowwise_supported`), so that e.g. SASRec with
        dropout or NeuMF with two hidden layers on a large catalogue trains on the dense path instead of
        failing in the first fit().  The choice is logged once: row-wise differs from the reference for
        Adam / Adagrad / l2 > 0 (untouched rows neither decay nor move on stale moments)."""
        mode = None
        if not hasattr(model, 'hip_train_step') or self.optimizer_name not in ('SGD', 'Adam', 'Adagrad'):
            mode = False
        elif self.engine == 'rowwise':
            mode = True   # explicit: an unsupported configuration raises in hip_train_step
        elif self.engine == 'dense':
            mode = False
        else:
            big = max(p.shape[0] for p in model.parameters() if p.dim() == 2) > (1 << 20)
            ok = getattr(model, 'hip_rowwise_supported', lambda: True)()
            mode = big and ok
            if big and not ok and not getattr(self, '_engine_logged', False):
                logging.warning('engine auto: tables exceed 2^20 rows but this configuration has no fused row-wise '
                                'step; using dense updates (every row of every table is streamed each step)')
        if not getattr(self, '_engine_logged', False):
            self._engine_logged = True
            logging.info('Engine: %s updates (--engine %s)%s', 'row-wise' if mode else 'dense', self.engine,
                         '; rows absent from a batch are not touched (differs from torch.optim for Adam / Adagrad / '
                         'l2 > 0)' if mode else '')
        return mode

    # ---- the epoch loop (reference contract: helpers/BaseRunner.py:116-172) ---------------------------------------
    # Observable behaviour kept: one log line per epoch in the format exp.py scrapes, the checkpoint written whenever
    # the dev metric reaches a new maximum (or the model is in `stage` 1), early stopping on the dev history, NaN loss
    # ends training, Ctrl-C offers to skip the final evaluation, the best checkpoint is loaded at the end.  The loop is
    # organised around a small record of the dev history instead of the reference's inline bookkeeping.
    class _DevHistory:
        def __init__(self, metric):
            self.metric, self.results = metric, []

        def add(self, result):
            self.results.append(result)
            return self.is_best()

        @property
        def curve(self):
            return [r[self.metric] for r in self.results]

        def is_best(self):
            c = self.curve
            return c[-1] == max(c)

        def best_index(self):
            c = self.curve
            return c.index(max(c))

    def _epoch(self, number, model, data_dict, history):
        """one epoch: fit, dev (and periodic test) evaluation, checkpoint; -> False when training has to stop"""
        self._check_time()
        gc.collect()
        loss = self.fit(data_dict['train'], epoch=number)
        if np.isnan(loss):
            logging.info('Loss is Nan. Stop training at %d.' % number)
            return False
        fit_seconds = self._check_time()
        if self.check_epoch > 0 and (number - 1) % self.check_epoch == 0 and len(model.check_list) > 0:
            utils.check(model.check_list)
        dev = self.evaluate(data_dict['dev'], [self.main_topk], self.metrics)
        improved = history.add(dev)
        parts = ['Epoch {:<5} loss={:<.4f} [{:<3.1f} s]\tdev=({})'.format(number, loss, fit_seconds, utils.format_metric(dev))]
        if self.test_epoch > 0 and (number - 1) % self.test_epoch == 0:
            parts.append(' test=({})'.format(utils.format_metric(self.evaluate(data_dict['test'], self.topk[:1], self.metrics))))
        parts.append(' [{:<.1f} s]'.format(self._check_time()))
        if improved or getattr(model, 'stage', None) == 1:
            model.save_model()
            parts.append(' *')
        logging.info(''.join(parts))
        if self.early_stop > 0 and self.eval_termination(history.curve):
            logging.info('Early stop at %d based on dev result.' % number)
            return False
        return True

    def train(self, data_dict: Dict[str, BaseModel.Dataset]):
        model = data_dict['train'].model
        history = self._DevHistory(self.main_metric)
        self._check_time(start=True)
        try:
            number = 1
            while number <= self.epoch and self._epoch(number, model, data_dict, history):
                number += 1
        except KeyboardInterrupt:
            logging.info('Early stop manually')
            if input('Exit completely without evaluation? (y/n) (default n):').lower().startswith('y'):
                logging.info(os.linesep + '-' * 45 + ' END: ' + utils.get_time() + ' ' + '-' * 45)
                exit(1)
        best = history.best_index()
        logging.info(os.linesep + 'Best Iter(dev)={:>5}\t dev=({}) [{:<.1f} s] '.format(
            best + 1, utils.format_metric(history.results[best]), self.time[1] - self.time[0]))
        model.load_model()

    def _on_device(self, dataset) -> bool:
        return (self.device_pipeline and torch.device(dataset.model.device).type == 'cuda'
                and pipeline.eligible(dataset))

    def _batches(self, dataset, batch_size, train):
        """feed dicts on the model's device: assembled there when the dataset is a standard one,
        else DataLoader -> collate_batch -> batch_to_gpu exactly like the reference (:182-186)"""
        model = dataset.model
        if self._on_device(dataset):
            dd = pipeline.device_dataset(dataset, torch.device(model.device))
            if train:
                dd.sample_negatives(seed=int(np.random.randint(0, 2 ** 31 - 1)))  # follows --random_seed
            yield from dd.batches(batch_size, shuffle=train)
            return
        if train:
            dataset.actions_before_epoch()  # negative sampling happens before workers fork
        dl = DataLoader(dataset, batch_size=batch_size, shuffle=train, num_workers=self.num_workers,
                        collate_fn=dataset.collate_batch, pin_memory=self.pin_memory)
        for batch in dl:
            yield utils.batch_to_gpu(batch, model.device)

    def fit(self, dataset: BaseModel.Dataset, epoch=-1) -> float:
        model = dataset.model
        rowwise = self._use_rowwise(model)
        if model.optimizer is None and not rowwise:
            model.optimizer = self._build_optimizer(model)

        model.train()
        losses = list()
        equivariant = getattr(model, 'candidate_permutation_equivariant', False)
        # hipGraph replay of the dense step: needs a host-free step (no host-side candidate shuffle; torch's
        # dropout is fine, its Philox offset advances per replay) and the capturable optimizer; one graph per
        # feed-dict shape
        if self.use_graph and not hgraph.usable() and not getattr(self, '_graph_warned', False):
            self._graph_warned = True
            logging.warning('--graph 1 requested but hipGraph replay is disabled: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 has to be in '
                            'the environment before HIP initialises (import rechorus_amd before torch touches the GPU, or '
                            'export it); training runs eagerly (same results, more launch overhead at small batches)')
        graphable = (self.use_graph and not rowwise and equivariant
                     and isinstance(model.optimizer, hnn.HipOptimizer) and model.optimizer.capturable
                     and torch.device(model.device).type == 'cuda' and hgraph.usable())
        if rowwise:
            # The engine's step is handed the FOLLOWING batch as well (the reference's DataLoader runs ahead of the loop
            # too, :182-186): its id grouping is prepared beside this step's row updates (rc_bprmf_train_step_ahead).
            # Models whose hip_train_step does not take `next_feed_dict` are called the plain way.
            takes_next = 'next_feed_dict' in inspect.signature(model.hip_train_step).parameters
            prev = None
            for nxt in itertools.chain(self._batches(dataset, self.batch_size, train=True), [None]):
                if prev is not None:
                    kw = {'next_feed_dict': nxt} if (takes_next and nxt is not None) else {}
                    losses.append(model.hip_train_step(prev, self.optimizer_name, self.learning_rate, self.l2, **kw).clone())
                prev = nxt
            return float(torch.cat([l.reshape(1) for l in losses]).mean().item()) if losses else float('nan')
        for batch in self._batches(dataset, self.batch_size, train=True):
            if graphable:
                key = (id(model), hgraph.GraphedStep.signature(batch))
                step = self._graphed.get(key)
                if step is None:
                    step = self._graphed[key] = hgraph.GraphedStep(model)
                losses.append(step.run(batch))
                continue
            item_ids = batch['item_id']
            indices = None
            if not equivariant:
                # shuffle candidate columns so a model cannot learn "column 0 is the target"
                indices = torch.argsort(torch.rand(*item_ids.shape), dim=-1).to(item_ids.device)
                batch['item_id'] = torch.gather(item_ids, 1, indices)
            model.optimizer.zero_grad()
            out_dict = model(batch)
            pred = out_dict['prediction']
            if indices is not None and pred.dim() == 2:
                restored = torch.zeros_like(pred)
                restored.scatter_(1, indices, pred)  # undo the shuffle: column 0 is the target again
                out_dict['prediction'] = restored
            loss = model.loss(out_dict)
            loss.backward()
            model.optimizer.step()
            losses.append(loss.detach().reshape(1))
        # epoch loss = mean of per-batch means (reference :207-208); one D2H copy per epoch
        return float(torch.cat([l.reshape(1) for l in losses]).mean().item()) if losses else float('nan')

    def eval_termination(self, criterion: List[float]) -> bool:
        if len(criterion) > self.early_stop and utils.non_increasing(criterion[-self.early_stop:]):
            return True
        return len(criterion) - criterion.index(max(criterion)) > self.early_stop

    def evaluate(self, dataset: BaseModel.Dataset, topks: list, metrics: list) -> Dict[str, float]:
        """same numbers as evaluate_method(predict(dataset)), computed where the scores are"""
        model = dataset.model
        if torch.device(model.device).type != 'cuda':
            return self.evaluate_method(self.predict(dataset), topks, metrics)
        if model.test_all and hasattr(model, 'full_catalogue_vectors'):
            rank = self._full_catalogue_ranks(dataset)
            if rank is not None:
                return engine.rank_metrics(rank, topks, metrics)
        if model.test_all:
            # any other head (NeuMF, a user's model file): the scores of ONE evaluation batch against the catalogue, masked and
            # ranked before the next batch is scored -- [eval_batch_size, n_items] alive at a time, not the reference's
            # [n_instances, n_items] matrix (helpers/BaseRunner.py:225-252)
            return engine.rank_metrics(self._streamed_test_all_ranks(dataset), topks, metrics)
        pred = self._predict_device(dataset)
        return engine.rank_metrics(engine.target_rank(pred.contiguous()), topks, metrics)

    def _streamed_test_all_ranks(self, dataset) -> torch.Tensor:
        """rank of the ground truth (column 0) of every instance under --test_all, clicked items masked (reference :243-250),
        one evaluation batch at a time"""
        model = dataset.model
        dev = torch.device(model.device)
        ptr, flat = pipeline.clicked_csr(dataset.corpus, dev, 'all')
        model.eval()
        ranks = list()
        with torch.no_grad():
            for batch in self._batches(dataset, self.eval_batch_size, train=False):
                out = model.inference(batch) if hasattr(model, 'inference') else model(batch)
                pred = out['prediction'].contiguous()
                users = batch['user_id'].to(dev).reshape(-1)
                lo, hi = ptr[users], ptr[users + 1]
                cnt = hi - lo
                rows = torch.repeat_interleave(torch.arange(users.numel(), device=dev), cnt)
                # position of every clicked item inside the flat CSR: lo[row] + (running index within the row)
                within = torch.arange(rows.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
                cols = flat[lo[rows] + within]
                pred[rows, cols] = -float('inf')
                ranks.append(engine.target_rank(pred))
        return torch.cat(ranks)

    def _predict_device(self, dataset) -> torch.Tensor:
        """[n_instances, n_candidates] scores on the GPU, ground truth in column 0"""
        model = dataset.model
        model.eval()
        chunks = list()
        with torch.no_grad():
            for batch in self._batches(dataset, self.eval_batch_size, train=False):
                out = model.inference(batch) if hasattr(model, 'inference') else model(batch)
                chunks.append(out['prediction'])
        return torch.cat(chunks)

    @staticmethod
    def _clicked_cells(dataset, device):
        """(row, column) of every item a user already interacted with (reference :243-250)"""
        rows, cols = list(), list()
        for i, u in enumerate(dataset.data['user_id']):
            seen = list(dataset.corpus.train_clicked_set[u] | dataset.corpus.residual_clicked_set[u])
            rows.extend([i] * len(seen))
            cols.extend(seen)
        return torch.tensor(rows, dtype=torch.long, device=device), torch.tensor(cols, dtype=torch.long, device=device)

    def _full_catalogue_ranks(self, dataset):
        """--test_all for dot-product heads: the model hands over its query vectors and item table
        (`full_catalogue_vectors`), rc_full_catalogue_rank streams the catalogue through MFMA tiles"""
        model = dataset.model
        if not self._on_device(dataset):
            return None
        dev = torch.device(model.device)
        dd = pipeline.device_dataset(dataset, dev)
        ptr, flat = pipeline.clicked_csr(dataset.corpus, dev, 'all')
        model.eval()
        ranks = list()
        with torch.no_grad():
            n = len(dd)
            for s in range(0, n, self.eval_batch_size):
                idx = torch.arange(s, min(n, s + self.eval_batch_size), device=dev)
                feed = dd.feed_without_candidates(idx)
                vec, table = model.full_catalogue_vectors(feed)
                if not engine.full_catalogue_rank_supported(vec.shape[1]):
                    return None
                rank, _ = engine.full_catalogue_rank(vec.contiguous(), table, feed['user_id'], dd.items[idx].contiguous(), ptr, flat)
                ranks.append(rank)
        return torch.cat(ranks)

    def predict(self, dataset: BaseModel.Dataset, save_prediction: bool = False) -> np.ndarray:
        """[n_instances, n_candidates] scores, ground truth in column 0 (reference :225-252)"""
        predictions = self._predict_device(dataset).cpu().numpy()  # one D2H copy
        if dataset.model.test_all:  # mask items the user already interacted with
            rows, cols = self._clicked_cells(dataset, 'cpu')
            predictions[rows.numpy(), cols.numpy()] = -np.inf
        return predictions

    def print_res(self, dataset: BaseModel.Dataset) -> str:
        return '(' + utils.format_metric(self.evaluate(dataset, self.topk, self.metrics)) + ')'
