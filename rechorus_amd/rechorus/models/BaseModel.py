"""Model base classes of the plugin surface (mirror of the reference's models/BaseModel.py:
BaseModel :16-152, GeneralModel :154-214, SequentialModel :216-245, CTRModel :247-288).

A model file written for ReChorus (class attrs `reader`/`runner`/`extra_log_args`, static
`parse_model_args`, `forward(feed_dict) -> {'prediction': [B, C]}`, `loss(out_dict)`) runs
unchanged on top of these classes.  What differs from the reference is underneath:
  * `GeneralModel.loss` is one HIP kernel with a closed-form backward (rechorus_amd.nn.bpr_loss);
  * the negative sampler is vectorised (same distribution, no Python double loop);
  * batches are collated without per-key Python list scans.
"""
import logging
from typing import List

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset as BaseDataset

from rechorus_amd import nn as hnn
from utils import utils

# Model files written for the reference's pinned stack (numpy 1.x, requirements.txt) spell dtypes `np.int`, `np.object`, `np.float`,
# `np.bool` (e.g. models/sequential/SASRec.py:69, models/BaseModel.py:141,146 of the reference); numpy >= 1.24 removed those
# aliases.  Every model file imports this module first, so they are restored here and such files run unmodified.
for _alias, _type in (("int", int), ("float", float), ("bool", bool), ("object", object)):
    if _alias not in np.__dict__:
        setattr(np, _alias, _type)


def task_variant(name, task_base, head, reader, runner, log_args, module, forward=None, parse_from=None, doc=None):
    """Build the class `name` = task base (GeneralModel, SequentialModel, ImpressionModel, ContextCTRModel, ...)
    + head mixin (the object that owns parameters and scoring: BPRMFBase, SASRecBase, DeepFMBase, ...).

    Every model file of the reference spells this combination out by hand once per task (TopK / Impression
    / CTR): class attributes, a parse_model_args that chains the two parsers, an __init__ that runs the task
    base and then the head's `_base_init`, a forward that delegates to the head.  Here it is one call;
    the resulting class is what main.py looks up by name, with the same attributes.
      forward    : optional wrapper `f(self, feed_dict, head_forward)` (the CTR variants squash and flatten)
      parse_from : class whose parse_model_args supplies the task flags (defaults to task_base)"""
    flags_from = parse_from or task_base

    def parse_model_args(parser):
        return flags_from.parse_model_args(head.parse_model_args(parser))

    def __init__(self, args, corpus):
        task_base.__init__(self, args, corpus)
        self._base_init(args, corpus)

    if forward is None:
        def run(self, feed_dict):
            return head.forward(self, feed_dict)
    else:
        def run(self, feed_dict):
            return forward(self, feed_dict, head.forward)
    body = dict(reader=reader, runner=runner, extra_log_args=list(log_args), __init__=__init__, forward=run,
                parse_model_args=staticmethod(parse_model_args), __module__=module, __doc__=doc)
    return type(name, (task_base, head), body)


class BaseModel(nn.Module):
    reader, runner = None, None  # helper class names, chosen by concrete models
    extra_log_args = []

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--model_path', type=str, default='', help='Model save path.')
        parser.add_argument('--buffer', type=int, default=1, help='Whether to buffer feed dicts for dev/test')
        return parser

    @staticmethod
    def init_weights(m):
        # normal(0, 0.01) for every Linear / Embedding weight and Linear bias (reference :29-35);
        # HipEmbedding is matched by the same 'Embedding' substring test
        kind = str(type(m))
        if 'Linear' in kind:
            nn.init.normal_(m.weight, mean=0.0, std=0.01)
            if m.bias is not None:
                nn.init.normal_(m.bias, mean=0.0, std=0.01)
        elif 'Embedding' in kind:
            nn.init.normal_(m.weight, mean=0.0, std=0.01)

    def __init__(self, args, corpus):
        super().__init__()
        self.device = args.device
        self.model_path = args.model_path
        self.buffer = args.buffer
        self.optimizer = None
        self.check_list = list()  # (name, tensor) pairs logged every check_epoch

    # ---- to be provided by concrete models -------------------------------------------------
    def _define_params(self):
        pass

    def forward(self, feed_dict: dict) -> dict:
        """-> out_dict with 'prediction' [batch_size, n_candidates] (column 0 = ground truth)"""
        pass

    def loss(self, out_dict: dict) -> torch.Tensor:
        pass

    # ---- shared services ----------------------------------------------------------------------
    def customize_parameters(self) -> list:
        """two param groups: weights (weight decay applies) and anything named '*bias*' (no decay)"""
        decay, no_decay = [], []
        for name, p in self.named_parameters():
            if p.requires_grad:
                (no_decay if 'bias' in name else decay).append(p)
        return [{'params': decay}, {'params': no_decay, 'weight_decay': 0}]

    def save_model(self, model_path=None):
        path = model_path or self.model_path
        utils.check_dir(path)
        torch.save(self.state_dict(), path)

    def load_model(self, model_path=None):
        path = model_path or self.model_path
        self.load_state_dict(torch.load(path, map_location=self.device))
        logging.info('Load model from ' + path)

    def count_variables(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def actions_after_train(self):
        pass

    class Dataset(BaseDataset):
        def __init__(self, model, corpus, phase: str):
            self.model, self.corpus, self.phase = model, corpus, phase
            self.buffer_dict = dict()
            self.data = corpus.data_df[phase].to_dict('list')  # column -> python list

        def __len__(self):
            for col in self.data:
                return len(self.data[col])
            return 0

        def __getitem__(self, index: int) -> dict:
            if self.model.buffer and self.phase != 'train':
                return self.buffer_dict[index]
            return self._get_feed_dict(index)

        def _get_feed_dict(self, index: int) -> dict:
            pass

        def prepare(self):
            if self.model.buffer and self.phase != 'train':
                for i in range(len(self)):
                    self.buffer_dict[i] = self._get_feed_dict(i)

        def actions_before_epoch(self):
            pass

        def collate_batch(self, feed_dicts: List[dict]) -> dict:
            """list of per-instance dicts -> dict of CPU tensors; ragged arrays (histories) are
            right-padded with 0 like the reference's pad_sequence(batch_first=True)"""
            batch = dict()
            for key, first in feed_dicts[0].items():
                vals = [d[key] for d in feed_dicts]
                if isinstance(first, np.ndarray) and any(len(v) != len(first) for v in vals):
                    batch[key] = pad_sequence([torch.from_numpy(np.asarray(v)) for v in vals], batch_first=True)
                else:
                    batch[key] = torch.from_numpy(np.array(vals))
            batch['batch_size'] = len(feed_dicts)
            batch['phase'] = self.phase
            return batch


class GeneralModel(BaseModel):
    reader, runner = 'BaseReader', 'BaseRunner'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--num_neg', type=int, default=1, help='The number of negative items during training.')
        parser.add_argument('--dropout', type=float, default=0, help='Dropout probability for each deep layer')
        parser.add_argument('--test_all', type=int, default=0, help='Whether testing on all the items.')
        return BaseModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.user_num, self.item_num = corpus.n_users, corpus.n_items
        self.num_neg, self.dropout, self.test_all = args.num_neg, args.dropout, args.test_all

    def loss(self, out_dict: dict) -> torch.Tensor:
        """softmax-weighted multi-negative BPR (reference :175-189) as ONE HIP kernel:
        -mean_b log clamp( sum_k softmax(neg)_k * sigmoid(pos - neg_k) )"""
        pred = out_dict['prediction']
        if pred.is_cuda:
            return hnn.bpr_loss(pred)
        raise RuntimeError('GeneralModel.loss: the HIP engine needs CUDA tensors (no CPU path)')

    class Dataset(BaseModel.Dataset):
        def _get_feed_dict(self, index):
            user, target = self.data['user_id'][index], self.data['item_id'][index]
            if self.phase != 'train' and self.model.test_all:
                negs = np.arange(1, self.corpus.n_items)
            else:
                negs = self.data['neg_items'][index]
            return {'user_id': user, 'item_id': np.concatenate([[target], negs]).astype(int)}

        def actions_before_epoch(self):
            """uniform negatives in [1, n_items) that the user has not clicked in TRAIN (dev/test
            positives may appear, as in the reference :206-214); vectorised rejection sampling"""
            users = np.asarray(self.data['user_id'])
            n, k, n_items = len(users), self.model.num_neg, self.corpus.n_items
            negs = np.random.randint(1, n_items, size=(n, k))
            clicked = getattr(self, '_clicked_codes', None)
            if clicked is None:
                pairs = [u * n_items + i for u, items in self.corpus.train_clicked_set.items() for i in items]
                clicked = self._clicked_codes = np.unique(np.asarray(pairs, dtype=np.int64))
            codes = users[:, None].astype(np.int64) * n_items + negs
            bad = np.isin(codes, clicked)
            while bad.any():
                fresh = np.random.randint(1, n_items, size=int(bad.sum()))
                negs[bad] = fresh
                rows = np.nonzero(bad)[0]
                still = np.isin(users[rows].astype(np.int64) * n_items + fresh, clicked)
                nb = np.zeros_like(bad)
                nb[bad] = still
                bad = nb
            self.data['neg_items'] = negs


class SequentialModel(GeneralModel):
    reader = 'SeqReader'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--history_max', type=int, default=20, help='Maximum length of history.')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.history_max = args.history_max

    class Dataset(GeneralModel.Dataset):
        def __init__(self, model, corpus, phase):
            super().__init__(model, corpus, phase)
            keep = np.array(self.data['position']) > 0  # instances with an empty history are dropped
            for col in self.data:
                self.data[col] = np.array(self.data[col], dtype=object)[keep].tolist()

        def _get_feed_dict(self, index):
            feed = super()._get_feed_dict(index)
            pos = self.data['position'][index]
            seq = self.corpus.user_his[feed['user_id']][:pos]
            if self.model.history_max > 0:
                seq = seq[-self.model.history_max:]
            feed['history_items'] = np.array([x[0] for x in seq])
            feed['history_times'] = np.array([x[1] for x in seq])
            feed['lengths'] = len(seq)
            return feed


class CTRModel(GeneralModel):
    reader, runner = 'BaseReader', 'CTRRunner'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--loss_n', type=str, default='BCE', help='Type of loss functions.')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.loss_n = args.loss_n

    def loss(self, out_dict: dict) -> torch.Tensor:
        """BCE / MSE on (prediction, label), reference :262-274 (torch ops: not on the ranking path)"""
        if self.loss_n == 'BCE':  # one HIP kernel, closed-form backward
            if 'loss' in out_dict:   # the context models' training forward computed it with the head (rc_ctr_head_fwd_bwd)
                return out_dict['loss']
            if not out_dict['prediction'].is_cuda:
                raise RuntimeError('CTRModel.loss: the HIP engine needs CUDA tensors (no CPU path)')
            return hnn.bce_loss(out_dict['prediction'], out_dict['label'])
        if self.loss_n == 'MSE':
            return ((out_dict['prediction'] - out_dict['label']) ** 2).mean()
        raise ValueError('Undefined loss function: {}'.format(self.loss_n))

    class Dataset(BaseModel.Dataset):
        def _get_feed_dict(self, index):
            return {'user_id': self.data['user_id'][index], 'item_id': [self.data['item_id'][index]],
                    'label': [self.data['label'][index]]}

        def actions_before_epoch(self):  # labelled data: no negative sampling
            pass
