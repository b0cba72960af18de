"""Impression-based ranking base (mirror of the reference's models/BaseImpressionModel.py:9-214): each
instance scores a list of up to --*_max_pos_item positives followed by up to --*_max_neg_item negatives
(right-padded with item 0), `loss(out_dict, target)` takes the runner's {1, 0, -1} labels.

Losses: every name the reference's substring rules resolve -- list-level BPR (the default of every *Impression
model) with its 'hard' / 'after' / 'before' / 'simple' variants, 'listnet', 'softmaxCE', 'attention_rank' -- is one
HIP kernel family with closed-form backward (rc_list_loss_fwd_bwd); 'BPR...simple' returns its per-row sums
unreduced, as the reference does.  ImpressionSeqModel adds the clicked / skipped item histories
(reader ImpressionSeqReader) for sequential heads such as SASRecImpression.
"""
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

from models.BaseModel import GeneralModel, SequentialModel
from rechorus_amd import engine, nn as hnn


class ImpressionModel(GeneralModel):
    reader = 'ImpressionReader'
    runner = 'ImpressionRunner'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--loss_n', type=str, default='BPR',
                            help='BPR[hard][after|before|simple], listnet, softmaxCE or attention_rank')
        parser.add_argument('--train_max_pos_item', type=int, default=20, help='max positive item sample for training')
        parser.add_argument('--train_max_neg_item', type=int, default=20, help='max negative item sample for training')
        parser.add_argument('--test_max_pos_item', type=int, default=20, help='max positive item sample for evaluation')
        parser.add_argument('--test_max_neg_item', type=int, default=20, help='max negative item sample for evaluation')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.loss_n = args.loss_n
        self.train_max_pos_item, self.train_max_neg_item = args.train_max_pos_item, args.train_max_neg_item
        self.test_max_pos_item, self.test_max_neg_item = args.test_max_pos_item, args.test_max_neg_item

    def loss(self, out_dict: dict, target=None):
        pred, P = out_dict['prediction'], self.train_max_pos_item
        name = self.loss_n
        kind = engine.list_kind(name)   # the reference's substring rules (:50-89)
        if kind is not None:
            if not pred.is_cuda:
                raise RuntimeError('ImpressionModel.loss: the HIP engine needs CUDA tensors (no CPU path)')
            return hnn.list_loss(pred, target.long(), P, kind)
        raise ValueError('Undefined loss function: {}'.format(self.loss_n))

    class Dataset(GeneralModel.Dataset):
        def __init__(self, model, corpus, phase: str):
            super().__init__(model, corpus, phase)
            train = self.phase == 'train'
            self.pos_len = model.train_max_pos_item if train else model.test_max_pos_item
            self.neg_len = model.train_max_neg_item if train else model.test_max_neg_item

        def _get_feed_dict(self, index):
            if self.phase != 'train' and self.model.test_all:
                negs = np.arange(1, self.corpus.n_items)
            else:
                negs = self.data['neg_items'][index]
            return {'user_id': self.data['user_id'][index],
                    'pos_items': np.array(self.data['pos_items'][index][:self.pos_len]),
                    'neg_items': np.array(negs[:self.neg_len]),
                    'pos_num': min(self.data['pos_num'][index], self.pos_len),
                    'neg_num': min(self.data['neg_num'][index], self.neg_len)}

        def collate_batch(self, feed_dicts: List[dict]):
            return _fixed_width_lists(super().collate_batch(feed_dicts), self.pos_len, self.neg_len)

        def actions_before_epoch(self):  # impressions bring their own negatives: nothing to sample
            pass


def _fixed_width_lists(batch, pos_len, neg_len):
    """item_id = [positives padded to pos_len | negatives padded to neg_len] (reference :190-201)"""
    def padded(x, width):  # ragged lists were padded to the batch maximum; now to the fixed width
        x = x.long()
        if x.shape[-1] < width:
            x = torch.cat([x, torch.zeros(x.shape[0], width - x.shape[-1], dtype=torch.long)], dim=-1)
        return x
    batch['item_id'] = torch.cat([padded(batch.pop('pos_items'), pos_len), padded(batch.pop('neg_items'), neg_len)], dim=-1)
    return batch


class ImpressionSeqModel(ImpressionModel):
    reader = 'ImpressionSeqReader'
    runner = 'ImpressionRunner'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--history_max', type=int, default=20, help='Maximum length of history.')
        return ImpressionModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.history_max = args.history_max

    class Dataset(SequentialModel.Dataset):
        def __init__(self, model, corpus, phase):
            super().__init__(model, corpus, phase)  # drops impressions with an empty click history
            train = self.phase == 'train'
            self.pos_len = model.train_max_pos_item if train else model.test_max_pos_item
            self.neg_len = model.train_max_neg_item if train else model.test_max_neg_item

        def _get_feed_dict(self, index):
            feed = ImpressionModel.Dataset._get_feed_dict(self, index)
            his = self.corpus.user_his[feed['user_id']]
            clicked = his['pos'][:self.data['position'][index]]
            skipped = his['neg'][:self.data['neg_position'][index]]
            if self.model.history_max > 0:
                clicked, skipped = clicked[-self.model.history_max:], skipped[-self.model.history_max:]
            feed['history_items'] = np.array([x[0] for x in clicked])
            feed['neg_history_items'] = np.array([x[0] for x in skipped])
            feed['history_times'] = np.array([x[1] for x in clicked])
            feed['neg_history_times'] = np.array([x[1] for x in skipped])
            feed['lengths'], feed['neg_lengths'] = len(clicked), len(skipped)
            return feed

        def collate_batch(self, feed_dicts: List[dict]):
            batch = _fixed_width_lists(super().collate_batch(feed_dicts), self.pos_len, self.neg_len)
            batch['history_items'] = batch['history_items'].long()
            batch['neg_history_items'] = batch['neg_history_items'].long()
            return batch

        def actions_before_epoch(self):
            pass

