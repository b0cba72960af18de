"""Context-aware model bases (mirror of the reference's models/BaseContextModel.py:16-89):
ContextModel (top-k ranking with BPR or BCE loss) and ContextCTRModel (labelled click data).
Both add the corpus' user / item / situation features to every feed dict; the model sees them
as extra int (or float) tensors keyed by feature name next to 'user_id' / 'item_id'.

The history-aware ContextSeq* bases (reference :91-166, reader ContextSeqReader) are not part of
this engine's path.
"""
import numpy as np
import torch

from models.BaseModel import GeneralModel, CTRModel
from rechorus_amd import nn as hnn


def get_context_feature(feed_dict, index, corpus, data):
    """user features by user id, situation features by row, item features per candidate"""
    for c in corpus.user_feature_names:
        feed_dict[c] = corpus.user_features[feed_dict['user_id']][c]
    for c in corpus.situation_feature_names:
        feed_dict[c] = data[c][index]
    items = feed_dict['item_id']
    for c in corpus.item_feature_names:
        if isinstance(items, (int, np.integer)):
            feed_dict[c] = corpus.item_features[items][c]
        else:
            feed_dict[c] = np.array([corpus.item_features[i][c] for i in items])
    return feed_dict


def _feature_list(corpus):
    return (corpus.user_feature_names + corpus.item_feature_names + corpus.situation_feature_names
            + ['user_id', 'item_id'])


class ContextModel(GeneralModel):
    reader = 'ContextReader'

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--loss_n', type=str, default='BPR', help='Type of loss functions.')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.loss_n = args.loss_n
        self.context_features = _feature_list(corpus)
        self.feature_max = corpus.feature_max

    def loss(self, out_dict: dict):
        """BPR (the HIP loss kernel of GeneralModel) or point-wise BCE over pos + negs (:49-63)"""
        if self.loss_n == 'BPR':
            return super().loss(out_dict)
        if self.loss_n == 'BCE':  # one HIP kernel, closed-form backward
            if not out_dict['prediction'].is_cuda:
                raise RuntimeError('ContextModel.loss: the HIP engine needs CUDA tensors (no CPU path)')
            return hnn.bce_ranking_loss(out_dict['prediction'])
        raise ValueError('Undefined loss function: {}'.format(self.loss_n))

    class Dataset(GeneralModel.Dataset):
        def _get_feed_dict(self, index):
            feed = super()._get_feed_dict(index)
            return get_context_feature(feed, index, self.corpus, self.data)


class ContextCTRModel(CTRModel):
    reader = 'ContextReader'

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.context_features = _feature_list(corpus)
        self.feature_max = corpus.feature_max

    class Dataset(CTRModel.Dataset):
        def _get_feed_dict(self, index):
            feed = super()._get_feed_dict(index)
            return get_context_feature(feed, index, self.corpus, self.data)
