"""DeepFM: first-order term + FM pairwise term + an MLP, all over the same field vectors; torch layers only"""
import importlib.util
import os

from models.BaseContextModel import ContextCTRModel, ContextModel

_spec = importlib.util.spec_from_file_location('rechorus_user_models._WideDeep', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'WideDeep.py'))
_wide_deep = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_wide_deep)      # (`from models.context.WideDeep import ...` would find the framework's own WideDeep.py)
WideDeepBase = _wide_deep.WideDeepBase


class DeepFMBase(WideDeepBase):
    def _scores(self, feed_dict):
        vectors, first = self._vectors_and_first_order(feed_dict)
        pairwise = 0.5 * (vectors.sum(dim=2).square() - vectors.square().sum(dim=2)).sum(-1)
        return first + pairwise + self.deep_layers(vectors.flatten(start_dim=-2)).squeeze(-1)


class DeepFMCTR(ContextCTRModel, DeepFMBase):
    reader, runner = 'ContextReader', 'CTRRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextModel.parse_model_args(DeepFMBase.parse_model_args_WD(parser))

    def __init__(self, args, corpus):
        ContextCTRModel.__init__(self, args, corpus)
        self._build_tower(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict).view(-1).sigmoid(), 'label': feed_dict['label'].view(-1)}


class DeepFMTopK(ContextModel, DeepFMBase):
    reader, runner = 'ContextReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextModel.parse_model_args(DeepFMBase.parse_model_args_WD(parser))

    def __init__(self, args, corpus):
        ContextModel.__init__(self, args, corpus)
        self._build_tower(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict)}
