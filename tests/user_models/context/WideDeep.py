"""wide & deep: the first-order (wide) term of the FM file plus an MLP over the concatenated field vectors; torch layers only"""
import ast
import importlib.util
import os

from models.BaseContextModel import ContextCTRModel, ContextModel
from utils.layers import MLP_Block


def _sibling(name):
    """the model file of the same directory (`from models.context.FM import ...` would find the framework's own FM.py)"""
    spec = importlib.util.spec_from_file_location('rechorus_user_models._' + name, os.path.join(os.path.dirname(os.path.abspath(__file__)), name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


FMBase = _sibling('FM').FMBase


class WideDeepBase(FMBase):
    @staticmethod
    def parse_model_args_WD(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='width of the field vectors')
        parser.add_argument('--layers', type=str, default='[64]', help='hidden sizes of the deep part')
        return parser

    def _build_tower(self, args):
        self._build_fields(args)
        self.layers = list(ast.literal_eval(args.layers))
        self.deep_layers = MLP_Block(len(self.context_features) * self.vec_size, self.layers, hidden_activations='ReLU', batch_norm=False,
                                     dropout_rates=self.dropout, output_dim=1)

    def _scores(self, feed_dict):
        vectors, wide = self._vectors_and_first_order(feed_dict)
        return wide + self.deep_layers(vectors.flatten(start_dim=-2)).squeeze(-1)


class WideDeepCTR(ContextCTRModel, WideDeepBase):
    reader, runner = 'ContextReader', 'CTRRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextModel.parse_model_args(WideDeepBase.parse_model_args_WD(parser))     # (--loss_n comes with the ranking task's flags)

    def __init__(self, args, corpus):
        ContextCTRModel.__init__(self, args, corpus)
        self._build_tower(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict).view(-1).sigmoid(), 'label': feed_dict['label'].view(-1)}


class WideDeepTopK(ContextModel, WideDeepBase):
    reader, runner = 'ContextReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextModel.parse_model_args(WideDeepBase.parse_model_args_WD(parser))

    def __init__(self, args, corpus):
        ContextModel.__init__(self, args, corpus)
        self._build_tower(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict)}
