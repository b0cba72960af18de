"""factorization machine over the context features of a (user, item, situation) row: a vector table and a first-order table per
categorical feature, a bias-free Linear on the value of a numeric one; torch layers only.  FMCTR: click probability (BCE through the
framework's CTR task); FMTopK: ranking scores"""
import torch
import torch.nn as nn

from models.BaseContextModel import ContextCTRModel, ContextModel


def _categorical(name):
    return name.endswith('_c') or name.endswith('_id')


class FMBase(object):
    @staticmethod
    def parse_model_args_FM(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='width of the field vectors')
        return parser

    def _build_fields(self, args):
        self.vec_size = args.emb_size
        self.context_embedding, self.linear_embedding = nn.ModuleDict(), nn.ModuleDict()
        for name in self.context_features:
            if _categorical(name):
                self.context_embedding[name] = nn.Embedding(self.feature_max[name], self.vec_size)
                self.linear_embedding[name] = nn.Embedding(self.feature_max[name], 1)
            else:
                self.context_embedding[name] = nn.Linear(1, self.vec_size, bias=False)
                self.linear_embedding[name] = nn.Linear(1, 1, bias=False)
        self.overall_bias = nn.Parameter(torch.tensor([0.01]))

    def _fields(self, family, feed_dict):
        """per feature [B, C, width]: a per-row feature is repeated over the row's candidates"""
        C = feed_dict['item_id'].shape[1]
        out = []
        for name in self.context_features:
            x = feed_dict[name]
            v = family[name](x) if _categorical(name) else family[name](x.float()[..., None])
            out.append(v if v.dim() == 3 else v[:, None, :].expand(-1, C, -1))
        return out

    def _vectors_and_first_order(self, feed_dict):
        vectors = torch.stack(self._fields(self.context_embedding, feed_dict), dim=2)                 # [B, C, F, d]
        first = torch.cat(self._fields(self.linear_embedding, feed_dict), dim=-1).sum(-1)            # [B, C]
        return vectors, self.overall_bias + first

    def _scores(self, feed_dict):
        vectors, first = self._vectors_and_first_order(feed_dict)
        pairwise = 0.5 * (vectors.sum(dim=2).square() - vectors.square().sum(dim=2)).sum(-1)
        return first + pairwise


class FMCTR(ContextCTRModel, FMBase):
    reader, runner = 'ContextReader', 'CTRRunner'
    extra_log_args = ['emb_size', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextCTRModel.parse_model_args(FMBase.parse_model_args_FM(parser))

    def __init__(self, args, corpus):
        ContextCTRModel.__init__(self, args, corpus)
        self._build_fields(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict).view(-1).sigmoid(), 'label': feed_dict['label'].view(-1)}


class FMTopK(ContextModel, FMBase):
    reader, runner = 'ContextReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        return ContextModel.parse_model_args(FMBase.parse_model_args_FM(parser))

    def __init__(self, args, corpus):
        ContextModel.__init__(self, args, corpus)
        self._build_fields(args)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        return {'prediction': self._scores(feed_dict)}
