""" FM on the HIP engine
Reference: 'Factorization Machines', Steffen Rendle, ICDM 2010.
Mirror of the reference's models/context/FM.py (same class / arg / state_dict names):
    python main.py --model_name FM --model_mode CTR --emb_size 64 --lr 5e-4 --l2 0 --dataset MIND_Large/MINDCTR \
        --include_item_features 1 --include_situation_features 1 --metric AUC,ACC
Every categorical field ('*_c', '*_id') owns two tables: context_embedding[f] [feature_max[f], d]
and linear_embedding[f] [feature_max[f], 1]; all F lookups of a table family (:49-55) are ONE
rc_gather_fields launch writing the stacked [B, C, F, d] block, their gradients ONE atomic-free
sort + segmented sum over the composite (field, id) key.  The pairwise-interaction term (:61)
    0.5 * sum_k ((sum_f v_fk)^2 - sum_f v_fk^2)
is one kernel pair (rc_fm_second_order_fwd / _bwd) over the stacked field vectors.  Numeric
fields (:38-41: Linear(1, d, bias=False) on the feature's value, e.g. MIND's c_day_f) keep the
reference's parameters and ride in the same gather launch (rc_gather_fields_mixed: the field's
"row" is x * W[:, 0]); their weight gradients are one weighted column sum (rc_numeric_field_grads).
"""
import torch
import torch.nn as nn

from models.BaseContextModel import ContextCTRModel, ContextModel
from models.BaseModel import task_variant
from rechorus_amd import engine, nn as hnn


def is_categorical(feature_name):
    return feature_name.endswith('_c') or feature_name.endswith('_id')


class FMBase(object):
    @staticmethod
    def parse_model_args_FM(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        return parser

    parse_model_args = parse_model_args_FM  # the head's flags, under the name task_variant chains

    def _base_init(self, args, corpus):
        self._define_init(args, corpus)

    def _define_init_params(self, args, corpus):
        self.vec_size = args.emb_size
        self._define_params_FM()
        self.apply(self.init_weights)

    def _define_init(self, args, corpus):
        self._define_init_params(args, corpus)
        self._define_params_FM()
        self.apply(self.init_weights)

    def _define_params_FM(self):
        self.context_embedding = nn.ModuleDict()
        self.linear_embedding = nn.ModuleDict()
        for f in self.context_features:
            if is_categorical(f):
                self.context_embedding[f] = hnn.HipEmbedding(self.feature_max[f], self.vec_size)
                self.linear_embedding[f] = hnn.HipEmbedding(self.feature_max[f], 1)
            else:
                self.context_embedding[f] = nn.Linear(1, self.vec_size, bias=False)
                self.linear_embedding[f] = nn.Linear(1, 1, bias=False)
        self.overall_bias = torch.nn.Parameter(torch.tensor([0.01]), requires_grad=True)

    def _lookup(self, tables, feed_dict, n_cand):
        """per field: [B, C, w] (per-user / per-row fields broadcast over the C candidates)"""
        out = []
        for f in self.context_features:
            x = feed_dict[f]
            v = tables[f](x) if is_categorical(f) else tables[f](x.float().unsqueeze(-1))
            out.append(v if v.dim() == 3 else v.unsqueeze(-2).expand(-1, n_cand, -1))
        return out

    def _field_kinds(self, feed_dict):
        """per field engine.FIELD_IDS, or the value type of a numeric feature (what `.float()` of :47-48 starts from); None when
        every field is categorical"""
        if all(is_categorical(f) for f in self.context_features):
            return None
        return [engine.FIELD_IDS if is_categorical(f) else engine.field_kind(feed_dict[f]) for f in self.context_features]

    def _get_embeddings_FM(self, feed_dict):
        """-> field vectors [B, C, F, d], first-order term [B, C]"""
        if self.overall_bias.is_cuda:
            fm_vectors, linear_value = self._fused_fields(feed_dict)[:2]
            return fm_vectors, self.overall_bias + linear_value.squeeze(-1).sum(dim=-1)
        n_cand = feed_dict['item_id'].shape[1]
        fm_vectors = torch.stack(self._lookup(self.context_embedding, feed_dict, n_cand), dim=-2)
        linear_value = torch.cat(self._lookup(self.linear_embedding, feed_dict, n_cand), dim=-1)
        return fm_vectors, self.overall_bias + linear_value.sum(dim=-1)

    fm_term = True      # the head adds the pairwise term of :61 (WideDeep: no)

    def forward(self, feed_dict):
        if self.overall_bias.is_cuda:
            fm_vectors, linear_value, fm = self._fused_fields(feed_dict)
            first_order = self.overall_bias + linear_value.squeeze(-1).sum(dim=-1)
            return {'prediction': first_order + sum(self._head_terms(fm_vectors, fm))}
        fm_vectors, linear_value = self._get_embeddings_FM(feed_dict)
        return {'prediction': linear_value + hnn.fm_second_order(fm_vectors)}

    def _fused_fields(self, feed_dict):
        """(field vectors [B, C, F, d], first-order values [B, C, F, 1], FM pairwise term [B, C] | None): every field of both
        families (:49-55) -- the [vocab, d] / [vocab, 1] tables of the categorical fields and the Linear(1, d) / Linear(1, 1) of
        the numeric ones -- in ONE gather launch, which also forms the pairwise term of :61 for the heads that add it (and, at small
        batches, groups the composite (field, id) keys for the backward pass); the backward is one row-sums launch for all dense
        table gradients with the numeric fields' weighted column sums and the pairwise term's backward inside it.  None where the
        model is not on the GPU"""
        if not self.overall_bias.is_cuda:
            return None
        n_cand = feed_dict['item_id'].shape[1]
        ids = [feed_dict[f] for f in self.context_features]
        out = hnn.gather_fields_pair([self.context_embedding[f].weight for f in self.context_features],
                                     [self.linear_embedding[f].weight for f in self.context_features], ids, n_cand,
                                     rows_opt=self._rows_opt(), kinds=self._field_kinds(feed_dict), fm=self.fm_term,
                                     bump=self._early_seed())
        return out if self.fm_term else out + (None,)

    def _early_seed(self):
        """a device counter whose per-forward increment the gather's launch can perform on its owner's behalf (WideDeep / DeepFM:
        the dropout seed of the deep tower, which runs right after the gather)"""
        return None

    def _rows_opt(self):
        """the optimizer, while this forward is part of a whole training step driven by graph.GraphedStep (forward, backward and
        optimizer.step() as one unit): small batches then take HipOptimizer's rows mode"""
        return getattr(self, '_step_optimizer', None) if self.training else None

    def _head_terms(self, field_vectors, fm=None):
        """what `forward` adds to the first-order term, as a list of [B, C] tensors (at most two); fm: the pairwise term where the
        gather formed it"""
        return [fm if fm is not None else hnn.fm_second_order(field_vectors)]


def ctr_forward(self, feed_dict, head_forward):
    """CTR variants: one candidate per row, probability out, label passed through (reference :74-78).
    Training with --loss_n BCE: the sum of the head's terms, the sigmoid, nn.BCELoss and their backward are ONE kernel
    (rc_ctr_head_fwd_bwd); the loss travels in the output dict and CTRModel.loss hands it on."""
    if self.training and getattr(self, 'loss_n', None) == 'BCE' and feed_dict['label'].dtype == torch.int64:
        fused = self._fused_fields(feed_dict)
        if fused is not None:
            field_vectors, lin, fm = fused
            p, loss = hnn.ctr_head(self.overall_bias, lin, feed_dict['label'], self._head_terms(field_vectors, fm))
            return {'prediction': p, 'label': feed_dict['label'].view(-1), 'loss': loss}
    out = head_forward(self, feed_dict)
    out['prediction'] = out['prediction'].view(-1).sigmoid()
    out['label'] = feed_dict['label'].view(-1)
    return out


_LOG = ['emb_size', 'loss_n']
FMCTR = task_variant('FMCTR', ContextCTRModel, FMBase, 'ContextReader', 'CTRRunner', _LOG, __name__, forward=ctr_forward)
FMTopK = task_variant('FMTopK', ContextModel, FMBase, 'ContextReader', 'BaseRunner', _LOG, __name__)
FMCTR.candidate_permutation_equivariant = True  # one candidate per row: nothing to shuffle in fit()
