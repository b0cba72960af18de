""" DeepFM on the HIP engine
Reference: 'DeepFM: A Factorization-Machine based Neural Network for CTR Prediction', Guo et al., IJCAI 2017.
Counterpart of the reference's models/context/DeepFM.py (same class / flag / state_dict names), e.g.
    python main.py --model_name DeepFM --model_mode CTR --emb_size 64 --layers '[512,64]' --loss_n BCE --lr 5e-4 \
        --dataset MIND_Large/MINDCTR --include_item_features 1 --include_situation_features 1 --metric AUC,ACC
prediction = first-order term + FM pairwise term (rc_fm_second_order_*) + MLP, all three over the SAME
stacked field vectors (:19-28), which are gathered once.
"""
from models.BaseContextModel import ContextCTRModel, ContextModel
from models.BaseModel import task_variant
from models.context.FM import ctr_forward
from models.context.WideDeep import WideDeepBase
from rechorus_amd import nn as hnn


class DeepFMBase(WideDeepBase):
    fm_term = True

    def forward(self, feed_dict):
        if self.overall_bias.is_cuda:
            field_vectors, linear_value, fm = self._fused_fields(feed_dict)
            first_order = self.overall_bias + linear_value.squeeze(-1).sum(dim=-1)
        else:
            (field_vectors, first_order), fm = self._get_embeddings_FM(feed_dict), None
        fm, deep = self._head_terms(field_vectors, fm)
        return {'prediction': first_order + fm + deep}

    def _head_terms(self, field_vectors, fm=None):
        if fm is not None:          # the gather formed the pairwise term; its backward meets the tower's gradient in the row sums
            return [fm, self._deep(field_vectors)]
        if field_vectors.is_cuda:   # one autograd node for both consumers of the field vectors (their gradients meet in one pass)
            fm, flat = hnn.fm_second_order_and_flat(field_vectors)
            return [fm, self.deep_layers(flat).squeeze(dim=-1)]
        return [hnn.fm_second_order(field_vectors), self._deep(field_vectors)]


_LOG = ['emb_size', 'layers', 'loss_n']
DeepFMCTR = task_variant('DeepFMCTR', ContextCTRModel, DeepFMBase, 'ContextReader', 'CTRRunner', _LOG, __name__,
                         forward=ctr_forward, parse_from=ContextModel)  # (--loss_n quirk: see WideDeep.py)
DeepFMTopK = task_variant('DeepFMTopK', ContextModel, DeepFMBase, 'ContextReader', 'BaseRunner', _LOG, __name__)
DeepFMCTR.candidate_permutation_equivariant = True  # one candidate per row: nothing to shuffle in fit()
