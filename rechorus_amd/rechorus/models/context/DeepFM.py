""" DeepFM on the HIP engine
Reference: 'DeepFM: A Factorization-Machine based Neural Network for CTR Prediction', Guo et al., IJCAI 2017.
Mirror of the reference's models/context/DeepFM.py (same class / arg / state_dict names):
    python main.py --model_name DeepFM --model_mode CTR --emb_size 64 --layers '[64,64]' --lr 5e-4 --l2 0 \
        --dataset MIND_Large/MINDCTR --include_item_features 1 --include_situation_features 1 --metric AUC,ACC
prediction = first-order + FM pairwise term (rc_fm_second_order_*) + MLP over the same stacked
field vectors (:19-28); the field vectors are gathered once and shared by both branches.
"""
from models.context.WideDeep import WideDeepBase, WideDeepCTR, WideDeepTopK
from rechorus_amd import nn as hnn


class DeepFMBase(WideDeepBase):
    def forward(self, feed_dict):
        context_vectors, linear_vectors = self._get_embeddings_FM(feed_dict)
        fm_prediction = hnn.fm_second_order(context_vectors) + linear_vectors
        deep_prediction = self.deep_layers(context_vectors.flatten(start_dim=-2)).squeeze(dim=-1)
        return {'prediction': fm_prediction + deep_prediction}


class DeepFMCTR(WideDeepCTR, DeepFMBase):
    reader, runner = 'ContextReader', 'CTRRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    def __init__(self, args, corpus):
        WideDeepCTR.__init__(self, args, corpus)

    def forward(self, feed_dict):
        out_dict = DeepFMBase.forward(self, feed_dict)
        out_dict['prediction'] = out_dict['prediction'].view(-1).sigmoid()
        out_dict['label'] = feed_dict['label'].view(-1)
        return out_dict


class DeepFMTopK(WideDeepTopK, DeepFMBase):
    reader, runner = 'ContextReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    def __init__(self, args, corpus):
        WideDeepTopK.__init__(self, args, corpus)

    def forward(self, feed_dict):
        return DeepFMBase.forward(self, feed_dict)
