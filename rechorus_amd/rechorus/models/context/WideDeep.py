""" WideDeep on the HIP engine
Reference: 'Wide & Deep Learning for Recommender Systems', Cheng et al., DLRS 2016.
Mirror of the reference's models/context/WideDeep.py (same class / arg / state_dict names):
    python main.py --model_name WideDeep --model_mode CTR --emb_size 64 --layers '[64,64]' --lr 5e-4 --l2 0 \
        --dataset MIND_Large/MINDCTR --include_item_features 1 --include_situation_features 1
wide = FM's first-order term (HIP gathers of the [*, 1] tables), deep = MLP_Block over the
flattened field vectors (HIP gathers feeding rocBLAS GEMMs).
"""
from models.BaseContextModel import ContextModel, ContextCTRModel
from models.context.FM import FMBase
from utils.layers import MLP_Block


class WideDeepBase(FMBase):
    @staticmethod
    def parse_model_args_WD(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        parser.add_argument('--layers', type=str, default='[64]', help="Size of each layer.")
        return parser

    def _define_init(self, args, corpus):
        self._define_init_params(args, corpus)
        self.layers = eval(args.layers)
        self._define_params_WD()
        self.apply(self.init_weights)

    def _define_params_WD(self):
        self._define_params_FM()
        self.deep_layers = MLP_Block(len(self.context_features) * self.vec_size, self.layers,
                                     hidden_activations="ReLU", batch_norm=False, dropout_rates=self.dropout,
                                     output_dim=1)

    def forward(self, feed_dict):
        deep_vectors, wide_prediction = self._get_embeddings_FM(feed_dict)
        deep_prediction = self.deep_layers(deep_vectors.flatten(start_dim=-2)).squeeze(dim=-1)
        return {'prediction': deep_prediction + wide_prediction}


class WideDeepCTR(ContextCTRModel, WideDeepBase):
    reader, runner = 'ContextReader', 'CTRRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        parser = WideDeepBase.parse_model_args_WD(parser)
        # like the reference (:53-55) the CTR variant takes ContextModel's args: --loss_n defaults to 'BPR' ...
        return ContextModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        ContextCTRModel.__init__(self, args, corpus)
        self._define_init(args, corpus)

    def forward(self, feed_dict):
        out_dict = WideDeepBase.forward(self, feed_dict)
        out_dict['prediction'] = out_dict['prediction'].view(-1).sigmoid()
        out_dict['label'] = feed_dict['label'].view(-1)
        return out_dict


class WideDeepTopK(ContextModel, WideDeepBase):
    reader, runner = 'ContextReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'layers', 'loss_n']

    @staticmethod
    def parse_model_args(parser):
        parser = WideDeepBase.parse_model_args_WD(parser)
        return ContextModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        ContextModel.__init__(self, args, corpus)
        self._define_init(args, corpus)

    def forward(self, feed_dict):
        return WideDeepBase.forward(self, feed_dict)
