""" WideDeep on the HIP engine
Reference: 'Wide & Deep Learning for Recommender Systems', Cheng et al., DLRS 2016.
Counterpart of the reference's models/context/WideDeep.py (same class / flag / state_dict names), e.g.
    python main.py --model_name WideDeep --model_mode CTR --emb_size 64 --layers '[64,64]' --loss_n BCE \
        --dataset MIND_Large/MINDCTR --include_item_features 1 --include_situation_features 1
wide part = FM's first-order term (the [vocab, 1] tables, one rc_gather_fields launch), deep part =
MLP_Block over the flattened field vectors (a second rc_gather_fields launch feeding the fp32 MFMA GEMMs of csrc/mlp.hip: rc_linear_fwd / rc_linear_bwd).
"""
from models.BaseContextModel import ContextCTRModel, ContextModel
from models.BaseModel import task_variant
from models.context.FM import FMBase, ctr_forward
from utils.layers import MLP_Block


class WideDeepBase(FMBase):
    fm_term = False

    @staticmethod
    def parse_model_args_WD(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        parser.add_argument('--layers', type=str, default='[64]', help="Size of each layer.")
        return parser

    parse_model_args = parse_model_args_WD

    def _define_init(self, args, corpus):
        self._define_init_params(args, corpus)
        self.layers = eval(args.layers)
        self._define_params_WD()
        self.apply(self.init_weights)

    def _define_params_WD(self):
        self._define_params_FM()
        width = len(self.context_features) * self.vec_size
        self.deep_layers = MLP_Block(width, self.layers, hidden_activations="ReLU", batch_norm=False,
                                     dropout_rates=self.dropout, output_dim=1)

    def _early_seed(self):
        # MLP_Block.forward bumps its dropout seed before its first layer (engine.step_increment); every forward of this head runs
        # the tower after the gather, so the gather's launch does it (engine.step_increment then finds it done)
        return getattr(self.deep_layers, 'drop_seed', None) if (self.training and self._rows_opt() is not None) else None

    def _deep(self, field_vectors):
        return self.deep_layers(field_vectors.flatten(start_dim=-2)).squeeze(dim=-1)

    def forward(self, feed_dict):
        field_vectors, wide = self._get_embeddings_FM(feed_dict)
        return {'prediction': self._deep(field_vectors) + wide}

    def _head_terms(self, field_vectors, fm=None):
        return [self._deep(field_vectors)]


_LOG = ['emb_size', 'layers', 'loss_n']
# like the reference (:51-54) the CTR variant takes its task flags from ContextModel, so --loss_n defaults to 'BPR'
# there and every CTR script passes --loss_n BCE
WideDeepCTR = task_variant('WideDeepCTR', ContextCTRModel, WideDeepBase, 'ContextReader', 'CTRRunner', _LOG, __name__,
                           forward=ctr_forward, parse_from=ContextModel)
WideDeepTopK = task_variant('WideDeepTopK', ContextModel, WideDeepBase, 'ContextReader', 'BaseRunner', _LOG, __name__)
WideDeepCTR.candidate_permutation_equivariant = True  # one candidate per row: nothing to shuffle in fit()
