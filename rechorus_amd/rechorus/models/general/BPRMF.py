""" BPRMF on the HIP engine
Reference: "Bayesian personalized ranking from implicit feedback", Rendle et al., UAI'2009.
Mirror of the reference's models/general/BPRMF.py (same class / arg / state_dict names):
    python main.py --model_name BPRMF --emb_size 64 --lr 1e-3 --l2 1e-6 --dataset 'Grocery_and_Gourmet_Food'
The gather + dot of :39-42 is one HIP kernel (rc_gather_dot_fwd) with a HIP backward; the
`u_v` repeat of :43 is not materialised (only the Impression/reranker variants consume it).
"""
import torch

from models.BaseImpressionModel import ImpressionModel
from models.BaseModel import GeneralModel
from rechorus_amd import engine, nn as hnn


class BPRMFBase(object):
    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        return parser

    def _base_init(self, args, corpus):
        self.emb_size = args.emb_size
        self._base_define_params()
        self.apply(self.init_weights)

    def _base_define_params(self):
        self.u_embeddings = hnn.HipEmbedding(self.user_num, self.emb_size)
        self.i_embeddings = hnn.HipEmbedding(self.item_num, self.emb_size)

    def forward(self, feed_dict):
        self.check_list = []
        u_ids = feed_dict['user_id']  # [batch_size]
        i_ids = feed_dict['item_id']  # [batch_size, n_candidates]
        pred = hnn.bprmf_scores(self.u_embeddings.weight, self.i_embeddings.weight, u_ids, i_ids)
        return {'prediction': pred.view(feed_dict['batch_size'], -1)}


class BPRMF(GeneralModel, BPRMFBase):
    reader = 'BaseReader'
    runner = 'BaseRunner'
    extra_log_args = ['emb_size', 'batch_size']
    candidate_permutation_equivariant = True  # each candidate is scored independently

    @staticmethod
    def parse_model_args(parser):
        parser = BPRMFBase.parse_model_args(parser)
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        GeneralModel.__init__(self, args, corpus)
        self._base_init(args, corpus)
        self._trainer = None

    def forward(self, feed_dict):
        return BPRMFBase.forward(self, feed_dict)

    def full_catalogue_vectors(self, feed_dict):
        """(query vectors [B, d], item table) of the dot-product head, for --test_all ranking"""
        return engine.gather_rows(self.u_embeddings.weight.detach(), feed_dict['user_id']), self.i_embeddings.weight.detach()

    # ---- large-table mode: the whole fit() iteration as one C-ABI call ---------------------
    def hip_train_step(self, feed_dict, opt_name, lr, l2):
        """forward + BPR loss + backward + row-wise optimizer update (rc_bprmf_train_step);
        returns the device loss tensor without synchronising."""
        if self._trainer is None or self._trainer.opt != opt_name:
            self._trainer = engine.BprmfTrainer(self.u_embeddings.weight.data, self.i_embeddings.weight.data,
                                                opt=opt_name, lr=lr, l2=l2)
        with torch.no_grad():
            return self._trainer.step(feed_dict['user_id'].contiguous(), feed_dict['item_id'].contiguous())


class BPRMFImpression(ImpressionModel, BPRMFBase):
    """BPRMF scored over impression lists (reference :65-80).  The `u_v` / `i_v` tensors the reference also
    returns are consumed by reranker models only and are not materialised here."""
    reader = 'ImpressionReader'
    runner = 'ImpressionRunner'
    extra_log_args = ['emb_size', 'batch_size']

    @staticmethod
    def parse_model_args(parser):
        parser = BPRMFBase.parse_model_args(parser)
        return ImpressionModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        ImpressionModel.__init__(self, args, corpus)
        self._base_init(args, corpus)

    def forward(self, feed_dict):
        return BPRMFBase.forward(self, feed_dict)

