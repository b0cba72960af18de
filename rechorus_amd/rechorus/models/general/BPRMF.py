""" BPRMF on the HIP engine
Reference: "Bayesian personalized ranking from implicit feedback", Rendle et al., UAI'2009.
Counterpart of the reference's models/general/BPRMF.py (same class / flag / state_dict names), e.g.
    python main.py --model_name BPRMF --emb_size 64 --lr 1e-3 --l2 1e-6 --dataset Grocery_and_Gourmet_Food
    python main.py --model_name BPRMF --model_mode Impression --loss_n BPR --dataset MINDCTR ...
Scoring (:39-42 of the reference: two gathers, broadcast multiply, sum) is one HIP kernel with a HIP
backward (rc_gather_dot_fwd + sort / segmented sum); the `u_v` / `i_v` tensors the reference also returns
(:43-45) are consumed by reranker models only and are not materialised.
"""
import torch

from models.BaseImpressionModel import ImpressionModel
from models.BaseModel import GeneralModel, task_variant
from rechorus_amd import engine, nn as hnn


class BPRMFBase(object):
    """user / item tables and the dot-product head, shared by the task variants below"""
    candidate_permutation_equivariant = True  # every candidate is scored on its own

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        return parser

    def _base_init(self, args, corpus):
        self.emb_size, self._trainer = args.emb_size, None
        self._base_define_params()
        self.apply(self.init_weights)

    def _base_define_params(self):
        self.u_embeddings, self.i_embeddings = (hnn.HipEmbedding(rows, self.emb_size) for rows in (self.user_num, self.item_num))

    def forward(self, feed_dict):
        self.check_list = []
        scores = hnn.bprmf_scores(self.u_embeddings.weight, self.i_embeddings.weight,
                                  feed_dict['user_id'], feed_dict['item_id'])  # ids [B], [B, n_candidates]
        out = {'prediction': scores.view(feed_dict['batch_size'], -1)}
        if isinstance(self, ImpressionModel):
            # the reference's BPRMFImpression.forward hands back the base's whole dict (BPRMF.py:43-45, 79-80): the
            # rerankers built on a frozen base ranker read u_v / i_v.  BPRMF proper drops them (:61-63), so only the
            # impression variant pays for the two extra gathers.
            i_ids = feed_dict['item_id']
            out['u_v'] = self.u_embeddings(feed_dict['user_id'])[:, None, :].expand(-1, i_ids.shape[1], -1)
            out['i_v'] = self.i_embeddings(i_ids)
        return out

    def full_catalogue_vectors(self, feed_dict):
        """(query vectors [B, d], item table) of the dot-product head, for --test_all ranking"""
        return engine.gather_rows(self.u_embeddings.weight.detach(), feed_dict['user_id']), self.i_embeddings.weight.detach()

    # ---- large-table mode: the whole fit() iteration as one C-ABI call ---------------------
    def hip_train_step(self, feed_dict, opt_name, lr, l2, next_feed_dict=None):
        """forward + BPR loss + backward + row-wise optimizer update (rc_bprmf_train_step_ahead);
        returns the device loss tensor without synchronising.  next_feed_dict: the batch the following call will bring
        (BaseRunner.fit passes it): its id grouping is prepared beside this step's row updates."""
        if self._trainer is None or self._trainer.opt != opt_name:
            self._trainer = engine.BprmfTrainer(self.u_embeddings.weight.data, self.i_embeddings.weight.data,
                                                opt=opt_name, lr=lr, l2=l2)
        nxt = None
        if next_feed_dict is not None:
            nu, ni = next_feed_dict['user_id'], next_feed_dict['item_id']
            if nu.is_contiguous() and ni.is_contiguous():   # (the following call has to bring these very tensors)
                nxt = (nu, ni)
        with torch.no_grad():
            return self._trainer.step(feed_dict['user_id'].contiguous(), feed_dict['item_id'].contiguous(), next_batch=nxt)


_LOG = ['emb_size', 'batch_size']
BPRMF = task_variant('BPRMF', GeneralModel, BPRMFBase, 'BaseReader', 'BaseRunner', _LOG, __name__,
                     doc='top-k recommendation with sampled negatives (BPR loss)')
BPRMFImpression = task_variant('BPRMFImpression', ImpressionModel, BPRMFBase, 'ImpressionReader', 'ImpressionRunner', _LOG,
                               __name__, doc='ranking inside impression lists (list-level BPR by default)')
