""" NeuMF on the HIP engine
Reference: "Neural Collaborative Filtering", Xiangnan He et al., WWW'2017.
Mirror of the reference's models/general/NeuMF.py (same class / arg / state_dict names):
    python main.py --model_name NeuMF --emb_size 64 --layers '[64]' --lr 5e-4 --l2 1e-7 --dataset 'Grocery_and_Gourmet_Food'
With one hidden layer and emb_size and layer size in {32, 64, 128}, the whole head (:61-75: four
gathers, GMF product, MLP, dropout, prediction layer) is the fp32-MFMA kernel pair rc_neumf_fwd(_dropout)
/ rc_neumf_bwd(_dropout); the dropout mask comes from a counter-based stream keyed by a device-side seed
that is bumped every training forward (so a captured step replays with fresh masks).  Any other
configuration runs the same parameters through HipEmbedding gathers + torch layers.
"""
import torch
import torch.nn as nn

from models.BaseModel import GeneralModel
from rechorus_amd import engine, nn as hnn


class NeuMF(GeneralModel):
    reader = 'BaseReader'
    runner = 'BaseRunner'
    extra_log_args = ['emb_size', 'layers']
    candidate_permutation_equivariant = True

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        parser.add_argument('--layers', type=str, default='[64]', help="Size of each layer.")
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.emb_size = args.emb_size
        self.layers = eval(args.layers)
        self._define_params()
        self.apply(self.init_weights)

    def _define_params(self):
        self.mf_u_embeddings = hnn.HipEmbedding(self.user_num, self.emb_size)
        self.mf_i_embeddings = hnn.HipEmbedding(self.item_num, self.emb_size)
        self.mlp_u_embeddings = hnn.HipEmbedding(self.user_num, self.emb_size)
        self.mlp_i_embeddings = hnn.HipEmbedding(self.item_num, self.emb_size)
        self.mlp = nn.ModuleList([])
        pre_size = 2 * self.emb_size
        for layer_size in self.layers:
            self.mlp.append(nn.Linear(pre_size, layer_size))
            pre_size = layer_size
        self.dropout_layer = nn.Dropout(p=self.dropout)
        self.prediction = nn.Linear(pre_size + self.emb_size, 1, bias=False)
        # key of the dropout mask stream; not a parameter and not in the state_dict (the reference has no such key)
        self.register_buffer('drop_seed', hnn.fresh_drop_seed(), persistent=False)

    def _fused_ok(self):
        return len(self.layers) == 1 and engine.neumf_supported(self.emb_size, self.layers[0])

    def _drop_p(self):
        return float(self.dropout) if self.training else 0.0

    def forward(self, feed_dict):
        self.check_list = []
        u_ids = feed_dict['user_id']  # [batch_size]
        i_ids = feed_dict['item_id']  # [batch_size, n_candidates]
        if self._fused_ok():
            p = self._drop_p()
            if p > 0:
                engine.step_increment(self.drop_seed)  # new mask for this forward; its backward reads the same value
            pred = hnn.neumf_scores(self.mf_u_embeddings.weight, self.mf_i_embeddings.weight,
                                    self.mlp_u_embeddings.weight, self.mlp_i_embeddings.weight,
                                    self.mlp[0].weight, self.mlp[0].bias, self.prediction.weight, u_ids, i_ids,
                                    p, self.drop_seed if p > 0 else None)
            return {'prediction': pred.view(feed_dict['batch_size'], -1)}
        # any other tower (--layers '[64,32]', sizes outside {32, 64, 128}): table lookups on HipEmbedding, every
        # Linear -> ReLU -> Dropout of the loop at reference :69-72 and the prediction layer as fp32 MFMA GEMMs with
        # the epilogue fused (rc_linear_fwd / rc_linear_bwd); torch only concatenates
        u_rep = u_ids.unsqueeze(-1).repeat((1, i_ids.shape[1]))
        mf = self.mf_u_embeddings(u_rep) * self.mf_i_embeddings(i_ids)
        h = torch.cat([self.mlp_u_embeddings(u_rep), self.mlp_i_embeddings(i_ids)], dim=-1)
        p = self._drop_p()
        if p > 0:
            engine.step_increment(self.drop_seed)
        for k, layer in enumerate(self.mlp):
            h = hnn.linear(h, layer.weight, layer.bias, relu=True, drop_p=p, seed=self.drop_seed if p > 0 else None, site=k)
        pred = hnn.linear(torch.cat([mf, h], dim=-1), self.prediction.weight, None)
        return {'prediction': pred.view(feed_dict['batch_size'], -1)}

    # ---- large-table mode: row-wise update of the four tables, no dense [n_rows, d] gradient -----------
    def _step_ok(self):
        """engine.NeumfTrainer can train this tower: the three-kernel chain covers it (_fused_ok), or -- hidden 16, a shape of the
        one-kernel step only -- NeumfTrainer.step would really select that step for this model's 1 + num_neg candidates per tuple
        and the process's switches (the same tests it runs: engine.neumf_fused_step_selected)"""
        return len(self.layers) == 1 and (self._fused_ok() or
                                          engine.neumf_fused_step_selected(1 + int(self.num_neg), self.emb_size, self.layers[0]))

    def hip_rowwise_supported(self):
        return self._step_ok()

    def hip_train_step(self, feed_dict, opt_name, lr, l2, next_feed_dict=None):
        """one fit iteration on engine.NeumfTrainer: ONE kernel for forward (dropout mask included), BPR loss, backward and the in-place
        update of single-occurrence item rows (rc_neumf_train_step) + the plan's pair updates + the dense step of the MLP;
        returns the device loss tensor.  next_feed_dict: the batch the following call will bring (BaseRunner.fit passes it): its
        bucket plan is built beside this step's table updates."""
        if not self._step_ok():
            raise RuntimeError('NeuMF --engine rowwise needs the fused head: one hidden layer, emb_size in {32, 64, 128} and layer '
                               'size in {16, 32, 64, 128}')
        tr = getattr(self, '_trainer', None)
        if tr is None or tr.opt != opt_name:
            P = {'mf_u': self.mf_u_embeddings.weight.data, 'mf_i': self.mf_i_embeddings.weight.data,
                 'mlp_u': self.mlp_u_embeddings.weight.data, 'mlp_i': self.mlp_i_embeddings.weight.data,
                 'W1': self.mlp[0].weight.data, 'b1': self.mlp[0].bias.data,
                 'w_out': self.prediction.weight.data.view(-1)}
            tr = self._trainer = engine.NeumfTrainer(P, opt=opt_name, lr=lr, l2=l2, rowwise=True,
                                                     dropout=self.dropout, seed=int(self.drop_seed.item()))
        nxt = None
        if next_feed_dict is not None:
            nu, ni = next_feed_dict['user_id'], next_feed_dict['item_id']
            if nu.is_contiguous() and ni.is_contiguous():   # (the following call has to bring these very tensors)
                nxt = (nu, ni)
        with torch.no_grad():
            return tr.step(feed_dict['user_id'].contiguous(), feed_dict['item_id'].contiguous(), next_batch=nxt)

