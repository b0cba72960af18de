""" SASRec on the HIP engine
Reference: "Self-attentive Sequential Recommendation", Kang et al., IEEE ICDM'2018.
Mirror of the reference's models/sequential/SASRec.py (same class / arg / state_dict names):
    python main.py --model_name SASRec --emb_size 64 --num_layers 1 --num_heads 1 --lr 1e-4 --l2 1e-6 \
        --history_max 20 --dataset 'Grocery_and_Gourmet_Food'
The encoder of :58-76 (history + position gather, causal self-attention blocks, last valid row) is
rc_sasrec_fwd / rc_sasrec_bwd; the candidate scoring of :80-81 is the BPRMF gather-dot kernel with
the encoder output as the "user" row.  Training with --dropout p runs the batch-level kernels with the
two nn.Dropout sites of every TransformerLayer (utils/layers.py:104-117) inside them: the mask comes from a
counter-based stream keyed by a device-side seed (rc_sasrec_batch_fwd_dropout), never from torch's RNG.
Shapes outside those kernels' envelope (emb_size other than 32 / 64, history beyond 64 -- 128 with one block and no dropout --,
more than four blocks) run the same parameters layer by layer on the shape-generic HIP kernels of csrc/seq_layers.hip and the
fp32 MFMA GEMMs of csrc/mlp.hip (rechorus_amd.nn.sasrec_encode_layers): any emb_size that is a multiple of 4, any head / block
count, history up to 1,024, dropout.  Torch layers remain for CPU tensors only.
"""
import numpy as np
import torch
import torch.nn as nn

from models.BaseImpressionModel import ImpressionSeqModel
from models.BaseModel import SequentialModel, task_variant
from rechorus_amd import engine, nn as hnn
from utils import layers


class SASRecBase(object):
    candidate_permutation_equivariant = True  # every candidate is scored on its own

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        parser.add_argument('--num_layers', type=int, default=1, help='Number of self-attention layers.')
        parser.add_argument('--num_heads', type=int, default=4, help='Number of attention heads.')
        return parser

    def _base_init(self, args, corpus):
        self.emb_size, self.max_his = args.emb_size, args.history_max
        self.num_layers, self.num_heads = args.num_layers, args.num_heads
        self._base_define_params()
        self.apply(self.init_weights)

    def _base_define_params(self):
        self.i_embeddings = hnn.HipEmbedding(self.item_num, self.emb_size)
        self.p_embeddings = hnn.HipEmbedding(self.max_his + 1, self.emb_size)
        self.transformer_block = nn.ModuleList([
            layers.TransformerLayer(d_model=self.emb_size, d_ff=self.emb_size, n_heads=self.num_heads,
                                    dropout=self.dropout, kq_same=False)
            for _ in range(self.num_layers)])
        # key of the dropout mask stream; not a parameter and not in the state_dict (the reference has no such key)
        self.register_buffer('drop_seed', hnn.fresh_drop_seed(), persistent=False)

    def _encode_torch(self, history, lengths):
        batch_size, seq_len = history.shape
        valid = (history > 0).long()
        position = (lengths[:, None] - torch.arange(seq_len, device=history.device)[None, :]) * valid
        his = self.i_embeddings(history) + self.p_embeddings(position)
        mask = torch.tril(torch.ones((1, 1, seq_len, seq_len), dtype=torch.long, device=history.device))
        for block in self.transformer_block:
            his = block(his, mask)
        his = his * valid[:, :, None].float()
        return his[torch.arange(batch_size, device=history.device), lengths - 1, :]

    def _encode(self, feed_dict):
        history = feed_dict['history_items']    # [batch_size, <= history_max], right padded with 0
        lengths = feed_dict['lengths']          # [batch_size]
        p = float(self.dropout) if self.training else 0.0
        if engine.sasrec_supported(self.emb_size, self.num_layers, self.num_heads, history.shape[1], p):
            if p > 0:
                engine.step_increment(self.drop_seed)  # new mask for this forward; its backward reads the same value
            return hnn.sasrec_encode(self.i_embeddings.weight, self.p_embeddings.weight,
                                     self.transformer_block, self.num_heads, history, lengths,
                                     p, self.drop_seed if p > 0 else None)
        if history.is_cuda:
            if not hnn.sasrec_layers_supported(self.emb_size, self.num_heads, history.shape[1]):
                raise RuntimeError('SASRec on the HIP engine: emb_size {} / {} heads / history {} is outside every kernel (emb_size a '
                                   'multiple of 4 up to 1024 that the heads divide, head width <= 256, history <= 1024)'.format(
                                       self.emb_size, self.num_heads, history.shape[1]))
            if p > 0:
                engine.step_increment(self.drop_seed)
            return hnn.sasrec_encode_layers(self.i_embeddings.weight, self.p_embeddings.weight, self.transformer_block, self.num_heads,
                                            history, lengths, p, self.drop_seed if p > 0 else None)
        return self._encode_torch(history, lengths)     # CPU tensors (construction-time checks, tests without a GPU)

    def full_catalogue_vectors(self, feed_dict):
        """(sequence vectors [B, d], item table) of the dot-product head, for --test_all ranking"""
        return self._encode(feed_dict).detach(), self.i_embeddings.weight.detach()

    def forward(self, feed_dict):
        self.check_list = []
        i_ids = feed_dict['item_id']            # [batch_size, n_candidates]
        batch_size = i_ids.shape[0]
        his_vector = self._encode(feed_dict)
        rows = torch.arange(batch_size, device=i_ids.device)
        prediction = hnn.bprmf_scores(his_vector, self.i_embeddings.weight, rows, i_ids)
        return {'prediction': prediction.view(batch_size, -1)}

    # ---- large-table mode: row-wise update of the item table, dense step of everything small -----------
    def hip_rowwise_supported(self):
        return bool(engine.sasrec_supported(self.emb_size, self.num_layers, self.num_heads, self.max_his, float(self.dropout)))

    def hip_train_step(self, feed_dict, opt_name, lr, l2):
        """encoder fwd/bwd (MFMA) + scoring + BPR loss + ONE segmented pass over candidate and history
        occurrences of the item table (engine.SasrecTrainer); returns the device loss tensor"""
        history, lengths = feed_dict['history_items'], feed_dict['lengths']
        if not engine.sasrec_supported(self.emb_size, self.num_layers, self.num_heads, history.shape[1], float(self.dropout)):
            raise RuntimeError('SASRec --engine rowwise needs the fused encoder: emb_size in {32, 64}, history <= 64 '
                               '(<= 128 with one block, 1 / 2 / 4 heads and no dropout)')
        tr = getattr(self, '_trainer', None)
        if tr is None or tr.opt != opt_name:
            P = {'item_emb': self.i_embeddings.weight.data, 'pos_emb': self.p_embeddings.weight.data,
                 'layers': hnn.sasrec_layer_params(self.transformer_block)}
            from rechorus_amd import graph as hgraph
            # the step replays from a hipGraph where that is possible in this process (rechorus_amd/graph.py); Adam's step
            # count then lives in device memory
            tr = self._trainer = engine.SasrecTrainer(P, self.num_heads, opt=opt_name, lr=lr, l2=l2, rowwise=True,
                                                      dropout=self.dropout, seed=int(self.drop_seed.item()),
                                                      graph=hgraph.usable() and history.is_cuda and opt_name in ('SGD', 'Adam', 'Adagrad'))
        with torch.no_grad():
            return tr.step(history.contiguous(), lengths.contiguous(), feed_dict['item_id'].contiguous())


_LOG = ['emb_size', 'num_layers', 'num_heads']
SASRec = task_variant('SASRec', SequentialModel, SASRecBase, 'SeqReader', 'BaseRunner', _LOG, __name__,
                      doc='next-item recommendation with sampled negatives (BPR loss)')
SASRecImpression = task_variant('SASRecImpression', ImpressionSeqModel, SASRecBase, 'ImpressionSeqReader', 'ImpressionRunner',
                                _LOG, __name__, doc='SASRec scored over impression lists (list-level BPR by default)')
