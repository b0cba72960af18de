"""self-attentive sequential recommendation: item + position vectors, causal transformer blocks, the last valid position scores
the candidates; torch layers only (the blocks are the framework's utils.layers.TransformerLayer)"""
import torch
import torch.nn as nn

from models.BaseModel import SequentialModel
from utils.layers import TransformerLayer


class SASRec(SequentialModel):
    reader, runner = 'SeqReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'num_layers', 'num_heads']

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='model width')
        parser.add_argument('--num_layers', type=int, default=1, help='transformer blocks')
        parser.add_argument('--num_heads', type=int, default=4, help='attention heads')
        return SequentialModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.emb_size, self.max_his = args.emb_size, args.history_max
        self.num_layers, self.num_heads = args.num_layers, args.num_heads
        self.i_embeddings = nn.Embedding(self.item_num, self.emb_size)
        self.p_embeddings = nn.Embedding(self.max_his + 1, self.emb_size)
        self.transformer_block = nn.ModuleList(
            TransformerLayer(d_model=self.emb_size, d_ff=self.emb_size, n_heads=self.num_heads, dropout=self.dropout, kq_same=False)
            for _ in range(self.num_layers))
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        self.check_list = []
        history, lengths = feed_dict['history_items'], feed_dict['lengths']       # [B, L] right-padded with 0, [B]
        B, L = history.shape
        steps = torch.arange(L, device=history.device)
        real = history > 0
        # position 1 = the most recent item of the sequence; padding slots look up position 0
        positions = (lengths[:, None] - steps[None, :]) * real
        x = self.i_embeddings(history) + self.p_embeddings(positions)
        causal = (steps[None, :] <= steps[:, None]).view(1, 1, L, L).long()
        for block in self.transformer_block:
            x = block(x, causal)
        x = x * real[:, :, None].float()
        last = x[torch.arange(B, device=history.device), lengths - 1]            # [B, d]
        candidates = self.i_embeddings(feed_dict['item_id'])                      # [B, C, d]
        return {'prediction': (last[:, None, :] * candidates).sum(-1).reshape(B, -1)}
