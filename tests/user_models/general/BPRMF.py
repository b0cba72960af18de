"""matrix factorisation with the BPR loss, the way a user of the framework writes a model file: torch layers only"""
import torch
import torch.nn as nn

from models.BaseModel import GeneralModel


class BPRMF(GeneralModel):
    reader, runner = 'BaseReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'batch_size']

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='width of the user / item vectors')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.emb_size = args.emb_size
        self.u_embeddings = nn.Embedding(self.user_num, self.emb_size)
        self.i_embeddings = nn.Embedding(self.item_num, self.emb_size)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        users = self.u_embeddings(feed_dict['user_id'])          # [B, d]
        items = self.i_embeddings(feed_dict['item_id'])          # [B, C, d]
        scores = torch.einsum('bd,bcd->bc', users, items)
        return {'prediction': scores.reshape(feed_dict['batch_size'], -1)}
