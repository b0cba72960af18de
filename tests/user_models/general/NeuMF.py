"""neural collaborative filtering (a GMF branch and an MLP branch over separate tables, one prediction layer), torch layers only"""
import ast

import torch
import torch.nn as nn

from models.BaseModel import GeneralModel


class NeuMF(GeneralModel):
    reader, runner = 'BaseReader', 'BaseRunner'
    extra_log_args = ['emb_size', 'layers']

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='width of every table')
        parser.add_argument('--layers', type=str, default='[64]', help='hidden sizes of the MLP branch')
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.emb_size = args.emb_size
        self.layers = list(ast.literal_eval(args.layers))
        d = self.emb_size
        self.mf_u_embeddings = nn.Embedding(self.user_num, d)
        self.mf_i_embeddings = nn.Embedding(self.item_num, d)
        self.mlp_u_embeddings = nn.Embedding(self.user_num, d)
        self.mlp_i_embeddings = nn.Embedding(self.item_num, d)
        widths = [2 * d] + self.layers
        self.mlp = nn.ModuleList([nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])])
        self.dropout_layer = nn.Dropout(p=self.dropout)
        self.prediction = nn.Linear(widths[-1] + d, 1, bias=False)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        items = feed_dict['item_id']                                         # [B, C]
        users = feed_dict['user_id'][:, None].expand_as(items)               # the user id of a row, once per candidate
        gmf = self.mf_u_embeddings(users) * self.mf_i_embeddings(items)
        hidden = torch.cat((self.mlp_u_embeddings(users), self.mlp_i_embeddings(items)), dim=-1)
        for fc in self.mlp:
            hidden = self.dropout_layer(torch.relu(fc(hidden)))
        scores = self.prediction(torch.cat((gmf, hidden), dim=-1)).squeeze(-1)
        return {'prediction': scores.reshape(feed_dict['batch_size'], -1)}
