# -*- coding: UTF-8 -*-
""" DropinGRU
A sequential recommender written the way a ReChorus user writes a model file -- against the reference's plugin
surface only: `from models.BaseModel import SequentialModel`, plain torch.nn layers, `self.apply(self.init_weights)`,
a `forward(feed_dict)` that returns {'prediction': [batch_size, n_candidates]}, the inherited BPR loss.  It knows
nothing about rechorus_amd (no HipEmbedding, no engine calls, no hooks): tests/test_gpu_dropin.py trains it through
the plugin's main.py to show that such a file drops in.
CMD example:
    python main.py --model_name DropinGRU --emb_size 32 --hidden_size 48 --history_max 10 --dataset synth
"""
import torch
import torch.nn as nn

from models.BaseModel import SequentialModel


class DropinGRU(SequentialModel):
    reader = 'SeqReader'
    runner = 'BaseRunner'
    extra_log_args = ['emb_size', 'hidden_size']

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--emb_size', type=int, default=64, help='Size of embedding vectors.')
        parser.add_argument('--hidden_size', type=int, default=64, help='Size of the GRU state.')
        return SequentialModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.emb_size = args.emb_size
        self.hidden_size = args.hidden_size
        self.i_embeddings = nn.Embedding(self.item_num, self.emb_size)
        self.u_embeddings = nn.Embedding(self.user_num, self.emb_size)
        self.rnn = nn.GRU(input_size=self.emb_size, hidden_size=self.hidden_size, batch_first=True)
        self.out = nn.Linear(self.hidden_size, self.emb_size)
        self.apply(self.init_weights)

    def forward(self, feed_dict):
        self.check_list = []
        i_ids = feed_dict['item_id']            # [batch_size, n_candidates]
        history = feed_dict['history_items']    # [batch_size, history_max], right padded with 0
        lengths = feed_dict['lengths']          # [batch_size]
        his_vectors = self.i_embeddings(history)
        states, _ = self.rnn(his_vectors)       # padded steps run too; the state at the last VALID step is picked
        last = states[torch.arange(len(lengths), device=states.device), lengths - 1]
        query = self.out(last) + self.u_embeddings(feed_dict['user_id'])
        prediction = (query[:, None, :] * self.i_embeddings(i_ids)).sum(-1)
        return {'prediction': prediction.view(feed_dict['batch_size'], -1)}
