"""BaseReader: csv -> corpus object (mirror of the reference's helpers/BaseReader.py:13-65).

CPU / pandas plumbing, outside the hot path; kept so model files and runners written for
ReChorus find the same attributes: data_df{train,dev,test}, all_df, n_users, n_items,
train_clicked_set, residual_clicked_set."""
import logging
import os

import numpy as np
import pandas as pd

from utils import utils


class BaseReader(object):
    @staticmethod
    def parse_data_args(parser):
        parser.add_argument('--path', type=str, default='data/', help='Input data dir.')
        parser.add_argument('--dataset', type=str, default='Grocery_and_Gourmet_Food', help='Choose a dataset.')
        parser.add_argument('--sep', type=str, default='\t', help='sep of csv file.')
        return parser

    def __init__(self, args):
        self.sep, self.prefix, self.dataset = args.sep, args.path, args.dataset
        self._read_data()
        self._build_clicked_sets()

    def _read_data(self):
        logging.info('Reading data from "{}", dataset = "{}" '.format(self.prefix, self.dataset))
        self.data_df = {}
        for phase in ('train', 'dev', 'test'):
            path = os.path.join(self.prefix, self.dataset, phase + '.csv')
            df = pd.read_csv(path, sep=self.sep).reset_index(drop=True).sort_values(by=['user_id', 'time'])
            self.data_df[phase] = utils.eval_list_columns(df)

        logging.info('Counting dataset statistics...')
        cols = ['user_id', 'item_id', 'time']
        if 'label' in self.data_df['train'].columns:  # CTR data carries labels
            cols.append('label')
        self.all_df = pd.concat([self.data_df[p][cols] for p in ('train', 'dev', 'test')])
        # ids start at 1; row 0 of every table is the padding row (still a trainable row)
        self.n_users = int(self.all_df['user_id'].max()) + 1
        self.n_items = int(self.all_df['item_id'].max()) + 1
        for phase in ('dev', 'test'):
            if 'neg_items' in self.data_df[phase]:
                negs = np.array(self.data_df[phase]['neg_items'].tolist())
                assert (negs >= self.n_items).sum() == 0, 'negative items must be known items'
        logging.info('"# user": {}, "# item": {}, "# entry": {}'.format(
            self.n_users - 1, self.n_items - 1, len(self.all_df)))
        if 'label' in cols:
            pos = int((self.all_df.label == 1).sum())
            logging.info('"# positive interaction": {} ({:.1f}%)'.format(pos, 100.0 * pos / len(self.all_df)))

    def _build_clicked_sets(self):
        """per user: items clicked in train, and in dev/test ("residual")"""
        self.train_clicked_set, self.residual_clicked_set = {}, {}
        for phase in ('train', 'dev', 'test'):
            target = self.train_clicked_set if phase == 'train' else self.residual_clicked_set
            df = self.data_df[phase]
            for uid, items in df.groupby('user_id')['item_id']:
                self.train_clicked_set.setdefault(uid, set())
                self.residual_clicked_set.setdefault(uid, set())
                target[uid].update(items.tolist())
