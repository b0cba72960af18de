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

    PHASES = ('train', 'dev', 'test')

    def _load_phase(self, phase):
        """<path>/<dataset>/<phase>.csv -> DataFrame ordered by (user, time), list-valued columns parsed"""
        frame = pd.read_csv(os.path.join(self.prefix, self.dataset, phase + '.csv'), sep=self.sep)
        frame = frame.reset_index(drop=True).sort_values(by=['user_id', 'time'])
        return utils.eval_list_columns(frame)

    def _read_data(self):
        """Attributes the plugin surface promises (reference helpers/BaseReader.py:24-65): data_df[phase], all_df
        (the interaction columns of all phases), n_users / n_items = max id + 1 (ids start at 1; row 0 of every table
        is the padding row, still a trainable row)."""
        logging.info('Reading data from "{}", dataset = "{}" '.format(self.prefix, self.dataset))
        self.data_df = {phase: self._load_phase(phase) for phase in self.PHASES}
        logging.info('Counting dataset statistics...')
        labelled = 'label' in self.data_df['train'].columns   # CTR data
        wanted = ['user_id', 'item_id', 'time'] + (['label'] if labelled else [])
        self.all_df = pd.concat([self.data_df[phase][wanted] for phase in self.PHASES])
        self.n_users = int(self.all_df['user_id'].max()) + 1
        self.n_items = int(self.all_df['item_id'].max()) + 1
        self._check_stored_negatives()
        logging.info('"# user": {}, "# item": {}, "# entry": {}'.format(self.n_users - 1, self.n_items - 1, len(self.all_df)))
        if labelled:
            n_pos = int((self.all_df.label == 1).sum())
            logging.info('"# positive interaction": {} ({:.1f}%)'.format(n_pos, 100.0 * n_pos / len(self.all_df)))

    def _check_stored_negatives(self):
        """dev / test rows carry their evaluation negatives: every one of them has to be a known item"""
        for phase in ('dev', 'test'):
            frame = self.data_df[phase]
            if 'neg_items' not in frame:
                continue
            worst = max((max(row) for row in frame['neg_items'] if len(row)), default=0)
            assert worst < self.n_items, 'negative items must be known items ({} >= {} in {})'.format(worst, self.n_items, phase)

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
