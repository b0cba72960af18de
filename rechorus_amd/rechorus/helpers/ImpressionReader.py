"""ImpressionReader: impression (request-level) data, every impression carrying its own positive and
negative item lists (mirror of the reference's helpers/ImpressionReader.py:12-128: same flag, same
corpus attributes).  CPU / pandas plumbing either side of the hot path.

Rows of train/dev/test.csv are (user_id, item_id, time[, impression_id], label); rows sharing
(user_id, impression_idkey) form one impression.  Per impression: pos_items = the distinct items with
label 1, neg_items = those with label 0; impressions without a positive or without a negative are
dropped; the impression is represented by its LAST row (the reference attaches the lists there, :62-93).
Item order inside a list is ascending here (the reference iterates a Python set; the order only matters
when a list is longer than --train/test_max_*_item and gets truncated)."""
import logging
import os

import numpy as np
import pandas as pd

from helpers.BaseReader import BaseReader
from utils import utils


class ImpressionReader(BaseReader):
    @staticmethod
    def parse_data_args(parser):
        parser.add_argument('--impression_idkey', type=str, default='time',
                            help='The key for impression identification, [time, impression_id]')
        return BaseReader.parse_data_args(parser)

    def __init__(self, args):
        self.impression_idkey = args.impression_idkey
        super().__init__(args)
        self._append_impression_info()

    def _read_data(self):
        logging.info('Reading data from "{}", dataset = "{}" '.format(self.prefix, self.dataset))
        self.data_df = {}
        for phase in ('train', 'dev', 'test'):
            df = pd.read_csv(os.path.join(self.prefix, self.dataset, phase + '.csv'), sep=self.sep)
            df = df.reset_index(drop=True).sort_values(by=['user_id', self.impression_idkey])
            self.data_df[phase] = utils.eval_list_columns(df)
        logging.info('Counting dataset statistics...')
        cols = ['user_id', 'item_id', 'time'] + ([] if self.impression_idkey == 'time' else [self.impression_idkey])
        if 'label' not in self.data_df['train'].columns:
            raise KeyError('Impression data must have binary labels')
        cols.append('label')
        self.all_df = pd.concat([self.data_df[p][cols] for p in ('train', 'dev', 'test')])
        self.n_users = int(self.all_df['user_id'].max()) + 1
        self.n_items = int(self.all_df['item_id'].max()) + 1
        logging.info('Update impression data -- "# user": {}, "# item": {}, "# entry": {}'.format(
            self.n_users - 1, self.n_items - 1, len(self.all_df)))
        pos = int((self.all_df.label == 1).sum())
        logging.info('"# positive interaction": {} ({:.1f}%)'.format(pos, 100.0 * pos / len(self.all_df)))

    @staticmethod
    def _valid_len(items):
        return items.index(0) if 0 in items else len(items)

    def _append_impression_info(self):
        logging.info('Merging positive items by timestamp/impression_idkey...')
        key = ['user_id', self.impression_idkey]
        totals = [0, 0, 0]
        for phase in ('train', 'dev', 'test'):
            df = self.data_df[phase]
            groups = df.groupby(key, sort=False)
            last = groups.tail(1).copy()                      # the row that represents the impression
            lists = groups[['item_id', 'label']].apply(lambda g: pd.Series({
                'pos_items': sorted(set(g['item_id'][g['label'] != 0].tolist())),
                'neg_items': sorted(set(g['item_id'][g['label'] == 0].tolist()))}))
            last = last.merge(lists.reset_index(), on=key, how='left')
            last = last[last['pos_items'].map(len) > 0]       # impressions with only negatives are dropped
            last['neg_num'] = last['neg_items'].map(self._valid_len)
            last['pos_num'] = last['pos_items'].map(self._valid_len)
            last = last.loc[last.neg_num > 0].reset_index(drop=True)  # ... and those without negatives
            self.data_df[phase] = last
            totals[0] += int(last['neg_num'].sum())
            totals[1] += int(last['pos_num'].sum())
            totals[2] += len(last)
        logging.info('train, dev, test request num: ' + ' '.join(str(len(self.data_df[p])) for p in ('train', 'dev', 'test')))
        logging.info('Average positive items / impression = %.3f, negative items / impression = %.3f' % (
            totals[1] / max(totals[2], 1), totals[0] / max(totals[2], 1)))
