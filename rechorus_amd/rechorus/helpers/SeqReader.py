"""SeqReader: BaseReader + per-user interaction history (mirror of helpers/SeqReader.py:9-33).
Adds `user_his[uid] = [(item, time), ...]` in global time order and a `position` column (index
of each interaction inside its user's history) to every data_df."""
import logging

import pandas as pd

from helpers.BaseReader import BaseReader


class SeqReader(BaseReader):
    def __init__(self, args):
        super().__init__(args)
        self._append_his_info()

    def _append_his_info(self):
        logging.info('Appending history info...')
        ordered = self.all_df.sort_values(by=['time', 'user_id'], kind='mergesort')
        ordered['position'] = ordered.groupby('user_id').cumcount()
        self.user_his = {}
        for uid, grp in ordered.groupby('user_id', sort=False):
            self.user_his[uid] = list(zip(grp['item_id'].tolist(), grp['time'].tolist()))
        for phase in ('train', 'dev', 'test'):
            self.data_df[phase] = pd.merge(left=self.data_df[phase], right=ordered, how='left',
                                           on=['user_id', 'item_id', 'time'])
