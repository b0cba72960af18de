"""ImpressionSeqReader: ImpressionReader + per-user histories of clicked AND skipped items (mirror of the
reference's helpers/ImpressionSeqReader.py:12-55).  `user_his[uid] = {'pos': [(item, time), ...],
'neg': [...]}` in impression order; every impression row gets `position` / `neg_position` = the lengths
of the two histories BEFORE it.  CPU / pandas plumbing."""
import logging

import pandas as pd

from helpers.ImpressionReader import ImpressionReader


class ImpressionSeqReader(ImpressionReader):
    def __init__(self, args):
        super().__init__(args)
        self._append_his_info()

    def _append_his_info(self):
        logging.info('Appending history info with corresponding impressions...')
        by_time = self.impression_idkey == 'time'
        cols = ['user_id', 'pos_items', 'neg_items', 'time'] + ([] if by_time else [self.impression_idkey])
        order = ['user_id', 'time'] if by_time else ['user_id', self.impression_idkey, 'time']
        frames = [self.data_df[p][cols] for p in ('train', 'dev', 'test')]
        ordered = pd.concat(frames).sort_values(by=order, kind='mergesort')
        self.user_his = {}
        position, neg_position = [], []
        for uid, pos, neg, t in zip(ordered['user_id'], ordered['pos_items'], ordered['neg_items'], ordered['time']):
            his = self.user_his.setdefault(uid, {'pos': [], 'neg': []})
            position.append(len(his['pos']))
            neg_position.append(len(his['neg']))
            his['pos'].extend((i, t) for i in pos)
            his['neg'].extend((i, t) for i in neg)
        ordered = ordered.drop(columns=['pos_items', 'neg_items'])
        ordered['position'], ordered['neg_position'] = position, neg_position
        for phase in ('train', 'dev', 'test'):
            self.data_df[phase] = pd.merge(left=self.data_df[phase], right=ordered, how='left',
                                           on=['user_id', self.impression_idkey])
