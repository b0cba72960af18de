"""ContextReader: BaseReader + user / item / situation context features (mirror of the
reference's helpers/ContextReader.py:15-84: same flags, same corpus attributes).

CPU / pandas plumbing either side of the hot path.  Attributes model files read:
item_feature_names / user_feature_names / situation_feature_names (sorted column names with the
i_ / u_ / c_ prefixes), item_features / user_features (id -> {feature: value}) and feature_max
(feature -> vocabulary size = max value + 1, the row count of that feature's embedding table).
"""
import logging
import os

import pandas as pd

from helpers.BaseReader import BaseReader


class ContextReader(BaseReader):
    @staticmethod
    def parse_data_args(parser):
        parser.add_argument('--include_item_features', type=int, default=0,
                            help='Whether include item context features (0 or 1).')
        parser.add_argument('--include_user_features', type=int, default=0,
                            help='Whether include user context features (0 or 1).')
        parser.add_argument('--include_situation_features', type=int, default=0,
                            help='Whether include situation (i.e., dynamic context) features (0 or 1).')
        return BaseReader.parse_data_args(parser)

    def __init__(self, args):
        super().__init__(args)
        self.include_item_features = args.include_item_features
        self.include_user_features = args.include_user_features
        self.include_situation_features = args.include_situation_features
        self._load_ui_metadata()
        self._collect_context()

    def _read_meta(self, fname, wanted, prefix):
        path = os.path.join(self.prefix, self.dataset, fname)
        if not (wanted and os.path.exists(path)):
            return None, []
        df = pd.read_csv(path, sep=self.sep)
        return df, sorted(c for c in df.columns if c[:2] == prefix)

    def _load_ui_metadata(self):
        self.item_meta_df, self.item_feature_names = self._read_meta('item_meta.csv', self.include_item_features, 'i_')
        self.user_meta_df, self.user_feature_names = self._read_meta('user_meta.csv', self.include_user_features, 'u_')
        self.situation_feature_names = []
        if self.include_situation_features:
            self.situation_feature_names = sorted(c for c in self.data_df['train'].columns if c[:2] == 'c_')

    def _grow(self, name, column):
        self.feature_max[name] = max(self.feature_max.get(name, 0), int(column.max()) + 1)

    def _collect_context(self):
        logging.info('Collect context features...')
        self.item_features, self.user_features = None, None
        self.feature_max = dict()
        for phase in ('train', 'dev', 'test'):
            logging.info('Loading context for %s set...' % phase)
            df = self.data_df[phase]
            for f in ['user_id', 'item_id'] + self.situation_feature_names:
                self._grow(f, df[f])
        if self.item_meta_df is not None:
            item_df = self.item_meta_df[['item_id'] + self.item_feature_names]
            self.item_features = item_df.set_index('item_id').to_dict(orient='index')
            for f in self.item_feature_names:
                self._grow(f, item_df[f])
            logging.info('# Item Features: %d' % item_df.shape[1])
        if self.user_meta_df is not None:
            user_df = self.user_meta_df[['user_id'] + self.user_feature_names].set_index('user_id')
            self.user_features = user_df.to_dict(orient='index')
            for f in self.user_feature_names:
                self._grow(f, user_df[f])
            logging.info('# User Features: %d' % user_df.shape[1])
