"""Device-resident datasets: the reference's Dataset -> DataLoader -> collate_batch -> batch_to_gpu chain
(models/BaseModel.py:98-152,192-245; helpers/BaseRunner.py:182-186,230-233) for the standard
GeneralModel / SequentialModel datasets, rebuilt so that a batch never exists on the host.

The reference assembles every training instance in Python (`_get_feed_dict`), samples negatives in a
double Python loop once per epoch (`actions_before_epoch`) and pads / stacks in `collate_batch`; that
is 30-50 % of its CPU epoch at K = 99 and three orders of magnitude slower than the
HIP training step.  Here the interaction columns, the users' train-clicked sets (CSR) and their
time-ordered histories (CSR) are uploaded once; per epoch one kernel samples all negatives
(rc_sample_negatives) and per batch one or two kernels emit the feed dict on the device
(rc_assemble_candidates, rc_gather_history).

Only datasets whose feed dict is exactly one of the reference's standard ones are eligible (`dataset_kind()`:
General, Sequential, Context top-k, Context CTR -- the latter two add the user / item / situation feature
columns, gathered on the device from dense per-id tables); a model file that overrides `_get_feed_dict` /
`collate_batch` keeps the DataLoader path.
"""
import numpy as np
import torch

from . import engine


def _csr(sets_by_id, n_rows, device, sort=True):
    """{id: iterable of ints} -> (ptr [n_rows+1], flat values) on the device"""
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    chunks = []
    for r in range(n_rows):
        vals = sets_by_id.get(r)
        if vals:
            arr = np.fromiter(vals, dtype=np.int64, count=len(vals))
            if sort:
                arr.sort()
            chunks.append(arr)
            ptr[r + 1] = len(arr)
    np.cumsum(ptr, out=ptr)
    flat = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
    return torch.from_numpy(ptr).to(device), torch.from_numpy(flat).to(device)


def _corpus_cache(corpus, device):
    cache = getattr(corpus, '_hip_device_cache', None)
    if cache is None or cache.get('device') != device:
        cache = {'device': device}
        # not pickled with the corpus (main.py saves the corpus before any model exists, but be safe)
        object.__setattr__(corpus, '_hip_device_cache', cache)
    return cache


def clicked_csr(corpus, device, which='train'):
    """train clicked sets (negative sampling) or train + residual (the --test_all mask)"""
    cache = _corpus_cache(corpus, device)
    key = 'clicked_' + which
    if key not in cache:
        if which == 'train':
            sets = corpus.train_clicked_set
        else:
            sets = {u: set(s) | set(corpus.residual_clicked_set.get(u, ())) for u, s in corpus.train_clicked_set.items()}
        cache[key] = _csr(sets, corpus.n_users, device)
    return cache[key]


def _padded(lists, width):
    out = np.zeros((len(lists), width), dtype=np.int64)
    for r, items in enumerate(lists):
        items = items[:width]
        out[r, :len(items)] = items
    return out


def history_csr(corpus, device, which=None):
    """time-ordered (item, time) histories as CSR; `which` = 'pos' / 'neg' for the impression readers, whose
    user_his holds clicked and skipped items separately (helpers/ImpressionSeqReader.py)"""
    if which is not None:
        cache = _corpus_cache(corpus, device)
        key = 'history_' + which
        if key not in cache:
            class _View:  # same shape as SeqReader.user_his
                pass
            view = _View()
            view.n_users = corpus.n_users
            view.user_his = {u: h[which] for u, h in corpus.user_his.items()}
            cache[key] = history_csr(view, device)
        return cache[key]
    cache = _corpus_cache(corpus, device)
    if 'history' not in cache:
        n = corpus.n_users
        ptr = np.zeros(n + 1, dtype=np.int64)
        for u, seq in corpus.user_his.items():
            ptr[u + 1] = len(seq)
        np.cumsum(ptr, out=ptr)
        items = np.zeros(int(ptr[-1]), dtype=np.int64)
        times = np.zeros(int(ptr[-1]), dtype=np.int64)
        for u, seq in corpus.user_his.items():
            if seq:
                a = np.asarray(seq, dtype=np.int64)
                items[ptr[u]:ptr[u + 1]] = a[:, 0]
                times[ptr[u]:ptr[u + 1]] = a[:, 1]
        cache['history'] = tuple(torch.from_numpy(x).to(device) for x in (ptr, items, times))
    return cache['history']


def dataset_kind(dataset):
    """'general' | 'sequential' | 'context' | 'ctr' when the dataset produces exactly one of the reference's
    standard feed dicts (so it can be assembled on the device), else None (DataLoader path)"""
    from models.BaseModel import BaseModel, GeneralModel, SequentialModel  # plugin surface (on sys.path)
    from models.BaseModel import CTRModel
    from models.BaseContextModel import ContextCTRModel, ContextModel
    cls = type(dataset)
    if cls.__getitem__ is not BaseModel.Dataset.__getitem__:
        return None
    feed = cls._get_feed_dict
    if cls.collate_batch is not BaseModel.Dataset.collate_batch:
        return _impression_kind(cls, feed)
    sampled = cls.actions_before_epoch is GeneralModel.Dataset.actions_before_epoch
    if feed is GeneralModel.Dataset._get_feed_dict and sampled:
        return 'general'
    if feed is SequentialModel.Dataset._get_feed_dict and sampled:
        return 'sequential'
    if feed is ContextModel.Dataset._get_feed_dict and sampled:
        return 'context'
    unsampled = cls.actions_before_epoch in (BaseModel.Dataset.actions_before_epoch, CTRModel.Dataset.actions_before_epoch)
    if feed in (ContextCTRModel.Dataset._get_feed_dict, CTRModel.Dataset._get_feed_dict) and unsampled:
        return 'ctr'
    return None


def _impression_kind(cls, feed):
    from models.BaseImpressionModel import ImpressionModel, ImpressionSeqModel
    listed = cls.actions_before_epoch in (ImpressionModel.Dataset.actions_before_epoch,
                                          ImpressionSeqModel.Dataset.actions_before_epoch)
    if listed and feed is ImpressionModel.Dataset._get_feed_dict and cls.collate_batch is ImpressionModel.Dataset.collate_batch:
        return 'impression'
    if listed and feed is ImpressionSeqModel.Dataset._get_feed_dict and cls.collate_batch is ImpressionSeqModel.Dataset.collate_batch:
        return 'impression_seq'
    return None


def eligible(dataset):
    return dataset_kind(dataset) is not None


def _feature_table(by_id, name, n_rows, device):
    """{id: {feature: value}} -> dense column [n_rows] on the device (ids without metadata read 0)"""
    vals = [v[name] for v in by_id.values()]
    col = np.zeros(n_rows, dtype=np.float64 if any(isinstance(x, float) for x in vals) else np.int64)
    for i, v in by_id.items():
        col[i] = v[name]
    return torch.from_numpy(col).to(device)


def context_tables(corpus, device):
    cache = _corpus_cache(corpus, device)
    if 'context' not in cache:
        user = {f: _feature_table(corpus.user_features, f, corpus.n_users, device) for f in corpus.user_feature_names}
        item = {f: _feature_table(corpus.item_features, f, corpus.n_items, device) for f in corpus.item_feature_names}
        cache['context'] = (user, item)
    return cache['context']


class DeviceDataset:
    """The columns of one phase of a General / Sequential dataset, resident on the device."""

    def __init__(self, dataset, device):
        from models.BaseModel import SequentialModel
        self.dataset, self.device = dataset, device
        model, corpus = dataset.model, dataset.corpus
        self.phase, self.train = dataset.phase, dataset.phase == 'train'
        self.n_items, self.num_neg = corpus.n_items, model.num_neg

        def col(name):
            return torch.from_numpy(np.asarray(dataset.data[name], dtype=np.int64)).to(device)

        def ids(name, n_rows):
            """an id column, range-checked once on the host: the kernels index tables with these ids unchecked (nn.Embedding
            would raise a device assert), e.g. after a stale corpus pickle that does not match the model"""
            a = np.asarray(dataset.data[name], dtype=np.int64)
            if a.size and (a.min() < 0 or a.max() >= n_rows):
                raise ValueError('{} of the {} set outside [0, {}): min {}, max {} (does the corpus match the dataset? '
                                 'try --regenerate 1)'.format(name, dataset.phase, n_rows, int(a.min()), int(a.max())))
            return torch.from_numpy(a).to(device)

        self.users, self.items = ids('user_id', corpus.n_users), ids('item_id', corpus.n_items)
        self.kind = dataset_kind(dataset)
        self.sequential = self.kind == 'sequential'
        self.labels = col('label') if self.kind == 'ctr' else None
        self.user_feat = self.item_feat = self.situation = None
        from models.BaseModel import CTRModel
        with_features = self.kind == 'context' or (self.kind == 'ctr' and type(dataset)._get_feed_dict
                                                    is not CTRModel.Dataset._get_feed_dict)
        if with_features:  # models/BaseContextModel.py:16-30: features next to the ids
            self.user_feat, self.item_feat = context_tables(corpus, device)
            self.situation = {f: torch.from_numpy(np.asarray(dataset.data[f])).to(device) for f in corpus.situation_feature_names}
        if self.sequential:
            self.position = col('position')
            self.his_ptr, self.his_items, self.his_times = history_csr(corpus, device)
            self.max_his = model.history_max if model.history_max > 0 else max(1, int(self.position.max()))
        self.impression = self.kind in ('impression', 'impression_seq')
        if self.impression:  # fixed-width positive / negative lists, right-padded with item 0 (collate_batch :190-201)
            self.lists = torch.from_numpy(np.concatenate(
                [_padded(dataset.data['pos_items'], dataset.pos_len), _padded(dataset.data['neg_items'], dataset.neg_len)],
                axis=1)).to(device)
            self.pos_num = torch.clamp(col('pos_num'), max=dataset.pos_len)
            self.neg_num = torch.clamp(col('neg_num'), max=dataset.neg_len)
        if self.kind == 'impression_seq':
            self.position, self.neg_position = col('position'), col('neg_position')
            self.his_pos, self.his_neg = history_csr(corpus, device, 'pos'), history_csr(corpus, device, 'neg')
            self.max_his = model.history_max if model.history_max > 0 else max(1, int(self.position.max()))
        self.neg = None
        self.test_all = bool(getattr(model, 'test_all', 0)) and not self.train
        if not self.train and not self.test_all and self.kind != 'ctr' and not self.impression:
            negs = np.asarray(dataset.data['neg_items'], dtype=np.int64)
            if negs.size and (negs.min() < 0 or negs.max() >= corpus.n_items):
                raise ValueError('neg_items of the {} set outside [0, {})'.format(dataset.phase, corpus.n_items))
            self.neg = torch.from_numpy(negs).to(device).contiguous()
        self._draws = 0

    def __len__(self):
        return self.users.numel()

    def sample_negatives(self, seed):
        """all negatives of one epoch, like actions_before_epoch (models/BaseModel.py:206-214)"""
        if self.kind == 'ctr' or self.impression:  # labelled data / impressions bring their own negatives
            return None
        ptr, flat = clicked_csr(self.dataset.corpus, self.device, 'train')
        self.neg = engine.sample_negatives(self.users, self.num_neg, self.n_items, ptr, flat, seed=seed,
                                           base_index=self._draws, out=self.neg)
        self._draws += self.neg.numel()  # successive epochs draw from disjoint counter ranges
        return self.neg

    def feed(self, idx):
        """the feed dict of rows `idx` (int64 device tensor), as collate_batch would build it"""
        B = idx.numel()
        if self.impression:
            return self._impression_feed(idx)
        if self.kind == 'ctr':  # one labelled (user, item) pair per row (models/BaseModel.py:276-284)
            user_id, item_id = self.users[idx], self.items[idx, None]
        elif self.test_all:  # candidates = target + every item (models/BaseModel.py:194-195)
            user_id = self.users[idx]
            everything = torch.arange(1, self.n_items, device=self.device).expand(B, -1)
            item_id = torch.cat([self.items[idx, None], everything], dim=1)
        else:
            user_id, item_id = engine.assemble_candidates(idx, self.users, self.items, self.neg)
        feed = {'user_id': user_id, 'item_id': item_id}
        if self.sequential:
            hist, times, lengths = engine.gather_history(idx, self.users, self.position, self.his_ptr, self.his_items,
                                                         self.max_his, his_times=self.his_times)
            feed.update(history_items=hist, history_times=times, lengths=lengths)
        if self.labels is not None:
            feed['label'] = self.labels[idx, None]
        if self.user_feat is not None:
            for f, col in self.user_feat.items():
                feed[f] = col[user_id]              # [B]
            for f, col in self.situation.items():
                feed[f] = col[idx]                  # [B]
            for f, col in self.item_feat.items():
                feed[f] = col[item_id]              # [B, C]
        feed['batch_size'] = B
        feed['phase'] = self.phase
        return feed

    def _impression_feed(self, idx):
        """models/BaseImpressionModel.py Dataset._get_feed_dict + collate_batch: positives then negatives"""
        feed = {'user_id': self.users[idx], 'item_id': self.lists[idx], 'pos_num': self.pos_num[idx],
                'neg_num': self.neg_num[idx]}
        if self.kind == 'impression_seq':
            for prefix, position, (ptr, items, times) in (('', self.position, self.his_pos),
                                                          ('neg_', self.neg_position, self.his_neg)):
                hist, tms, lengths = engine.gather_history(idx, self.users, position, ptr, items, self.max_his, his_times=times)
                feed[prefix + 'history_items'], feed[prefix + 'history_times'], feed[prefix + 'lengths'] = hist, tms, lengths
        feed['batch_size'] = idx.numel()
        feed['phase'] = self.phase
        return feed

    def feed_without_candidates(self, idx):
        """user ids (+ history) only: what a dot-product head needs to build its query vectors"""
        feed = {'user_id': self.users[idx]}
        if self.sequential:
            hist, times, lengths = engine.gather_history(idx, self.users, self.position, self.his_ptr, self.his_items,
                                                         self.max_his, his_times=self.his_times)
            feed.update(history_items=hist, history_times=times, lengths=lengths)
        feed['batch_size'] = idx.numel()
        feed['phase'] = self.phase
        return feed

    def batches(self, batch_size, shuffle):
        """generator over one pass (DataLoader(shuffle=..., drop_last=False) semantics)"""
        n = len(self)
        order = torch.randperm(n, device=self.device) if shuffle else torch.arange(n, device=self.device)
        for s in range(0, n, batch_size):
            yield self.feed(order[s:s + batch_size].contiguous())


def device_dataset(dataset, device):
    """cached DeviceDataset of a plugin Dataset (one per phase)"""
    dd = getattr(dataset, '_hip_device_dataset', None)
    if dd is None or dd.device != device:
        dd = DeviceDataset(dataset, device)
        dataset._hip_device_dataset = dd
    return dd
