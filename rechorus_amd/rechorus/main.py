"""Command-line driver with the reference's flags and log lines (mirror of src/main.py):

    cd rechorus_amd/rechorus
    python main.py --model_name BPRMF --emb_size 64 --lr 1e-3 --l2 1e-6 --dataset Grocery_and_Gourmet_Food

Classes are resolved by NAME exactly like the reference does (`model_name` -> models/*/<name>.py,
the model's `reader` / `runner` strings -> helpers/<name>.py), so a model file written for
ReChorus can be dropped into models/general or models/sequential.  `run(argv)` is the same
program as a function (used by the tests).  Host-side control plane only; every FLOP of the
chosen model runs in librechorus_hip.so on the GPU.
"""
import argparse
import importlib
import importlib.util
import logging
import os
import pickle
import sys

import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from utils import utils  # noqa: E402

MODEL_PACKAGES = ('models.general', 'models.sequential', 'models.context')


def parse_global_args(parser):
    parser.add_argument('--gpu', type=str, default='0', help='Set CUDA_VISIBLE_DEVICES, default for CPU only')
    parser.add_argument('--verbose', type=int, default=logging.INFO, help='Logging Level, 0, 10, ..., 50')
    parser.add_argument('--log_file', type=str, default='', help='Logging file path')
    parser.add_argument('--random_seed', type=int, default=0, help='Random seed of numpy and pytorch')
    parser.add_argument('--load', type=int, default=0, help='Whether load model and continue to train')
    parser.add_argument('--train', type=int, default=1, help='To train the model or not.')
    parser.add_argument('--save_final_results', type=int, default=1, help='To save the final validation and test results or not.')
    parser.add_argument('--regenerate', type=int, default=0, help='Whether to regenerate intermediate files')
    return parser


def find_class(kind, name):
    """kind 'model' -> models/<pkg>/<name>.py:<name><mode>; kind 'helper' -> helpers/<name>.py"""
    if kind == 'helper':
        return getattr(importlib.import_module('helpers.' + name), name)
    base, mode = name
    # user model files outside the package: directories listed in RECHORUS_MODEL_DIRS (os.pathsep separated), each
    # holding <name>.py written against the reference's plugin surface (`from models.BaseModel import ...`)
    for d in filter(None, os.environ.get('RECHORUS_MODEL_DIRS', '').split(os.pathsep)):
        path = os.path.join(d, base + '.py')
        if os.path.exists(path):
            spec = importlib.util.spec_from_file_location('rechorus_user_models.' + base, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return getattr(mod, base + mode)
    for pkg in MODEL_PACKAGES:
        try:
            mod = importlib.import_module('{}.{}'.format(pkg, base))
        except ModuleNotFoundError:
            continue
        return getattr(mod, base + mode)
    raise ValueError('unknown model {!r} (looked in {})'.format(base + mode, MODEL_PACKAGES))


def save_rec_results(args, init_args, dataset, runner, topk):
    """per-instance predictions of the dev / test set as csv (reference main.py:98-153): top-k lists for
    ranking models, pCTR + label for CTR models, the scored positive / negative lists for impression models"""
    name = init_args.model_name + init_args.model_mode
    path = os.path.join(runner.log_path, runner.save_appendix, 'rec-{}-{}.csv'.format(name, dataset.phase))
    utils.check_dir(path)
    mode = init_args.model_mode
    if mode == 'CTR':
        logging.info('Saving CTR prediction results to: {}'.format(path))
        predictions, labels = runner.predict(dataset)
        df = pd.DataFrame({'user_id': dataset.data['user_id'], 'item_id': dataset.data['item_id'],
                           'pCTR': predictions, 'label': labels})
    elif mode in ('TopK', ''):
        logging.info('Saving top-{} recommendation results to: {}'.format(topk, path))
        predictions = runner.predict(dataset)
        rows = []
        for i in range(len(dataset)):
            info = dataset[i]
            order = (-predictions[i]).argsort(kind='stable')[:topk]
            rows.append((info['user_id'], [info['item_id'][j] for j in order], [predictions[i][j] for j in order]))
        df = pd.DataFrame(rows, columns=['user_id', 'rec_items', 'rec_predictions'])
    elif mode in ('Impression', 'General', 'Sequential'):
        logging.info('Saving all recommendation results to: {}'.format(path))
        predictions = runner.predict(dataset)
        rows = []
        for i in range(len(dataset)):
            info = dataset[i]
            # (the reference writes predictions[i][:neg_len] into neg_predictions, i.e. the POSITIVE slots
            # again, main.py:143; the negatives' own scores start at pos_len)
            rows.append((info['user_id'], list(info['pos_items']), list(predictions[i][:dataset.pos_len]),
                         list(info['neg_items']), list(predictions[i][dataset.pos_len:dataset.pos_len + dataset.neg_len])))
        df = pd.DataFrame(rows, columns=['user_id', 'pos_items', 'pos_predictions', 'neg_items', 'neg_predictions'])
    else:
        return 0
    df.to_csv(path, sep=args.sep, index=False)
    logging.info('{} Prediction results saved!'.format(dataset.phase))


def run(argv=None):
    init_parser = argparse.ArgumentParser(description='Model')
    init_parser.add_argument('--model_name', type=str, default='BPRMF', help='Choose a model to run.')
    init_parser.add_argument('--model_mode', type=str, default='', help='Model mode (class-name suffix).')
    init_args, _ = init_parser.parse_known_args(argv)

    model_cls = find_class('model', (init_args.model_name, init_args.model_mode))
    reader_cls = find_class('helper', model_cls.reader)   # the model names its reader ...
    runner_cls = find_class('helper', model_cls.runner)   # ... and its runner

    parser = parse_global_args(argparse.ArgumentParser(description=''))
    parser = reader_cls.parse_data_args(parser)
    parser = runner_cls.parse_runner_args(parser)
    parser = model_cls.parse_model_args(parser)
    args, _ = parser.parse_known_args(argv)
    # cached corpus / log / model files of the context readers depend on which feature groups are loaded
    # (reference: main.py:176-180)
    args.data_appendix = ''
    if 'Context' in model_cls.reader:
        args.data_appendix = '_context%d%d%d' % (args.include_item_features, args.include_user_features,
                                                 args.include_situation_features)

    tag = init_args.model_name + init_args.model_mode
    log_args = [tag, args.dataset + args.data_appendix, str(args.random_seed)]
    log_args += ['{}={}'.format(a, getattr(args, a)) for a in ['lr', 'l2'] + model_cls.extra_log_args]
    log_name = '__'.join(log_args).replace(' ', '__')
    if args.log_file == '':
        args.log_file = '../log/{}/{}.txt'.format(tag, log_name)
    if args.model_path == '':
        args.model_path = '../model/{}/{}.pt'.format(tag, log_name)
    utils.check_dir(args.log_file)
    for h in list(logging.getLogger().handlers):
        logging.getLogger().removeHandler(h)
    logging.basicConfig(filename=args.log_file, level=args.verbose)
    logging.getLogger().addHandler(logging.StreamHandler(sys.stdout))
    logging.info(init_args)

    logging.info('-' * 45 + ' BEGIN: ' + utils.get_time() + ' ' + '-' * 45)
    exclude = ['check_epoch', 'log_file', 'model_path', 'path', 'pin_memory', 'load', 'regenerate', 'sep',
               'train', 'verbose', 'metric', 'test_epoch', 'buffer']
    logging.info(utils.format_arg_str(args, exclude_lst=exclude))

    utils.init_seed(args.random_seed)
    os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    if args.gpu == '' or not torch.cuda.is_available():
        raise SystemExit('rechorus_amd runs on an MI355X only: no GPU visible (--gpu {!r})'.format(args.gpu))
    args.device = torch.device('cuda')
    logging.info('Device: {}'.format(args.device))

    corpus_path = os.path.join(args.path, args.dataset, model_cls.reader + args.data_appendix + '.pkl')
    if not args.regenerate and os.path.exists(corpus_path):
        logging.info('Load corpus from {}'.format(corpus_path))
        corpus = pickle.load(open(corpus_path, 'rb'))
    else:
        corpus = reader_cls(args)
        logging.info('Save corpus to {}'.format(corpus_path))
        pickle.dump(corpus, open(corpus_path, 'wb'))

    model = model_cls(args, corpus).to(args.device)
    from rechorus_amd import nn as hnn
    # plain nn.Embedding tables of a model file written for the reference move onto the HIP engine -- on a GPU only: with
    # --gpu '' an unmodified model file keeps running on torch, as it does in the reference
    adopted = hnn.adopt_embeddings(model) if torch.device(args.device).type == 'cuda' else 0
    if adopted:
        logging.info('Adopted {} nn.Embedding table(s) onto the HIP engine'.format(adopted))
    if torch.device(args.device).type == 'cuda' and (adopted or type(model).__module__.startswith('rechorus_user_models')):
        # a model file whose head is one of the hot path's (the reference's own BPRMF.py / NeuMF.py / SASRec.py, unmodified; its
        # FM.py / WideDeep.py / DeepFM.py) gets the fused forward -- and, for the first three, the one-call fit() iteration and the
        # --test_all scorer (rechorus_amd/dropin.py)
        from rechorus_amd import dropin
        dropin.bind_known_head(model, log=logging.getLogger())
    logging.info('#params: {}'.format(model.count_variables()))
    logging.info(model)

    data_dict = dict()
    for phase in ('train', 'dev', 'test'):
        data_dict[phase] = model_cls.Dataset(model, corpus, phase)
        data_dict[phase].prepare()

    runner = runner_cls(args)
    logging.info('Test Before Training: ' + runner.print_res(data_dict['test']))
    if args.load > 0:
        model.load_model()
    if args.train > 0:
        runner.train(data_dict)

    results = {}
    for phase in ('dev', 'test'):
        results[phase] = runner.print_res(data_dict[phase])
        logging.info(os.linesep + '{} After Training: '.format('Dev ' if phase == 'dev' else 'Test') + results[phase])
    if args.save_final_results == 1:
        for phase in ('dev', 'test'):
            save_rec_results(args, init_args, data_dict[phase], runner, 100)
    model.actions_after_train()
    logging.info(os.linesep + '-' * 45 + ' END: ' + utils.get_time() + ' ' + '-' * 45)
    return results


if __name__ == '__main__':
    run()
