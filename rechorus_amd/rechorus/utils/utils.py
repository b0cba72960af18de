"""Helper functions of the plugin surface (mirror of the reference's utils/utils.py API:
same names and behaviour, re-written).  Pure host-side plumbing."""
import datetime
import logging
import os
import random

import numpy as np
import pandas as pd
import torch


def init_seed(seed):
    """reference utils/utils.py:13-19"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def df_to_dict(df: pd.DataFrame) -> dict:
    return {k: np.array(v) for k, v in df.to_dict("list").items()}


def batch_to_gpu(batch: dict, device) -> dict:
    """reference utils/utils.py:30-34; non_blocking so pinned batches overlap with compute"""
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            batch[k] = v.to(device, non_blocking=True)
    return batch


def check(check_list) -> None:
    logging.info("")
    for name, tensor in check_list:
        arr = np.array(tensor.detach().cpu())
        logging.info(os.linesep.join([name + "\t" + str(arr.shape), np.array2string(arr, threshold=20)]) + os.linesep)


def eval_list_columns(df: pd.DataFrame) -> pd.DataFrame:
    """list-valued csv columns arrive as strings (data/README.md of the reference)"""
    for col in df.columns:
        if pd.api.types.is_string_dtype(df[col]):
            df[col] = df[col].apply(lambda x: eval(str(x)))
    return df


def format_metric(result_dict) -> str:
    """'HR@5:0.1234,NDCG@5:0.0567' -- exp.py of the reference scrapes this format"""
    assert isinstance(result_dict, dict)
    metrics = sorted({k.split("@")[0] for k in result_dict})
    topks = sorted({int(k.split("@")[1]) for k in result_dict if "@" in k})
    parts = []
    for topk in (topks or ["All"]):
        for metric in metrics:
            name = metric if topk == "All" else "{}@{}".format(metric, topk)
            val = result_dict[name]
            if isinstance(val, (float, np.floating)):
                parts.append("{}:{:<.4f}".format(name, val))
            elif isinstance(val, (int, np.integer)):
                parts.append("{}:{}".format(name, val))
    return ",".join(parts)


def format_arg_str(args, exclude_lst, max_len=20) -> str:
    items = {k: v for k, v in vars(args).items() if k not in exclude_lst}
    kw = max(len("Arguments"), max(len(str(k)) for k in items))
    vw = max(len("Values"), min(max_len, max(len(str(v)) for v in items.values())))
    bar = "=" * (kw + vw + 5)
    lines = ["", bar, " " + "Arguments".ljust(kw) + " | " + "Values".ljust(vw) + " ", bar]
    for k in sorted(items):
        if items[k] is None:
            continue
        v = str(items[k]).replace("\t", "\\t")
        if len(v) > max_len:
            v = v[:max_len - 3] + "..."
        lines.append(" " + str(k).ljust(kw) + " | " + v.ljust(vw))
    lines.append(bar)
    return os.linesep.join(lines)


def check_dir(file_name: str):
    d = os.path.dirname(file_name)
    if d and not os.path.exists(d):
        print("make dirs:", d)
        os.makedirs(d)


def non_increasing(lst: list) -> bool:
    return all(lst[0] >= y for y in lst[1:])


def get_time():
    return datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
