"""Known-head recognition: an UNMODIFIED reference model file reaches the fused engine steps.

`hnn.adopt_embeddings` moves the tables of any model file onto the engine; its head then still runs as torch ops and its
backward as autograd.  For the three heads of the hot path -- the reference's own `models/general/BPRMF.py:34-63`,
`models/general/NeuMF.py:56-76`, `models/sequential/SASRec.py:51-86` -- this module recognises the instance and puts the
plugin's head in front of the model file's in the method resolution order: the fused `forward` (one kernel per head with a
HIP backward), `hip_train_step` (the one-call fit() iteration BaseRunner.fit uses for row-wise updates:
rc_bprmf_train_step_ahead / rc_neumf_train_step / engine.SasrecTrainer), `full_catalogue_vectors` (--test_all on the MFMA
scorer) and `candidate_permutation_equivariant` (no host-side candidate shuffle).  SURVEY.md 8(b2): "fused engine modules
selected when a model class matches a known head".

The CTR path's model files are recognised the same way: the reference's own `models/context/FM.py:33-78`, `WideDeep.py:29-60`,
`DeepFM.py:18-41` (their CTR and top-k classes, by class name -- WideDeep and DeepFM own the same parameters -- plus exactly the
parameter set `_define_params_FM` / `_define_params_WD` create, the inherited context loss, syntax tree and probe) get the plugin's
head methods of the class of the same name: every field of a table family in one gather launch, the fused FM term, the
one-kernel CTR head under `--loss_n BCE`, and through them HipOptimizer's rows mode at small batches.

A model is bound only when ALL of these hold, otherwise it keeps the adopted-tables route unchanged:
  * structure: the state_dict keys and parameter shapes are exactly the head's (`u_embeddings / i_embeddings`;
    `mf_* / mlp_* / mlp.k / prediction`; `i_embeddings / p_embeddings / transformer_block.*`), the hyper-parameter attributes
    the head reads exist (`emb_size`, `layers`, `num_heads`, ...), and the loss is the inherited `GeneralModel.loss` (BPR);
  * behaviour: on a probe batch, in eval mode, the model file's own forward and the fused forward agree to 1e-4 relative;
  * source: the syntax tree of the model file's forward is one of the known ones (the reference at the surveyed commit) --
    or, if the file was edited, dropout is 0, where the probe is a complete check of the function being replaced (a moved
    dropout layer is the one change the eval-mode probe cannot see).
"""
import ast
import hashlib
import importlib
import inspect
import logging
import os
import sys
import textwrap

import torch
import torch.nn as nn

PLUGIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rechorus")

# sha256 of ast.dump of the forward functions of the reference's model files (tests/golden/make_reference_model_copies.py prints them;
# comments, blank lines and indentation style do not enter)
KNOWN_FORWARD_HASHES = {
    "BPRMF": {"346a573d92bfe65d1176f7f1ebcbc5c9313108ce9e082bb4b34c5077823c43d6"},
    "NeuMF": {"74e9a03825ddedf9be6b9476f8ea1465e1c62a4a5ff6bd0dfb642ff76cc0fe4c"},
    "SASRec": {"0be9d019aa2162328ce8b550d2b3ccd98c9c3ef15303891e386ac9ffc91219b0"},
    # the CTR / top-k heads of models/context/{FM,WideDeep,DeepFM}.py (keyed by class name: WideDeep and DeepFM own the same parameters)
    "FMCTR": {"adf56d37116f128e90e9984a53191ac3e24581997ad7177b1a574632732cf2a3"},
    "FMTopK": {"f96e3b561c6c16f4ef5fd79e2680164f59e02222bcde71ce6acac227fd374434"},
    "WideDeepCTR": {"80156254c2d77f7c92f8b7fec0cafd2780929f9f214866cdf524f558d778d892"},
    "WideDeepTopK": {"41b8c7f594e5ac79146add432a6a7675e734f4447dbea0e61ec08860c2f410d6"},
    "DeepFMCTR": {"6893fc755c37fee261fa41c0bf58ea6821bdf670f0af3710add6d35f77f2c02a"},
    "DeepFMTopK": {"f5d19151bbf5e245ca2d368e69a6fd33f878fd71b40a51f6382afef93bdaef48"},
}
CONTEXT_HEADS = ("FMCTR", "FMTopK", "WideDeepCTR", "WideDeepTopK", "DeepFMCTR", "DeepFMTopK")

SAS_BLOCK_KEYS = ("masked_attn_head.q_linear.weight", "masked_attn_head.q_linear.bias", "masked_attn_head.k_linear.weight",
                  "masked_attn_head.k_linear.bias", "masked_attn_head.v_linear.weight", "masked_attn_head.v_linear.bias",
                  "layer_norm1.weight", "layer_norm1.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                  "layer_norm2.weight", "layer_norm2.bias")


def forward_hash(cls):
    """sha256 over the syntax trees of every `forward` defined along the class's own chain up to (excluding) the plugin's base
    classes and nn.Module -- for the reference files: <Head>Base.forward + <Head>.forward"""
    parts = []
    for k in cls.__mro__:
        mod = getattr(k, "__module__", "") or ""
        if k is object or mod.startswith("torch.") or mod.startswith("models.") or mod.startswith("rechorus_amd"):
            continue
        fn = k.__dict__.get("forward")
        if fn is None:
            continue
        try:
            tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
        except (OSError, TypeError, SyntaxError, IndentationError):
            return None
        parts.append(k.__name__ + ":" + ast.dump(tree, annotate_fields=False, include_attributes=False))
    if not parts:
        return None
    return hashlib.sha256("\n".join(parts).encode()).hexdigest()


def _mirror(module):
    if PLUGIN not in sys.path:
        sys.path.insert(0, PLUGIN)
    return importlib.import_module(module)


def _shapes(model):
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


def _is_int(x):
    return isinstance(x, int) and not isinstance(x, bool)


def _kind(model):
    """'BPRMF' | 'NeuMF' | 'SASRec' | None from state_dict keys, shapes and the attributes the plugin's head reads"""
    sh = _shapes(model)
    keys = set(sh)
    d = getattr(model, "emb_size", None)
    if not _is_int(d) or not hasattr(model, "dropout"):
        return None
    if keys == {"u_embeddings.weight", "i_embeddings.weight"}:
        ok = sh["u_embeddings.weight"][1:] == (d,) and sh["i_embeddings.weight"][1:] == (d,)
        return "BPRMF" if ok else None
    tables = {"mf_u_embeddings.weight", "mf_i_embeddings.weight", "mlp_u_embeddings.weight", "mlp_i_embeddings.weight"}
    if tables <= keys and "prediction.weight" in keys:
        layers = getattr(model, "layers", None)
        if not (isinstance(layers, (list, tuple)) and len(layers) >= 1 and all(_is_int(x) for x in layers)):
            return None
        want = set(tables) | {"prediction.weight"}
        pre = 2 * d
        for k, size in enumerate(layers):
            want |= {"mlp.%d.weight" % k, "mlp.%d.bias" % k}
            if sh.get("mlp.%d.weight" % k) != (size, pre) or sh.get("mlp.%d.bias" % k) != (size,):
                return None
            pre = size
        ok = (keys == want and sh["prediction.weight"] == (1, pre + d) and all(sh[t][1:] == (d,) for t in tables)
              and sh["mf_u_embeddings.weight"][0] == sh["mlp_u_embeddings.weight"][0]
              and sh["mf_i_embeddings.weight"][0] == sh["mlp_i_embeddings.weight"][0]
              and isinstance(getattr(model, "mlp", None), nn.ModuleList) and all(type(m) is nn.Linear for m in model.mlp)
              and type(getattr(model, "prediction", None)) is nn.Linear and model.prediction.bias is None
              and type(getattr(model, "dropout_layer", None)) is nn.Dropout
              and float(model.dropout_layer.p) == float(model.dropout))
        return "NeuMF" if ok else None
    if {"i_embeddings.weight", "p_embeddings.weight"} <= keys:
        nl, nh, max_his = (getattr(model, a, None) for a in ("num_layers", "num_heads", "max_his"))
        if not (_is_int(nl) and _is_int(nh) and _is_int(max_his) and nl >= 1 and nh >= 1 and d % nh == 0):
            return None
        want = {"i_embeddings.weight", "p_embeddings.weight"} | {"transformer_block.%d.%s" % (l, k) for l in range(nl) for k in SAS_BLOCK_KEYS}
        blocks = getattr(model, "transformer_block", None)
        layers_mod = _mirror("utils.layers")
        ok = (keys == want and sh["i_embeddings.weight"][1:] == (d,) and sh["p_embeddings.weight"] == (max_his + 1, d)
              and all(sh["transformer_block.%d.%s" % (l, k)] == ((d, d) if k.endswith("linear.weight") or k in ("linear1.weight", "linear2.weight") else (d,))
                      for l in range(nl) for k in SAS_BLOCK_KEYS)
              and isinstance(blocks, nn.ModuleList) and all(type(b) is layers_mod.TransformerLayer for b in blocks)
              and all(getattr(b.masked_attn_head, "h", nh) == nh and not getattr(b.masked_attn_head, "kq_same", False) for b in blocks))
        return "SASRec" if ok else None
    return None


def _context_base(kind):
    return kind[:-3] if kind.endswith("CTR") else kind[:-4]


def _context_kind(model):
    """'FMCTR' | ... | 'DeepFMTopK' | None: a model file's FM / WideDeep / DeepFM class over the plugin's context task bases, with
    exactly the parameters the plugin's class of the same name owns (models/context/FM.py:33-41, WideDeep.py:34-39)"""
    kind = type(model).__name__
    if kind not in CONTEXT_HEADS:
        return None
    bc = _mirror("models.BaseContextModel")
    task = bc.ContextCTRModel if kind.endswith("CTR") else bc.ContextModel
    if not isinstance(model, task) or type(model).loss is not task.loss:
        return None
    feats, fmax, d = getattr(model, "context_features", None), getattr(model, "feature_max", None), getattr(model, "vec_size", None)
    if not (isinstance(feats, (list, tuple)) and len(feats) >= 1 and isinstance(fmax, dict) and _is_int(d) and hasattr(model, "dropout")):
        return None
    cat = lambda f: f.endswith("_c") or f.endswith("_id")
    sh = _shapes(model)
    want = {"overall_bias": (1,)}
    for f in feats:
        if cat(f) and not _is_int(fmax.get(f)):
            return None
        want["context_embedding.%s.weight" % f] = (fmax[f], d) if cat(f) else (d, 1)
        want["linear_embedding.%s.weight" % f] = (fmax[f], 1) if cat(f) else (1, 1)
    rest = {k: v for k, v in sh.items() if k not in want}
    if any(sh.get(k) != v for k, v in want.items()):
        return None
    if _context_base(kind) == "FM":
        return kind if not rest else None
    layers_mod = _mirror("utils.layers")
    layers = getattr(model, "layers", None)
    ok = (isinstance(layers, (list, tuple)) and len(layers) >= 1 and all(_is_int(x) for x in layers)
          and type(getattr(model, "deep_layers", None)) is layers_mod.MLP_Block and rest and all(k.startswith("deep_layers.") for k in rest))
    return kind if ok else None


def _context_mixin(kind):
    """the plugin's class of the same name as a mixin: every method of its head chain (FMBase .. DeepFMBase) and its task forward,
    without construction (the model file built its own parameters, with the same names)"""
    m = getattr(_mirror("models.context." + _context_base(kind)), kind)
    skip = ("parse_model_args", "parse_model_args_FM", "parse_model_args_WD", "_base_init", "_define_init", "_define_init_params",
            "_define_params_FM", "_define_params_WD", "reader", "runner", "extra_log_args")
    body = {}
    for k in reversed(m.__mro__):       # base-most first: the derived head's methods win
        if (getattr(k, "__module__", "") or "").startswith("models.context."):
            body.update({n: v for n, v in k.__dict__.items() if n not in skip and not (n.startswith("__") and n.endswith("__"))})
    body["forward"] = m.__dict__["forward"]
    return type(kind + "Head", (object,), body)


def _context_probe_feed(model, kind, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(20240917)
    B, C = 6, (1 if kind.endswith("CTR") else 4)
    feed = {}
    for f in list(model.context_features) + ["user_id", "item_id"]:
        if f in feed:
            continue
        shape = (B, C) if (f == "item_id" or f.startswith("i_")) else (B,)
        if f.endswith("_c") or f.endswith("_id"):
            feed[f] = torch.randint(0, max(int(model.feature_max.get(f, 2)), 1), shape, generator=g).to(device)
        else:
            feed[f] = torch.rand(shape, generator=g).to(device)
    if kind.endswith("CTR"):
        feed["label"] = torch.randint(0, 2, (B, C), generator=g).to(device)
    feed.update(batch_size=B, phase="test")
    return feed


def _head_mixin(kind):
    """the plugin's head of this kind as a mixin class: the methods of the mirror's model file that replace the model file's"""
    if kind in CONTEXT_HEADS:
        return _context_mixin(kind)
    if kind == "BPRMF":
        return _mirror("models.general.BPRMF").BPRMFBase
    if kind == "SASRec":
        return _mirror("models.sequential.SASRec").SASRecBase
    m = _mirror("models.general.NeuMF").NeuMF
    # everything the plugin's class defines except construction (the model file built its own parameters, with the same names)
    skip = ("__init__", "__module__", "__doc__", "__qualname__", "__dict__", "__weakref__", "parse_model_args", "_define_params",
            "reader", "runner", "extra_log_args")
    return type("NeuMFHead", (object,), {n: v for n, v in m.__dict__.items() if n not in skip})


def _probe_feed(model, kind, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(20240917)
    B, C = 6, 4
    sh = _shapes(model)
    if kind == "SASRec":
        n_items, L = sh["i_embeddings.weight"][0], max(1, min(model.max_his, 9))
        lengths = torch.randint(1, L + 1, (B,), generator=g)
        hist = torch.randint(1, max(n_items, 2), (B, L), generator=g) * (torch.arange(L)[None, :] < lengths[:, None])
        feed = {"history_items": hist.to(device), "lengths": lengths.to(device)}
        n_users = 2
    else:
        n_users = sh["u_embeddings.weight" if kind == "BPRMF" else "mf_u_embeddings.weight"][0]
        n_items = sh["i_embeddings.weight" if kind == "BPRMF" else "mf_i_embeddings.weight"][0]
        feed = {}
    feed.update(user_id=torch.randint(0, max(n_users, 1), (B,), generator=g).to(device),
                item_id=torch.randint(0, max(n_items, 1), (B, C), generator=g).to(device), batch_size=B, phase="test")
    return feed


last_miss_reason = None     # why the last bind_known_head call left a model file on its own route (None: it bound, or was not asked)


def _miss(log, model, reason):
    """a model file keeps its own (adopted-tables) route: say so once, with the reason, where main.py's log shows it"""
    global last_miss_reason
    last_miss_reason = reason
    log.info("Known-head recognition: %s keeps the model file's own head (torch layers over the engine's tables): %s",
             type(model).__name__, reason)
    return None


def _overridden_hooks(model, base_cls):
    """training hooks of the model file that the fused step would bypass: per-group optimizer settings (customize_parameters, e.g.
    the reference's Chorus.py:179-196) and per-epoch actions"""
    out = []
    for hook in ("customize_parameters", "actions_before_epoch", "actions_after_train"):
        theirs, ours = getattr(type(model), hook, None), getattr(base_cls, hook, None)
        if theirs is not None and ours is not None and getattr(theirs, "__func__", theirs) is not getattr(ours, "__func__", ours):
            out.append(hook)
    return out


def bind_known_head(model, log=logging.getLogger(__name__)):
    """Recognise a BPRMF / NeuMF / SASRec / FM-family head (see the module docstring) and bind the plugin's fused head to the instance.
    Returns the kind that was bound, or None (the model is left exactly as it was; for a model file the reason is logged and kept
    in `last_miss_reason`).  The model must be on the GPU with its tables adopted (hnn.adopt_embeddings)."""
    global last_miss_reason
    last_miss_reason = None
    if hasattr(model, "hip_train_step") or getattr(type(model), "_rc_bound_head", None) is not None:
        return None     # the plugin's own class (or a model file that brings its own fused step); bound already
    if (type(model).__module__ or "").startswith("models."):
        return None     # the plugin's own model files
    try:
        kind = _context_kind(model)
        base = _mirror("models.BaseModel")
        if kind is None:
            if not isinstance(model, base.GeneralModel):
                return _miss(log, model, "not a GeneralModel / ContextModel / ContextCTRModel head")
            if type(model).loss is not base.GeneralModel.loss:
                return _miss(log, model, "its loss is not GeneralModel's BPR loss (list-wise / custom losses have no fused step)")
            kind = _kind(model)
    except Exception as e:     # a model file this module does not understand keeps its route
        return _miss(log, model, "recognition raised %r" % (e,))
    if kind is None:
        return _miss(log, model, "parameters / attributes match none of BPRMF, NeuMF, SASRec, FM, WideDeep, DeepFM")
    if kind not in CONTEXT_HEADS:      # hip_train_step trains with one (lr, l2) pair and calls none of these hooks
        hooks = _overridden_hooks(model, base.BaseModel)
        if hooks:
            return _miss(log, model, "%s-shaped, but it overrides %s, which the fused step would bypass" % (kind, ", ".join(hooks)))
    p = next(model.parameters())
    if not p.is_cuda:
        return _miss(log, model, "%s-shaped, but the model is not on the GPU" % kind)
    h = forward_hash(type(model))
    known = h is not None and h in KNOWN_FORWARD_HASHES[kind]
    if not known and float(model.dropout) > 0:
        return _miss(log, model, "%s-shaped, but its forward is not one of the known syntax trees and dropout > 0 (a probe batch cannot "
                                 "verify a stochastic forward)" % kind)
    cls = type(model)
    was_training = model.training
    seed_added = False
    try:
        model.eval()
        feed = _context_probe_feed(model, kind, p.device) if kind in CONTEXT_HEADS else _probe_feed(model, kind, p.device)
        with torch.no_grad():
            ref = model(dict(feed))["prediction"].float().clone()
        if kind in ("NeuMF", "SASRec") and not hasattr(model, "drop_seed"):
            from . import nn as hnn
            # key of the heads' counter-based dropout masks; not a parameter and not in the state_dict (checkpoints keep the
            # reference's keys)
            model.register_buffer("drop_seed", hnn.fresh_drop_seed().to(p.device), persistent=False)
            seed_added = True
        if kind == "BPRMF" and not hasattr(model, "_trainer"):
            model._trainer = None      # (the plugin's _base_init creates the slot its hip_train_step fills)
        model.__class__ = type(cls.__name__, (_head_mixin(kind), cls), {"__module__": cls.__module__, "_rc_bound_head": kind,
                                                                         "_rc_model_file_class": cls})
        with torch.no_grad():
            got = model(dict(feed))["prediction"].float()
        torch.cuda.synchronize(p.device)
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max()) if got.shape == ref.shape else float("inf")
        if not (err <= 1e-4 * scale + 1e-7):
            raise RuntimeError("probe mismatch: max |fused - model file| = %.3e at scale %.3e" % (err, scale))
    except Exception as e:
        model.__class__ = cls
        if seed_added:
            del model._buffers["drop_seed"]
            model._non_persistent_buffers_set.discard("drop_seed")
        model.train(was_training)
        return _miss(log, model, "%s-shaped, but the fused head does not reproduce its forward on a probe batch (%s)" % (kind, e))
    model.train(was_training)
    what = "one-launch field gathers / fused FM term and CTR head" if kind in CONTEXT_HEADS else "fused forward / hip_train_step / --test_all scorer"
    log.info("Recognised the %s head%s: %s bound to %s", kind, "" if known else " (edited forward, verified on a probe batch)", what,
             cls.__name__)
    return kind
