"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement (float64 accumulation) of one hidden layer of the reference's dense towers and of the gradients autograd
derives from it: utils/layers.py:201-243 (MLP_Block: nn.Linear -> nn.ReLU -> nn.Dropout) and models/general/NeuMF.py:69-72
(`layer(h).relu()` then `dropout_layer`).  Training-mode dropout enters as an explicit keep-and-scale mask
(dropout_keep: the counter scheme of rc_linear_fwd, include/rechorus_hip.h); torch's own random stream cannot be
reproduced by another implementation.  Pinned through the model-level goldens generated FROM the reference
(tests/golden/neumfml_*.npz, deepfm_*.npz: test_gpu_plugin.py, test_gpu_deepfm.py run the mirror's model files, whose
Linear layers are these kernels, against them) and against torch's fp32 modules (tests/test_gpu_mlp.py)."""
import numpy as np

from . import sampler_oracle

F32 = np.float32


def dropout_keep(seed, site, M, N, p):
    """keep-and-scale factors [M, N] of rc_linear_fwd: element (m, n) is dropped iff word (m & 3) of
    Philox4x32-10(key = seed, counter = (m >> 2, site * 65536 + n)) < p * 2^32; kept values carry 1 / (1 - p)"""
    M4 = (M + 3) // 4
    m4 = np.arange(M4, dtype=np.uint64)[:, None] + np.zeros((1, N), dtype=np.uint64)
    blk = np.zeros((M4, 1), dtype=np.uint32) + (np.uint32(site * 65536) + np.arange(N, dtype=np.uint32))[None]
    ctr = np.stack([(m4 & sampler_oracle.MASK32).astype(np.uint32), (m4 >> np.uint64(32)).astype(np.uint32), blk, np.zeros_like(blk)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), m4.shape + (2,))
    words = sampler_oracle.philox4x32_10(ctr, key)              # [M4, N, 4]: word e belongs to row 4 * m4 + e
    words = words.transpose(0, 2, 1).reshape(M4 * 4, N)[:M]
    thresh = np.uint32(int(float(F32(p)) * 4294967296.0))
    return np.where(words < thresh, F32(0), F32(1) / (F32(1) - F32(p))).astype(F32)


def linear_fwd(X, W, b, relu, keep=None):
    """-> (Y [M, N] fp32, cache)"""
    z = X.astype(np.float64) @ W.T.astype(np.float64)
    if b is not None:
        z = z + b
    y = np.maximum(z, 0) if relu else z
    mask = (z > 0) if relu else np.ones_like(z, dtype=bool)
    if keep is not None:
        y = y * keep
    return y.astype(F32), {"mask": mask, "keep": keep}


def linear_bwd(X, W, cache, dY):
    """-> (dX [M, K], dW [N, K], db [N]) of sum(dY * Y)"""
    dz = dY.astype(np.float64) * cache["mask"]
    if cache["keep"] is not None:
        dz = dz * cache["keep"]
    return (dz @ W.astype(np.float64)).astype(F32), (dz.T @ X.astype(np.float64)).astype(F32), dz.sum(0).astype(F32)


def linear_bwd_chain(X, W, dZ, x_mask=None, x_scale=1.0):
    """rc_linear_bwd_chain in float64: the layer receives dZ (its dY already masked by the layer above's dX product) and hands
    down dX multiplied by the mask of the layer BELOW -- x_mask = (X > 0) of the saved activation X = drop(relu(.)), x_scale =
    1 / (1 - p) of that layer's dropout -- i.e. the dZ of the layer below (utils/layers.py:201-243 chains Linear -> ReLU -> Dropout).
    -> (dX masked [M, K], dW [N, K], db [N])"""
    dz = dZ.astype(np.float64)
    dX = dz @ W.astype(np.float64)
    if x_mask is not None:
        dX = np.where(x_mask, dX * x_scale, 0.0)
    return dX.astype(F32), (dz.T @ X.astype(np.float64)).astype(F32), dz.sum(0).astype(F32)

