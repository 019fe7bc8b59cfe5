"""Shared helpers for the parity tests (oracle construction, golden loading)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import seeded
from oracle.head import OracleCrossHead2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def head_cfg():
    from pairnet_amd import pairnet_head_cfg
    cfg = pairnet_head_cfg()
    cfg.pop("type")
    return cfg


def oracle_head(weight_seed, overrides=None):
    """Oracle head with seeded weights (+ fixture ops); returns (head, state_dict, crc of
    the seeded weights before the ops)."""
    head = OracleCrossHead2(**head_cfg()).eval()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, weight_seed)
    crc = seeded.checksum(sd)
    seeded.apply_ops(sd, overrides or {})
    head.load_state_dict(sd, strict=True)
    return head, sd, crc


def baseline_cfg():
    from pairnet_amd import baseline_head_cfg
    cfg = baseline_head_cfg()
    cfg.pop("type")
    return cfg


def oracle_baseline_head(weight_seed, overrides=None):
    """Oracle of the sibling head CrossHeadBaseline with seeded weights."""
    from oracle.baseline_head import OracleCrossHeadBaseline
    head = OracleCrossHeadBaseline(**baseline_cfg()).eval()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, weight_seed)
    crc = seeded.checksum(sd)
    seeded.apply_ops(sd, overrides or {})
    head.load_state_dict(sd, strict=True)
    return head, sd, crc


def psgtr2_cfg():
    from pairnet_amd import psgtr2_head_cfg
    cfg = psgtr2_head_cfg()
    cfg.pop("type")
    return cfg


def oracle_psgtr2_head(weight_seed, overrides=None):
    """Oracle of the sibling head PSGTrHead2 with seeded weights."""
    from oracle.psgtr_head2 import OraclePSGTrHead2
    head = OraclePSGTrHead2(**psgtr2_cfg()).eval()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, weight_seed)
    crc = seeded.checksum(sd)
    seeded.apply_ops(sd, overrides or {})
    head.load_state_dict(sd, strict=True)
    return head, sd, crc


def overrides_of(fx):
    """The weight edits a fixture stores on top of its seed (oracle/seeded.py "ops")."""
    return seeded.ops_of(fx)


def tie_aware_topk_match(ref_scores, ref_idx, got_idx, tol):
    """Compare two descending top-k index lists where `ref_scores` (flat, the
    reference's scores) may contain near-ties: positions whose reference score is
    separated from both neighbours by more than `tol` must agree exactly; runs of
    near-tied positions must hold the same index multiset.  Returns (ok, n_exact)."""
    ref_idx = np.asarray(ref_idx).reshape(-1)
    got_idx = np.asarray(got_idx).reshape(-1)
    k = len(ref_idx)
    s = np.sort(ref_scores.reshape(-1))[::-1][:k + 1]
    gaps = s[:-1] - s[1:]               # gaps[i] between rank i and i+1
    ok, i = True, 0
    while i < k:
        j = i
        while j < k - 1 and gaps[j] <= tol:
            j += 1
        if j == k - 1 and gaps[k - 1] <= tol:
            # the run reaches past rank k: membership itself is ambiguous; accept any
            # candidates whose reference score is within the run
            lo = s[k] - tol
            ok &= all(ref_scores.reshape(-1)[g] >= lo for g in got_idx[i:k])
        else:
            ok &= sorted(ref_idx[i:j + 1].tolist()) == sorted(got_idx[i:j + 1].tolist())
        i = j + 1
    return bool(ok), int((ref_idx == got_idx).sum())
