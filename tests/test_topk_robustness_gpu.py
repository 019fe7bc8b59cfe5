"""How robust is "top-k pair indices bit-exact" WITHOUT an engineered score gap?

Every strict fixture of this repo carries separation edits (oracle/make_golden.py:301-398:
the gap between consecutive top-k scores is made >= 10 x the fp32-vs-fp64 noise), because the
semantics (pairnet_head.py:334-340; consumer sgg_metrics.py:95-99: the evaluator takes the
pairs IN ORDER) make the list a discrete function of scores that two correct fp32
implementations round differently.  This test measures the other case: plain seeded
("unseparated") weights -- the only proxy for a released checkpoint that exists offline -- 20
seeds at 256 x 320 and 2 at 800 x 1333, through the CPU oracle in fp32 and in fp64 and through
the HIP path.  Per seed it records the smallest gap between consecutive top-k scores, the
fp32-vs-fp64 noise of the scores, which of the three lists agree and the first differing rank,
writes the table to gpurun_out/r06_topk_flip_rate.json (committed as
profiles/r06_topk_flip_rate.json; round 5 under the fp32 arithmetic: r05_...) and asserts the one property a correct implementation can
promise: wherever the GPU list differs from the oracle's, the scores at that rank are closer
to a neighbour than 4 x the measured noise.  Seeded weights, not trained ones.
"""
import copy
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import seeded
from oracle.head import OracleCrossHead2
from helpers import head_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(seed, H, W, sf):
    from pairnet_amd import CrossHead2
    cfg = head_cfg()
    head_o = OracleCrossHead2(**cfg).eval()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in head_o.state_dict().items())
    sd = seeded.seeded_state_dict(shapes, seed)          # no separation edits
    head_o.load_state_dict(sd)
    feats = seeded.seeded_feats(1000 + seed, 1, H, W)
    metas = [dict(img_shape=(H, W, 3), scale_factor=[sf] * 4)]
    t32, t64 = {}, {}
    with torch.no_grad():
        c32, _ = head_o.forward(feats, metas, trace=t32)
        c64, _ = copy.deepcopy(head_o).double().forward([f.double() for f in feats], metas,
                                                         trace=t64)
    head = CrossHead2(**cfg)
    head.load_state_dict(sd)
    head.to(DEV)
    cg, _ = head.forward([f.to(DEV) for f in feats], metas)
    torch.cuda.synchronize()
    k = head.num_rel_query
    i32, i64 = c32["importance"][0].numpy().ravel(), c64["importance"][0].numpy().ravel()
    ig = cg["importance"][0].cpu().numpy().ravel()
    l32, l64 = t32["topk_idx"][0].numpy(), t64["topk_idx"][0].numpy()
    lg = head._last_plan.topk_idx[0].cpu().numpy()
    s = np.sort(i64)[::-1][:k + 1]
    gaps = s[:-1] - s[1:]                                # gaps[r]: rank r to rank r + 1 (fp64)
    noise32, noiseg = float(np.abs(i32 - i64).max()), float(np.abs(ig - i64).max())
    noise = max(noise32, noiseg)

    def first_diff(a, b):
        d = np.nonzero(a != b)[0]
        return int(d[0]) if len(d) else None

    def near_tie(r):     # rank r is closer than 4 x noise to a neighbour in the fp64 order
        lo = gaps[r - 1] if r > 0 else np.inf
        return bool(min(lo, gaps[r]) < 4.0 * noise)
    diff_g32 = np.nonzero(lg != l32)[0].tolist()
    diff_g64 = np.nonzero(lg != l64)[0].tolist()
    # membership: indices the GPU selected that the fp64 oracle did not
    extra = sorted(set(lg.tolist()) - set(l64.tolist()))
    rec = dict(seed=seed, height=H, width=W,
               min_gap=float(gaps[:k].min()), median_gap=float(np.median(gaps[:k])),
               score_range=[float(s[0]), float(s[k - 1])],
               noise_oracle_fp32_vs_fp64=noise32, noise_gpu_vs_fp64=noiseg,
               rel_err_gpu_vs_oracle_fp32=float((cg["rel"].cpu() - c32["rel"]).abs().max())
               if not diff_g32 else None,
               oracle_fp32_eq_fp64=bool((l32 == l64).all()),
               gpu_eq_oracle_fp32=not diff_g32, gpu_eq_oracle_fp64=not diff_g64,
               first_diff_fp32_vs_fp64=first_diff(l32, l64),
               first_diff_gpu_vs_fp32=first_diff(lg, l32),
               first_diff_gpu_vs_fp64=first_diff(lg, l64),
               ranks_differing_gpu_vs_fp32=len(diff_g32),
               ranks_differing_gpu_vs_fp64=len(diff_g64),
               members_not_in_fp64_list=len(extra),
               ranks_with_gap_below_4x_noise=int(sum(near_tie(r) for r in range(k))))
    # the promise: every differing rank sits in a near-tie of the reference scores.  (Both
    # orders are compared: a rank can differ from the fp32 oracle where the fp32 oracle itself
    # differs from fp64.)
    bad = [r for r in diff_g32 if not near_tie(r)]
    bad64 = [r for r in diff_g64 if not near_tie(r)]
    return rec, bad, bad64


def test_topk_flip_rate_on_unseparated_seeded_weights():
    # the committed study (profiles/r05_topk_flip_rate.json) is the FULL set: 20 seeds at
    # 256 x 320 + 2 at 800 x 1333, PAIRNET_TOPK_FULL=1 (two fp64 oracle passes at full size take
    # most of its two minutes); the default run asserts the same property on a subset
    full = bool(os.environ.get("PAIRNET_TOPK_FULL"))
    cases = [(100 + i, 256, 320, 1.0) for i in range(20 if full else 8)] + \
            [(900 + i, 800, 1333, 2.083) for i in range(2 if full else 1)]
    recs, failures = [], []
    for seed, H, W, sf in cases:
        rec, bad, bad64 = _case(seed, H, W, sf)
        recs.append(rec)
        if bad or bad64:
            failures.append((seed, bad, bad64, rec))
    n = len(recs)
    summary = dict(
        what="top-k pair lists (k = 100 of 10 000 scores) on UNSEPARATED seeded weights: CPU "
             "oracle fp32, CPU oracle fp64, HIP path; seeded weights, not a trained checkpoint",
        cases=n,
        gpu_list_identical_to_oracle_fp32=sum(r["gpu_eq_oracle_fp32"] for r in recs),
        gpu_list_identical_to_oracle_fp64=sum(r["gpu_eq_oracle_fp64"] for r in recs),
        oracle_fp32_identical_to_fp64=sum(r["oracle_fp32_eq_fp64"] for r in recs),
        flip_rate_gpu_vs_oracle_fp32=1.0 - sum(r["gpu_eq_oracle_fp32"] for r in recs) / n,
        flip_rate_oracle_fp32_vs_fp64=1.0 - sum(r["oracle_fp32_eq_fp64"] for r in recs) / n,
        differing_ranks_outside_near_ties=sum(len(b) + len(b64) for _, b, b64, _ in failures),
        per_seed=recs)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        if not full:
            raise OSError("subset run: the committed record is written by the full set only")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r06_topk_flip_rate.json"), "w") as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    print(json.dumps({k: v for k, v in summary.items() if k != "per_seed"}))
    assert not failures, failures[:2]
