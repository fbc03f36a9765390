"""pinv truncation ranks by day and the eks_pinv grid's time for the headline batch (75 000 chains x 520 days), on the
survey's series and on the living epidemic:   python profiles/pinv_rank_histogram.py [out.json]

For each workload: share of the 75 000 covariances P(k+1|k) at each rank 0..6 in 20-day bins, the day by which half the
chains have left full rank, and the HIP-event time of the three stages (forward, pinv grid, smoother) enqueued one by one."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import batch, synth  # noqa: E402

out = {}
for tag, live in (("cfg4 (survey's series)", False), ("cfg4-live (living epidemic)", True)):
    w = synth.make_cfg4(live=live)
    r = batch.EkfRunner(batch.DeviceWorkload(w, "cuda:0"), lane_block="auto", extras=True)
    for ph in (1, 3, 4):
        r.run(phase=ph)
    torch.cuda.synchronize()
    ms = {}
    for name, ph in (("ekf_fwd", 1), ("eks_pinv", 3), ("eks_bwd", 4)):
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r.run(phase=ph); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms[name] = float(np.median(ts))
    rk = r.unblocked("pinv_rank")[:-1]                      # [T-1, B]: rank kept for P(k+1|k), k = 0 .. T-2
    T1 = rk.shape[0]
    bins = {}
    for d0 in range(0, T1, 20):
        blk = rk[d0:min(d0 + 20, T1)].flatten()
        h = torch.bincount(blk.clamp(min=0).long(), minlength=7).float()
        bins[f"{d0}-{min(d0 + 20, T1) - 1}"] = [round(float(v), 4) for v in (h / h.sum())]
    full = (rk == 6).float().mean(dim=1).cpu().numpy()
    half = int(np.argmax(full < 0.5)) if (full < 0.5).any() else None
    tot = torch.bincount(rk.flatten().clamp(min=0).long(), minlength=7).float()
    x, R = w.x[:400], w.R_series[:400]
    S = r.unblocked_at("S_SMOOTH", 399)[1]
    out[tag] = {"stage_ms": ms, "share_of_all_steps_by_rank_0_to_6": [round(float(v), 4) for v in (tot / tot.sum())],
                "first_day_on_which_fewer_than_half_the_chains_have_rank_6": half,
                "rank_shares_by_20_day_bin": bins,
                "smoothed_i_at_day_400_min_max": [float(S.min()), float(S.max())],
                "share_of_observed_days_with_x_gt_0_min_over_regions": float((x > 0).mean(axis=0).min()),
                "share_of_observed_days_with_R_v_gt_0_min_over_regions": float((R > 0).mean(axis=0).min())}
    print(tag, ms, out[tag]["share_of_all_steps_by_rank_0_to_6"], "half-rank day", half, flush=True)
    del r
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
