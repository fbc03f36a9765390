"""Tracker CSV in, XPRIZE prescription CSV out -- the device stages of Tools/TrainPredictPrescribeNPI.m for all regions
of the file at once (epidemicmodeling_amd/pipeline.py).

    python examples/prescribe_from_csv.py OxCGRT_latest.csv populations.csv 2020-03-01 2020-12-31 30 out.csv

Without arguments a small synthetic tracker file is generated first (there is no data set in this repository)."""
import os
import sys
import tempfile

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epidemicmodeling_amd import dataio, pipeline, synth  # noqa: E402


def synthetic_files(tmp, S=8, T=200):
    raw = synth.make_raw_counts(S, T, seed=1)
    raw["cases"][:, -1] = np.cumsum(np.full(T, 25.0))
    dates = [int(d.strftime("%Y%m%d")) for d in pd.date_range("2020-03-01", periods=T)]
    rows = []
    for s in range(S):
        for t in range(T):
            rows.append([f"Country{s}", "", dates[t], raw["cases"][t, s], raw["deaths"][t, s]] + raw["ip"][t, :, s].tolist())
    data = os.path.join(tmp, "OxCGRT_latest.csv")
    pd.DataFrame(rows, columns=["CountryName", "RegionName", "Date", "ConfirmedCases", "ConfirmedDeaths"] + dataio.IP_COLUMNS).to_csv(data, index=False)
    pops = os.path.join(tmp, "populations.csv")
    pd.DataFrame({"CountryName": [f"Country{s}" for s in range(S)], "RegionName": [""] * S, "Population2020": raw["population"]}).to_csv(pops, index=False)
    return data, pops, dates[0], dates[-1]


def main():
    if len(sys.argv) >= 7:
        data, pops, start, end, horizon, dst = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    else:
        tmp = tempfile.mkdtemp()
        data, pops, start, end = synthetic_files(tmp)
        horizon, dst = 30, os.path.join(tmp, "prescriptions.csv")
    d = dataio.read_oxcgrt(data, start, end)
    N = dataio.read_populations(pops, d["geo_ids"])
    keep = np.flatnonzero(np.isfinite(N) & np.isfinite(d["cases"]).any(axis=0))
    out = pipeline.prescribe(d["cases"][:, keep], d["deaths"][:, keep], N[keep], d["ip"][:, :, keep], horizon=horizon, n_eps=50)
    last = pd.Timestamp(str(d["dates"][-1]))
    days = [int((last + pd.Timedelta(days=k + 1)).strftime("%Y%m%d")) for k in range(horizon)]
    dataio.write_prescriptions(dst, out["prescription"][None], [d["countries"][k] for k in keep], [d["regions"][k] for k in keep], days)
    print(f"{len(keep)} regions x {horizon} days -> {dst}")
    print("Pareto optimum per region (index into the cost-weight grid):", out["i_opt"].tolist())


if __name__ == "__main__":
    main()
