"""On-disk formats either side of the hot path, host side.

* Oxford COVID-19 Government Response Tracker table (`OxCGRT_latest.csv`): the columns the reference reads
  (Tools/TrainPredictPrescribeNPI.m:73-83, 98-117, 131-145): CountryName, RegionName, Date (YYYYMMDD as a number),
  ConfirmedCases, ConfirmedDeaths and the 12 NPI columns (`included_IP`, Tools/PrescribeNPI.m:24-36).  Regions are
  keyed `CountryName + " " + RegionName` in first-appearance order (`unique(AllGeoIDs, 'stable')`, :84).
* populations file (`xprize-sample-data/populations.csv`), NPI cost file (`fixed_equal_costs.csv`), intervention-plan
  file (`future_ip.csv`) and the XPRIZE prescription file
  (`2020-08-01_2020-08-04_prescriptions_example.csv`: PrescriptionIndex, CountryName, RegionName, Date, 12 NPI levels).
* the trained-parameter cell array the reference saves (`save(trained_model_params_file, 'TrainedModelParams')`,
  Tools/TrainPredictPrescribeNPI.m:912) -- read and written through scipy.io.

Everything comes back in the device layout of include/epiekf.h: series [T, S] (day-major, region-minor), NPI levels
[T, n_npi, S]; missing cells are NaN, exactly what `batch.preprocess` expects."""
from __future__ import annotations

import numpy as np
import pandas as pd

IP_COLUMNS = ["C1_School closing", "C2_Workplace closing", "C3_Cancel public events", "C4_Restrictions on gatherings",
              "C5_Close public transport", "C6_Stay at home requirements", "C7_Restrictions on internal movement",
              "C8_International travel controls", "H1_Public information campaigns", "H2_Testing policy",
              "H3_Contact tracing", "H6_Facial Coverings"]


def geo_id(country, region) -> str:
    """strcat(string(CountryName), " ", string(RegionName)) -- an empty RegionName leaves the trailing blank."""
    region = "" if region is None or (isinstance(region, float) and np.isnan(region)) else str(region)
    return f"{country} {region}"


def date_number(s) -> int:
    """'2020-08-01' -> 20200801 (Tools/TrainPredictPrescribeNPI.m:25-29); numbers pass through."""
    if isinstance(s, (int, np.integer)):
        return int(s)
    return int(str(s).replace("-", ""))


def _geo_series(df):
    reg = df["RegionName"].astype(object).where(df["RegionName"].notna(), "")
    return df["CountryName"].astype(str) + " " + reg.astype(str)


def read_oxcgrt(path, start_date=None, end_date=None, ip_columns=IP_COLUMNS, geo_ids=None):
    """Read the tracker table between two dates (inclusive).  Returns dict: geo_ids [S] (first-appearance order, or the
    requested `geo_ids`), countries, regions, dates [T] (YYYYMMDD ints), cases / deaths [T, S], ip [T, n_npi, S].
    Days a region has no row for, and empty cells, are NaN."""
    df = pd.read_csv(path, dtype={"CountryName": str, "RegionName": str}, low_memory=False)
    if not np.issubdtype(df["Date"].dtype, np.number):
        df["Date"] = df["Date"].map(date_number)
    lo = -np.inf if start_date is None else date_number(start_date)
    hi = np.inf if end_date is None else date_number(end_date)
    df = df[(df["Date"] >= lo) & (df["Date"] <= hi)].copy()
    df["_geo"] = _geo_series(df)
    order = list(pd.unique(df["_geo"])) if geo_ids is None else list(geo_ids)
    dates = np.sort(df["Date"].unique()).astype(np.int64)
    T, S, n = len(dates), len(order), len(ip_columns)
    cases = np.full((T, S), np.nan); deaths = np.full((T, S), np.nan); ip = np.full((T, n, S), np.nan)
    col = {g: i for i, g in enumerate(order)}
    row = {d: i for i, d in enumerate(dates)}
    sel = df[df["_geo"].isin(col)]
    ti = sel["Date"].map(row).to_numpy(); si = sel["_geo"].map(col).to_numpy()
    num = lambda c: pd.to_numeric(sel[c], errors="coerce").to_numpy(dtype=np.float64) if c in sel else np.full(len(sel), np.nan)
    cases[ti, si] = num("ConfirmedCases"); deaths[ti, si] = num("ConfirmedDeaths")
    for j, c in enumerate(ip_columns):
        ip[ti, j, si] = num(c)
    first = sel.drop_duplicates("_geo").set_index("_geo")
    countries = [first.loc[g, "CountryName"] if g in first.index else g for g in order]
    regions = [("" if (g not in first.index or pd.isna(first.loc[g, "RegionName"])) else first.loc[g, "RegionName"]) for g in order]
    return {"geo_ids": order, "countries": countries, "regions": regions, "dates": dates, "cases": cases,
            "deaths": deaths, "ip": ip}


def read_populations(path, geo_ids=None):
    """populations.csv -> dict geo_id -> Population2020 (or an array aligned with `geo_ids`, NaN where unknown)."""
    df = pd.read_csv(path, dtype={"CountryName": str, "RegionName": str})
    table = dict(zip(_geo_series(df), df["Population2020"].astype(float)))
    if geo_ids is None:
        return table
    return np.array([table.get(g, np.nan) for g in geo_ids])


def read_costs(path, geo_ids, ip_columns=IP_COLUMNS):
    """NPI cost file (fixed_equal_costs.csv / uniform_random_costs.csv) -> weights [n_npi, S]; unknown regions get 1."""
    df = pd.read_csv(path, dtype={"CountryName": str, "RegionName": str})
    df["_geo"] = _geo_series(df)
    df = df.set_index("_geo")
    w = np.ones((len(ip_columns), len(geo_ids)))
    for s, g in enumerate(geo_ids):
        if g in df.index:
            w[:, s] = df.loc[g, ip_columns].to_numpy(dtype=np.float64)
    return w


def read_ip_file(path, geo_ids=None, start_date=None, end_date=None, ip_columns=IP_COLUMNS):
    """Intervention-plan file (future_ip.csv: CountryName, RegionName, Date 'YYYY-MM-DD', 12 levels) -> same dict as
    read_oxcgrt without cases / deaths."""
    df = pd.read_csv(path, dtype={"CountryName": str, "RegionName": str})
    df["Date"] = df["Date"].map(date_number)
    tmp = df.assign(ConfirmedCases=np.nan, ConfirmedDeaths=np.nan)
    import io
    buf = io.StringIO(); tmp.to_csv(buf, index=False); buf.seek(0)
    out = read_oxcgrt(buf, start_date, end_date, ip_columns, geo_ids)
    del out["cases"], out["deaths"]
    return out


def _fmt_date(d: int) -> str:
    d = int(d)
    return f"{d // 10000:04d}-{d // 100 % 100:02d}-{d % 100:02d}"


def write_prescriptions(path, plans, countries, regions, dates, ip_columns=IP_COLUMNS):
    """XPRIZE prescription file.  plans [P, T, n_npi, S] (prescription index, day, NPI, region) of integer levels;
    rows ordered by prescription index, then region, then date, as in the sample file."""
    plans = np.asarray(plans)
    P, T, n, S = plans.shape
    if n != len(ip_columns) or T != len(dates) or S != len(countries):
        raise ValueError("plans must be [P, len(dates), len(ip_columns), len(countries)]")
    rows = []
    lv = np.rint(plans).astype(np.int64)
    for p in range(P):
        for s in range(S):
            for t in range(T):
                rows.append([p, countries[s], regions[s] if regions[s] else "", _fmt_date(dates[t])] + lv[p, t, :, s].tolist())
    pd.DataFrame(rows, columns=["PrescriptionIndex", "CountryName", "RegionName", "Date"] + list(ip_columns)).to_csv(path, index=False)


def read_prescriptions(path, ip_columns=IP_COLUMNS):
    """Inverse of write_prescriptions: dict with plans [P, T, n_npi, S], geo_ids, countries, regions, dates."""
    df = pd.read_csv(path, dtype={"CountryName": str, "RegionName": str})
    df["Date"] = df["Date"].map(date_number)
    df["_geo"] = _geo_series(df)
    geos = list(pd.unique(df["_geo"])); dates = np.sort(df["Date"].unique()).astype(np.int64)
    idx = np.sort(df["PrescriptionIndex"].unique())
    plans = np.full((len(idx), len(dates), len(ip_columns), len(geos)), np.nan)
    pi = df["PrescriptionIndex"].map({v: i for i, v in enumerate(idx)}).to_numpy()
    ti = df["Date"].map({d: i for i, d in enumerate(dates)}).to_numpy()
    si = df["_geo"].map({g: i for i, g in enumerate(geos)}).to_numpy()
    for j, c in enumerate(ip_columns):
        plans[pi, ti, j, si] = df[c].to_numpy(dtype=np.float64)
    first = df.drop_duplicates("_geo")
    return {"plans": plans, "geo_ids": geos, "countries": first["CountryName"].tolist(),
            "regions": ["" if pd.isna(r) else r for r in first["RegionName"]], "dates": dates,
            "prescription_index": idx}


def save_trained_params(path, rows):
    """rows: list of (CountryName, RegionName, N_population, reg_coef_b, reg_coef_a, reg_coef_b2, reg_coef_a2) ->
    the `TrainedModelParams` cell array with its header row (Tools/TrainPredictPrescribeNPI.m:90, 910-912)."""
    from scipy.io import savemat
    header = ["CountryName", "RegionName", "N_population", "reg_coef_b", "reg_coef_a", "reg_coef_b2", "reg_coef_a2"]
    cell = np.empty((len(rows) + 1, 7), dtype=object)
    cell[0] = header
    for i, r in enumerate(rows):
        c, g, N, b, a, b2, a2 = r
        cell[i + 1] = [c, g, float(N), float(b), np.asarray(a, dtype=np.float64).reshape(-1, 1), float(b2),
                       np.asarray(a2, dtype=np.float64).reshape(-1, 1)]
    savemat(path, {"TrainedModelParams": cell})


def load_trained_params(path):
    """-> list of dicts (one per region) from a `TrainedModelParams` .mat file."""
    from scipy.io import loadmat
    cell = loadmat(path, squeeze_me=True)["TrainedModelParams"]
    out = []
    for r in cell[1:]:
        out.append({"CountryName": str(r[0]), "RegionName": "" if np.size(r[1]) == 0 else str(r[1]),
                    "N_population": float(r[2]), "reg_coef_b": float(r[3]), "reg_coef_a": np.atleast_1d(r[4]).astype(float),
                    "reg_coef_b2": float(r[5]), "reg_coef_a2": np.atleast_1d(r[6]).astype(float)})
    return out
