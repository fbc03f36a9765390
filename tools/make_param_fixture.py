"""Extracts the 236 trained SI-alpha parameter sets of the reference's sample data into a
small .npz that travels with the repo (the reference tree does not exist on the GPU box).

Source (read-only): /root/reference/xprize-sample-data/prescription_trained_params_nonnegls.mat
  TrainedModelParams: 236 x {CountryName, RegionName, N_population, coef0, coef(12), coef0_2, coef_2(12)}
  (written by Tools/TrainPredictPrescribeNPI.m:910-913, read by testScripts/testPrescribeXPRIZE01.m:59)
Output: epidemicmodeling_amd/data/trained_params_nonnegls.npz  (data only: numbers and names)

Run in the build container:  python tools/make_param_fixture.py
"""
import os
import numpy as np
import scipy.io as sio

SRC = "/root/reference/xprize-sample-data/prescription_trained_params_nonnegls.mat"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "epidemicmodeling_amd", "data",
                   "trained_params_nonnegls.npz")

tp = sio.loadmat(SRC)["TrainedModelParams"]
hdr = [str(tp[0, j][0]) for j in range(tp.shape[1])]
assert hdr == ["CountryName", "RegionName", "N_population", "coef0", "coef", "coef0_2", "coef_2"], hdr
rows = tp[1:]
names = np.array([(str(r[0][0]) if r[0].size else "") + "|" + (str(r[1][0]) if r[1].size else "") for r in rows])
N = np.array([float(r[2].reshape(-1)[0]) for r in rows])
b1 = np.array([float(r[3].reshape(-1)[0]) for r in rows])
a1 = np.stack([np.asarray(r[4], dtype=np.float64).reshape(-1) for r in rows])
b2 = np.array([float(r[5].reshape(-1)[0]) for r in rows])
a2 = np.stack([np.asarray(r[6], dtype=np.float64).reshape(-1) for r in rows])
print(len(rows), a1.shape, a2.shape, N.min(), N.max())
np.savez_compressed(DST, names=names, N_population=N, coef0=b1, coef=a1, coef0_2=b2, coef_2=a2)
print("wrote", os.path.normpath(DST), os.path.getsize(DST), "bytes")
