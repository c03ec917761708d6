"""Hypothesis grids for the fine sweep (BASELINE.json configs[2]):
200 hypotheses per frame = f0 in f_c + {-2..2}*0.25 Hz  x  lag in shift_c +
{-96,-64,-32,0,32,64,96,128}  x  drift in {-1,-0.5,0,0.5,1} Hz, linear model."""
import numpy as np

from .native import HYP_DTYPE, LINEAR

DF_STEPS = (-2, -1, 0, 1, 2)
LAGS = (-96, -64, -32, 0, 32, 64, 96, 128)
DRIFTS = (-1.0, -0.5, 0.0, 0.5, 1.0)


def sweep_grid(cands, frames_idx=None):
    """cands: one candidate record per frame (freq, shift used). -> HYP_DTYPE [len*200],
    grouped by frame in order."""
    cands = np.atleast_1d(cands)
    if frames_idx is None:
        frames_idx = range(len(cands))
    out = np.zeros(len(cands) * 200, HYP_DTYPE)
    q = 0
    for c, fi in zip(cands, frames_idx):
        fc = np.float32(c["freq"])
        for k in DF_STEPS:
            f0 = np.float32(fc + np.float32(k) * np.float32(0.25))
            for lag in LAGS:
                for d in DRIFTS:
                    out[q]["frame"] = fi
                    out[q]["m_type"] = LINEAR
                    out[q]["f0"] = f0
                    out[q]["lag"] = int(c["shift"]) + lag
                    out[q]["drift"] = d
                    q += 1
    return out


def sweep_grid_uniform(B, f_c=0.0, shift_c=368):
    """Same 200-point grid around a fixed (f_c, shift_c) for every one of B frames
    (vectorised; bench input)."""
    k, lag, d = np.meshgrid(np.array(DF_STEPS, np.float32), np.array(LAGS, np.int32),
                            np.array(DRIFTS, np.float32), indexing="ij")
    one = np.zeros(200, HYP_DTYPE)
    one["m_type"] = LINEAR
    one["f0"] = (np.float32(f_c) + k * np.float32(0.25)).reshape(-1)
    one["lag"] = (shift_c + lag).reshape(-1)
    one["drift"] = d.reshape(-1)
    out = np.tile(one, B)
    out["frame"] = np.repeat(np.arange(B, dtype=np.int32), 200)
    return out
