#!/usr/bin/env python3
"""Measure what of the FDR output survives a second binary32 FFT (tests/fft_gap_common.py) -> profiles/r04_fft_gap.json
(CPU only; 8 processes, about a minute)."""
import json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fft_gap_common as F

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    jobs = F.workload(n)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        rs = list(ex.map(F.compare_frame, jobs, chunksize=8))
    tot = F.merge(rs)
    tot["seconds"] = time.time() - t0
    print(json.dumps(tot, indent=1))
    json.dump(tot, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r04_fft_gap.json"), "w"), indent=1)
