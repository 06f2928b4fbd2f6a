"""Adds the host calibration beside the reference's CPU timing in ref_cpu_timing.json: the time of a fixed fp32 matmul
(2048 x 2048 x 2048) on the same thread count, min of 7, measured in the container that produced `ref_cpu_fwd_b1_s`
(make_golden.py).  tests/test_oracle.py::test_oracle_is_not_a_strawman scales its bound by (the same matmul now) / (this number), so a
loaded or slower host moves the bound with it instead of turning the suite red.  Run on an otherwise idle container."""
import json
import os
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def calib(threads):
    torch.set_num_threads(threads)
    a, b = torch.randn(2048, 2048), torch.randn(2048, 2048)
    torch.mm(a, b)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        torch.mm(a, b)
        ts.append(time.perf_counter() - t0)
    return min(ts)


if __name__ == "__main__":
    p = os.path.join(HERE, "ref_cpu_timing.json")
    rec = json.load(open(p))
    rec["calib_mm2048_s"] = calib(rec["threads"])
    json.dump(rec, open(p, "w"))
    print(rec)
