#!/usr/bin/env python3
"""Golden vectors for the mini-clip loader, produced by the REFERENCE's own dataset/vod_clip.py vodClipDataset
behind the shims of make_golden.py (build container only): the mini-clip tuples of the training partition under a fixed numpy
seed (resampling to num_points, clips cut into mini-clips with the remainder dropped, a clip whose name does not start with
'delft' -- this loader has no such filter) and the per-frame tuples + clips_info of the evaluation partition, from sample files
written by cmflow_amd.dataset.write_synthetic_split.  Small clouds (70-130 points, num_points = 96) keep the fixture small.

    python tests/golden/make_golden_clip.py
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims  # noqa: E402

CLIPS = (("train", "delft_1", (70, 96, 130, 88, 120)), ("train", "delft_12", (101, 75, 96)), ("train", "other_7", (90, 110)),
         ("test", "delft_2", (83, 128)), ("test", "delft_9", (96,)))


class Args:
    num_points = 96
    eval = False
    mini_clip_len = 2
    update_len = 1


class Textio:
    def __init__(self):
        self.lines = []

    def cprint(self, s):
        self.lines.append(s)


def main():
    tmp = tempfile.mkdtemp(prefix="cmf_clip_")
    from cmflow_amd import dataset as D                               # writer only; imported before the chdir
    names = D.write_synthetic_split(tmp, seed=5, clips=CLIPS)
    install_shims()
    from dataset.vod_clip import vodClipDataset
    out = {}
    for part, ev_flag in (("train", False), ("test", True)):
        a = Args()
        a.eval = ev_flag
        tio = Textio()
        ds = vodClipDataset(a, root=tmp + "/", partition=part, textio=tio)
        np.random.seed(23)
        out["%s/len" % part] = np.int64(len(ds))
        out["%s/textio" % part] = np.array(json.dumps(tio.lines))
        out["%s/samples" % part] = np.array(json.dumps([os.path.relpath(p, tmp) for p in ds.samples]))
        for i in range(len(ds)):
            for j, v in enumerate(ds[i]):
                out["%s/%d/%d" % (part, i, j)] = np.asarray(v).copy()
        if ev_flag:
            out["%s/clips_info" % part] = np.array(json.dumps(ds.clips_info))
        else:
            out["%s/mini_samples" % part] = np.array(json.dumps([[os.path.relpath(p, tmp) for p in m] for m in ds.mini_samples]))
    for rel in names:
        out["file::" + rel] = np.array(open(os.path.join(tmp, rel)).read())
    np.savez_compressed(os.path.join(HERE, "vod_clip_kat.npz"), **out)
    print("clip dataset:", {k: int(out[k]) for k in out if k.endswith("/len")}, "files", len(names),
          "%.0f KB" % (os.path.getsize(os.path.join(HERE, "vod_clip_kat.npz")) / 1e3))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
