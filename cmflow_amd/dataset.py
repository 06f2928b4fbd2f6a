"""Sample I/O: the View-of-Delft scene-flow sample format and its Datasets -- mirror of ``dataset/vod.py``
(vodDataset :14-137; format described in src/GETTING_STARTED.md:97-106), of ``dataset/vod_clip.py`` (vodClipDataset
:14-198, the mini-clip loader CMFlow-T trains on) and of ``extract_data_info`` / ``extract_data_info_clip``
(main_util.py:21-36, clip_util.py:81-96).

One sample = one JSON file ``<root>/<partition>/<clip>/<k>_*.json`` with
    pc1, pc2          [n][5]  x, y, z, RCS, v_r          (features fed to the net: [v_r, RCS, RCS], vod.py:62-63)
    gt_labels, pse_labels [n1][3], gt_mask, pse_mask [n1]  scene-flow labels / static masks (ground truth, pseudo)
    trans             [4][4]  frame-2 -> frame-1 ego transform (the loader inverts it, vod.py:92)
    opt_info          {opt_flow [n1][2], radar_u [n1], radar_v [n1]}   (training partitions only)
``__getitem__`` returns the reference's 11-tuple; in training mode clouds are resampled to ``args.num_points``
(duplicate padding below, random subset above; same numpy RNG call sequence as vod.py:95-121, so a seeded
run reproduces the reference's batches).  The calibration constants are those of dataset/vod_radar_calib.txt.
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synth


class vodDataset(Dataset):

    def __init__(self, args, root, partition='train', textio=None):
        self.npoints = args.num_points
        self.textio = textio
        self.res = {'r_res': 0.2, 'theta_res': 1.5 * np.pi / 180, 'phi_res': 1.5 * np.pi / 180}
        self.camera_projection_matrix = np.array(synth.CAMERA_PROJECTION, dtype=np.float32)
        self.t_camera_radar = np.array(synth.T_CAMERA_RADAR, dtype=np.float32)
        self.eval = args.eval
        self.partition = partition
        self.root = os.path.join(root, self.partition)
        self.interval = 0.10
        self.clips = sorted(os.listdir(self.root), key=lambda x: int(x.split("_")[1]))
        self.samples = []
        self.clips_info = []
        for clip in self.clips:
            clip_path = os.path.join(self.root, clip)
            files = sorted(os.listdir(clip_path), key=lambda x: int(x.split("_")[0]))
            if self.eval:
                self.clips_info.append({'clip_name': clip, 'index': [len(self.samples), len(self.samples) + len(files)]})
            if clip[:5] == 'delft':
                self.samples.extend(os.path.join(clip_path, f) for f in files)
        if self.textio is not None:
            self.textio.cprint(self.partition + ' : ' + str(len(self.samples)))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        with open(self.samples[index], 'rb') as fp:
            data = json.load(fp)
        return _sample_item(self, data)


def _resample(npoints, n):
    """vod.py:99-111 / vod_clip.py:183-191: keep all n points and pad with random duplicates, or draw a random subset."""
    if n < npoints:
        return np.append(np.arange(0, n), np.random.choice(n, npoints - n, replace=True))
    return np.random.choice(n, npoints, replace=False)


def _sample_item(ds, data):
    """One decoded sample -> the reference's 11-tuple (vod.py:54-124 = vod_clip.py:77-127: the two loaders share this body).
    Training (``not ds.eval``): both clouds resampled to ``ds.npoints`` -- cloud 1's index draw first, then cloud 2's, the numpy RNG
    call order of ``sample_points`` (vod_clip.py:181-193)."""
    d1 = np.array(data["pc1"]).astype('float32')
    d2 = np.array(data["pc2"]).astype('float32')
    pos_1, pos_2 = d1[:, 0:3], d2[:, 0:3]
    feature_1, feature_2 = d1[:, [4, 3, 3]], d2[:, [4, 3, 3]]
    if ds.partition in ('test', 'val', 'train_anno'):            # ground truth for evaluation
        labels = np.array(data["gt_labels"]).astype('float32')
        mask = np.array(data["gt_mask"])
        n1 = pos_1.shape[0]
        opt_flow = np.zeros((n1, 2)).astype('float32')
        radar_u, radar_v = np.zeros(n1).astype('float32'), np.zeros(n1).astype('float32')
    else:                                                         # pseudo labels + optical flow for training
        labels = np.array(data["pse_labels"]).astype('float32')
        mask = np.array(data["pse_mask"])
        info = data["opt_info"]
        opt_flow = np.array(info["opt_flow"]).astype('float32')
        radar_u = np.array(info["radar_u"]).astype('float32')
        radar_v = np.array(info["radar_v"]).astype('float32')
    trans = np.linalg.inv(np.array(data["trans"])).astype('float32')
    if not ds.eval:
        i1 = _resample(ds.npoints, pos_1.shape[0])
        i2 = _resample(ds.npoints, pos_2.shape[0])
        pos_1, pos_2 = pos_1[i1, :], pos_2[i2, :]
        feature_1, feature_2 = feature_1[i1, :], feature_2[i2, :]
        radar_u, radar_v, opt_flow = radar_u[i1], radar_v[i1], opt_flow[i1, :]
        labels, mask = labels[i1, :], mask[i1]
    return pos_1, pos_2, feature_1, feature_2, trans, labels, mask, ds.interval, radar_u, radar_v, opt_flow


class vodClipDataset(Dataset):
    """``dataset/vod_clip.py:14-198``.  Training (``args.eval`` false): item i is the i-th MINI-CLIP -- ``args.mini_clip_len``
    consecutive samples of one clip (clips cut into floor(len / mini_clip_len) mini-clips, the remainder dropped, :40-50) -- as
    eleven arrays with a leading ``(mini_clip_len, ...)`` axis (:131-170); every frame is resampled to ``args.num_points`` in
    file order (one numpy RNG stream).  Evaluation: item i is ONE frame (:69-74), ragged, in clip order, with ``clips_info``
    giving each clip's [first, last) frame range (:34-39)."""

    def __init__(self, args, root, partition='train', textio=None):
        self.npoints = args.num_points
        self.textio = textio
        self.res = {'r_res': 0.2, 'theta_res': 1.5 * np.pi / 180, 'phi_res': 1.5 * np.pi / 180}
        self.camera_projection_matrix = np.array(synth.CAMERA_PROJECTION, dtype=np.float32)
        self.t_camera_radar = np.array(synth.T_CAMERA_RADAR, dtype=np.float32)
        self.eval = args.eval
        self.partition = partition
        self.root = os.path.join(root, self.partition)
        self.interval = 0.10
        self.mini_clip_len = args.mini_clip_len
        self.update_len = args.update_len
        self.clips = sorted(os.listdir(self.root), key=lambda x: int(x.split("_")[1]))
        self.mini_samples = []
        self.samples = []
        self.clips_info = []
        self.mini_clips_info = []
        for clip in self.clips:
            clip_path = os.path.join(self.root, clip)
            files = sorted(os.listdir(clip_path), key=lambda x: int(x.split("/")[-1].split("_")[0]))
            if self.eval:
                self.clips_info.append({'clip_name': clip, 'index': [len(self.samples), len(self.samples) + len(files)]})
                self.samples.extend(os.path.join(clip_path, f) for f in files)
            else:
                for i in range(int(np.floor(len(files) / self.mini_clip_len))):
                    mini = [os.path.join(clip_path, files[i * self.mini_clip_len + j]) for j in range(self.mini_clip_len)]
                    self.samples.extend(mini)
                    self.mini_samples.append(mini)
        if self.textio is not None:
            if self.eval:
                self.textio.cprint(self.partition + ' : ' + str(len(self.samples)) + ' frames')
            else:
                self.textio.cprint(self.partition + ' : ' + str(len(self.mini_samples)) + ' mini_clips')

    def __len__(self):
        return len(self.samples) if self.eval else len(self.mini_samples)

    def get_sample_item(self, data):
        return _sample_item(self, data)

    def get_clip_item(self, index):
        mini = self.mini_samples[index]
        L, n = self.mini_clip_len, self.npoints
        z = lambda *shape: np.zeros(shape).astype('float32')
        out = (z(L, n, 3), z(L, n, 3), z(L, n, 3), z(L, n, 3), z(L, 4, 4), z(L, n, 3), z(L, n), z(L), z(L, n), z(L, n), z(L, n, 2))
        for i, path in enumerate(mini):
            with open(path, 'rb') as fp:
                item = _sample_item(self, json.load(fp))
            for dst, v in zip(out, item):
                dst[i] = v
        return out

    def __getitem__(self, index):
        if not self.eval:
            return self.get_clip_item(index)
        with open(self.samples[index], 'rb') as fp:
            return _sample_item(self, json.load(fp))


def extract_data_info(data, device="cuda"):
    """main_util.py:21-36: a collated batch -> model-layout device tensors
    (pc1, pc2, ft1, ft2 (B,3,N); trans (B,4,4); gt (B,N,3); mask (B,N); interval (B); radar_u/v (B,N); opt_flow (B,N,2))."""
    pc1, pc2, ft1, ft2, trans, gt, mask, interval, radar_u, radar_v, opt_flow = data
    cm = lambda t: torch.as_tensor(t).to(device).transpose(2, 1).contiguous()
    fl = lambda t: torch.as_tensor(t).to(device).float()
    return (cm(pc1), cm(pc2), cm(ft1), cm(ft2), fl(trans), fl(gt), fl(mask), fl(interval), fl(radar_u), fl(radar_v),
            fl(opt_flow))


def extract_data_info_clip(seq_data, idx, device="cuda"):
    """clip_util.py:81-96: frame ``idx`` of a collated mini-clip batch -> the tuple of extract_data_info."""
    return extract_data_info(tuple(t[:, idx] for t in seq_data), device=device)


def as_batch_dict(info):
    """The tuple of extract_data_info as the dict TrainStep / RadarFlowLoss take (train partitions: mask = fg_mask)."""
    keys = ("pc1", "pc2", "ft1", "ft2", "gt_trans", "flow_label", "fg_mask", "interval", "radar_u", "radar_v", "opt_flow")
    return dict(zip(keys, info))


def write_sample(path, pc1, pc2, gt_labels, pse_labels, gt_mask, pse_mask, trans, opt_flow=None, radar_u=None, radar_v=None):
    """Write one sample file in the format above (pc1, pc2: (n,5) arrays x,y,z,RCS,v_r)."""
    ls = lambda a: np.asarray(a).tolist()
    data = {"pc1": ls(pc1), "pc2": ls(pc2), "gt_labels": ls(gt_labels), "pse_labels": ls(pse_labels),
            "gt_mask": ls(gt_mask), "pse_mask": ls(pse_mask), "trans": ls(trans)}
    if opt_flow is not None:
        data["opt_info"] = {"opt_flow": ls(opt_flow), "radar_u": ls(radar_u), "radar_v": ls(radar_v)}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(data, f)


def write_synthetic_split(root, seed=0, clips=(("train", "delft_1", (180, 256, 400)), ("train", "delft_12", (300,)),
                                               ("test", "delft_2", (210, 330)))):
    """A few synthetic samples with ragged point counts in the reference's directory layout (for tests and
    dry runs).  Returns the relative file names written."""
    names = []
    for ci, (part, clip, sizes) in enumerate(clips):
        for k, n in enumerate(sizes):
            b = synth.make_batch(1, N=n + 37, seed=seed + 100 * ci + k, train_extras=True)
            n2 = n + 37 - 11 * k
            xyz1, xyz2 = b["pc1"][0].t().numpy()[:n], b["pc2"][0].t().numpy()[:n2]
            f1, f2 = b["ft1"][0].t().numpy()[:n], b["ft2"][0].t().numpy()[:n2]
            pc1 = np.concatenate([xyz1, f1[:, 1:2], f1[:, 0:1]], axis=1)          # x, y, z, RCS, v_r
            pc2 = np.concatenate([xyz2, f2[:, 1:2], f2[:, 0:1]], axis=1)
            lab = b["flow_label"][0].numpy()[:n]
            rng = np.random.default_rng(seed + k)
            pse = lab + rng.normal(0, 0.02, lab.shape)
            mask = b["fg_mask"][0].numpy()[:n]
            rel = os.path.join(part, clip, "%d_%s.json" % (k + 5 * ci, clip))
            write_sample(os.path.join(root, rel), pc1, pc2, lab, pse, mask, (rng.random(n) < 0.8).astype(np.float64),
                         np.linalg.inv(b["gt_trans"][0].numpy().astype(np.float64)), b["opt_flow"][0].numpy()[:n],
                         b["radar_u"][0].numpy()[:n], b["radar_v"][0].numpy()[:n])
            names.append(rel)
    return names
