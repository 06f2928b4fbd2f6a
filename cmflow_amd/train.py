"""The reference's training-step sequence (main_util.py:63-76 / clip_util.py:34-62) on device.

    labels -> net(..., mseg_gt, 'train') -> RadarFlowLoss -> zero_grad / backward / [all-reduce] / step

``TrainStep`` owns the optimizer (Adam lr 1e-3, weight decay 1e-4: main.py:107), the flat
gradient bucket and the loss module; ``__call__(batch)`` runs one optimizer step and returns
(loss, items) as device tensors.
"""
import os

import torch

from . import synth
from .dp import FlatAdam, FlatGradBucket, SegmentedReducer
from .fused_blocks import join_side_streams
from .losses import RadarFlowLoss, make_labels


class TrainStep:
    def __init__(self, net, vr_thres=0.3, lr=0.001, weight_decay=1e-4, camera_projection=None, t_camera_radar=None):
        self.net = net
        dev = next(net.parameters()).device
        self.vr_thres = vr_thres
        self.loss_obj = RadarFlowLoss(camera_projection or synth.CAMERA_PROJECTION,
                                      t_camera_radar or synth.T_CAMERA_RADAR).to(dev)
        self.bucket = FlatGradBucket(net)
        # the reference's Adam (main.py:107) as one launch over the flat bucket (dp.FlatAdam; torch's fused Adam: six launches at the
        # tail of the step); on a CPU model torch's own
        self.opt = (FlatAdam(self.bucket, lr=lr, weight_decay=weight_decay) if dev.type == "cuda"
                    else torch.optim.Adam(self.bucket.params, lr=lr, weight_decay=weight_decay))
        self.recurrent = hasattr(net, "gru")
        self.self_supervised = hasattr(net, "fd_layer")
        # the scales of an encoder run on side streams while the gradient bucket lives on the main stream
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self.gfeat = None
        # diagnostics (bench.py --force-allreduce): run the gradient all-reduce even with a single rank
        self.force_allreduce = False
        # The all-reduce can be cut into three segments in the order backward completes them -- heads + second encoder
        # (+ GRU), cost volume, first encoder -- each launched from a tensor hook as soon as its chains have been enqueued
        # (models/model.py:40-42: nn.DataParallel reduces inside backward too).  Opt-in (CMF_OVERLAP_ALLREDUCE=1 or
        # overlap_allreduce = True): measured on one MI355X with a world-1 RCCL group inside every step (bench.py
        # --force-allreduce, same box) the step is 22.22 ms without a collective, 22.79 ms with ONE all-reduce after backward
        # and 23.00 ms with the three overlapped segments -- RCCL's stream is a fifth busy hardware queue next to the four
        # the step uses (more queues are slower on this part), so running it DURING backward costs more
        # than the 0.2-0.3 ms of ring time it could hide at 8 GPUs.  Both forms leave bit-identical buckets (tests/test_dp.py,
        # the two-rank GPU tests run the overlapped one).
        self.overlap_allreduce = os.environ.get("CMF_OVERLAP_ALLREDUCE") == "1"
        self.reducer = None
        enc2 = net._second_encoder() if hasattr(net, "_second_encoder") else None
        if enc2 is not None and hasattr(net, "fc_layer") and hasattr(net, "mse_layer") and not self.self_supervised:
            try:
                late = [m for m in (enc2, getattr(net, "gru", None), net.fp, net.mp) if m is not None]
                segs = [self.bucket.segment_of(late), self.bucket.segment_of([net.fc_layer]), self.bucket.segment_of([net.mse_layer])]
                self.reducer = SegmentedReducer(self.bucket, segs)
            except ValueError:                                   # a model whose parameter order does not follow the data flow
                self.reducer = None

    def reset_clip(self):
        """clip_util.py:51-52: the first frame of a mini-clip starts from gfeat=None."""
        self.gfeat = None

    def forward_loss(self, batch):
        pc1, pc2, ft1, ft2 = batch["pc1"], batch["pc2"], batch["ft1"], batch["ft2"]
        if self.self_supervised:                                   # model 'raflow': main_util.py:57-60
            output, pred_f, pre_trans, mask_s = self.net(pc1, pc2, ft1, ft2, batch["interval"])
            loss, items = self.loss_obj(pc1, pc2, pred_f, ft1[:, 0])
            return loss, items, (pred_f, output, pre_trans, mask_s), (None, None)
        dyn_mask, mseg_gt = make_labels(batch, self.vr_thres)
        if self.recurrent:
            g = self.gfeat.detach() if self.gfeat is not None else None          # clip_util.py:54
            pred_f, mseg_pre, pre_trans, mask, self.gfeat = self.net(pc1, pc2, ft1, ft2, mseg_gt, 'train', g)
        else:
            pred_f, mseg_pre, pre_trans, mask = self.net(pc1, pc2, ft1, ft2, mseg_gt, 'train')
        loss, items = self.loss_obj(pc1, pc2, pred_f, ft1[:, 0], batch["flow_label"].transpose(2, 1), pre_trans,
                                    mseg_pre, batch["gt_trans"], mseg_gt, dyn_mask, batch["radar_u"],
                                    batch["radar_v"], batch["opt_flow"])
        return loss, items, (pred_f, mseg_pre, pre_trans, mask), (dyn_mask, mseg_gt)

    def _segment_ready(self, i):
        """Tensor hook (autograd thread): everything that writes segment i has been enqueued.  Parameter gradients are
        accumulated in place by kernels on the side streams (fused_blocks.grad_sink), so the pool is joined into this
        stream first; the collective then waits for this stream."""
        if self.reducer is not None and self.reducer.active:
            join_side_streams()
            self.reducer.launch(i)

    def __call__(self, batch):
        overlap = self.overlap_allreduce and self.reducer is not None and self.reducer.begin(self.force_allreduce)
        self.net._grad_ready = self._segment_ready if overlap else None
        try:
            loss, items, outs, labels = self.forward_loss(batch)
        finally:
            self.net._grad_ready = None
        self.bucket.zero()
        loss.backward()
        if loss.is_cuda:
            join_side_streams()                     # gradient sinks written on side streams (fused_blocks.grad_sink)
        if overlap:
            self.reducer.finish()
        else:
            self.bucket.all_reduce_mean(force=self.force_allreduce)
        self.opt.step()
        return loss.detach(), items, outs, labels
