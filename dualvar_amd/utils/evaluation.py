"""The classifier's 10-clip test (reference classifier.py:657-738 `temporal_test_10clip`, summary `:762-784`): a video
is `num_seq` temporally uniform clips; each goes through the eval-mode `LinearClassifier`, the class probabilities
are averaged over the clips and the video is scored top-1 / top-5 on the mean.  Softmax, the clip average and the
target's rank run in the HIP library; the loop over a dataset stays with the caller (the reference's DataLoader)."""
import torch

from .. import _lib as L
from .. import ops


def clips_from_sequence(input_seq, num_seq, seq_len):
    """`tr()` of classifier.py:670-677: [B, 3, num_seq * seq_len, H, W] -> [B * num_seq, 3, seq_len, H, W]"""
    B, C, TT, H, W = input_seq.shape
    if TT != num_seq * seq_len:
        raise ValueError('expected %d x %d frames per video, got %d' % (num_seq, seq_len, TT))
    return input_seq.view(B, C, num_seq, seq_len, H, W).permute(0, 2, 1, 3, 4, 5).contiguous().view(B * num_seq, C, seq_len, H, W)


def softmax_rows(logit):
    lg = logit.float().contiguous()
    R, K = lg.shape
    pr = torch.empty_like(lg)
    L.check(L.load().dv_softmax_rows_f32(lg.data_ptr(), K, R, K, pr.data_ptr(), K, ops.stream_ptr()), 'dv_softmax_rows_f32')
    return pr


def ten_clip_probabilities(model, input_seq, num_seq=10, seq_len=16):
    """-> (prob_per [B, num_seq, K], prob_mean [B, K]); the model must be in eval mode (classifier.py:661)"""
    L.require_device()
    if model.training:
        raise RuntimeError('ten_clip_probabilities expects model.eval() (classifier.py:661)')
    B = input_seq.shape[0]
    with torch.no_grad():
        logit, _ = model(clips_from_sequence(input_seq, num_seq, seq_len))
    prob = softmax_rows(logit)
    K = prob.shape[1]
    mean = torch.empty(B, K, dtype=torch.float32, device=prob.device)
    L.check(L.load().dv_group_mean_f32(prob.data_ptr(), B, num_seq, K, mean.data_ptr(), ops.stream_ptr()), 'dv_group_mean_f32')
    return prob.view(B, num_seq, K), mean


def topk_of_mean(prob_mean, target, topk=(1, 5)):
    """calc_topk_accuracy(mean_prob, target, (1, 5)) (classifier.py:772) without a sort: the number of classes whose mean
    probability exceeds the target's, from dv_knn_rank on a one-hot label table."""
    pm = prob_mean.float().contiguous()
    R, K = pm.shape
    cls = torch.arange(K, dtype=torch.int32, device=pm.device)
    tgt = target.to(device=pm.device, dtype=torch.int32).contiguous()
    rank = torch.empty(R, dtype=torch.int32, device=pm.device)
    L.check(L.load().dv_knn_rank(pm.data_ptr(), K, R, K, cls.data_ptr(), tgt.data_ptr(), rank.data_ptr(), ops.stream_ptr()),
            'dv_knn_rank')
    rk = rank.cpu()
    return [float((rk < k).float().mean()) for k in topk]
