"""Nearest-neighbour video retrieval on extracted features (reference classifier.py:787-995 `test_retrieval`): per-video
feature = mean over its clips, centre both sets, L2-normalise, similarity = test . train^T, k-NN accuracy for
k in (1, 5, 10, 20, 50).  All arithmetic runs in the HIP library; the accuracy needs no sort (`dv_knn_rank`)."""
import torch

from .. import _lib as L
from .. import ops
from ..ops import DV_F32


def _chk(rc, what):
    L.check(rc, what)


def video_features(clip_features, num_clips):
    """[B * num_clips, D] clip features (the classifier's second output) -> [B, D] per-video mean (classifier.py:889-890)"""
    x = clip_features.float().contiguous()
    R, D = x.shape[0] // num_clips, x.shape[1]
    assert R * num_clips == x.shape[0]
    y = torch.empty(R, D, dtype=torch.float32, device=x.device)
    _chk(L.load().dv_group_mean_f32(x.data_ptr(), R, num_clips, D, y.data_ptr(), ops.stream_ptr()), 'dv_group_mean_f32')
    return y


def _centre_normalise(x):
    lib, s = L.load(), ops.stream_ptr()
    x = x.float().contiguous()
    M, D = x.shape
    assert D % 8 == 0
    col = torch.zeros(D, dtype=torch.float32, device=x.device)
    _chk(lib.dv_colsum_f32(x.data_ptr(), D, M, D, col.data_ptr(), s), 'dv_colsum_f32')
    one, shift = torch.ones(D, dtype=torch.float32, device=x.device), torch.zeros(D, dtype=torch.float32, device=x.device)
    _chk(lib.dv_addcmul_f32(shift.data_ptr(), col.data_ptr(), 0, -1.0 / M, D, s), 'dv_addcmul_f32')       # -mean
    c = torch.empty_like(x)
    _chk(lib.dv_bn_apply(DV_F32, x.data_ptr(), D, one.data_ptr(), shift.data_ptr(), 0, 0, c.data_ptr(), D, M, D, 0, s), 'centre')
    y, nrm = torch.empty_like(x), torch.empty(M, dtype=torch.float32, device=x.device)
    _chk(lib.dv_l2norm_fwd(c.data_ptr(), M, D, 1e-12, y.data_ptr(), nrm.data_ptr(), s), 'dv_l2norm_fwd')
    return y


def nn_retrieval(test_feature, test_label, train_feature, train_label, ks=(1, 5, 10, 20, 50)):
    """-> ({k: accuracy}, sim [n_test, n_train]); classifier.py:964-981"""
    L.require_device()
    te, tr = _centre_normalise(test_feature), _centre_normalise(train_feature)
    R, Nt, D = te.shape[0], tr.shape[0], te.shape[1]
    lib, s = L.load(), ops.stream_ptr()
    sim = torch.empty(R, Nt, dtype=torch.float32, device=te.device)
    _chk(lib.dv_gemm_f32(R, Nt, D, te.data_ptr(), D, 1, tr.data_ptr(), 1, D, sim.data_ptr(), Nt, 1.0, 0, s), 'similarity')
    rank = torch.empty(R, dtype=torch.int32, device=te.device)
    tl = train_label.to(device=te.device, dtype=torch.int32).contiguous()
    ql = test_label.to(device=te.device, dtype=torch.int32).contiguous()
    _chk(lib.dv_knn_rank(sim.data_ptr(), Nt, R, Nt, tl.data_ptr(), ql.data_ptr(), rank.data_ptr(), s), 'dv_knn_rank')
    rk = rank.cpu()
    return {k: float((rk < k).float().mean()) for k in ks}, sim
