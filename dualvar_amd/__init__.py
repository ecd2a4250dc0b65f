"""dualvar_amd -- MI355X (gfx950) native implementation of DualVar's data-parallel pretrain hot path.

Python host mirroring the reference's operator surface (select_backbone, SimCLR_* / MoCo_* models,
GatherLayer, pretrain.py loop) over hand-written HIP kernels behind the C ABI in
include/dualvar_hip.h.  There is no CPU / PyTorch fallback: without the built
libdualvar_hip.so and a gfx950 device every compute entry point raises."""
__version__ = '0.1.0'
