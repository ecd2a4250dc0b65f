"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DualVar pretrain hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker (never as the thing measured or shipped).
The product package ``dualvar_amd`` must never import from here.

Contents
--------
torch_ref.py   fp32 PyTorch-CPU restatement of the reference's hot path
               (backbones, SimCLR / MoCo objectives, GatherLayer, top-k).  Each
               symbol cites the reference file:line it follows.  Pinned against
               the reference itself by ``gen_golden.py`` (run in the build
               container, where /root/reference is importable) and against the
               committed fixtures in ``tests/golden`` everywhere else.
procedural.py  RNG-independent parameter / input fill shared by the fixture
               generator, the oracle and the tests.
harness.py     import shim for the *reference* (stubs absent third-party
               modules).  Used only by gen_golden.py / the pinning tests, only in
               the build container.
gen_golden.py  writes tests/golden/*.npz from the reference's own classes.
"""
