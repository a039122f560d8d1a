"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the hot path, used as the parity
checker by tests/, by __graft_entry__.smoke() and by bench.py's ``cpu_baseline`` leg.  Nothing in the
product package ``sst_amd`` imports, links or executes anything from this directory.

Pinning status:
  dynamic_voxelize         pinned against the reference's own C++ (oracle/_ref, built from
                           /root/reference/mmdet3d/ops/voxel/src/*.cpp by oracle/build_ref.py) and the golden
                           vectors generated from it (tests/golden/voxelize_*.npz).
  DynamicScatter fwd/bwd   the reference has no CPU implementation ("do not support cpu yet",
                           voxelization.h:106); restated from scatter_points_cuda.cu:183-303 and pinned with the
                           brute-force construction of the reference's own test
                           (tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:8-93).
  window / region batching / pos-embed / SRA block / SIR
                           no reference test or fixture exists: pinned against the reference's own Python
                           executed unmodified under stubs (oracle/ref_loader.py) in the build container and
                           against golden tensors generated from it (tests/golden/make_golden.py).
                           TorchEx ingroup_indices and torch_scatter are un-vendored third-party natives:
                           "parity unpinned" for the in-window order (any bijection is valid per the reference's
                           fallback) — the oracle fixes it to ascending voxel index.
"""
