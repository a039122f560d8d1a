// TEST INFRASTRUCTURE ONLY.  Python binding for the reference's own CPU points-in-boxes routine, which is compiled
// from mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cpu.cpp where it lies under /root/reference (see
// oracle/build_ref.py; the reference binds it in roiaware_pool3d.cpp together with CUDA-only entry points, which
// cannot be built here).  Only the declaration of that routine appears in this file.
#include <torch/extension.h>

int points_in_boxes_cpu(at::Tensor boxes_tensor, at::Tensor pts_tensor, at::Tensor pts_indices_tensor);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("points_in_boxes_cpu", &points_in_boxes_cpu, "boxes [N,7] fp32, pts [P,3] fp32, out flags [N,P] int32");
}
