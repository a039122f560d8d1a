// TEST INFRASTRUCTURE ONLY.  Python binding for the reference's own CPU rulebook ("indice pair") routines of its
// vendored spconv: the templates getIndicePairsConv / getIndicePairsSubM / getIndicePairsDeConv of
// mmdet3d/ops/spconv/include/spconv/geometry.h are instantiated from that header where it lies under
// /root/reference (oracle/build_ref.py adds its include directory); the reference binds them through
// spconv_ops.h / all.cc together with CUDA-only code that cannot be built here.  This file contains only the glue:
// tensor -> tv::TensorView, the output-shape formulas of mmdet3d/ops/spconv/ops.py:20-52 are NOT restated here
// (the caller passes out_shape).
#include <torch/extension.h>

#include <spconv/geometry.h>
#include <spconv/maxpool.h>
#include <tensorview/tensorview.h>

namespace {

tv::TensorView<int> view(at::Tensor t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back((int)s);
  return tv::TensorView<int>(t.data_ptr<int>(), shape);
}

tv::TensorView<const int> cview(at::Tensor t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back((int)s);
  return tv::TensorView<const int>(t.data_ptr<int>(), shape);
}

// indices [N, 4] int32 (b, z, y, x); returns (out_indices [M, 4], indice_pairs [K, 2, N], indice_num [K])
std::vector<at::Tensor> get_indice_pairs_3d(at::Tensor indices, int64_t batch_size, std::vector<int64_t> out_shape,
                                            std::vector<int64_t> ksize, std::vector<int64_t> stride,
                                            std::vector<int64_t> padding, std::vector<int64_t> dilation, bool subm,
                                            bool transpose) {
  TORCH_CHECK(indices.dtype() == at::kInt && indices.is_contiguous() && indices.size(1) == 4);
  const int64_t n = indices.size(0);
  int kv = 1, vol = 1;
  int ks[3], st[3], pd[3], dl[3], os[3];
  for (int i = 0; i < 3; ++i) {
    ks[i] = (int)ksize[i];
    st[i] = (int)stride[i];
    pd[i] = (int)padding[i];
    dl[i] = (int)dilation[i];
    os[i] = (int)out_shape[i];
    kv *= ks[i];
    vol *= os[i];
  }
  auto pairs = at::full({kv, 2, n}, -1, indices.options());
  auto num = at::zeros({kv}, indices.options());
  auto grid = at::full({batch_size * vol}, -1, indices.options());
  at::Tensor out;
  int m;
  if (subm) {
    m = spconv::getIndicePairsSubM<int, int, 3>(cview(indices), view(grid), view(pairs), view(num), ks, st, pd, dl, os);
    out = indices.clone();
  } else {
    out = at::zeros({n * kv, 4}, indices.options());
    if (transpose)
      m = spconv::getIndicePairsDeConv<int, int, 3>(cview(indices), view(out), view(grid), view(pairs), view(num), ks,
                                                    st, pd, dl, os);
    else
      m = spconv::getIndicePairsConv<int, int, 3>(cview(indices), view(out), view(grid), view(pairs), view(num), ks, st,
                                                  pd, dl, os);
    out = out.slice(0, 0, m).clone();
  }
  return {out, pairs, num};
}

// Sparse max pooling: the per-offset functors are the reference's (src/maxpool.cc, compiled next to this file); the
// loop over the kernel offsets and the zero-filled start follow include/spconv/pool_ops.h:24-97.
tv::TensorView<float> fview(at::Tensor t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back((int)s);
  return tv::TensorView<float>(t.data_ptr<float>(), shape);
}

tv::TensorView<const float> cfview(at::Tensor t) {
  tv::Shape shape;
  for (auto s : t.sizes()) shape.push_back((int)s);
  return tv::TensorView<const float>(t.data_ptr<float>(), shape);
}

at::Tensor indice_maxpool(at::Tensor features, at::Tensor pairs, at::Tensor num, int64_t num_act) {
  auto out = at::zeros({num_act, features.size(1)}, features.options());
  spconv::functor::SparseMaxPoolForwardFunctor<tv::CPU, float, int> f;
  for (int64_t k = 0; k < pairs.size(0); ++k) {
    const int n = num.data_ptr<int>()[k];
    if (n > 0) f(tv::CPU(), fview(out), cfview(features), cview(pairs).subview(k), n);
  }
  return out;
}

at::Tensor indice_maxpool_backward(at::Tensor features, at::Tensor out_features, at::Tensor out_grad, at::Tensor pairs,
                                   at::Tensor num) {
  auto din = at::zeros_like(features);
  spconv::functor::SparseMaxPoolBackwardFunctor<tv::CPU, float, int> f;
  for (int64_t k = 0; k < pairs.size(0); ++k) {
    const int n = num.data_ptr<int>()[k];
    if (n > 0) f(tv::CPU(), cfview(out_features), cfview(features), cfview(out_grad), fview(din), cview(pairs).subview(k), n);
  }
  return din;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("get_indice_pairs_3d", &get_indice_pairs_3d);
  m.def("indice_maxpool", &indice_maxpool);
  m.def("indice_maxpool_backward", &indice_maxpool_backward);
}
