// TEST INFRASTRUCTURE ONLY.  Stand-in for the CUDA runtime header that the reference's
// mmdet3d/ops/spconv/include/tensorview/tensorview.h includes unconditionally; the CPU rulebook templates compiled by
// oracle/build_ref.build_spconv_rulebook() only need the stream type to be declared (tv::GPU holds one).
#pragma once
typedef void* cudaStream_t;
