// conv_igemm_patch_nchw_f16.hip -- the binary16 NCHW instantiations of the row-patch kernel (3x3 stride 1 "same", read and
// written NCHW: no re-layout pass; conv_igemm_patch.hip describes the kernel, conv_igemm_patch_kernel.h:patch_body what
// kF16 and kNchw change).  A translation unit of its own: the build compiles .hip files in parallel.
// Restates shl_ref_conv2d_nchw_f32 on binary16 tensors (source/reference/convolution.c:91-139, conversions
// source/nn2/utils.c:576-643, relu variants convolution_relu.c); parity bar 1e-3 relative (fp32 summation order).
#include "conv_igemm_patch_kernel.h"

namespace shl {

int patch_launch_nchw_f16(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s) { return patch_launch_layout<true, true>(a, tiles, lds, s); }
int patch_read_trace_nchw_f16(unsigned long long *host, int count) { return patch_read_trace_layout<true>(host, count); }

}  // namespace shl
