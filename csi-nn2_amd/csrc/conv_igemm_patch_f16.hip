// conv_igemm_patch_f16.hip -- the binary16 instantiations of the row-patch kernel (NHWC, 3x3 stride 1 "same";
// conv_igemm_patch.hip describes the kernel, conv_igemm_patch_kernel.h:patch_body what kF16 changes).  A translation unit
// of its own: the build compiles .hip files in parallel.
// Restates shl_ref_conv2d_nhwc_f32 on binary16 tensors (source/reference/convolution.c:28-89, conversions
// source/nn2/utils.c:576-643, relu variants convolution_relu.c); parity bar 1e-3 relative (fp32 summation order).
#include "conv_igemm_patch_kernel.h"

namespace shl {

int patch_launch_nhwc_f16(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s) { return patch_launch_layout<false, true>(a, tiles, lds, s); }
int patch_read_trace_nhwc_f16(unsigned long long *host, int count) { return patch_read_trace_layout<false>(host, count); }

}  // namespace shl
