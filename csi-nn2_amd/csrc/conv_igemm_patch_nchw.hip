// conv_igemm_patch_nchw.hip -- the NCHW instantiations of the row-patch kernel (conv_igemm_patch.hip describes it; a
// translation unit per layout halves the longest compile of the build).
#include "conv_igemm_patch_kernel.h"

namespace shl {

int patch_launch_nchw(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s) { return patch_launch_layout<true>(a, tiles, lds, s); }
int patch_read_trace_nchw(unsigned long long *host, int count) { return patch_read_trace_layout<true>(host, count); }

}  // namespace shl
