// render_feat16.hip — the kernels of render_impl.h instantiated for feature set 16 (recipe W, untextured, touched-bytes mode).
// One translation unit per feature set so that the library builds in parallel (make -j).
#include "render_impl.h"

int rtu_launch_feat16(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    // (one stack size: the touched-bytes mode is not timed, and neither images nor counters depend on the size of a stack that is large enough)
    (void)bvh_stack_needed;
    return launch_all<RTU_MAX_BVH_STACK, 16>(args, n_tiles, stats, stream, mode, probe);
}
