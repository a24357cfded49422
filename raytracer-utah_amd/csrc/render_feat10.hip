// render_feat10.hip — the kernels of render_impl.h instantiated for feature set 10 (recipe P, untextured).
// One translation unit per feature set so that the library builds in parallel (make -j).
#include "render_impl.h"

int rtu_launch_feat10(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    (void)bvh_stack_needed;  // recipe P has one stack size (the largest)
    return launch_all<RTU_MAX_BVH_STACK, 10>(args, n_tiles, stats, stream, mode, probe);
}
