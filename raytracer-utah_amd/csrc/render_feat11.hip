// render_feat11.hip — the kernels of render_impl.h instantiated for feature set 11 (recipe P, textured).
// One translation unit per feature set so that the library builds in parallel (make -j).
#include "render_impl.h"

int rtu_launch_feat11(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    (void)bvh_stack_needed;  // recipe P has one stack size (the largest)
    return launch_all<RTU_MAX_BVH_STACK, 11>(args, n_tiles, stats, stream, mode, probe);
}
