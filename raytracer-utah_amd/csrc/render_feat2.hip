// render_feat2.hip — the kernels of render_impl.h instantiated for feature set 2 (recipe S, untextured).
// One translation unit per feature set so that the library builds in parallel (make -j).
#include "render_impl.h"

int rtu_launch_feat2(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    if (bvh_stack_needed <= 16) return launch_all<16, 2>(args, n_tiles, stats, stream, mode, probe);
    if (bvh_stack_needed <= 24) return launch_all<24, 2>(args, n_tiles, stats, stream, mode, probe);
    if (bvh_stack_needed <= 32) return launch_all<32, 2>(args, n_tiles, stats, stream, mode, probe);
    return launch_all<RTU_MAX_BVH_STACK, 2>(args, n_tiles, stats, stream, mode, probe);
}
